import os as _os

_ref = _os.environ.get("MD_REFERENCE_SRC")
if _ref and _os.path.isdir(_os.path.join(_ref, "models")):
    __path__.append(_os.path.join(_ref, "models"))
