"""MI355X-native drop-in for the hot path of the reference's ``src`` package.

Put this directory's PARENT on ``PYTHONPATH`` and the reference's training scripts import
``src.models.*``, ``src.loss``, ``src.train``, ``src.GradientBlending`` and ``src.distributed`` from here.
All compute runs in the gfx950 shared library ``csrc/libmi355x_disrupt.so`` (C ABI: include/mi355x_disrupt.h);
there is no CPU fallback -- a missing library or a CPU tensor raises.
"""

import os as _os

# Optional: let src.* modules that are NOT part of the hot path (dataset, evaluate, utils.utility, ...) resolve from
# the reference checkout, so its unchanged training scripts find everything under one `src` package.  Modules
# mirrored here come first on __path__ and win.
_ref = _os.environ.get("MD_REFERENCE_SRC")
if _ref and _os.path.isdir(_ref):
    __path__.append(_ref)
