"""MI355X-native pieces of the reference's ``src/models/resnet.py``.

Built here: ``SwishEfficient`` / ``Swish`` (resnet.py:63-81) on ``md_swish_fwd`` / ``md_swish_bwd``.
Everything else of that module (BasicBlock3D, Bottleneck3D, ResNet3D, SubBatchNorm3d, ...) is not rebuilt yet
(SURVEY section 8a rows a6-a7).  When ``MD_REFERENCE_SRC`` points at the reference's ``src`` directory those names are
loaded from the reference file and re-exported with ITS ``Swish`` / ``SwishEfficient`` replaced by the ones below, so
``from src.models.resnet import *`` (slowfast.py:5) keeps working and the reference's SlowFast runs its Swish on the
gfx950 kernel.  Without it this module exports the two native names only.
"""
import importlib.util as _ilu
import os as _os

import torch
import torch.nn as nn

from .. import _native as N
from .. import ops


class SwishEfficient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ops.require_cuda(x)
        x = ops.f32(x).contiguous()
        y = torch.empty_like(x)
        N.check(N.lib().md_swish_fwd(ops._p(x), x.numel(), ops._p(y), ops._stream()), "md_swish_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        g = ops.f32(grad_output).contiguous()
        dx = torch.empty_like(x)
        N.check(N.lib().md_swish_bwd(ops._p(x), ops._p(g), x.numel(), ops._p(dx), ops._stream()), "md_swish_bwd")
        return dx


class Swish(nn.Module):
    def __init__(self):
        super(Swish, self).__init__()

    def forward(self, x):
        return SwishEfficient.apply(x)


def _adopt_reference():
    ref = _os.environ.get("MD_REFERENCE_SRC")
    path = _os.path.join(ref, "models", "resnet.py") if ref else None
    if not path or not _os.path.isfile(path):
        return
    spec = _ilu.spec_from_file_location("src.models._reference_resnet", path)
    mod = _ilu.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.Swish = Swish
    mod.SwishEfficient = SwishEfficient
    g = globals()
    for name in dir(mod):
        if not name.startswith("_") and name not in ("Swish", "SwishEfficient"):
            g.setdefault(name, getattr(mod, name))


_adopt_reference()
