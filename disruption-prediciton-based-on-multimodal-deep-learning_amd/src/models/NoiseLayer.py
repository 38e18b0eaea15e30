"""MI355X-native mirror of the reference's ``src/models/NoiseLayer.py`` (same class, same constructor).

Training: ``x + (mean + randn * std)``.  The reference draws the noise with the CPU default generator
(``torch.randn(x.size())``, NoiseLayer.py:13) and moves it to the input's device; this mirror keeps exactly that, so
seeded runs see the same noise, and does the arithmetic in one launch (``md_add_noise``).  Eval: identity.
The 0D encoders of the reference (CnnLSTM.py:38, MLSTM_FCN.py:113, transformer.py:57) import this module by name.
"""
import torch
import torch.nn as nn

from .. import _native as N
from .. import ops


class _AddNoise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, noise, mean, std):
        out = torch.empty_like(x)
        N.check(N.lib().md_add_noise(ops._p(x), ops._p(noise), float(mean), float(std), x.numel(), ops._p(out), ops._stream()),
                "md_add_noise")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return grad_out, None, None, None


class NoiseLayer(nn.Module):
    def __init__(self, mean: float = 0, std: float = 1e-2):
        super().__init__()
        self.mean = mean
        self.std = std

    def forward(self, x: torch.Tensor):
        if not self.training:
            return x
        ops.require_cuda(x)
        static = self.__dict__.get("_graph_mode")
        if static and not torch.cuda.is_current_stream_capturing():
            # an eager step of another batch size beside a captured one (the short last batch of an epoch): the staging buffer the
            # graph uploads from must stay as it is
            st = self.__dict__.get("_static")
            static = st is None or st[0].shape == x.shape
        noise = self._draw_static(x) if static else self._draw(x)
        return _AddNoise.apply(ops.f32(x), noise, self.mean, self.std)

    # -- whole-step HIP graphs (src/utils/graphed.py): the upload must read the SAME pinned buffer on every replay, and the host
    #    refills it from the CPU generator before each replay (refresh_static), so the noise stream stays the reference's
    def _draw_static(self, x):
        st = self.__dict__.get("_static")
        capturing = torch.cuda.is_current_stream_capturing()
        if st is None or st[0].shape != x.shape:
            if capturing:
                raise RuntimeError("NoiseLayer: graph mode needs one eager forward (warm-up) before the capture")
            st = (torch.empty(x.shape, dtype=torch.float32).pin_memory(), torch.empty(x.shape, dtype=torch.float32, device=x.device),
                  torch.cuda.Event())
            self.__dict__["_static"] = st
        host, dev, done = st
        if not capturing:
            done.synchronize()
            torch.randn(x.size(), out=host)
        dev.copy_(host, non_blocking=True)
        if not capturing:
            done.record(torch.cuda.current_stream(x.device))
        return dev

    def refresh_static(self):
        """Draw the next noise tensor into the pinned buffer a captured step uploads from (call before each replay, after the
        previous replay has finished)."""
        st = self.__dict__.get("_static")
        if st is not None and self.training:
            torch.randn(st[0].size(), out=st[0])

    def __getstate__(self):                                      # the staging ring (pinned memory, events) is per process
        return {k: v for k, v in self.__dict__.items() if k not in ("_ring", "_static", "_graph_mode")}

    def __deepcopy__(self, memo):
        return NoiseLayer(self.mean, self.std).train(self.training)

    def _draw(self, x):
        """torch.randn(x.size()) from the CPU default generator, as the reference, drawn into a pinned staging buffer (ring of 4,
        one event each) and uploaded without blocking the host: the reference's pageable `.to(device)` waits for everything
        queued on the stream before it."""
        ring = self.__dict__.setdefault("_ring", {"i": 0, "slots": [None] * 4})
        i = ring["i"] = (ring["i"] + 1) % 4
        slot = ring["slots"][i]
        if slot is None or slot[0].shape != x.shape:
            slot = ring["slots"][i] = (torch.empty(x.shape, dtype=torch.float32).pin_memory(), torch.cuda.Event())
        else:
            slot[1].synchronize()                                # the upload that last used this buffer has finished
        torch.randn(x.size(), out=slot[0])
        noise = slot[0].to(x.device, non_blocking=True)
        slot[1].record(torch.cuda.current_stream(x.device))
        return noise
