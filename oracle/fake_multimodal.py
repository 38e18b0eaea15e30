"""Test infrastructure: a tiny two-stream model with the interface the Gradient-Blending loops expect from
MultiModalModel_GB (reference src/models/MultiModal.py:62-151): ``update_use_stream(task)`` and a forward over
(video, 0D) that returns the fused logits, or (fused, video, 0D) logits in "multi-GB" mode.  Used by
tests/golden/make_golden.py (driving the REFERENCE loops) and by tests/test_gb_loops.py (driving the mirrored loops)."""
import torch
import torch.nn as nn


class FakeMultiModalGB(nn.Module):
    def __init__(self, d_vis: int = 6, d_ts: int = 4, hidden: int = 5, seed: int = 3):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.vis = nn.Linear(d_vis, hidden); self.ts = nn.Linear(d_ts, hidden)
        self.head_vis = nn.Linear(hidden, 2); self.head_ts = nn.Linear(hidden, 2); self.head = nn.Linear(2 * hidden, 2)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
        self.use_stream = "multi"

    def update_use_stream(self, use_stream):
        self.use_stream = use_stream

    def forward(self, x_vis, x_ts):
        hv, ht = torch.tanh(self.vis(x_vis)), torch.tanh(self.ts(x_ts))
        if self.use_stream == "video":
            return self.head_vis(hv)
        if self.use_stream == "0D":
            return self.head_ts(ht)
        fused = self.head(torch.cat([hv, ht], dim=1))
        if self.use_stream == "multi-GB":
            return fused, self.head_vis(hv), self.head_ts(ht)
        return fused


class DictSet(torch.utils.data.Dataset):
    def __init__(self, n: int, seed: int, d_vis: int = 6, d_ts: int = 4):
        g = torch.Generator().manual_seed(seed)
        self.v = torch.randn(n, d_vis, generator=g); self.t = torch.randn(n, d_ts, generator=g)
        self.y = ((self.v[:, 0] + 0.5 * self.t[:, 1] + 0.3 * torch.randn(n, generator=g)) > 0).long()

    def __len__(self):
        return self.y.numel()

    def __getitem__(self, i):
        return {"video": self.v[i], "0D": self.t[i]}, self.y[i]


def loaders(seed: int = 11):
    tr = torch.utils.data.DataLoader(DictSet(48, seed), batch_size=8, shuffle=False)
    va = torch.utils.data.DataLoader(DictSet(24, seed + 1), batch_size=8, shuffle=False)
    return tr, va


class OracleMultiModalGB(nn.Module):
    """Test infrastructure: oracle.multimodal's functional MultiModalModel_GB restatement (itself pinned by
    tests/golden/multimodal.npz) behind the module interface the Gradient-Blending loops need -- parameters that an optimizer can
    step, state_dict()/load_state_dict() for the checkpoint reload, update_use_stream() and the four forward variants of the
    reference (src/models/MultiModal.py:131-149).  Built from a state dict with the reference's keys; float64 on the CPU."""

    def __init__(self, state: dict, param_names, pool: str = "cls", alpha: float = 1.0):
        super().__init__()
        self._keys = list(state)
        self._pool, self._alpha = pool, alpha
        for k, v in state.items():
            v = v.detach().clone()
            v = v.double() if v.is_floating_point() else v
            if k in param_names:
                self.register_parameter(k.replace(".", "__"), nn.Parameter(v))
            else:
                self.register_buffer(k.replace(".", "__"), v)
        self.use_stream = "multi-GB"

    def update_use_stream(self, use_stream):
        self.use_stream = use_stream

    def _sd(self):
        return {k: getattr(self, k.replace(".", "__")) for k in self._keys}

    def forward(self, x_vis, x_ts):
        from . import multimodal as om
        from . import transformer0d as ot
        from . import vivit as ov
        sd = self._sd()
        x_vis, x_ts = x_vis.double(), x_ts.double()
        if self.use_stream == "video":
            return ov.vivit_forward(x_vis, om._sub(sd, "vis_model."), om.VIDEO["patch_size"], om.VIDEO["depth"], om.VIDEO["n_heads"],
                                    self._pool, 3, self._alpha, with_mlp=True)
        if self.use_stream == "0D":
            return ot.transformer0d_forward(x_ts, om._sub(sd, "ts_model."), om.TS["n_layers"], om.TS["n_heads"], om.TS["kernel_size"],
                                            self.training, with_classifier=True)
        outs = om.multimodal_gb_forward(x_vis, x_ts, sd, self._pool, self._alpha, self.training)
        return outs[0] if self.use_stream == "multi" else outs


class ClipSet(torch.utils.data.Dataset):
    """(video clip, 0D series) pairs in the reference's multimodal batch format ({"video": (C,T,H,W), "0D": (L,F)}, label)."""

    def __init__(self, n: int, seed: int, frames: int = 5, size: int = 32, feats: int = 6):
        g = torch.Generator().manual_seed(seed)
        self.y = (torch.rand(n, generator=g) > 0.5).long()
        s = (self.y.float() * 2 - 1).view(n, 1, 1, 1, 1)
        self.v = torch.randn(n, 3, frames, size, size, generator=g) + 0.4 * s
        self.t = torch.randn(n, frames, feats, generator=g) + 0.4 * s.view(n, 1, 1)

    def __len__(self):
        return self.y.numel()

    def __getitem__(self, i):
        return {"video": self.v[i], "0D": self.t[i]}, self.y[i]


def clip_loaders(seed: int = 21, n_train: int = 32, n_valid: int = 16, batch: int = 8):
    tr = torch.utils.data.DataLoader(ClipSet(n_train, seed), batch_size=batch, shuffle=False)
    va = torch.utils.data.DataLoader(ClipSet(n_valid, seed + 1), batch_size=batch, shuffle=False)
    return tr, va
