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
