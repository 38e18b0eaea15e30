"""Derived fusion model for BASELINE configs 4 and 5 (R2Plus1D + Transformer-0D with GradientBlending; SlowFast +
MLSTM_FCN): the reference's ``MultiModalModel_GB`` (src/models/MultiModal.py:56-168) is hard-wired to ViViT + Transformer, so
this class applies ITS recipe to any pair of the mirrored models - each stream keeps its own head, the latents (the inputs of the
vision head's first Linear and of ``classifier[0]``, which the reference captures with forward hooks, :96-97) are concatenated
and go through ``connector`` (Linear, ReLU) and ``classifier`` (Linear, LayerNorm, ReLU, Linear).  Same child names
(``vis_model``, ``ts_model``, ``connector``, ``classifier``), ``use_stream`` protocol and return convention as the reference
class, so ``src.train`` / ``src.GradientBlending.train_GB`` / ``src.distributed`` drive it unchanged.  Not a reference class:
"derived"; pinned by tests/golden/fusion_derived.npz, recorded from the reference's model classes combined with hooks.
"""
from typing import Literal

import torch
import torch.nn as nn

import os

from ..utils import streams
from .MLSTM_FCN import MLSTM_FCN
from .MultiModal import _classifier, _connector, _make_classifier, _make_connector, _param_table
from .R2Plus1D import R2Plus1DClassifier, R2Plus1DNet
from .slowfast import SlowFast
from .transformer import Transformer
from .ViViT import ViViT


_GRAPH_BRANCH = os.environ.get("MD_GRAPH_BRANCH") == "1"
_GRAPH_BRANCH_FORK = os.environ.get("MD_GRAPH_BRANCH_FORK") == "1"     # the graphed branch on the side stream as well


def _vision_adapter(m: nn.Module):
    """(latent(x), head(latent), latent width) of a mirrored vision classifier."""
    if isinstance(m, R2Plus1DClassifier):
        return m.res2plus1d, m.linear, R2Plus1DNet.output_channels()
    if isinstance(m, SlowFast):
        return (lambda x: m.encoder(x).reshape(x.shape[0], -1)), m.classifier, m.classifier.input_dim
    if isinstance(m, ViViT):
        return m._encode, m._head, m.dim
    raise TypeError("FusionGB: unsupported vision model %s" % type(m).__name__)


def _ts_adapter(m: nn.Module):
    if isinstance(m, Transformer):
        return m.encoder, m._head, m.feature_dims
    if isinstance(m, MLSTM_FCN):
        return m._features, m._head, m.converter.out_features
    raise TypeError("FusionGB: unsupported 0D model %s" % type(m).__name__)


class FusionGB(nn.Module):
    def __init__(self, n_classes: int, vis_model: nn.Module, ts_model: nn.Module,
                 use_stream: Literal["video", "0D", "multi", "multi-GB"] = "multi-GB"):
        super(FusionGB, self).__init__()
        self.n_classes = n_classes
        self.vis_model = vis_model
        self.ts_model = ts_model
        self._vis = _vision_adapter(vis_model)
        self._ts = _ts_adapter(ts_model)
        linear_input_dims = self._ts[2] + self._vis[2]
        self.connector = _make_connector(linear_input_dims, linear_input_dims // 2)
        self.classifier = _make_classifier(linear_input_dims // 2, linear_input_dims // 2, n_classes)
        self.vis_latent = None
        self.ts_latent = None
        self.use_stream = use_stream

    def update_use_stream(self, use_stream: Literal["video", "0D", "multi", "multi-GB"]):
        self.use_stream = use_stream

    def remove_my_hooks(self):
        pass

    def forward(self, x_vis: torch.Tensor, x_ts: torch.Tensor):
        from .. import ops
        from ._unit import deferred_bn_counters
        with deferred_bn_counters(), ops.prepacked(self):   # BatchNorm step counters and weight packs of both encoders, batched
            return self.forward_stream(x_vis, x_ts)

    def forward_stream(self, x_vis: torch.Tensor, x_ts: torch.Tensor):
        if self.use_stream == "video":
            return self.vis_model(x_vis)
        elif self.use_stream == "0D":
            return self.ts_model(x_ts)
        if streams.enabled(x_vis) and x_ts.is_cuda and not (_GRAPH_BRANCH and not _GRAPH_BRANCH_FORK and torch.is_grad_enabled()):
            # the 0D encoder and its head on a side stream, beside the video encoder (src/utils/streams.py)
            with streams.fork(x_vis.device, 1, (x_ts,)) as f:
                ts_latent, out_ts = self._ts_branch(x_ts)
            vis_latent = self._vis[0](x_vis)
            out_vis = self._vis[1](vis_latent)
            f.join(ts_latent, out_ts)
        else:
            vis_latent = self._vis[0](x_vis)
            ts_latent, out_ts = self._ts_branch(x_ts)
            out_vis = self._vis[1](vis_latent)
        self.vis_latent = (vis_latent.detach(),)       # values only: a kept graph would outlive the step (src/utils/graphed.py)
        self.ts_latent = (ts_latent.detach(),)
        x = _connector(torch.cat([vis_latent, ts_latent], axis=1), self.connector)
        out_multi = _classifier(x, self.classifier)
        return out_multi if self.use_stream == 'multi' else (out_multi, out_vis, out_ts)

    def _ts_eager(self, x_ts: torch.Tensor):
        ts_latent = self._ts[0](x_ts)
        return ts_latent, self._ts[1](ts_latent)

    def _ts_branch(self, x_ts: torch.Tensor):
        """(latent, logits) of the 0D encoder.  MD_GRAPH_BRANCH=1: its forward and its backward are replayed from HIP graphs
        (src/utils/graphed.py::GraphedBranch) while the rest of the step stays eager -- for the pairs whose video trunk runs from the
        C++ executor and whose 0D encoder is a few hundred tiny launches (cfg4)."""
        if not (_GRAPH_BRANCH and x_ts.is_cuda and torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()):
            return self._ts_eager(x_ts)
        gbr = self.__dict__.get("_md_ts_graph")
        if gbr is None:
            from ..utils.graphed import GraphedBranch
            try:
                gbr = GraphedBranch(self.ts_model, self._ts_eager, (x_ts,))
            except RuntimeError as e:
                print("FusionGB | MD_GRAPH_BRANCH: capture refused, 0D encoder stays eager (%s)" % str(e).split("\n")[0][:200])
                gbr = False
            self.__dict__["_md_ts_graph"] = gbr
        if gbr is False or not gbr.matches((x_ts,), self.training):
            return self._ts_eager(x_ts)
        return gbr(x_ts)

    def encode(self, x_vis: torch.Tensor, x_0D: torch.Tensor):
        with torch.no_grad():
            vis_latent = self._vis[0](x_vis)
            ts_latent = self._ts[0](x_0D)
            x = _connector(torch.cat([vis_latent, ts_latent], axis=1), self.connector)
        return (x, vis_latent, ts_latent)

    def summary(self, *args, **kwargs):
        _param_table(self)
