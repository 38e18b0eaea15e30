"""MI355X-native mirror of the reference's ``src/models/MultiModal.py`` (same classes, constructor arguments, child-module names
and state-dict keys).  All four wrappers are hard-wired to ViViT + the 0D Transformer, as in the reference:

* ``MultiModalModel`` (:10-53): ViViTEncoder | TransformerEncoder latents concatenated -> Linear, ReLU -> Linear, LayerNorm, ReLU, Linear.
* ``MultiModalModel_GB`` (:56-168): ViViT and Transformer with their own heads plus the fused head on the latents (the inputs of
  ``vis_model.mlp[0]`` / ``ts_model.classifier[0]``, which the reference captures with forward hooks; here the encoders hand
  them over directly and ``vis_latent`` / ``ts_latent`` hold the same one-element tuples).
* ``TFN`` (:173-243) / ``TFN_GB`` (:246-331): tensor fusion [1 | h_vis] (x) [1 | h_0D] (``md_outer_*``) -> connector / classifier.

The reference's ``*_GB`` classes read ``Transformer.feature_dims``, which its Transformer never sets; the mirror
of ``Transformer`` exposes it, so these construct.  Arithmetic: the encoders' kernels, rows-major MFMA Linears +
``md_channel_bias_*``, ``md_add_layernorm_*``, ``md_elu`` with alpha 0 for the ReLUs, the Linear+BatchNorm1d+ReLU conv unit for
TFN_GB's classifier; torch only concatenates.
"""
from typing import Dict, Literal

import torch
import torch.nn as nn

from ._unit import AddLayerNormFunction, EluFunction, OuterFusionFunction, dropout, linear_bn_leaky, linear_wb
from .transformer import Transformer, TransformerEncoder
from .ViViT import ViViT, ViViTEncoder


def _lin(x, lin: nn.Linear):
    return linear_wb(x, lin.weight, lin.bias)


def _relu(x):
    return EluFunction.apply(x, 0.0)


def _connector(x, seq: nn.Sequential):                 # Linear, ReLU
    return _relu(_lin(x, seq[0]))


def _classifier(x, seq: nn.Sequential):                # Linear, LayerNorm, ReLU, Linear
    h = AddLayerNormFunction.apply(_lin(x, seq[0]), None, seq[1].weight, seq[1].bias, seq[1].eps)
    return _lin(_relu(h), seq[3])


def _param_table(m: nn.Module):
    rows = ["%-70s %-22s %d" % (k, tuple(v.shape), v.numel()) for k, v in m.named_parameters()]
    print("\n".join(rows + ["total parameters: %d" % sum(p.numel() for p in m.parameters())]))


def _make_connector(d_in: int, d_out: int) -> nn.Sequential:
    """Linear, ReLU - children 0 and 1, as the reference's ``connector``."""
    return nn.Sequential(nn.Linear(d_in, d_out), nn.ReLU())


def _make_classifier(d_in: int, d_hidden: int, n_classes: int, norm: str = "layer") -> nn.Sequential:
    """Linear, LayerNorm (BatchNorm1d for TFN_GB), ReLU, Linear - children 0..3, as the reference's ``classifier``."""
    mid = nn.LayerNorm(d_hidden) if norm == "layer" else nn.BatchNorm1d(d_hidden)
    return nn.Sequential(nn.Linear(d_in, d_hidden), mid, nn.ReLU(), nn.Linear(d_hidden, n_classes))


class _NoHook:
    def remove(self):
        pass


class MultiModalModel(nn.Module):
    def __init__(self, n_classes: int, args_video: Dict, args_0D: Dict):
        super(MultiModalModel, self).__init__()
        self.n_classes = n_classes
        self.args_video = args_video
        self.args_0D = args_0D
        self.encoder_video = ViViTEncoder(**args_video)
        self.encoder_0D = TransformerEncoder(**args_0D)
        width = self.encoder_0D.feature_dims + self.encoder_video.dim          # concatenated latents
        self.connector = _make_connector(width, width // 2)
        self.classifier = _make_classifier(width // 2, width // 2, n_classes)

    def forward(self, x_video: torch.Tensor, x_0D: torch.Tensor):
        x = torch.cat([self.encoder_video(x_video), self.encoder_0D(x_0D)], axis=1)
        return _classifier(_connector(x, self.connector), self.classifier)

    def encode(self, x_vis: torch.Tensor, x_0D: torch.Tensor):
        with torch.no_grad():
            h_vis = self.encoder_video(x_vis)
            h_0D = self.encoder_0D(x_0D)
            h_concat = _connector(torch.cat([h_vis, h_0D], axis=1), self.connector)
        return (h_concat, h_vis, h_0D)

    def summary(self, *args, **kwargs):
        _param_table(self)


class MultiModalModel_GB(nn.Module):
    def __init__(self, n_classes: int, args_video: Dict, args_0D: Dict,
                 use_stream: Literal["video", "0D", "multi", "multi-GB"] = "multi-GB"):
        super(MultiModalModel_GB, self).__init__()
        self.n_classes = n_classes
        self.args_video = args_video
        self.args_0D = args_0D
        self.vis_model = ViViT(**args_video)
        self.ts_model = Transformer(**args_0D)
        width = self.ts_model.feature_dims + self.vis_model.dim
        self.connector = _make_connector(width, width // 2)
        self.classifier = _make_classifier(width // 2, width // 2, n_classes)
        self.vis_latent = None
        self.ts_latent = None
        self.vis_hook = _NoHook()
        self.ts_hook = _NoHook()
        self.update_use_stream(use_stream)

    def remove_my_hooks(self):
        self.vis_hook.remove()
        self.ts_hook.remove()

    def update_use_stream(self, use_stream: Literal["video", "0D", "multi", "multi-GB"]):
        # the reference flips only the top-level .training flags of the three children (:83-93, :106-119)
        self.use_stream = use_stream
        ts_on, vis_on, cls_on = {"video": (False, True, False), "0D": (True, False, False)}.get(use_stream, (True, True, True))
        self.ts_model.training, self.vis_model.training, self.classifier.training = ts_on, vis_on, cls_on

    def _both(self, x_vis, x_ts):
        vis_latent = self.vis_model._encode(x_vis)
        ts_latent = self.ts_model.encoder(x_ts)
        self.vis_latent = (vis_latent.detach(),)       # values only: a kept graph would outlive the step (src/utils/graphed.py)
        self.ts_latent = (ts_latent.detach(),)
        return vis_latent, ts_latent

    def forward(self, x_vis: torch.Tensor, x_ts: torch.Tensor):
        return self.forward_stream(x_vis, x_ts)

    def forward_stream(self, x_vis: torch.Tensor, x_ts: torch.Tensor):
        if self.use_stream == "video":
            return self.vis_model(x_vis)
        elif self.use_stream == "0D":
            return self.ts_model(x_ts)
        vis_latent, ts_latent = self._both(x_vis, x_ts)
        out_vis = self.vis_model._head(vis_latent)
        out_ts = self.ts_model._head(ts_latent)
        x = _connector(torch.cat([vis_latent, ts_latent], axis=1), self.connector)
        out_multi = _classifier(x, self.classifier)
        return out_multi if self.use_stream == 'multi' else (out_multi, out_vis, out_ts)

    def encode(self, x_vis: torch.Tensor, x_0D: torch.Tensor):
        with torch.no_grad():
            vis_latent, ts_latent = self._both(x_vis, x_0D)
            x = _connector(torch.cat([vis_latent, ts_latent], axis=1), self.connector)
        return (x, vis_latent, ts_latent)

    def summary(self, *args, **kwargs):
        _param_table(self)


class TFN(nn.Module):
    def __init__(self, n_classes: int, args_video: Dict, args_0D: Dict):
        super(TFN, self).__init__()
        self.n_classes = n_classes
        self.args_video = args_video
        self.args_0D = args_0D

        # the reference caps both latent widths at 128, in the callers' dicts (:181-185)
        if args_video['dim'] > 128:
            args_video['dim'] = 128
        if args_0D['feature_dims'] > 128:
            args_0D['feature_dims'] = 128

        self.encoder_video = ViViTEncoder(**args_video)
        self.encoder_0D = TransformerEncoder(**args_0D)
        self.encoder_0D_dim = self.encoder_0D.feature_dims
        self.encoder_video_dim = self.encoder_video.dim
        assert self.encoder_0D_dim == self.encoder_video_dim, "two encoder should be the same latent dims"

        self.fusion_input_dims = (self.encoder_0D_dim + 1) * (self.encoder_video_dim + 1)
        self.linear_input_dims = self.encoder_0D_dim + self.encoder_video_dim

        self.connector = _make_connector(self.fusion_input_dims, self.linear_input_dims)
        self.classifier = _make_classifier(self.linear_input_dims, self.linear_input_dims // 2, n_classes)

    def forward(self, x_vis: torch.Tensor, x_0D: torch.Tensor):
        fusion = OuterFusionFunction.apply(self.encoder_video(x_vis), self.encoder_0D(x_0D))
        return _classifier(_connector(fusion, self.connector), self.classifier)

    def encode(self, x_vis: torch.Tensor, x_0D: torch.Tensor):
        with torch.no_grad():
            h_vis = self.encoder_video(x_vis)
            h_0D = self.encoder_0D(x_0D)
            fusion = _connector(OuterFusionFunction.apply(h_vis, h_0D), self.connector)
        return (fusion, h_vis, h_0D)

    def summary(self, *args, **kwargs):
        _param_table(self)


class TFN_GB(nn.Module):
    def __init__(self, n_classes: int, args_video: Dict, args_0D: Dict):
        super(TFN_GB, self).__init__()
        self.n_classes = n_classes
        self.args_video = args_video
        self.args_0D = args_0D

        self.embedd_subnet = nn.ModuleDict({
            "network_video": ViViT(**args_video),
            "network_0D": Transformer(**args_0D)
        })
        self.network_0D_dims = self.embedd_subnet['network_0D'].feature_dims
        self.network_video_dims = self.embedd_subnet['network_video'].dim
        assert self.network_0D_dims == self.network_video_dims, "two encoder should be the same latent dims"
        self.encoder_dims = self.network_video_dims
        self.fusion_input_dims = (self.network_0D_dims + 1) * (self.network_video_dims + 1)

        self.dropout = nn.Dropout(0)
        self.classifier = _make_classifier(self.fusion_input_dims, self.fusion_input_dims // 2, n_classes, norm="batch")
        self.h_vis = None
        self.h_0D = None
        self.vis_hook = _NoHook()
        self.ts_hook = _NoHook()

    def remove_my_hooks(self):
        self.vis_hook.remove()
        self.ts_hook.remove()

    def forward(self, x_vis: torch.Tensor, x_0D: torch.Tensor):
        vis, ts = self.embedd_subnet['network_video'], self.embedd_subnet['network_0D']
        h_vis = vis._encode(x_vis)
        h_0D = ts.encoder(x_0D)
        self.h_vis = (h_vis.detach(),)                 # values only (see vis_latent)
        self.h_0D = (h_0D.detach(),)
        out_vis = vis._head(h_vis)
        out_0D = ts._head(h_0D)
        fusion = dropout(OuterFusionFunction.apply(h_vis, h_0D), self.dropout.p, self.dropout.training)
        bn = self.classifier[1]
        h = linear_bn_leaky(fusion, self.classifier[0], bn, 0.0, bn.training)
        return (_lin(h, self.classifier[3]), out_vis, out_0D)

    def encode(self, x_vis: torch.Tensor, x_0D: torch.Tensor):
        with torch.no_grad():
            latent_vis = self.embedd_subnet['network_video'].encode(x_vis)
            latent_0D = self.embedd_subnet['network_0D'].encode(x_0D)
            fusion = OuterFusionFunction.apply(latent_vis, latent_0D)
        return (fusion, latent_vis, latent_0D)

    def summary(self, *args, **kwargs):
        _param_table(self)
