"""CPU restatement (test infrastructure only) of the DERIVED fusion models for BASELINE configs 4 and 5: the recipe of
the reference's MultiModalModel_GB (src/models/MultiModal.py:65-77, 131-149) applied to encoder pairs the reference never wires
- R2Plus1DClassifier + Transformer (cfg4) and SlowFast + MLSTM_FCN (cfg5).  The vision latent is the input of the vision head's
first Linear (R2Plus1DClassifier.linear[0], 128-d; SlowFast.classifier.classifier[0], 640-d), the 0D latent the input of
classifier[0]; both go through connector (Linear, ReLU) and classifier (Linear, LayerNorm, ReLU, Linear).  Composed from the
per-model restatements.  Pinned by tests/golden/fusion_derived.npz, which is recorded from the reference's own model classes
combined by forward hooks exactly as MultiModal.py:96-97 does ("derived", not reference-verbatim)."""
import torch
import torch.nn.functional as F

from . import mlstm_fcn as om
from . import r2plus1d as orc
from . import slowfast as osf
from . import transformer0d as ot


def _sub(sd, prefix, new_prefix=""):
    return {new_prefix + k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def _put_back(sd, sub, prefix, strip=""):
    for k, v in sub.items():
        if "running" in k or "num_batches" in k:
            sd[prefix + k[len(strip):]] = v


def fusion_state(shapes, seed: int):
    """oracle.slowfast.synth_state over every key but the sinusoidal positional table (a constant buffer)."""
    return osf.synth_state({k: v for k, v in shapes.items() if not k.endswith("pos_enc.pe")}, seed)


def _fused_head(h_vis, h_ts, sd):
    x = F.relu(F.linear(torch.cat([h_vis, h_ts], 1), sd["connector.0.weight"], sd["connector.0.bias"]))
    h = F.linear(x, sd["classifier.0.weight"], sd["classifier.0.bias"])
    h = F.relu(F.layer_norm(h, (h.shape[1],), sd["classifier.1.weight"], sd["classifier.1.bias"], 1e-5))
    return F.linear(h, sd["classifier.3.weight"], sd["classifier.3.bias"])


def r2p1d_transformer_forward(x_vis, x_ts, sd, layer_sizes, alpha, n_layers, n_heads, kernel_size, training=True):
    """(out_multi, out_vis, out_ts); running statistics in `sd` move as in training mode."""
    sv = _sub(sd, "vis_model.")
    st = _sub(sd, "ts_model.")
    h_vis = orc.trunk_forward(x_vis, sv, sv, layer_sizes, alpha, training)
    out_vis = orc.head_forward(h_vis, sv, sv, alpha, training)
    h_ts = ot.transformer0d_forward(x_ts, st, n_layers, n_heads, kernel_size, training, with_classifier=False)
    out_ts = ot.classifier_head(h_ts, st)
    _put_back(sd, sv, "vis_model."); _put_back(sd, st, "ts_model.")
    return _fused_head(h_vis, h_ts, sd), out_vis, out_ts


def slowfast_mlstm_forward(x_vis, x_ts, sd, layers, alpha, alpha_elu, mlstm_cfg, training=True):
    sv = _sub(sd, "vis_model.")
    st = _sub(sd, "ts_model.")
    h_vis = osf.slowfast_forward(x_vis, sv, layers, alpha, 1, alpha_elu, training, return_latent=True)
    out_vis = osf.slowfast_head(h_vis, sv, alpha_elu, training)
    h_ts = om.mlstm_fcn_forward(x_ts, st, training=training, return_latent=True, **mlstm_cfg)
    out_ts = om.mlstm_fcn_head(h_ts, st, mlstm_cfg["alpha"], training)
    _put_back(sd, sv, "vis_model."); _put_back(sd, st, "ts_model.")
    return _fused_head(h_vis, h_ts, sd), out_vis, out_ts
