"""CPU restatement (test infrastructure only) of the reference's fusion wrappers (src/models/MultiModal.py): MultiModalModel
(:33-39), MultiModalModel_GB.forward_stream (:131-149), TFN (:213-224) and TFN_GB (:294-311) as functions of a state dict with the
reference's keys, on top of oracle.vivit and oracle.transformer0d (noise std 0, dropout 0).  Pinned by tests/golden/multimodal.npz."""
import torch
import torch.nn.functional as F

from . import transformer0d as ot
from . import vivit as ov

VIDEO = dict(patch_size=8, depth=1, n_heads=2)        # the fixture's encoder geometry
TS = dict(n_layers=1, n_heads=2, kernel_size=3)


def _sub(sd, prefix, new_prefix=""):
    return {new_prefix + k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def _classifier(x, sd, p="classifier."):
    h = F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])
    h = F.relu(F.layer_norm(h, (h.shape[1],), sd[p + "1.weight"], sd[p + "1.bias"], 1e-5))
    return F.linear(h, sd[p + "3.weight"], sd[p + "3.bias"])


def _latents(x_vis, x_ts, sd, pv, pt, pool, training):
    sv = _sub(sd, pv)
    st = _sub(sd, pt, "encoder.")
    h_vis = ov.vivit_forward(x_vis, sv, VIDEO["patch_size"], VIDEO["depth"], VIDEO["n_heads"], pool, 3, with_mlp=False)
    h_ts = ot.transformer0d_forward(x_ts, st, TS["n_layers"], TS["n_heads"], TS["kernel_size"], training, with_classifier=False)
    for k, v in st.items():                              # running statistics of the 0D filter move in training mode
        if "running" in k or "num_batches" in k:
            sd[pt + k[len("encoder."):]] = v
    return h_vis, h_ts


def _fusion(h_vis, h_ts):                                                                             # :214-220
    one = torch.ones(h_vis.shape[0], 1)
    return torch.bmm(torch.cat((one, h_vis), 1).unsqueeze(2), torch.cat((one, h_ts), 1).unsqueeze(1)).reshape(h_vis.shape[0], -1)


def multimodal_forward(x_vis, x_ts, sd, pool="mean", training=True):
    h_vis, h_ts = _latents(x_vis, x_ts, sd, "encoder_video.", "encoder_0D.", pool, training)
    x = F.relu(F.linear(torch.cat([h_vis, h_ts], 1), sd["connector.0.weight"], sd["connector.0.bias"]))
    return _classifier(x, sd)


def multimodal_gb_forward(x_vis, x_ts, sd, pool="cls", alpha=1.0, training=True):
    h_vis, h_ts = _latents(x_vis, x_ts, sd, "vis_model.", "ts_model.encoder.", pool, training)
    out_vis = ov.vivit_head(h_vis, _sub(sd, "vis_model."), alpha)
    out_ts = ot.classifier_head(h_ts, _sub(sd, "ts_model."))
    x = F.relu(F.linear(torch.cat([h_vis, h_ts], 1), sd["connector.0.weight"], sd["connector.0.bias"]))
    return _classifier(x, sd), out_vis, out_ts


def tfn_forward(x_vis, x_ts, sd, pool="mean", training=True):
    h_vis, h_ts = _latents(x_vis, x_ts, sd, "encoder_video.", "encoder_0D.", pool, training)
    x = F.relu(F.linear(_fusion(h_vis, h_ts), sd["connector.0.weight"], sd["connector.0.bias"]))
    return _classifier(x, sd)


def tfn_gb_forward(x_vis, x_ts, sd, pool="cls", alpha=1.0, training=True):
    pv, pt = "embedd_subnet.network_video.", "embedd_subnet.network_0D."
    h_vis, h_ts = _latents(x_vis, x_ts, sd, pv, pt + "encoder.", pool, training)
    out_vis = ov.vivit_head(h_vis, _sub(sd, pv), alpha)
    out_ts = ot.classifier_head(h_ts, _sub(sd, pt))
    h = F.linear(_fusion(h_vis, h_ts), sd["classifier.0.weight"], sd["classifier.0.bias"])
    h = F.relu(F.batch_norm(h, sd["classifier.1.running_mean"], sd["classifier.1.running_var"], sd["classifier.1.weight"],
                            sd["classifier.1.bias"], training, 0.1, 1e-5))
    return F.linear(h, sd["classifier.3.weight"], sd["classifier.3.bias"]), out_vis, out_ts
