"""CPU restatement (test infrastructure only) of the reference's SlowFast forward (src/models/slowfast.py:11-196 over
src/models/resnet.py:202-273) as a function of a state dict with the reference's keys.  Pinned by
tests/golden/slowfast_tiny.npz (reference outputs)."""
import torch
import torch.nn.functional as F

from .bottleneck3d import bottleneck3d_forward


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}          # same tensor objects: running stats update in place


def _stem(x, sd, p, training):
    x = F.conv3d(x, sd[p + "layer0.0.weight"], sd[p + "layer0.0.bias"], (1, 2, 2), (0, 3, 3))              # resnet.py:222
    x = F.relu(F.batch_norm(x, sd[p + "layer0.1.running_mean"], sd[p + "layer0.1.running_var"], sd[p + "layer0.1.weight"],
                            sd[p + "layer0.1.bias"], training, 0.1, 1e-5))                                    # :223-224
    return F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))                                                   # :225


def _layer(x, sd, p, nblocks, stride, head_conv, training):
    for i in range(nblocks):
        x = bottleneck3d_forward(x, _sub(sd, f"{p}{i}."), stride if i == 0 else 1, head_conv, 0, training)   # :259-264 (index 0)
    return x


def slowfast_forward(x, sd, layers, alpha: int = 4, tau_fast: int = 1, alpha_elu: float = 1.0, training: bool = True,
                     return_latent: bool = False):
    xs, xf = x[:, :, ::tau_fast * alpha], x[:, :, ::tau_fast]                                                 # slowfast.py:120-128
    pf, ps = "encoder.fastnet.", "encoder.slownet."
    lat = []
    f = _stem(xf, sd, pf, training)
    lat.append(F.conv3d(f, sd[pf + "l_maxpool.weight"], None, (alpha, 1, 1), (1, 0, 0)))                     # :71-72
    f = _layer(f, sd, pf + "layer1.", layers[0], 1, 3, training); lat.append(F.conv3d(f, sd[pf + "l_layer1.weight"], None, (alpha, 1, 1), (1, 0, 0)))
    f = _layer(f, sd, pf + "layer2.", layers[1], 2, 3, training); lat.append(F.conv3d(f, sd[pf + "l_layer2.weight"], None, (alpha, 1, 1), (1, 0, 0)))
    f = _layer(f, sd, pf + "layer3.", layers[2], 2, 3, training); lat.append(F.conv3d(f, sd[pf + "l_layer3.weight"], None, (alpha, 1, 1), (1, 0, 0)))
    f = _layer(f, sd, pf + "layer4.", layers[3], 2, 3, training)
    f = F.adaptive_avg_pool3d(f, 1).view(-1, f.size(1))                                                       # :86-87
    s = _stem(xs, sd, ps, training)
    s = _layer(torch.cat([s, lat[0]], 1), sd, ps + "layer1.", layers[0], 1, 1, training)                      # :21-22
    s = _layer(torch.cat([s, lat[1]], 1), sd, ps + "layer2.", layers[1], 2, 1, training)
    s = _layer(torch.cat([s, lat[2]], 1), sd, ps + "layer3.", layers[2], 2, 3, training)
    s = _layer(torch.cat([s, lat[3]], 1), sd, ps + "layer4.", layers[3], 2, 3, training)
    s = F.adaptive_avg_pool3d(s, 1).view(-1, s.size(1))                                                       # :33-34
    feat = torch.cat([s, f], dim=1)                                                                           # :134
    if return_latent:
        return feat
    return slowfast_head(feat, sd, alpha_elu, training)


def slowfast_head(feat, sd, alpha_elu: float = 1.0, training: bool = True):
    c = "classifier.classifier."
    h = F.linear(feat, sd[c + "0.weight"], sd[c + "0.bias"])                                                  # :157
    h = F.elu(F.batch_norm(h, sd[c + "1.running_mean"], sd[c + "1.running_var"], sd[c + "1.weight"], sd[c + "1.bias"],
                           training, 0.1, 1e-5), alpha_elu)                                                   # :158-159
    return F.linear(h, sd[c + "3.weight"], sd[c + "3.bias"])                                                  # :160


def synth_state(shapes, seed: int):
    """Deterministic NumPy recipe for a SlowFast state dict (so fixtures store a seed, not megabytes of weights):
    `shapes` = {key: shape} in the module's own key order.  Convolution / linear weights ~ N(0, 2/fan_in), 1-D `.weight`
    (normalisation scales) ~ U(0.5, 1.5), biases ~ N(0, 0.3^2), running_mean 0, running_var 1, counters 0."""
    import numpy as np
    rng = np.random.default_rng(seed)
    sd = {}
    for k, shp in shapes.items():
        shp = tuple(shp)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros(shp)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(shp)
        elif len(shp) == 1 and k.endswith(".weight"):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype("float32"))
        elif len(shp) == 1:
            sd[k] = torch.from_numpy((rng.standard_normal(shp) * 0.3).astype("float32"))
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            sd[k] = torch.from_numpy((rng.standard_normal(shp) * (2.0 / fan_in) ** 0.5).astype("float32"))
    return sd


def synth_clip(B, T, S, seed):
    import numpy as np
    return torch.from_numpy(np.random.default_rng(seed).standard_normal((B, 3, T, S, S)).astype("float32"))
