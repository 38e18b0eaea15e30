"""CPU restatement (test infrastructure only) of the reference's Bottleneck3D.forward (src/models/resnet.py:170-200) as a
function of a state dict with the reference's keys.  Pinned by tests/golden/bottleneck3d_*.npz (reference outputs)."""
import torch
import torch.nn.functional as F


def _bn(x, sd, name, training, momentum=0.1, eps=1e-5):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        training, momentum, eps)


def bottleneck3d_forward(x, sd, stride: int, head_conv: int, index: int, training: bool = True):
    """sd: tensors keyed like Bottleneck3D.state_dict() (running stats are updated in place when training)."""
    residual = x
    pad1 = (1, 0, 0) if head_conv == 3 else (0, 0, 0)
    out = F.conv3d(x, sd["conv1.weight"], None, 1, pad1)                                   # :173
    out = F.relu(_bn(out, sd, "bn1", training))                                            # :174-175
    out = F.conv3d(out, sd["conv2.weight"], None, (1, stride, stride), (0, 1, 1))          # :177
    out = F.relu(_bn(out, sd, "bn2", training))                                            # :178-179
    if index % 2 == 0:
        se = out.mean(dim=(2, 3, 4), keepdim=True)                                         # :182
        se = F.relu(F.conv3d(se, sd["fc1.weight"], sd["fc1.bias"]))                        # :183-184
        se = torch.sigmoid(F.conv3d(se, sd["fc2.weight"], sd["fc2.bias"]))                 # :185-186
        out = out * se                                                                     # :187
    out = out * torch.sigmoid(out)                                                         # :189 (SwishEfficient)
    out = _bn(F.conv3d(out, sd["conv3.weight"], None), sd, "bn3", training)                # :190-191
    if "downsample.0.weight" in sd:
        residual = F.conv3d(x, sd["downsample.0.weight"], None, (1, stride, stride))       # :193-194, _make_layer :254-257
        residual = _bn(residual, sd, "downsample.1", training)
    return F.relu(out + residual)                                                          # :196-197
