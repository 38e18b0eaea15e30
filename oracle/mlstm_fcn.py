"""CPU restatement (test infrastructure only) of the reference's MLSTM_FCN.forward (src/models/MLSTM_FCN.py:122-139) as a
function of a state dict with the reference's keys (input already passed through the noise layer; LSTM inter-layer dropout
0).  Pinned by tests/golden/mlstm_fcn.npz (reference outputs)."""
import torch
import torch.nn.functional as F

from .cnnlstm import _lstm_dir


def mlstm_fcn_forward(x, sd, kernel_size: int, stride: int, lstm_n_layers: int, bidirectional: bool, alpha: float,
                      training: bool = True, return_latent: bool = False):
    # RNN branch (:54-68)
    out = x.permute(1, 0, 2)
    for layer in range(lstm_n_layers):
        dirs = []
        for rev in range(2 if bidirectional else 1):
            s = f"_l{layer}" + ("_reverse" if rev else "")
            dirs.append(_lstm_dir(out, sd["rnn.lstm.weight_ih" + s], sd["rnn.lstm.weight_hh" + s], sd["rnn.lstm.bias_ih" + s],
                                  sd["rnn.lstm.bias_hh" + s], bool(rev)))
        out = torch.cat(dirs, dim=2)
    lo = out.permute(1, 0, 2)
    att = F.softmax(F.linear(torch.tanh(F.linear(lo, sd["rnn.w_s1.weight"], sd["rnn.w_s1.bias"])), sd["rnn.w_s2.weight"],
                             sd["rnn.w_s2.bias"]), dim=2)
    x_rnn = torch.bmm(att.permute(0, 2, 1), lo).mean(dim=1)
    # FCN branch (:17-45, :133-134)
    y = x.permute(0, 2, 1)
    for blk, se in ((0, 1), (2, 3)):
        y = F.conv1d(y, sd[f"fcn.{blk}.conv.weight"], sd[f"fcn.{blk}.conv.bias"], stride)
        y = F.leaky_relu(F.batch_norm(y, sd[f"fcn.{blk}.bn.running_mean"], sd[f"fcn.{blk}.bn.running_var"], sd[f"fcn.{blk}.bn.weight"],
                                      sd[f"fcn.{blk}.bn.bias"], training, 0.1, 1e-5), alpha)
        g = torch.sigmoid(F.linear(F.relu(F.linear(y.mean(dim=2), sd[f"fcn.{se}.fc.0.weight"])), sd[f"fcn.{se}.fc.2.weight"]))
        y = y * g[:, :, None]
    x_fcn = y.mean(dim=2)
    f = F.linear(torch.cat([x_rnn, x_fcn], dim=1), sd["converter.weight"], sd["converter.bias"])              # :137-138
    return f if return_latent else mlstm_fcn_head(f, sd, alpha, training)


def mlstm_fcn_head(f, sd, alpha: float, training: bool = True):
    h = F.linear(f, sd["classifier.0.weight"], sd["classifier.0.bias"])
    h = F.leaky_relu(F.batch_norm(h, sd["classifier.1.running_mean"], sd["classifier.1.running_var"], sd["classifier.1.weight"],
                                  sd["classifier.1.bias"], training, 0.1, 1e-5), alpha)
    return F.linear(h, sd["classifier.3.weight"], sd["classifier.3.bias"])                                    # :139
