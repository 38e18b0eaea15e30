"""CPU restatement (test infrastructure only) of the reference's CnnLSTM.forward (src/models/CnnLSTM.py:89-103) as a function
of a state dict with the reference's keys (noise layer in eval position: callers pass the already-noised input).  Pinned by
tests/golden/cnnlstm.npz (reference outputs)."""
import torch
import torch.nn.functional as F


def _lstm_dir(x, w_ih, w_hh, b_ih, b_hh, reverse):
    S, B, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H); c = x.new_zeros(B, H)
    outs = [None] * S
    for step in range(S):
        t = S - 1 - step if reverse else step
        g = x[t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh               # nn.LSTM, gate order i, f, g, o
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, 0)


def cnnlstm_forward(x, sd, lstm_dim: int, n_layers: int, bidirectional: bool, training: bool = True):
    """x: (B, T, F) after the noise layer."""
    y = F.conv1d(x.permute(0, 2, 1), sd["conv.0.weight"], sd["conv.0.bias"], 1, 1)                           # :42
    y = F.conv1d(y, sd["conv.1.weight"], sd["conv.1.bias"], 1, 1)                                            # :43
    y = F.relu(F.batch_norm(y, sd["conv.2.running_mean"], sd["conv.2.running_var"], sd["conv.2.weight"], sd["conv.2.bias"],
                            training, 0.1, 1e-5))                                                            # :44-45
    out = y.permute(1, 0, 2)                                                                                 # :96 (seq = conv channels)
    for layer in range(n_layers):
        dirs = []
        for rev in range(2 if bidirectional else 1):
            s = f"_l{layer}" + ("_reverse" if rev else "")
            dirs.append(_lstm_dir(out, sd["lstm.weight_ih" + s], sd["lstm.weight_hh" + s], sd["lstm.bias_ih" + s],
                                  sd["lstm.bias_hh" + s], bool(rev)))
        out = torch.cat(dirs, dim=2)
    lo = out.permute(1, 0, 2)                                                                                # :97
    att = F.softmax(F.linear(torch.tanh(F.linear(lo, sd["w_s1.weight"], sd["w_s1.bias"])), sd["w_s2.weight"], sd["w_s2.bias"]), dim=2)  # :76-78
    hidden = torch.bmm(att.permute(0, 2, 1), lo).mean(dim=1)                                                 # :99
    h = F.linear(hidden, sd["classifier.0.weight"], sd["classifier.0.bias"])
    h = F.relu(F.batch_norm(h, sd["classifier.1.running_mean"], sd["classifier.1.running_var"], sd["classifier.1.weight"],
                            sd["classifier.1.bias"], training, 0.1, 1e-5))
    return F.linear(h, sd["classifier.3.weight"], sd["classifier.3.bias"])                                  # :101
