"""CPU restatement (test infrastructure only) of the reference's Transformer.forward (src/models/transformer.py:88-105, 142-147)
as a function of a state dict with the reference's keys (input already passed through the noise layer; dropout 0).  The
nn.TransformerEncoderLayer arithmetic (post-norm, additive mask) is written out.  Pinned by tests/golden/transformer0d.npz."""
import math

import torch
import torch.nn.functional as F


def _gelu_tanh(x):
    return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))       # transformer.py:36-37


def positional_table(max_len: int, d_model: int):
    """PositionalEncoding.pe (transformer.py:10-28): (max_len, 1, d_model), sin on even and cos on odd features."""
    pe = torch.zeros(max_len, d_model)
    pos = torch.arange(0, max_len).float().unsqueeze(1)
    div = (torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model)).exp()
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)[:, : d_model // 2]
    return pe.unsqueeze(1)


def transformer0d_forward(x, sd, n_layers: int, n_heads: int, kernel_size: int, training: bool = True, with_classifier: bool = True):
    e = "encoder."
    pad = (kernel_size - 1) // 2
    y = F.conv1d(x.permute(0, 2, 1), sd[e + "filter.0.weight"], sd[e + "filter.0.bias"], 1, pad)        # :64
    y = F.conv1d(y, sd[e + "filter.1.weight"], sd[e + "filter.1.bias"], 1, pad)                          # :65
    y = F.relu(F.batch_norm(y, sd[e + "filter.2.running_mean"], sd[e + "filter.2.running_var"], sd[e + "filter.2.weight"],
                            sd[e + "filter.2.bias"], training, 0.1, 1e-5))                               # :66-67
    h = y.permute(2, 0, 1)                                                                                # (T, B, D)  :94-97
    S, B, D = h.shape
    mask = torch.triu(torch.full((S, S), float("-inf")), diagonal=1)                                     # :107-110
    h = h + sd[e + "pos_enc.pe"][:S]                                                                      # :32
    dh = D // n_heads
    for l in range(n_layers):
        p = f"{e}transformer_encoder.layers.{l}."
        qkv = F.linear(h, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
        q, k, v = qkv.chunk(3, dim=2)
        def heads(t):
            return t.reshape(S, B * n_heads, dh).transpose(0, 1)                                          # (B*H, S, dh)
        att = torch.softmax(heads(q) @ heads(k).transpose(1, 2) / math.sqrt(dh) + mask, dim=2) @ heads(v)
        att = att.transpose(0, 1).reshape(S, B, D)
        att = F.linear(att, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(h + att, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        ff = F.linear(_gelu_tanh(F.linear(h, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        h = F.layer_norm(h + ff, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    z = h.permute(1, 0, 2).mean(dim=1)                                                                    # :103
    z = F.gelu(F.layer_norm(F.linear(z, sd[e + "connector.0.weight"], sd[e + "connector.0.bias"]), (D,), sd[e + "connector.1.weight"],
                            sd[e + "connector.1.bias"], 1e-5))                                            # :83-87
    return classifier_head(z, sd) if with_classifier else z


def classifier_head(z, sd):                                                                               # :132-137
    c = F.linear(z, sd["classifier.0.weight"], sd["classifier.0.bias"])
    c = _gelu_tanh(F.layer_norm(c, (c.shape[1],), sd["classifier.1.weight"], sd["classifier.1.bias"], 1e-5))
    return F.linear(c, sd["classifier.3.weight"], sd["classifier.3.bias"])
