"""MI355X-native mirror of the reference's ``src/models/CnnLSTM.py`` (same class, constructor arguments, child-module
names and state-dict keys).

forward (reference :89-103): NoiseLayer -> Conv1d(+bias) -> Conv1d(+bias) -> BatchNorm1d -> ReLU over (B, F, T) -> LSTM over
the conv-channel axis (bidirectional, zero initial state) -> attention pooling -> Linear, BatchNorm1d, ReLU, Linear.
Kernels: the convolutions run as (k,1,1) units of the conv kernels (the second one fused with its BatchNorm + ReLU, its bias
folded into the running mean), the LSTM on ``md_lstm_*``, the classifier on ``md_head_*`` (ELU slope 0 = ReLU).

Attention pooling.  The reference computes ``att = softmax(w_s2(tanh(w_s1(out))), dim=2)`` and
``hidden = bmm(att^T, out).mean(dim=1)`` (:76-79, :97-99).  The softmax normalises over the SAME axis the mean is then taken
over, so ``mean_h att[b,s,h] = 1/lstm_dim`` for every step and ``hidden[b,:] = sum_s out[b,s,:] / lstm_dim`` exactly: ``w_s1`` /
``w_s2`` cannot influence the output and their true gradients are zero (the reference's autograd returns round-off noise for
them).  This mirror evaluates that closed form (``md_seq_sum_*``) and gives ``w_s1`` / ``w_s2`` exact zero gradients.
"""
import torch
import torch.nn as nn

from .. import _native as N
from .. import ops
from .NoiseLayer import NoiseLayer
from ._unit import bump_batches_tracked
from ._unit import (ConvBnLeakyFunction, ConvFunction, HeadFunction, head_apply, lstm_forward, _AbsorbedBias, _ChannelBias,
                    _SeqSum)


class CnnLSTM(nn.Module):
    def __init__(self, seq_len: int = 21, n_features: int = 10, conv_dim: int = 32, conv_kernel: int = 3, conv_stride: int = 1,
                 conv_padding: int = 1, lstm_dim: int = 64, n_layers: int = 1, bidirectional: bool = True, n_classes: int = 2):
        super(CnnLSTM, self).__init__()
        self.n_features = n_features
        self.seq_len = seq_len
        self.conv_dim = conv_dim
        self.conv_kernel = conv_kernel
        self.conv_stride = conv_stride
        self.conv_padding = conv_padding
        self.lstm_dim = lstm_dim
        self.n_layers = n_layers
        self.bidirectional = bidirectional
        self.n_classes = n_classes
        self.noise = NoiseLayer(mean=0, std=1e-3)
        self.conv = nn.Sequential(
            nn.Conv1d(in_channels=n_features, out_channels=conv_dim, kernel_size=conv_kernel, stride=conv_stride, padding=conv_padding),
            nn.Conv1d(in_channels=conv_dim, out_channels=conv_dim, kernel_size=conv_kernel, stride=conv_stride, padding=conv_padding),
            nn.BatchNorm1d(conv_dim),
            nn.ReLU(),
        )
        lstm_input_dim = self.compute_conv1d_output_dim(
            self.compute_conv1d_output_dim(seq_len, conv_kernel, conv_stride, conv_padding, 1), conv_kernel, conv_stride, conv_padding, 1)
        self.lstm = nn.LSTM(lstm_input_dim, lstm_dim, bidirectional=bidirectional, batch_first=False, num_layers=n_layers)
        if bidirectional:
            self.w_s1 = nn.Linear(lstm_dim * 2, lstm_dim)
            linear_input_dims = lstm_dim * 2
        else:
            self.w_s1 = nn.Linear(lstm_dim, lstm_dim)
            linear_input_dims = lstm_dim
        self.w_s2 = nn.Linear(lstm_dim, lstm_dim)
        self.classifier = nn.Sequential(
            nn.Linear(linear_input_dims, linear_input_dims // 2),
            nn.BatchNorm1d(linear_input_dims // 2),
            nn.ReLU(),
            nn.Linear(linear_input_dims // 2, n_classes)
        )

    def compute_conv1d_output_dim(self, input_dim: int, kernel_size: int = 3, stride: int = 1, padding: int = 1, dilation: int = 1):
        return int((input_dim + 2 * padding - dilation * (kernel_size - 1) - 1) / stride + 1)

    def attention(self, lstm_output: torch.Tensor):
        raise NotImplementedError("mi355x hot path: the attention weights cancel in the pooled output (see the module docstring)")

    # -- pieces
    def _conv_stack(self, x_bft):
        """(B, F, T) -> (B, conv_dim, T'') : Conv1d + bias, then Conv1d (+bias) + BatchNorm1d + ReLU."""
        c1, c2, bn = self.conv[0], self.conv[1], self.conv[2]
        x5 = x_bft.contiguous()[:, :, :, None, None]
        y = ConvFunction.apply(x5, c1.weight[:, :, :, None, None], (c1.stride[0], 1, 1), (c1.padding[0], 0, 0))
        y = _ChannelBias.apply(y.squeeze(4).squeeze(3), c1.bias)
        b = c2.bias.detach()
        rmean = bn.running_mean if self.training else bn.running_mean - b
        z = ConvBnLeakyFunction.apply(y[:, :, :, None, None], c2.weight[:, :, :, None, None], bn.weight, bn.bias, rmean,
                                      bn.running_var, (c2.stride[0], 1, 1), (c2.padding[0], 0, 0), 0.0, bool(self.training),
                                      float(bn.eps), float(bn.momentum))
        if self.training:
            bn.running_mean.add_(b * bn.momentum)
            bump_batches_tracked(bn)
            z = _AbsorbedBias.apply(z, c2.bias)
        return z.squeeze(4).squeeze(3)

    def _hidden(self, x):
        x = self.noise(x)
        x_conv = self._conv_stack(x.permute(0, 2, 1))
        lstm_output = lstm_forward(x_conv.permute(1, 0, 2).contiguous(), self.lstm)          # (conv_dim, B, dirs*H)
        lstm_output = lstm_output.permute(1, 0, 2).contiguous()                               # (B, conv_dim, dirs*H)
        return _SeqSum.apply(lstm_output, 1.0 / self.lstm_dim, self.w_s1.weight, self.w_s1.bias, self.w_s2.weight, self.w_s2.bias)

    def encode(self, x: torch.Tensor):
        with torch.no_grad():
            hidden = self._hidden(x)
            return hidden.view(hidden.size()[0], -1)

    def forward(self, x: torch.Tensor):
        hidden = self._hidden(x)
        lin0, bn, lin1 = self.classifier[0], self.classifier[1], self.classifier[3]
        return head_apply(hidden, lin0, bn, lin1, 0.0, bool(self.training))

    def summary(self, device: str = 'cpu', show_input: bool = True, show_hierarchical: bool = False, print_summary: bool = False,
                show_parent_layers: bool = False):
        rows = ["%-40s %-20s %d" % (k, tuple(v.shape), v.numel()) for k, v in self.named_parameters()]
        text = "\n".join(rows + ["total parameters: %d" % sum(p.numel() for p in self.parameters())])
        print(text)
        return None
