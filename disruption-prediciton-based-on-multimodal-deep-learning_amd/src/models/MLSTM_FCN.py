"""MI355X-native mirror of the reference's ``src/models/MLSTM_FCN.py`` (same classes, constructor arguments, child-module
names and state-dict keys: ``fcn.{0,2}.{conv,bn}``, ``fcn.{1,3}.fc.{0,2}``, ``rnn.lstm``, ``rnn.w_s1/w_s2``, ``converter``,
``classifier``).

forward (reference :122-139): NoiseLayer; RNN branch = (bi)LSTM + attention pooling; FCN branch = [Conv1d (no padding, bias)
+ BatchNorm1d + LeakyReLU, squeeze-excitation] x 2, mean over time; concat -> Linear -> Linear, BatchNorm1d, LeakyReLU,
Linear.  Kernels: ``md_lstm_*``; (k,1,1) conv units with the bias folded into the running mean; ``md_se_scale_*`` (gate network
with bias-free Linears); ``md_rowmean_*``; ``linear`` = 1x1x1 convolution + ``md_channel_bias_*``; head on ``md_head_*`` with the
LeakyReLU activation.  The attention pooling evaluates to ``sum_s lstm_out[b,s,:] / hidden_dim`` exactly (the softmax and the
mean run over the same axis; see models/CnnLSTM.py), so ``w_s1`` / ``w_s2`` get exact zero gradients.
"""
import torch
import torch.nn as nn

from .. import _native as N
from .. import ops
from .NoiseLayer import NoiseLayer
from ._unit import GlobalAvgPoolFunction, HeadFunction, head_apply, _SeqSum, conv1d_bn_leaky, linear, lstm_forward


class _SEScaleFunction(torch.autograd.Function):
    """x (B, C, T) * sigmoid(W2 relu(W1 mean_t x))   (SqueezeExciteBlock.forward, MLSTM_FCN.py:29-33)."""

    @staticmethod
    def forward(ctx, a, w1, w2):
        ops.require_cuda(a.contiguous(), w1, w2)
        a = ops.f32(a).contiguous()
        B, Cc, T = a.shape
        Wd = w1.shape[0]
        dev = a.device
        z1 = torch.zeros(Wd, device=dev); z2 = torch.zeros(Cc, device=dev)
        pool = torch.empty(B, Cc, device=dev); hidden = torch.empty(B, Wd, device=dev); gate = torch.empty(B, Cc, device=dev)
        out = torch.empty_like(a)
        N.check(N.lib().md_se_scale_fwd(ops._p(a), B, Cc, T, Wd, ops._p(w1.contiguous()), ops._p(z1), ops._p(w2.contiguous()),
                                        ops._p(z2), ops._p(pool), ops._p(hidden), ops._p(gate), ops._p(out), ops._stream()),
                "md_se_scale_fwd")
        ctx.save_for_backward(a, w1, w2, pool, hidden, gate)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, w1, w2, pool, hidden, gate = ctx.saved_tensors
        B, Cc, T = a.shape
        Wd = w1.shape[0]
        dev = a.device
        g = ops.f32(dout).contiguous()
        da = torch.empty_like(a); dw1 = torch.empty_like(w1); dw2 = torch.empty_like(w2)
        db1 = torch.empty(Wd, device=dev); db2 = torch.empty(Cc, device=dev)
        scratch = torch.empty(3 * B * Cc + B * Wd, device=dev)
        N.check(N.lib().md_se_scale_bwd(ops._p(a), ops._p(g), B, Cc, T, Wd, ops._p(w1.contiguous()), ops._p(w2.contiguous()),
                                        ops._p(pool), ops._p(hidden), ops._p(gate), ops._p(da), ops._p(dw1), ops._p(db1),
                                        ops._p(dw2), ops._p(db2), ops._p(scratch), ops._stream()), "md_se_scale_bwd")
        return da, dw1, dw2


class SqueezeExciteBlock(nn.Module):
    def __init__(self, in_channels: int, reduction: int = 16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Sequential(
            nn.Linear(in_channels, in_channels // reduction, bias=False),
            nn.ReLU(inplace=True),
            nn.Linear(in_channels // reduction, in_channels, bias=False),
            nn.Sigmoid()
        )

    def forward(self, x: torch.Tensor):
        return _SEScaleFunction.apply(x, self.fc[0].weight, self.fc[2].weight)


class ConvBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int, alpha: float = 1.0):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride)
        self.bn = nn.BatchNorm1d(out_channels)
        self.relu = nn.LeakyReLU(alpha)

    def forward(self, x: torch.Tensor):
        return conv1d_bn_leaky(x, self.conv, self.bn, self.relu.negative_slope, self.training)


class SelfAttentionRnn(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int, n_layers: int, bidirectional: bool = True, dropout: float = 0.1):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.n_layers = n_layers
        self.bidirectional = bidirectional
        self.lstm = nn.LSTM(input_dim, hidden_dim, bidirectional=bidirectional, batch_first=False, num_layers=n_layers, dropout=dropout)
        if bidirectional:
            self.w_s1 = nn.Linear(hidden_dim * 2, hidden_dim)
            output_dim = hidden_dim * 2
        else:
            self.w_s1 = nn.Linear(hidden_dim, hidden_dim)
            output_dim = hidden_dim
        self.output_dim = output_dim
        self.w_s2 = nn.Linear(hidden_dim, hidden_dim)

    def attention(self, lstm_output: torch.Tensor):
        raise NotImplementedError("mi355x hot path: the attention weights cancel in the pooled output (see the module docstring)")

    def forward(self, x: torch.Tensor):
        lstm_output = lstm_forward(x.permute(1, 0, 2).contiguous(), self.lstm).permute(1, 0, 2).contiguous()     # (B, T, dirs*H)
        return _SeqSum.apply(lstm_output, 1.0 / self.hidden_dim, self.w_s1.weight, self.w_s1.bias, self.w_s2.weight, self.w_s2.bias)


class MLSTM_FCN(nn.Module):
    def __init__(self, n_features: int, fcn_dim: int, kernel_size: int, stride: int, seq_len: int, lstm_dim: int,
                 lstm_n_layers: int = 1, lstm_bidirectional: bool = True, lstm_dropout: float = 0.1, reduction: int = 16,
                 alpha: float = 1.0, n_classes: int = 2):
        super().__init__()
        self.n_features = n_features
        self.seq_len = seq_len
        self.fcn = nn.Sequential(
            ConvBlock(n_features, fcn_dim, kernel_size, stride, alpha),
            SqueezeExciteBlock(fcn_dim, reduction),
            ConvBlock(fcn_dim, 2 * fcn_dim, kernel_size, stride, alpha),
            SqueezeExciteBlock(2 * fcn_dim, reduction),
        )
        self.noise = NoiseLayer(mean=0, std=1e-3)
        self.rnn = SelfAttentionRnn(n_features, lstm_dim, lstm_n_layers, lstm_bidirectional, lstm_dropout)
        feature_dims = self.rnn.output_dim + 2 * fcn_dim
        self.converter = nn.Linear(feature_dims, feature_dims)
        self.classifier = nn.Sequential(
            nn.Linear(feature_dims, feature_dims // 2),
            nn.BatchNorm1d(feature_dims // 2),
            nn.LeakyReLU(alpha),
            nn.Linear(feature_dims // 2, n_classes)
        )

    def _features(self, x):
        x = self.noise(x)
        x_rnn = self.rnn(x)
        x_fcn = self.fcn(self.shuffle(x))                                   # (B, 2*fcn_dim, T'')
        x_fcn = GlobalAvgPoolFunction.apply(x_fcn[:, :, :, None, None])      # mean over time (:134)
        return linear(torch.cat([x_rnn, x_fcn], dim=1), self.converter)

    def forward(self, x: torch.Tensor):
        return self._head(self._features(x))

    def _head(self, f: torch.Tensor):
        lin0, bn, act, lin1 = self.classifier[0], self.classifier[1], self.classifier[2], self.classifier[3]
        slope = float(act.negative_slope)
        return head_apply(f, lin0, bn, lin1, -slope if slope != 0.0 else 0.0, bool(self.training))   # alpha < 0: LeakyReLU(-alpha)

    def encode(self, x: torch.Tensor):
        with torch.no_grad():
            return self._features(x)

    def shuffle(self, x: torch.Tensor):
        return x.permute(0, 2, 1)

    def summary(self):
        rows = ["%-40s %-20s %d" % (k, tuple(v.shape), v.numel()) for k, v in self.named_parameters()]
        print("\n".join(rows + ["total parameters: %d" % sum(p.numel() for p in self.parameters())]))
