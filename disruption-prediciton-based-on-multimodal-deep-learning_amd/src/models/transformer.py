"""MI355X-native mirror of the reference's ``src/models/transformer.py`` (same classes, constructor arguments, child-module
names and state-dict keys: ``filter``, ``pos_enc.pe``, ``transformer_encoder.layers.N.{self_attn,linear1,linear2,norm1,norm2}``,
``connector``, ``classifier``).

forward (reference :88-105): NoiseLayer -> Conv1d, Conv1d, BatchNorm1d, ReLU over (B, F, T) -> positional encoding ->
n_layers x nn.TransformerEncoderLayer (post-norm, causal additive mask, the reference's tanh-GELU) -> mean over time -> Linear,
LayerNorm, nn.GELU [-> Linear, LayerNorm, tanh-GELU, Linear for ``Transformer``].  The ``nn.TransformerEncoder`` child is kept
as the parameter holder; the arithmetic runs on the gfx950 kernels: (k,1,1) conv units, 1x1x1 convolutions + ``md_channel_bias_*``
for every Linear, ``md_attention_*``, ``md_add_layernorm_*`` (residual add fused), ``md_gelu``, ``md_seq_sum_*``, ``md_mask_scale``
for the dropouts (masks from torch's device generator: same distribution as, not bit-compatible with, ATen's fused dropout).
"""
import math

import torch
import torch.nn as nn

from .. import _native as N
from .. import ops
from .NoiseLayer import NoiseLayer
from ._unit import (AddLayerNormFunction, AttentionFunction, ConvFunction, GeluFunction, _ChannelBias, _SeqSum, conv1d_bn_leaky,
                    dropout, linear, linear_bias_gelu_dropout, linear_wb)


class PositionalEncoding(nn.Module):
    def __init__(self, d_model: int, max_len: int = 128):
        super(PositionalEncoding, self).__init__()
        self.d_model = d_model
        self.max_len = max_len
        # sin on the even features, cos on the odd ones, wavelengths 10000^(2i/d); built as (max_len, 1, d_model) directly
        steps = torch.arange(max_len, dtype=torch.float32)[:, None]
        rates = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(steps * rates)
        pe[:, 0, 1::2] = torch.cos(steps * rates)[:, : d_model // 2]
        self.register_buffer('pe', pe)

    def forward(self, x: torch.Tensor):
        # x : (seq_len, batch, d_model); done in (batch, seq_len*d_model) rows so that the table is one bias vector
        S, B, D = x.shape
        xb = x.permute(1, 0, 2).contiguous().view(B, S * D, 1)
        out = _ChannelBias.apply(xb, self.pe[:S, 0, :].contiguous().view(S * D))
        return out.view(B, S, D).permute(1, 0, 2).contiguous()


class GELU(nn.Module):
    def forward(self, x):
        return GeluFunction.apply(x, 1)


class TransformerEncoder(nn.Module):
    def __init__(self, n_features: int = 11, kernel_size: int = 3, feature_dims: int = 256, max_len: int = 128, n_layers: int = 1,
                 n_heads: int = 8, dim_feedforward: int = 1024, dropout: float = 0.1):
        super(TransformerEncoder, self).__init__()
        self.src_mask = None
        self.n_features = n_features
        self.max_len = max_len
        self.feature_dims = feature_dims
        self.noise = NoiseLayer(mean=0, std=1e-3)
        if kernel_size // 2 == 0:            # (sic) the reference's check: only kernel_size 1 is bumped, to 2
            print("kernel sholud be odd number")
            kernel_size += 1
        same = dict(kernel_size=kernel_size, stride=1, padding=(kernel_size - 1) // 2)
        self.filter = nn.Sequential(nn.Conv1d(n_features, feature_dims, **same), nn.Conv1d(feature_dims, feature_dims, **same),
                                    nn.BatchNorm1d(feature_dims), nn.ReLU())
        self.pos_enc = PositionalEncoding(d_model=feature_dims, max_len=max_len)
        encoder = nn.TransformerEncoderLayer(d_model=feature_dims, nhead=n_heads, dropout=dropout, dim_feedforward=dim_feedforward,
                                             activation=GELU())
        self.transformer_encoder = nn.TransformerEncoder(encoder, num_layers=n_layers)
        self.connector = nn.Sequential(nn.Linear(feature_dims, feature_dims), nn.LayerNorm(feature_dims), nn.GELU())

    # -- one nn.TransformerEncoderLayer (norm_first = False): x = norm1(x + drop(SA(x))); x = norm2(x + drop(FF(x)))
    def _layer(self, x, layer, mask):
        S, B, D = x.shape
        sa = layer.self_attn
        tr = self.training
        qkv = linear_wb(x.view(S * B, D), sa.in_proj_weight, sa.in_proj_bias).view(S, B, 3 * D)
        drop = None
        if tr and sa.dropout > 0:
            keep = 1.0 - sa.dropout
            drop = torch.empty((B * sa.num_heads, S, S), device=x.device).bernoulli_(keep) / keep
        att = AttentionFunction.apply(qkv, mask, sa.num_heads, drop)
        att = linear_wb(att.view(S * B, D), sa.out_proj.weight, sa.out_proj.bias).view(S, B, D)
        x = AddLayerNormFunction.apply(x, dropout(att, layer.dropout1.p, tr), layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
        h = linear_bias_gelu_dropout(x.view(S * B, D), layer.linear1.weight, layer.linear1.bias, layer.dropout.p, tr, 1)
        h = linear_wb(h, layer.linear2.weight, layer.linear2.bias).view(S, B, D)
        return AddLayerNormFunction.apply(x, dropout(h, layer.dropout2.p, tr), layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)

    def _filter(self, x_bft):
        c1, c2, bn = self.filter[0], self.filter[1], self.filter[2]
        y = ConvFunction.apply(x_bft.contiguous()[:, :, :, None, None], c1.weight[:, :, :, None, None], (c1.stride[0], 1, 1),
                               (c1.padding[0], 0, 0))
        y = _ChannelBias.apply(y.squeeze(4).squeeze(3), c1.bias)
        return conv1d_bn_leaky(y, c2, bn, 0.0, self.training)

    def forward(self, x: torch.Tensor):
        from .. import ops
        with ops.counter_dropout(x.device, self.training):      # the layers' nn.Dropout without mask tensors (md_dropout_ctr)
            return self._forward(x)

    def _forward(self, x: torch.Tensor):
        x = self.noise(x)
        x = self._filter(x.permute(0, 2, 1)).permute(2, 0, 1).contiguous()          # (T, B, D)
        # the reference rebuilds the (constant) causal mask on the CPU and uploads it every forward (:98-99), which puts a
        # synchronous host-to-device copy into every step; it is built once per (length, device) here
        if self.src_mask is None or self.src_mask.shape[0] != len(x) or self.src_mask.device != x.device:
            self.src_mask = self._generate_square_subsequent_mask(len(x), x.device)
        x = self.pos_enc(x)
        for layer in self.transformer_encoder.layers:
            x = self._layer(x, layer, self.src_mask)
        S = x.shape[0]
        x = _SeqSum.apply(x.permute(1, 0, 2).contiguous(), 1.0 / S)                 # mean over time (:103)
        lin, ln = self.connector[0], self.connector[1]
        x = AddLayerNormFunction.apply(linear(x, lin), None, ln.weight, ln.bias, ln.eps)
        return GeluFunction.apply(x, 0)

    def _generate_square_subsequent_mask(self, size: int, device: str):
        mask = (torch.triu(torch.ones(size, size)) == 1).to(device).transpose(0, 1)
        mask = mask.float().masked_fill(mask == 0, float('-inf')).masked_fill(mask == 1, float(0.0))
        return mask.contiguous()

    def summary(self):
        rows = ["%-60s %-20s %d" % (k, tuple(v.shape), v.numel()) for k, v in self.named_parameters()]
        print("\n".join(rows + ["total parameters: %d" % sum(p.numel() for p in self.parameters())]))


class Transformer(nn.Module):
    def __init__(self, n_features: int = 11, kernel_size: int = 5, feature_dims: int = 256, max_len: int = 128, n_layers: int = 1,
                 n_heads: int = 8, dim_feedforward: int = 1024, dropout: float = 0.1, cls_dims: int = 128, n_classes: int = 2):
        super(Transformer, self).__init__()
        self.max_len = max_len
        self.n_features = n_features
        self.encoder = TransformerEncoder(n_features, kernel_size, feature_dims, max_len, n_layers, n_heads, dim_feedforward, dropout)
        self.classifier = nn.Sequential(nn.Linear(feature_dims, cls_dims), nn.LayerNorm(cls_dims), GELU(), nn.Linear(cls_dims, n_classes))

    def encode(self, x: torch.Tensor):
        with torch.no_grad():
            return self.encoder(x)

    @property
    def feature_dims(self):
        # the reference's MultiModalModel_GB / TFN_GB read this attribute (MultiModal.py:65,259), which its Transformer never
        # sets; exposed here so those classes construct
        return self.encoder.feature_dims

    def _head(self, latent: torch.Tensor):
        lin0, ln, lin1 = self.classifier[0], self.classifier[1], self.classifier[3]
        h = AddLayerNormFunction.apply(linear(latent, lin0), None, ln.weight, ln.bias, ln.eps)
        return linear(GeluFunction.apply(h, 1), lin1)

    def forward(self, x: torch.Tensor):
        return self._head(self.encoder(x))

    def summary(self):
        self.encoder.summary()
