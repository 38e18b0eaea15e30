"""MI355X-native mirror of the reference's ``src/models/ViViT.py`` (same classes, constructor arguments, child-module names and
state-dict keys: ``to_patch_embedding.1``, ``pos_embedding``, ``space_token``, ``temporal_token``,
``{space,temporal}_transformer.layers.N.{0,1}.{norm,fn...}``, ``{space,temporal}_transformer.norm``, ``mlp.{0,1,3}``).

forward (reference :171-194): (b, t, c, H, W) clip -> p x p patches, Linear -> [space token | patches] + positional table ->
dropout -> space Transformer over every frame's n+1 tokens -> the space token of each frame -> [temporal token | frames] ->
temporal Transformer -> token (or mean) -> Linear, LayerNorm, ELU, Linear.  Transformer (:96-113) is the pre-norm stack
``x = attn(norm(x)) + x; x = ff(norm(x)) + x`` with a closing LayerNorm.

The arithmetic runs on the gfx950 kernels: every Linear is the rows-major 1x1x1 convolution (MFMA) + ``md_channel_bias_*``;
``md_attention_*`` (batch-first, no mask) for the attention core; ``md_add_layernorm_*`` carries the residual stream (sum and
normalised branch from one kernel, the two gradients joined in its backward); ``md_gelu`` / ``md_elu``; the tokens and the
positional table are added as bias vectors by ``md_channel_bias_*``; ``md_mask_scale`` for the dropouts; ``md_seq_sum_*`` for
the mean pool.  torch only moves data (patch rearrangement, concatenation with the token slots, slicing).
"""
from typing import Union

import torch
import torch.nn as nn

from .. import ops
from ._unit import (AddLayerNormFunction, AttentionFunction, EluFunction, GeluFunction, LinearRowsFunction, PatchEmbedFunction,
                    ResidualLayerNormFunction, _ChannelBias, _SeqSum, branch_residual_layernorm, dropout, linear_bias_gelu_dropout, linear_wb)


def _layer_norm(x, norm: nn.LayerNorm):
    return AddLayerNormFunction.apply(x, None, norm.weight, norm.bias, norm.eps)


def _rows(x, lin: nn.Linear):
    """nn.Linear over the last dimension of x."""
    return linear_wb(x.reshape(-1, x.shape[-1]), lin.weight, lin.bias).reshape(x.shape[:-1] + (lin.out_features,))


class Residule(nn.Module):
    def __init__(self, fn):
        super(Residule, self).__init__()
        self.fn = fn

    def forward(self, x: torch.Tensor, **kwargs):
        y = self.fn(x, **kwargs)
        # y + x through the residual kernel; the normalised output is not used
        s, _ = ResidualLayerNormFunction.apply(y, x, x.new_ones(x.shape[-1]), x.new_zeros(x.shape[-1]), 1e-5)
        return s


class PreNorm(nn.Module):
    def __init__(self, dim: int, fn: Union[nn.Module, None]):
        super(PreNorm, self).__init__()
        self.dim = dim
        self.fn = fn
        self.norm = nn.LayerNorm(dim)

    def forward(self, x: torch.Tensor, **kwargs):
        return self.fn(_layer_norm(x, self.norm), **kwargs)


class FeedForward(nn.Module):
    def __init__(self, dim: int, hidden_dim: int, dropout: float = 0.5):
        super(FeedForward, self).__init__()
        self.dim = dim
        self.hidden_dim = hidden_dim
        self.dropout = dropout
        self.net = nn.Sequential(
            nn.Linear(dim, hidden_dim),
            nn.GELU(),
            nn.Dropout(dropout),
            nn.Linear(hidden_dim, dim),
            nn.Dropout(dropout)
        )

    def _hidden(self, x: torch.Tensor):
        lin = self.net[0]
        return linear_bias_gelu_dropout(x.reshape(-1, x.shape[-1]), lin.weight, lin.bias, self.net[2].p, self.net[2].training, 0)

    def forward(self, x: torch.Tensor):
        h = self._hidden(x).reshape(x.shape[:-1] + (self.net[0].out_features,))
        return dropout(_rows(h, self.net[3]), self.net[4].p, self.net[4].training)

    def branch_tail(self, x: torch.Tensor):
        """The branch up to the raw product of its last Linear: (y (rows, dim), bias, dropout p, training) -- Transformer.forward folds
        bias, dropout, the residual add and the next LayerNorm into one pass (branch_residual_layernorm)."""
        lin = self.net[3]
        return LinearRowsFunction.apply(self._hidden(x), lin.weight), lin.bias, self.net[4].p, self.net[4].training


class Attention(nn.Module):
    def __init__(self, dim: int, n_heads: int = 8, d_head: int = 64, dropout: float = 0.5):
        super(Attention, self).__init__()
        self.dim = dim
        self.n_heads = n_heads
        self.d_head = d_head
        self.dropout = dropout
        self.att_mat = None

        project_out = not (n_heads == 1 and d_head == dim)

        self.inner_dim = d_head * n_heads
        self.scale = d_head ** (-0.5)

        self.to_qkv = nn.Linear(dim, self.inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(
            nn.Linear(self.inner_dim, dim),
            nn.Dropout(dropout)
        ) if project_out else nn.Identity()

    def forward(self, x: torch.Tensor):
        # x (b, n, dim); q | k | v in thirds of the projection, heads as 'b n (h d)' (reference :72-76)
        out = AttentionFunction.apply(_rows(x, self.to_qkv), None, self.n_heads, None, True)
        if isinstance(self.to_out, nn.Identity):
            return out
        return dropout(_rows(out, self.to_out[0]), self.to_out[1].p, self.to_out[1].training)

    def branch_tail(self, x: torch.Tensor):
        """As FeedForward.branch_tail; None when there is no output projection (then forward() is the whole branch)."""
        if isinstance(self.to_out, nn.Identity):
            return None
        out = AttentionFunction.apply(_rows(x, self.to_qkv), None, self.n_heads, None, True)
        lin = self.to_out[0]
        return LinearRowsFunction.apply(out.reshape(-1, out.shape[-1]), lin.weight), lin.bias, self.to_out[1].p, self.to_out[1].training


class Transformer(nn.Module):
    def __init__(self, dim: int, depth: int, n_heads: int, d_head: int, mlp_dim: int, dropout: float = 0.0):
        super(Transformer, self).__init__()
        self.layers = nn.ModuleList([])
        self.norm = nn.LayerNorm(dim)

        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PreNorm(dim, Attention(dim=dim, n_heads=n_heads, d_head=d_head, dropout=dropout)),
                PreNorm(dim, FeedForward(dim=dim, hidden_dim=mlp_dim, dropout=dropout))
            ]))

    def forward(self, x: torch.Tensor):
        # res is the residual stream, h its LayerNorm for the next branch: each "branch + x, then norm" of the reference
        # (:108-112) is one md_add_layernorm call
        norms = [m.norm for pair in self.layers for m in pair] + [self.norm]
        res = x
        h = _layer_norm(res, norms[0])
        for i, (attn, ff) in enumerate(self.layers):
            for blk, norm in ((attn.fn, norms[2 * i + 1]), (ff.fn, norms[2 * i + 2])):
                tail = blk.branch_tail(h) if hasattr(blk, "branch_tail") else None
                if tail is None:
                    res, h = ResidualLayerNormFunction.apply(blk(h), res, norm.weight, norm.bias, norm.eps)
                else:           # Linear bias + dropout + residual add + the next LayerNorm in one pass
                    res, h = branch_residual_layernorm(tail[0], tail[1], tail[2], tail[3], res, norm)
        return h


class _PatchRearrange(nn.Module):
    """'b t c (h p1) (w p2) -> b t (h w) (p1 p2 c)' (reference :141)."""

    def __init__(self, patch_size: int):
        super().__init__()
        self.p = patch_size

    def forward(self, x):
        b, t, c, H, W = x.shape
        p = self.p
        return x.reshape(b, t, c, H // p, p, W // p, p).permute(0, 1, 3, 5, 4, 6, 2).reshape(b, t, (H // p) * (W // p), p * p * c)


def _with_token(x, token):
    """x (B, n, d) -> (B, n+1, d) with `token` (d) in front: the slot is concatenated as zeros and the token added as a bias
    vector, so its gradient (the sum over B) comes from md_channel_bias_bwd."""
    B, n, d = x.shape
    z = torch.cat((x.new_zeros(B, 1, d), x), dim=1).reshape(B, (n + 1) * d, 1)
    bias = torch.cat((token.reshape(d), token.new_zeros(n * d)))
    return _ChannelBias.apply(z, bias).reshape(B, n + 1, d)


class _ViViTBase(nn.Module):
    def _init_encoder(self, image_size, patch_size, n_frames, dim, depth, n_heads, pool, in_channels, d_head, dropout,
                      embedd_dropout, scale_dim):
        assert pool in {'cls', 'mean'}, 'pool type must be either cls(cls token) or mean(mean pooling)'
        assert image_size % patch_size == 0, "Image dimension(height and width) must be divisible by the patch_size"
        self.image_size = image_size
        self.n_frames = n_frames
        self.n_heads = n_heads
        self.d_head = d_head
        self.in_channels = in_channels

        n_patches = (image_size // patch_size) ** 2
        patch_dim = in_channels * patch_size ** 2

        self.to_patch_embedding = nn.Sequential(_PatchRearrange(patch_size), nn.Linear(patch_dim, dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, n_frames, n_patches + 1, dim))
        self.space_token = nn.Parameter(torch.randn(1, 1, dim))
        self.space_transformer = Transformer(dim, depth, n_heads, d_head, dim * scale_dim, dropout)
        self.temporal_token = nn.Parameter(torch.randn(1, 1, dim))
        self.temporal_transformer = Transformer(dim, depth, n_heads, d_head, dim * scale_dim, dropout)
        self.dropout = nn.Dropout(embedd_dropout)
        self.pool = pool
        self.dim = dim

    def _encode(self, x: torch.Tensor):
        if x.size()[1] == self.in_channels:
            x = torch.permute(x, (0, 2, 1, 3, 4))
        b, t, c, H, W = x.shape
        p = self.to_patch_embedding[0].p
        lin = self.to_patch_embedding[1]
        d = self.dim
        n = (H // p) * (W // p)
        pos = self.pos_embedding[0, :, :(n + 1)]
        if pos.shape[0] != t:
            raise RuntimeError("ViViT: the clip has %d frames, the positional table %d" % (t, pos.shape[0]))
        if self._fused_patch_embed(x, p, d):
            # one gather-GEMM: patches are read straight from the clip, bias + space token + positional table in the epilogue
            w_perm = lin.weight.view(d, p, p, c).permute(0, 3, 1, 2).reshape(d, c * p * p)         # columns (p1 p2 c) -> (c p1 p2)
            x = PatchEmbedFunction.apply(x, w_perm, lin.bias, pos.contiguous(), self.space_token.reshape(d), p)
        else:
            patches = self.to_patch_embedding[0](x)
            x = _rows(patches, lin)                                                     # (b, t, n, d)
            x = _with_token(x.reshape(b * t, n, d), self.space_token)                   # (b t, n+1, d)
            x = _ChannelBias.apply(x.reshape(b, t * (n + 1) * d, 1), pos.reshape(-1)).reshape(b * t, n + 1, d)
        x = dropout(x, self.dropout.p, self.dropout.training)
        x = self.space_transformer(x)
        x = x[:, 0].reshape(b, t, d)
        x = _with_token(x, self.temporal_token)                                     # (b, t+1, d)
        x = self.temporal_transformer(x)
        return _SeqSum.apply(x, 1.0 / (t + 1)) if self.pool == 'mean' else x[:, 0]

    @staticmethod
    def _fused_patch_embed(x, p: int, d: int) -> bool:
        from .. import _native as N
        ok_geo = p >= 8 and (p & (p - 1)) == 0 and x.shape[4] % 4 == 0 and all(s % 4 == 0 for s in x.stride()[:3])
        return (ok_geo and x.stride(4) == 1 and x.stride(3) == x.shape[4] and d % 4 == 0 and d <= 128 and x.dtype == torch.float32
                and not N.lib().md_get_exact_fp32())

    def summary(self, *args, **kwargs):
        rows = ["%-60s %-20s %d" % (k, tuple(v.shape), v.numel()) for k, v in self.named_parameters()]
        print("\n".join(rows + ["total parameters: %d" % sum(p.numel() for p in self.parameters())]))


class ViViT(_ViViTBase):
    def __init__(self, image_size: int, patch_size: int, n_frames: int = 21, n_classes: int = 2, dim: int = 192, depth: int = 4,
                 n_heads: int = 3, pool: str = 'cls', in_channels: int = 3, d_head: int = 64, dropout: float = 0.,
                 embedd_dropout: float = 0., scale_dim: int = 4, alpha: float = 1.0):
        super(ViViT, self).__init__()
        self._init_encoder(image_size, patch_size, n_frames, dim, depth, n_heads, pool, in_channels, d_head, dropout,
                           embedd_dropout, scale_dim)
        self.depth = depth
        self.mlp = nn.Sequential(
            nn.Linear(dim, dim // 2),
            nn.LayerNorm(dim // 2),
            nn.ELU(alpha),
            nn.Linear(dim // 2, n_classes)
        )

    def _head(self, latent: torch.Tensor):
        x = _layer_norm(_rows(latent, self.mlp[0]), self.mlp[1])
        return _rows(EluFunction.apply(x, self.mlp[2].alpha), self.mlp[3])

    def forward(self, x: torch.Tensor):
        # the GEMM operands of every Linear packed by one batched call (36 tiny launches per step otherwise); dropout without mask tensors
        with ops.prepacked(self), ops.counter_dropout(x.device, self.training):
            return self._head(self._encode(x))

    def encode(self, x: torch.Tensor):
        with torch.no_grad(), ops.prepacked(self):
            return self._encode(x)


class ViViTEncoder(_ViViTBase):
    def __init__(self, image_size: int, patch_size: int, n_frames: int, dim: int = 192, depth: int = 4, n_heads: int = 3,
                 pool: str = 'cls', in_channels: int = 3, d_head: int = 64, dropout: float = 0., embedd_dropout: float = 0.,
                 scale_dim: int = 4):
        super(ViViTEncoder, self).__init__()
        self._init_encoder(image_size, patch_size, n_frames, dim, depth, n_heads, pool, in_channels, d_head, dropout,
                           embedd_dropout, scale_dim)

    def forward(self, x: torch.Tensor):
        with ops.prepacked(self), ops.counter_dropout(x.device, self.training):
            return self._encode(x)
