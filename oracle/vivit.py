"""CPU restatement (test infrastructure only) of the reference's ViViT.forward / ViViTEncoder.forward (src/models/ViViT.py:171-194,
284-299) as a function of a state dict with the reference's keys (dropout 0).  The einops patterns are written out as
reshape / permute.  Pinned by tests/golden/vivit.npz."""
import torch
import torch.nn.functional as F


def _attention(x, sd, p, n_heads):                                                                   # ViViT.py:69-91
    b, n, _ = x.shape
    qkv = F.linear(x, sd[p + "to_qkv.weight"])
    inner = qkv.shape[-1] // 3
    dh = inner // n_heads
    q, k, v = (t.reshape(b, n, n_heads, dh).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))       # b h n d
    att = torch.softmax(q @ k.transpose(2, 3) * dh ** -0.5, dim=-1) @ v
    out = att.permute(0, 2, 1, 3).reshape(b, n, inner)
    if p + "to_out.0.weight" in sd:
        out = F.linear(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return out


def _transformer(x, sd, p, depth, n_heads):                                                          # ViViT.py:108-113
    D = x.shape[-1]
    for l in range(depth):
        a = f"{p}layers.{l}.0."
        f = f"{p}layers.{l}.1."
        x = _attention(F.layer_norm(x, (D,), sd[a + "norm.weight"], sd[a + "norm.bias"], 1e-5), sd, a + "fn.", n_heads) + x
        h = F.layer_norm(x, (D,), sd[f + "norm.weight"], sd[f + "norm.bias"], 1e-5)
        h = F.linear(F.gelu(F.linear(h, sd[f + "fn.net.0.weight"], sd[f + "fn.net.0.bias"])), sd[f + "fn.net.3.weight"], sd[f + "fn.net.3.bias"])
        x = h + x
    return F.layer_norm(x, (D,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)


def vivit_forward(x, sd, patch_size, depth, n_heads, pool, in_channels=3, alpha=1.0, with_mlp=True):
    if x.shape[1] == in_channels:                                                                    # :174-175
        x = x.permute(0, 2, 1, 3, 4)
    b, t, c, H, W = x.shape
    ps = patch_size
    x = x.reshape(b, t, c, H // ps, ps, W // ps, ps).permute(0, 1, 3, 5, 4, 6, 2).reshape(b, t, (H // ps) * (W // ps), ps * ps * c)
    x = F.linear(x, sd["to_patch_embedding.1.weight"], sd["to_patch_embedding.1.bias"])              # :177
    n, d = x.shape[2], x.shape[3]
    x = torch.cat((sd["space_token"].reshape(1, 1, 1, d).expand(b, t, 1, d), x), dim=2)              # :179,182
    x = x + sd["pos_embedding"][:, :, :(n + 1)]                                                      # :183
    x = _transformer(x.reshape(b * t, n + 1, d), sd, "space_transformer.", depth, n_heads)           # :186-187
    x = x[:, 0].reshape(b, t, d)                                                                      # :188
    x = torch.cat((sd["temporal_token"].reshape(1, 1, d).expand(b, 1, d), x), dim=1)                 # :190
    x = _transformer(x, sd, "temporal_transformer.", depth, n_heads)
    x = x.mean(dim=1) if pool == "mean" else x[:, 0]                                                 # :192
    return vivit_head(x, sd, alpha) if with_mlp else x


def vivit_head(latent, sd, alpha=1.0):                                                               # :163-168
    h = F.linear(latent, sd["mlp.0.weight"], sd["mlp.0.bias"])
    h = F.elu(F.layer_norm(h, (h.shape[1],), sd["mlp.1.weight"], sd["mlp.1.bias"], 1e-5), alpha)
    return F.linear(h, sd["mlp.3.weight"], sd["mlp.3.bias"])
