"""LeakyReLU kink diagnostics for the R(2+1)D gradient-parity tests (test infrastructure; imports the oracle).

The network is piecewise linear in its activations: LeakyReLU(0.01)'s derivative jumps by a factor 100 at zero, so two
correct evaluations that disagree about the SIGN of one near-zero pre-activation produce gradients that differ by far
more than rounding (the reference does this to itself: its fp32 and fp64 runs disagree by up to 2e-2 on some fixtures).
Instead of loosening the tolerance on such inputs, these helpers make the statement precise:

  1. run the HIP trunk, read every LeakyReLU pre-activation it actually used out of the executor workspace
     (md_plan_unit_layout / md_plan_z_layout) and record its sign pattern;
  2. compare with the fp64 oracle's pre-activations: list every element whose sign differs (unit, index, both values);
  3. evaluate the fp64 oracle ON THE HIP PATH'S ACTIVATION PATTERN (oracle.r2plus1d._leaky(force=...)): the exact
     gradient of the same piecewise-linear function the HIP path evaluated.  The HIP gradients must match THAT to 1e-3,
     and the flipped elements must be few and within rounding error of zero.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from oracle import r2plus1d as orc


def _to_ncthw(t2d: torch.Tensor, C: int, shape5) -> torch.Tensor:
    """[rows, Cp] channels-last -> (N, C, T, H, W)"""
    N, T, H, W = shape5
    return t2d[:, :C].reshape(N, T, H, W, C).permute(0, 4, 1, 2, 3).contiguous()


def hip_preactivations(model, x: torch.Tensor) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """Runs the trunk's executor forward (training mode, batch statistics) on its own workspace and returns
    ({name: pre-activation (N,C,T,H,W) fp32 on the CPU}, features).  Names as in the oracle: the unit's state-dict
    prefix, and ``<block prefix>.relu`` for the closing activation of a residual block.  Running statistics are given
    throw-away copies, so the model is not modified."""
    net = model.res2plus1d
    B, _, T, H, W = x.shape
    plan = net._plan(B, T, H, W)
    units = net.unit_modules()
    ws = plan.new_workspace(x.device)
    feat = plan.forward(x.contiguous().float(), ws, [u.conv.weight for u in units], [u.bn.weight for u in units],
                        [u.bn.bias for u in units], [u.bn.running_mean.clone() for u in units],
                        [u.bn.running_var.clone() for u in units], True)
    torch.cuda.synchronize()
    names = [u.name for u in orc.all_units(net.layer_sizes, net.alpha)]
    assert len(names) == plan.num_units
    pre: Dict[str, torch.Tensor] = {}
    for i, name in enumerate(names):
        d = plan.descs[i]
        raw, st = plan.unit_tensors(ws, i)
        p = raw * st[2].view(1, -1) + st[3].view(1, -1)                  # scale * y + shift, as the kernels evaluate it (fmaf)
        p = torch.addcmul(st[3].view(1, -1), raw, st[2].view(1, -1))
        pre[name] = _to_ncthw(p, d.Cout, (d.N, d.To, d.Ho, d.Wo)).cpu()
    # closing activations: z = leaky(skip + main, alpha) is materialised; the sign of the sum is the sign of z and
    # |sum| = |z| (z > 0) or |z| / alpha
    blocks = orc.block_prefixes(net.layer_sizes)
    # output geometry of block k = geometry of its last main-path unit (conv2.temporal_conv)
    idx = {n: i for i, n in enumerate(names)}
    for k, bp in enumerate(blocks):
        z, C = plan.z_tensor(ws, 2 + k)
        d = plan.descs[idx[bp + ".conv2.temporal_conv"]]
        zz = _to_ncthw(z, C, (d.N, d.To, d.Ho, d.Wo)).cpu()
        pre[bp + ".relu"] = torch.where(zz > 0, zz, zz / net.alpha)
    return pre, feat


def sign_masks(pre: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {k: v > 0 for k, v in pre.items()}


def flips(pre_a: Dict[str, torch.Tensor], pre_b: Dict[str, torch.Tensor]) -> List[tuple]:
    """Elements whose sign differs between two evaluations: (name, flat index, value in a, value in b, rms of the tensor)."""
    out = []
    for k, a in pre_a.items():
        b = pre_b[k].to(torch.float64)
        a = a.to(torch.float64)
        bad = ((a > 0) != (b > 0)).reshape(-1).nonzero().reshape(-1)
        rms = float(b.pow(2).mean().sqrt())
        for i in bad.tolist():
            out.append((k, i, float(a.reshape(-1)[i]), float(b.reshape(-1)[i]), rms))
    return out


def oracle_grads_on_pattern(x, y, params, bufs, layer_sizes, alpha, loss, masks):
    """fp64 oracle forward + backward with every LeakyReLU evaluated on the given activation pattern."""
    from oracle import step as ostep
    p64 = {k: v.double() for k, v in params.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in bufs.items()}
    return ostep.r2plus1d_loss_and_grads(x.double(), y, p64, b64, layer_sizes, alpha, loss, force=masks)


def oracle_preactivations(x, y, params, bufs, layer_sizes, alpha, loss):
    """fp64 oracle with its own activation pattern: (logits, loss, grads, {name: pre-activation})."""
    from oracle import step as ostep
    p64 = {k: v.double() for k, v in params.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in bufs.items()}
    tap: Dict[str, torch.Tensor] = {}
    logits, L, g = ostep.r2plus1d_loss_and_grads(x.double(), y, p64, b64, layer_sizes, alpha, loss, tap=tap)
    return logits, L, g, tap


def rel_l2(a: torch.Tensor, ref: torch.Tensor, floor: float) -> float:
    return float((a.double() - ref.double()).norm() / max(float(ref.double().norm()), floor))


def gradient_report(model, x_cpu, y_cpu, layer_sizes, alpha, seed, loss_weight, gamma, device):
    """Full comparison for a model that already has gradients from  FocalLoss(model(x), y).backward():
    returns dict(flips, worst_own, worst_pattern, median_pattern, worst_name) -- relative L2 error per parameter of the HIP
    gradients against the fp64 oracle with its own / with the HIP path's activation pattern (``linear.0.bias`` excluded:
    it feeds BatchNorm1d, its true gradient is zero)."""
    import numpy as np
    from oracle import losses as ol
    pre_hip, _ = hip_preactivations(model, x_cpu.to(device))
    w = loss_weight.double()
    lossf = lambda o, t: ol.focal_loss(o, t, w, gamma)
    params, bufs = orc.synth_state(layer_sizes, seed, alpha)
    _, _, g64, pre64 = oracle_preactivations(x_cpu, y_cpu, params, bufs, layer_sizes, alpha, lossf)
    fl = flips(pre_hip, pre64)
    if fl:
        params, bufs = orc.synth_state(layer_sizes, seed, alpha)
        _, _, g64p = oracle_grads_on_pattern(x_cpu, y_cpu, params, bufs, layer_sizes, alpha, lossf, sign_masks(pre_hip))
    else:
        g64p = g64
    gmax = max(float(v.norm()) for v in g64.values())
    own, pat, worst_name = [], [], ""
    for k, p in model.named_parameters():
        if k == "linear.0.bias":
            continue
        own.append(rel_l2(p.grad.cpu(), g64[k], 1e-6 * gmax))
        e = rel_l2(p.grad.cpu(), g64p[k], 1e-6 * gmax)
        if not pat or e > max(pat):
            worst_name = k
        pat.append(e)
    total = sum(v.numel() for v in pre_hip.values())
    return {"flips": fl, "worst_own": max(own), "worst_pattern": max(pat), "median_pattern": float(np.median(pat)),
            "worst_name": worst_name, "elements": total, "max_flips": max_flips(total)}


def max_flips(total_elements: int) -> int:
    """How many sign flips two correct fp32-level evaluations may show: pre-activations are O(1) numbers with O(1) density at
    zero, the two evaluations differ by ~1e-6 absolute (fp32 rounding accumulated through a convolution and BatchNorm), so
    about 1e-6 of all elements can land on different sides; 2e-6 of them (and at least 8) are allowed."""
    return max(8, int(2e-6 * total_elements))
