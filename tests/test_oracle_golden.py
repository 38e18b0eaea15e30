"""CPU: the oracle (oracle/) against the fixtures generated from the reference (tests/golden/).

Tolerances: 2e-5 relative-to-scale for fp32 forward values and gradients (same ATen CPU kernels in a
different composition order); bit-exact for predictions / label bookkeeping / the DRW table.
"""
import os

import numpy as np
import pytest
import torch

from oracle import losses, r2plus1d as orc, step

torch.set_num_threads(8)


def subsample(t, n=48):
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // n)
    return f[::stride][:n].numpy()


def close(a, b, tol=2e-5, floor=1e-6):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = max(floor, float(np.max(np.abs(b))))
    return float(np.max(np.abs(a - b))) / scale <= tol


TAGS = [f"r2p1d_1111_s{i}" for i in (1, 2, 3, 4)] + [f"r2p1d_1221_s{i}" for i in (11, 12, 13)]


@pytest.mark.parametrize("tag", TAGS)
def test_r2plus1d_oracle_matches_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    ls = [int(v) for v in g["layer_sizes"]]
    B, T, S, alpha, seed = int(g["B"]), int(g["T"]), int(g["S"]), float(g["alpha"]), int(g["seed"])
    params, bufs = orc.synth_state(ls, seed, alpha)
    x = orc.synth_clip(B, T, S, seed)
    y = orc.synth_labels(B, seed)
    w = torch.from_numpy(g["weight"])
    gamma = float(g["gamma"])
    logits, L, grads = step.r2plus1d_loss_and_grads(
        x, y, params, bufs, ls, alpha, lambda o, t: losses.focal_loss(o, t, w, gamma))
    assert close(logits.numpy(), g["logits"])
    assert close(L.numpy(), g["loss"])
    # gradients that are analytically zero (e.g. a bias in front of a BatchNorm) are pure round-off
    # noise in both implementations: compare those against the scale of the largest gradient.
    gmax = max(float(g["gnorm/" + str(k)]) for k in g["param_names"])
    for k in g["param_names"]:
        k = str(k)
        if k == "linear.0.bias":   # bias in front of BatchNorm1d: analytically zero gradient
            assert float(grads[k].abs().max()) < 1e-6 * gmax
            continue
        assert close(subsample(grads[k]), g["gsub/" + k], 5e-5, 1e-5 * gmax), k
        assert close(grads[k].norm().item(), g["gnorm/" + k], 5e-5, 1e-5 * gmax), k
    for k in [k for k in g.files if k.startswith("buf/")]:
        assert close(bufs[k[4:]].numpy(), g[k]), k
    assert int(bufs["linear.1.num_batches_tracked"]) == int(g["nbt"])


def test_losses_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    w = torch.from_numpy(g["w"])
    m = losses.ldam_margins([100, 2000], 0.5)
    assert np.array_equal(m.numpy(), g["m_list"])
    fns = {
        "focal_g2": lambda x, y: losses.focal_loss(x, y, w, 2.0),
        "focal_g0p5": lambda x, y: losses.focal_loss(x, y, w, 0.5),
        "ldam_s30": lambda x, y: losses.ldam_loss(x, y, m, w, 30.0),
        "ldam_s1_now": lambda x, y: losses.ldam_loss(x, y, m, None, 1.0),
        "ce": lambda x, y: losses.ce_loss(x, y, w),
    }
    for B in (1, 8, 33):
        y = torch.from_numpy(g[f"y{B}"])
        for name, fn in fns.items():
            x = torch.from_numpy(g[f"x{B}"]).requires_grad_(True)
            L = fn(x, y)
            L.backward()
            assert close(L.detach().numpy(), g[f"{name}/L{B}"], 1e-5), (name, B)
            assert close(x.grad.numpy(), g[f"{name}/g{B}"], 1e-5), (name, B)
    y = torch.from_numpy(g["gb/y"])
    xs = {n: torch.from_numpy(g[f"gb/x_{n}"]).requires_grad_(True) for n in ("multi", "vis", "ts")}
    one = torch.ones(2)
    L = losses.gradient_blending(losses.focal_loss(xs["multi"], y, one), losses.focal_loss(xs["vis"], y, one),
                                 losses.focal_loss(xs["ts"], y, one), 0.1, 0.4, 0.5)
    L.backward()
    assert close(L.detach().numpy(), g["gb/L"], 1e-6)
    for n in xs:
        assert close(xs[n].grad.numpy(), g[f"gb/g_{n}"], 1e-5)


def test_drw_schedule_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "drw.npz"))
    for n in (8, 50, 128):
        tab = np.stack([losses.drw_weights(e, n, [0, 0.25, 0.75, 0.9], [100, 2000]) for e in range(n)])
        assert np.array_equal(tab, g[f"w{n}"])


def test_step_bookkeeping(golden_dir):
    g = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    ls = [int(v) for v in g["layer_sizes"]]
    B, T, S, alpha, seed = int(g["B"]), int(g["T"]), int(g["S"]), float(g["alpha"]), int(g["seed"])
    params, bufs = orc.synth_state(ls, seed, alpha)
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    before = {k: v.detach().clone() for k, v in params.items()}
    opt = torch.optim.AdamW(list(params.values()), lr=2e-4)
    one = torch.ones(2)
    tot_loss, correct, n, preds, labels = 0.0, 0, 0, [], []
    for i in range(3):
        x = orc.synth_clip(B, T, S, seed + i); y = orc.synth_labels(B, seed + i, 0.4)
        opt.zero_grad()
        out = orc.classifier_forward(x, params, bufs, ls, alpha, True)
        L = losses.focal_loss(out, y, one, 2.0)
        L.backward()
        step.clip_grad_norm([p.grad for p in params.values()], 1.0)
        opt.step()
        tot_loss += L.item()
        p = step.predictions(out.detach())
        correct += int((p == y).sum()); n += B
        preds.append(p.numpy()); labels.append(y.numpy())
    assert np.array_equal(np.stack(preds), g["preds"])
    assert abs(tot_loss / n - float(g["train_loss"])) < 1e-5
    assert correct / n == float(g["train_acc"])
    assert abs(step.macro_f1(np.concatenate(labels), np.concatenate(preds)) - float(g["train_f1"])) < 1e-12
    for k in params:
        if k == "linear.0.bias":   # Adam step of a round-off-noise gradient: sign-chaotic, not comparable
            continue
        assert close(subsample(params[k].detach() - before[k]), g["dsub/" + k], 2e-3), k
