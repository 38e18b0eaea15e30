"""smoke(): one tiny forward + focal loss + backward of the R(2+1)D classifier on the GPU, checked
against the oracle (the oracle is the checker here, never the thing run)."""
import torch


def run_smoke(device):
    from oracle import losses as ol, r2plus1d as orc, step as ostep
    from src.models.R2Plus1D import R2Plus1DClassifier
    from src.loss import FocalLoss
    ls, B, T, S, alpha, seed = [1, 1, 1, 1], 2, 4, 32, 0.01, 3
    model = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=ls, alpha=alpha)
    params, bufs = orc.synth_state(ls, seed, alpha)
    sd = dict(params); sd.update(bufs)
    model.load_state_dict(sd, strict=True)
    model.to(device).train()
    x = orc.synth_clip(B, T, S, seed); y = orc.synth_labels(B, seed)
    loss_fn = FocalLoss(weight=torch.ones(2), gamma=2.0)
    logits = model(x.to(device))
    loss = loss_fn(logits, y.to(device))
    loss.backward()
    torch.cuda.synchronize()
    ref_logits, ref_loss, ref_g = ostep.r2plus1d_loss_and_grads(
        x, y, params, bufs, ls, alpha, lambda o, t: ol.focal_loss(o, t, torch.ones(2), 2.0))
    err = float((logits.detach().cpu() - ref_logits).abs().max() / ref_logits.abs().max())
    assert err < 1e-3, f"logits mismatch {err}"
    assert abs(loss.item() - float(ref_loss)) < 1e-3 * max(1.0, abs(float(ref_loss)))
    k = "res2plus1d.conv1.spatio_conv.conv.weight"
    g = dict(model.named_parameters())[k].grad.cpu()
    gerr = float((g - ref_g[k]).abs().max() / ref_g[k].abs().max())
    assert gerr < 1e-3, f"stem weight gradient mismatch {gerr}"
    print(f"smoke ok: logits relerr {err:.2e}, stem dW relerr {gerr:.2e}, loss {loss.item():.6f}")
