"""GPU: src.optim.ClipAdamW (md_opt_grad_norm + md_opt_adamw_step) against torch.nn.utils.clip_grad_norm_ +
torch.optim.AdamW on the same parameters and gradients, several steps, ragged tensor sizes (shorter than a vector,
not a multiple of 4, longer than one chunk), with and without clipping being active.
Tolerance: 2e-6 relative to the parameter scale per step (same formulas, different association of the fp32 ops);
the gradient norm to 1e-6 relative (fp64 tree here, fp32 per-tensor norms in torch)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src.optim import ClipAdamW

SHAPES = [(1,), (3,), (7, 5), (45, 3, 1, 7, 7), (4096,), (4097,), (64, 130), (3, 1), (288, 128, 3, 1, 1), (2,)]


@pytest.mark.parametrize("max_norm,gscale", [(1.0, 1.0), (1.0, 1e-4), (None, 1.0), (0.05, 3.0)])
def test_clip_adamw_matches_torch(max_norm, gscale):
    dev = "cuda:0"
    g = torch.Generator().manual_seed(7)
    ref = [torch.randn(s, generator=g).to(dev).requires_grad_(True) for s in SHAPES]
    mine = [p.detach().clone().requires_grad_(True) for p in ref]
    o_ref = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    o_my = ClipAdamW(mine, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    for step in range(6):
        grads = [torch.randn(s, generator=g).to(dev) * gscale * (1 + step) for s in SHAPES]
        for p, q, gr in zip(ref, mine, grads):
            p.grad = gr.clone(); q.grad = gr.clone()
        if max_norm:
            n_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm)
        o_ref.step()
        o_my.step(max_norm=max_norm)
        if max_norm:
            assert abs(float(o_my.last_grad_norm) - float(n_ref)) <= 1e-6 * float(n_ref)
            for p, q in zip(ref, mine):      # gradients are clipped in place, as clip_grad_norm_ does
                assert float((p.grad - q.grad).abs().max()) <= 1e-6 * max(1e-30, float(p.grad.abs().max()))
        for p, q in zip(ref, mine):
            assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(p.abs().max())), (step, tuple(p.shape))
    sd = o_my.state_dict()
    assert sd["param_groups"][0]["step"] == 6 and len(sd["state"]) == len(SHAPES)
    assert all(st["step"] == 6 for st in sd["state"].values())


def test_clip_adamw_scheduler_and_missing_grads():
    dev = "cuda:0"
    ps = [torch.ones(10, device=dev, requires_grad=True), torch.ones(5000, device=dev, requires_grad=True)]
    opt = ClipAdamW(ps, lr=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    ps[1].grad = torch.full((5000,), 2.0, device=dev)          # ps[0] has no gradient: left untouched, as torch does
    opt.step(max_norm=1.0)
    sched.step()
    assert opt.param_groups[0]["lr"] == pytest.approx(5e-3)
    assert float((ps[0] - 1).abs().max()) == 0.0
    assert float(ps[1].max()) < 1.0
    with pytest.raises(RuntimeError):
        q = torch.ones(4, requires_grad=True); q.grad = torch.ones(4)
        ClipAdamW([q]).step()


def test_state_dict_round_trip_and_late_first_gradient():
    """ load_state_dict replaces the moment tensors: the cached device table must follow them, and a parameter
    that receives its first gradient later than the others starts its own bias correction at step 1, as
    torch.optim.AdamW does (``step`` is kept per parameter)."""
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    shapes = [(33,), (5000,), (7, 9)]
    ref = [torch.randn(s, generator=g).to(dev).requires_grad_(True) for s in shapes]
    mine = [p.detach().clone().requires_grad_(True) for p in ref]
    o_ref = torch.optim.AdamW(ref, lr=1e-2)
    o_my = ClipAdamW(mine, lr=1e-2)

    def step(with_last, max_norm):
        grads = [torch.randn(s, generator=g).to(dev) for s in shapes]
        for i, (p, q, gr) in enumerate(zip(ref, mine, grads)):
            if i == 2 and not with_last:
                p.grad = None; q.grad = None
            else:
                p.grad = gr.clone(); q.grad = gr.clone()
        if max_norm:
            torch.nn.utils.clip_grad_norm_([p for p in ref if p.grad is not None], max_norm)
        o_ref.step(); o_my.step(max_norm=max_norm)

    for _ in range(3):
        step(False, 1.0)                      # the third parameter gets no gradient yet
    step(True, 1.0)                           # ... and now its first one: two step counts in one group, clipping active
    step(True, None)
    # round trip through state_dict into fresh optimizers that have already stepped once (so their tables are cached)
    sd_ref, sd_my = o_ref.state_dict(), o_my.state_dict()
    ref2 = [p.detach().clone().requires_grad_(True) for p in ref]; mine2 = [p.detach().clone().requires_grad_(True) for p in mine]
    o_ref2 = torch.optim.AdamW(ref2, lr=1e-2); o_my2 = ClipAdamW(mine2, lr=1e-2)
    for p, q in zip(ref2, mine2):
        p.grad = torch.zeros_like(p); q.grad = torch.zeros_like(q)
    o_my2.step(); o_ref2.step()
    with torch.no_grad():
        for a, b in zip(ref2, ref): a.copy_(b)
        for a, b in zip(mine2, mine): a.copy_(b)
    o_ref2.load_state_dict(sd_ref); o_my2.load_state_dict(sd_my)
    ref, mine, o_ref, o_my = ref2, mine2, o_ref2, o_my2
    for _ in range(2):
        step(True, 0.5)
    for p, q in zip(ref, mine):
        assert float((p - q).abs().max()) <= 5e-6 * max(1.0, float(p.abs().max())), tuple(p.shape)
