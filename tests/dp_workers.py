"""Worker bodies for the data-parallel tests (spawned processes import this module)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "disruption-prediciton-based-on-multimodal-deep-learning_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


class _FakeUnit(torch.nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = torch.nn.Conv3d(cin, cout, 1, bias=False)
        self.bn = torch.nn.BatchNorm3d(cout)


class _Blk(torch.nn.Module):
    def __init__(self, down):
        super().__init__()
        self.downsample = down


class _Layer(torch.nn.Module):
    def __init__(self, down):
        super().__init__()
        self.block1 = _Blk(down)
        self.blocks = torch.nn.ModuleList([])


class FakeTrunk(torch.nn.Module):
    """CPU stand-in that follows the trunk's gradient protocol exactly (flat buffer [w.., gamma.., beta..], stage hook
    called for stages 4..0) with a cheap differentiable function, so GradAllReducer can be exercised over gloo."""

    def __init__(self):
        super().__init__()
        self.grad_segment_hook = None
        self.conv2, self.conv3, self.conv4, self.conv5 = _Layer(False), _Layer(True), _Layer(True), _Layer(True)
        n = 2 + 4 + 6 + 6 + 6
        self.units = torch.nn.ModuleList([_FakeUnit(3, 4) for _ in range(n)])

    def unit_modules(self):
        return list(self.units)

    def forward(self, x):
        units = self.unit_modules()
        ps = [u.conv.weight for u in units] + [u.bn.weight for u in units] + [u.bn.bias for u in units]
        return _FakeFn.apply(self, x, *ps)


class _FakeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trunk, x, *params):
        ctx.trunk = trunk
        ctx.save_for_backward(x, *params)
        s = x.mean(dim=1, keepdim=True)                      # (B,1)
        out = sum((p * (i + 1)).sum() for i, p in enumerate(params)) * s
        return out.expand(-1, 2).contiguous()

    @staticmethod
    def backward(ctx, g):
        x, *params = ctx.saved_tensors
        coef = (g.sum(dim=1, keepdim=True) * x.mean(dim=1, keepdim=True)).sum()
        sizes = [p.numel() for p in params]
        flat = torch.empty(sum(sizes))
        grads, o = [], 0
        for i, (p, s) in enumerate(zip(params, sizes)):
            v = flat[o:o + s].view(p.shape); v.fill_(float(i + 1)); v.mul_(coef); grads.append(v); o += s
        hook = ctx.trunk.grad_segment_hook
        if hook is not None:
            for st in (4, 3, 2, 1, 0):
                hook(st, flat, grads)
        return (None, None) + tuple(grads)


def cpu_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from src.distributed import GradAllReducer, all_ranks_finite, broadcast_module_state, dp_train_step
    torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
    model = torch.nn.Sequential(FakeTrunk(), torch.nn.Linear(2, 2))
    broadcast_module_state(model, 0)
    red = GradAllReducer(model)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.randn(4, 5, generator=g); y = torch.randint(0, 2, (4,), generator=g)
    loss_fn = lambda o, t: torch.nn.functional.cross_entropy(o, t, reduction="sum")
    # local (un-reduced) gradients for the oracle
    model.zero_grad()
    model[0].grad_segment_hook = None
    loss_fn(model(x), y).backward()
    local = [p.grad.clone() for p in model.parameters()]
    model[0].grad_segment_hook = red._segment_hook
    loss, out, ok = dp_train_step(model, red, opt, loss_fn, x, y, max_norm_grad=None)
    ok = bool(ok.item() == 1.0)
    reduced = [p.grad.clone() for p in model.parameters()]
    finite_all = all_ranks_finite(torch.tensor(float("nan") if rank == 1 else 1.0))
    torch.save({"local": local, "reduced": reduced, "params": [p.detach().clone() for p in model.parameters()],
                "ok": ok, "finite_all": finite_all}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def hardening_worker(rank, world, port, out_dir):
    """World 3 / 4 over gloo: (1) the collective skip decision with the "not finite" indicator is exact at a world size that is
    not a power of two; (2) a padded last batch (DistributedSampler repeats samples to fill the ranks) and a rank whose shard holds
    no positive label still issue the SAME list of collectives, in the same order and with the same sizes, on every rank;
    (3) a parameter that takes no part keeps a zero gradient (documented deviation)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from src.distributed import GradAllReducer, broadcast_module_state, dp_train_step
    torch.manual_seed(300 + rank)
    model = torch.nn.Sequential(FakeTrunk(), torch.nn.Linear(2, 2))
    unused = torch.nn.Linear(3, 3)                          # registered, never called
    model.add_module("unused", unused)
    model.forward = lambda x: model[1](model[0](x))
    broadcast_module_state(model, 0)
    red = GradAllReducer(model)
    red.log_messages = True
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    loss_fn = lambda o, t: torch.nn.functional.cross_entropy(o, t, reduction="sum")
    # 10 samples over `world` ranks, batch 2: the sampler pads to a multiple of the world size; labels are positive only for
    # indices that rank 0 never draws without shuffling (index % world == 0 -> label 0)
    n = 10
    g = torch.Generator().manual_seed(5)
    X = torch.randn(n, 5, generator=g)
    Y = torch.tensor([0 if i % world == 0 else 1 for i in range(n)])
    ds = torch.utils.data.TensorDataset(X, Y)
    sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=False)
    loader = DataLoader(ds, batch_size=2, sampler=sampler)
    seen, oks, msgs = [], [], []
    for step, (x, y) in enumerate(loader):
        if step == 1 and rank == world - 1:
            x = x.clone(); x[0, 0] = float("nan")          # one rank's loss is not finite on step 1
        red.messages = []
        before = [p.detach().clone() for p in model.parameters()]
        _, _, ok = dp_train_step(model, red, opt, loss_fn, x, y, max_norm_grad=None)
        oks.append(float(ok.item()))
        msgs.append(list(red.messages))
        seen.append((y.tolist(), all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))))
    torch.save({"oks": oks, "msgs": msgs, "seen": seen, "params": [p.detach().clone() for p in model.parameters()],
                "unused_grad_zero": bool(unused.weight.grad is not None and float(unused.weight.grad.abs().sum()) == 0.0),
                "nsamples": len(sampler)}, os.path.join(out_dir, f"hard_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def gpu_worker(rank, world, port, out_dir):
    """Real trunk on the GPU (both ranks share cuda:0), gradients exchanged over gloo: checks the stage-hook /
    flat-buffer protocol of _plan.TrunkFunction end to end against the mean of per-shard gradients."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import r2plus1d as orc
    from src.distributed import GradAllReducer, broadcast_module_state, dp_train_step
    from src.loss import FocalLoss
    from src.models.R2Plus1D import R2Plus1DClassifier
    dev = torch.device("cuda:0")
    ls, B, T, S, alpha = [1, 1, 1, 1], 3, 4, 32, 0.01
    torch.manual_seed(5 + rank)
    model = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=ls, alpha=alpha).to(dev)
    broadcast_module_state(model, 0)
    red = GradAllReducer(model)
    x = orc.synth_clip(B, T, S, 40 + rank).to(dev); y = orc.synth_labels(B, 40 + rank).to(dev)
    loss_fn = FocalLoss(weight=torch.ones(2), gamma=2.0)
    model.train()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # local gradients with the hook off (restore BN buffers afterwards so both passes see the same state)
    model.res2plus1d.grad_segment_hook = None
    model.zero_grad()
    loss_fn(model(x), y).backward()
    local = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
    model.load_state_dict(sd0)
    model.res2plus1d.grad_segment_hook = red._segment_hook
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    dp_train_step(model, red, opt, loss_fn, x, y, max_norm_grad=None)
    torch.cuda.synchronize()
    reduced = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
    # ---- the sync-free step with src.optim.ClipAdamW: a batch whose loss is non-finite on ONE rank is skipped on BOTH
    # (device-side flag inside the gradient bucket, md_opt_adamw_step_if), the next batch is applied on both
    from src.optim import ClipAdamW
    model.load_state_dict(sd0)
    copt = ClipAdamW(model.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in model.parameters()]
    xbad = x.clone()
    if rank == 1:
        xbad[0, 0, 0, 0, 0] = float("nan")
    _, _, ok_bad = dp_train_step(model, red, copt, loss_fn, xbad, y, max_norm_grad=1.0)
    torch.cuda.synchronize()
    skipped_unchanged = all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
    model.load_state_dict(sd0)                  # (BatchNorm running statistics saw the NaN on rank 1: restore)
    _, _, ok_good = dp_train_step(model, red, copt, loss_fn, x, y, max_norm_grad=1.0)
    torch.cuda.synchronize()
    after = [p.detach().cpu().clone() for p in model.parameters()]
    changed = sum(0 if torch.equal(a.cpu(), p) else 1 for a, p in zip(before, after))
    sd = copt.state_dict()                      # settles the pending flags: the skipped step does not count
    steps = sorted({int(st["step"]) for st in sd["state"].values()})
    torch.save({"local": local, "reduced": reduced, "ok_bad": float(ok_bad.item()), "ok_good": float(ok_good.item()),
                "skipped_unchanged": skipped_unchanged, "changed": changed, "after": after, "steps": steps},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


class LoggedDataset(torch.utils.data.Dataset):
    """(x, y) pairs whose accesses are logged per process; sample `nan_index` is all-NaN (a non-finite loss on whichever rank
    draws it)."""

    def __init__(self, n, nan_index=-1, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(n, 6, generator=g)
        self.y = (self.x[:, 0] + 0.3 * self.x[:, 1] > 0).long()
        if nan_index >= 0:
            self.x[nan_index] = float("nan")
        self.log = []

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        self.log.append(int(i))
        return self.x[i], self.y[i]


def loop_worker(rank, world, port, out_dir):
    """src.distributed.train_per_proc end to end over gloo (CPU): the epoch loop, the rank partition, the rank-0 checkpoints,
    the collective non-finite skip.  The model is a plain torch module (the loop is model-agnostic; the GPU trunk's own
    exchange is covered by gpu_worker)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    from src import distributed as D
    torch.manual_seed(1000 + rank)                          # different init per rank: the broadcast must fix it
    model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 2))
    train = LoggedDataset(24, nan_index=5, seed=1); valid = LoggedDataset(8, seed=2)
    # snapshots of the parameters after every optimisation step (hook on the optimizer the loop creates)
    snaps = []
    orig_adamw = torch.optim.AdamW

    class Spy(orig_adamw):
        def step(self, *a, **k):
            r = super().step(*a, **k)
            snaps.append([p.detach().clone() for g in self.param_groups for p in g["params"]])
            return r

    torch.optim.AdamW = Spy
    try:
        hist = D.train_per_proc(rank, world, 4, model, train, valid, random_seed=7, resume=False,
                                loss_fn=torch.nn.CrossEntropyLoss(reduction="sum"),
                                model_filepath=os.path.join(out_dir, "last.pt"), num_epoch=3, verbose=None,
                                save_best_only=False, save_best_dir=os.path.join(out_dir, "best.pt"))
    finally:
        torch.optim.AdamW = orig_adamw
    torch.save({"hist": hist, "train_log": train.log, "valid_log": valid.log, "steps": len(snaps),
                "params": [p.detach().clone() for p in model.parameters()]}, os.path.join(out_dir, f"loop_rank{rank}.pt"))
