#!/usr/bin/env python3
"""Benchmark of the hot path: R(2+1)D `[1,2,2,1]` training step on synthetic IVIS-shaped clips.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one per-GPU batch (B=8, 3, 21, 128, 128) already resident in HBM:
forward + FocalLoss + backward (+ gradient all-reduce over RCCL when N>1) + clip_grad_norm(1.0) + AdamW, i.e. the
body of the reference's train_per_epoch (src/train.py:40-66).  Rank 0 prints ONE JSON line (see DESIGN.md for the
field definitions, the roofline accounting and the cpu_baseline sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "disruption-prediciton-based-on-multimodal-deep-learning_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

# The data-parallel step keeps three streams busy (compute chain, weight gradients, RCCL).  The HIP runtime multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4); with 4 the RCCL stream's cross-stream waits land in the compute chain's
# queue and stall it at every stage boundary (6.54 ms per step against 6.29 with 2, 6, 8, 12 or 16 queues; the single-process
# step does not care).  Must be set before the runtime initialises, i.e. before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Kernel arguments in device memory (the runtime's default on this image; stated here because the step is a chain of ~290 dependent
# launches and host-memory kernargs cost 0.5 ms of it: HIP_FORCE_DEV_KERNARG=0 measured 6.68 ms against 6.15).
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

B_PER_GPU, T, S = 8, 21, 128
LAYERS, ALPHA = [1, 2, 2, 1], 0.01
ALG_BYTES_PER_CLIP_FP32 = 951.9e6      # ideal-fusion conv I/O, fwd+bwd, fp32 storage
ALG_FLOP_PER_CLIP = 68.24e9            # 3 x 22.75 GFLOP
PEAK_16BIT_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16/fp16 MFMA
PRODUCTS_PER_MULTIPLY = 3              # split arithmetic: hi*hi + hi*lo + lo*hi
PEAK_SPLIT_TFLOPS = PEAK_16BIT_MFMA_TFLOPS / PRODUCTS_PER_MULTIPLY
PEAK_HBM_GBS = 8000.0
EVENT_EVERY = 20                        # timed steps between two steps whose conv launches are timed with HIP events


def synth_batch(device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randint(0, 256, (B_PER_GPU, 3, T, S, S), generator=g).float()
    x -= torch.tensor([90.0, 98.0, 102.0]).view(1, 3, 1, 1, 1)       # BGR means, src/dataset.py:201-205
    y = (torch.rand(B_PER_GPU, generator=g) >= 0.05).long()
    y[0], y[1] = 0, 1                                                  # both classes present
    return x.to(device), y.to(device)


def cpu_baseline(steps=5, warmup=2):
    """The oracle (CPU restatement of the reference, oracle/) timed on the host cores: same step definition,
    same shapes, bounded sample.  Reported beside the GPU number; never the thing optimised."""
    from oracle import losses as ol, r2plus1d as orc, step as ostep
    # the GPU box gives one GPU's share of the host (16 cores); more threads than cores only thrash
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncores)))
    params, bufs = orc.synth_state(LAYERS, 1234, ALPHA)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.AdamW(list(leaves.values()), lr=2e-4)
    x = orc.synth_clip(B_PER_GPU, T, S, 1234)
    y = orc.synth_labels(B_PER_GPU, 1234)
    one = torch.ones(2)

    def step():
        opt.zero_grad()
        out = orc.classifier_forward(x, leaves, bufs, LAYERS, ALPHA, True)
        loss = ol.focal_loss(out, y, one, 2.0)
        loss.backward()
        ostep.clip_grad_norm([p.grad for p in leaves.values()], 1.0)
        opt.step()

    for _ in range(warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return {"value": round(B_PER_GPU * steps / dt, 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} full steps (after {warmup} warm-up) of the same B=8 (3,21,128,128) workload, fp32, "
                      f"oracle/ on torch-CPU, anomaly mode off"}


def pmc_traffic(dom):
    """HBM bytes per launch of the dominant kernel family, from the newest committed PMC pass
    (profiles/*_pmc_hbm_traffic.json, written by tools/pmc_traffic.sh + tools/pmc_traffic_parse.py: separate
    FETCH_SIZE / WRITE_SIZE passes of this same bench, read bytes = 2 x FETCH_SIZE on gfx950).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")))
    if not files:
        return None, "no PMC pass committed"
    fams = json.load(open(files[-1]))["families"]
    doms = ("k_conv_patch", "k_conv_pers") if dom == "k_conv_patch" else (dom,)     # the persistent form is the same family
    sel = [v for k, v in fams.items() if k.startswith(doms)]
    n = sum(v["launches_per_step"] for v in sel)
    if not n:
        return None, "kernel family not in " + os.path.basename(files[-1])
    b = sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in sel)
    return round(b / n), "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) from profiles/" + os.path.basename(files[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="experiment: no HIP events around the conv launches (roofline fields become meaningless)")
    ap.add_argument("--torch-optimizer", action="store_true", help="torch clip_grad_norm_ + fused AdamW instead of src.optim.ClipAdamW")
    ap.add_argument("--composed-step", action="store_true", help="the step as model() / loss / backward() / optimizer.step() from Python instead of the one-call md_plan_train_step (src/_step.py; same kernels, bit-identical)")
    args = ap.parse_args()

    # The GPU box exposes every hardware thread of the host but grants one GPU's share of CPU time (cgroup quota: 16 CPUs).  A torch
    # CPU op that fans out over all visible threads spends that quota within milliseconds and the whole process - including the
    # thread that queues GPU work - is then throttled for the rest of the 100 ms scheduler period (measured: 88 ms stalls in
    # tools/latency.py until OMP threads were limited).  Bound the intra-op pool before anything runs.
    try:
        torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    except Exception:
        torch.set_num_threads(16)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    # MD_BENCH_FORCE_DP=1: run the data-parallel code path (RCCL process group, stage-wise all-reduce, sync-free skip) with
    # whatever WORLD_SIZE is -- with one rank it is how the RCCL path can be exercised on a one-GPU box
    dp = world > 1 or os.environ.get("MD_BENCH_FORCE_DP") == "1"
    saved_stdout = None
    if dp:
        # RCCL prints a five-line version banner on STDOUT when a communicator is created (lazily, at the first collective of a
        # stream): the contract is ONE JSON line on stdout, so file descriptor 1 points at stderr until rank 0 prints its line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from src.models.R2Plus1D import R2Plus1DClassifier
    from src.loss import FocalLoss
    from src.distributed import GradAllReducer, broadcast_module_state, dp_train_step

    torch.manual_seed(1234)
    model = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=LAYERS, alpha=ALPHA).to(device)
    model.train()
    reducer = None
    if dp:
        broadcast_module_state(model, 0)
        reducer = GradAllReducer(model)
    loss_fn = FocalLoss(weight=torch.ones(2), gamma=2.0)
    from src.optim import ClipAdamW
    if args.torch_optimizer:
        opt = torch.optim.AdamW(model.parameters(), lr=2e-4, fused=True)
    else:
        opt = ClipAdamW(model.parameters(), lr=2e-4)      # same arithmetic as clip_grad_norm_ + AdamW, three launches
    x, y = synth_batch(device, 1234 + rank)
    finite = torch.ones((), device=device)
    # N = 1: the step the training loop itself takes for this (model, loss, optimizer) triple (src/train.py::train_per_epoch ->
    # src/_step.py::FusedTrainStep -> md_plan_train_step): forward, Focal loss, backward, clip and AdamW queued by ONE C call
    fused = None
    if reducer is None and not args.torch_optimizer and not args.composed_step and os.environ.get("MD_FUSED_STEP", "1") != "0":
        from src._step import FusedTrainStep
        fused = FusedTrainStep(model, loss_fn, opt)

    def step():
        nonlocal finite
        if reducer is not None:
            # N > 1: the function the data-parallel loop itself uses (src/distributed.py: stage-wise all-reduce of the trunk
            # gradients during backward, one bucket for the rest + the collective finite flag, device-side skip; no host sync)
            loss, _, ok = dp_train_step(model, reducer, opt, loss_fn, x, y, max_norm_grad=1.0)
            finite = finite * ok.reshape(())
            return
        if fused is not None:
            _, _, _, ok = fused(x, y, max_norm=1.0)
            finite = finite * ok                                      # the device flag of the finite-loss guard: read after the timed region
            return
        opt.zero_grad(set_to_none=True)
        logits = model(x)
        loss = loss_fn(logits, y)
        loss.backward()
        if args.torch_optimizer:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
        else:
            opt.step(max_norm=1.0)
        finite = finite * torch.isfinite(loss.detach()).float()       # checked after the timed region, no host sync here

    for _ in range(args.warmup):
        step()
    rccl_messages = None
    if reducer is not None:
        # one extra untimed step with the reducer's message log on: the collectives of a step (tag, bytes) go to stderr next to the
        # JSON line, so that a scaling record can be read against DESIGN section 7 ("5 + 1 + 1" messages)
        reducer.log_messages = True; reducer.messages = []
        step()
        rccl_messages = list(reducer.messages)
        reducer.log_messages = False
    plan = model.res2plus1d._plans[(B_PER_GPU, T, S, S)]
    # Kernel durations for the roofline object come from HIP events stamped by the conv launches themselves (hipExtLaunchKernel's
    # start / stop events, csrc/common.h::md_klaunch).  A timed launch does not overlap its neighbours' dispatch, which costs a sampled
    # step ~0.45 ms (6.23 vs 5.76 ms), so the timed region samples one step in EVENT_EVERY (step 10, 30, ...: 63 launches of the
    # dominant family each): the events are live and inside the timed region, the perturbation ~0.4 %.
    plan.profile_enable(True); plan.profile_enable(False)        # forget anything recorded during warm-up
    if not args.no_kernel_events:
        # the event pairs of every sampled step exist before the clock starts (created on first use they cost the host 11-13 ms per
        # sampled step: 380 hipEventCreate calls, long enough for the GPU queue to run dry)
        plan.profile_reserve((args.steps // EVENT_EVERY + 1) * 3 * plan.num_units + 64)
    sampled = [0]

    def events_for(i):
        if args.no_kernel_events:
            return
        on = i % EVENT_EVERY == min(EVENT_EVERY // 2, args.steps - 1)      # (a step in the middle of each window: 10, 30, ...)
        plan.profile_enable(on, keep=True)
        if on:
            sampled[0] += 1

    def fence():
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    # per-step trace: one (timing) event per step on the compute stream + the host time at which the step was queued;
    # both are read after the timed region (no synchronisation inside it)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_t = [0.0] * (args.steps + 1)
    fence()
    t0 = time.perf_counter()
    step_ev[0].record(); host_t[0] = t0
    for i in range(args.steps):
        events_for(i)
        step()
        step_ev[i + 1].record(); host_t[i + 1] = time.perf_counter()
    fence()
    dt = time.perf_counter() - t0
    gpu_ms = sorted(step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps))
    host_ms = sorted((host_t[i + 1] - host_t[i]) * 1e3 for i in range(args.steps))
    first_gpu_ms = step_ev[0].elapsed_time(step_ev[1])

    def pct(v, q):
        return round(v[min(len(v) - 1, int(q * len(v)))], 4)
    step_trace = {"gpu_ms": {"first": round(first_gpu_ms, 4), "p10": pct(gpu_ms, 0.1), "median": pct(gpu_ms, 0.5), "p90": pct(gpu_ms, 0.9),
                             "max": round(gpu_ms[-1], 4)},
                  "host_queue_ms": {"p10": pct(host_ms, 0.1), "median": pct(host_ms, 0.5), "p90": pct(host_ms, 0.9),
                                    "max": round(host_ms[-1], 4)},
                  "note": "gpu_ms = interval between the per-step events on the compute stream; host_queue_ms = host time to queue one step"}
    prof = plan.profile_read()
    plan.profile_enable(False)
    if float(finite.item()) != 1.0:
        raise SystemExit("non-finite loss inside the timed region")
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if dp:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        clips = B_PER_GPU * world * args.steps
        value = clips / dt
        names = ["k_conv_patch|k_conv_pers<fp16 split>(forward)", "k_conv_patch|k_conv_pers<bf16 split>(data-gradient)",
                 "k_wgrad2|k_wgrad_patch (slab kernels; the batched slab reduction is one more launch per step)"]
        kern = []
        for (ms, n, fl), nm in zip(prof, names):
            if n:
                kern.append({"kernel": nm, "launches": int(n), "avg_ms": ms / n, "total_ms_per_step": ms / max(1, sampled[0]),
                             "tflops": fl / (ms * 1e-3) / 1e12})
        # Dominant kernel family = the one carrying most of the algorithmic work.  Forward + data-gradient launches are the
        # same kernel template (k_conv_patch / k_conv_pers, 2/3 of the FLOPs); the weight gradients (k_wgrad2, 1/3) run on the
        # same stream since round 3 (one-stream schedule), so the family with the most algorithmic work is also the one
        # with the largest summed duration.
        g_ms = prof[0][0] + prof[1][0]; g_n = prof[0][1] + prof[1][1]; g_fl = prof[0][2] + prof[1][2]
        w_ms, w_n, w_fl = prof[2]
        if g_fl >= w_fl:
            dom, d_ms, d_n, d_fl = "k_conv_patch", g_ms, g_n, g_fl
        else:
            dom, d_ms, d_n, d_fl = "k_wgrad_patch", w_ms, w_n, w_fl
        mfma_tflops = d_fl / (d_ms * 1e-3) / 1e12 if d_ms > 0 else 0.0
        traffic, traffic_note = pmc_traffic(dom)
        # this path is HBM-bound.  Algorithmic bytes of one launch of the dominant family = the family's share
        # of the step's ideal-fusion conv I/O (951.9 MB/clip fp32: forward 1/3, data gradient 1/3, weight gradient 1/3)
        # divided by its launches per step; achieved = those bytes / the average launch duration measured live with HIP events.
        fam_share = (2.0 / 3.0) if dom == "k_conv_patch" else (1.0 / 3.0)
        launches_per_step = d_n / max(1, sampled[0])
        alg_bytes_per_launch = fam_share * ALG_BYTES_PER_CLIP_FP32 * B_PER_GPU / max(1.0, launches_per_step)
        avg_launch_s = d_ms * 1e-3 / max(1, d_n)
        achieved = alg_bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        step_gbs = value / world * ALG_BYTES_PER_CLIP_FP32 / 1e9
        out = {
            "metric": "clips/sec (fwd+bwd) R2Plus1D T=21 128x128", "value": round(value, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 storage; products as 3 fp16 (forward) / bf16 (backward) MFMAs on hi+lo splits, f32 accumulate",
            "data": "synthetic",
            "config": {"workload": "R2Plus1D layer_sizes=[1,2,2,1] alpha=0.01, per-GPU clips (8,3,21,128,128) fp32, "
                                   "forward+FocalLoss(gamma=2)+backward+clip_grad_norm(1.0)+AdamW(2e-4), BN in train mode",
                       "per_gpu_batch": B_PER_GPU, "global_batch": B_PER_GPU * world, "parallelism": f"dp{world}" + (" (data-parallel code path forced)" if dp and world == 1 else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": dom, "launches": int(d_n), "avg_launch_ms": round(d_ms / max(1, d_n), 5),
                         "alg_bytes_per_launch": round(alg_bytes_per_launch),
                         "event_sampling": f"one timed step in {EVENT_EVERY} ({sampled[0]} of {args.steps})",
                         "whole_step_achieved": round(step_gbs, 1), "whole_step_frac": round(step_gbs / PEAK_HBM_GBS, 4),
                         "whole_step_note": "clips/s x 951.9e6 B (SURVEY 8(d), fp32 storage) against 8000 GB/s",
                         "mfma_view": {"achieved_tflops": round(mfma_tflops, 3), "peak_tflops": round(PEAK_SPLIT_TFLOPS, 1),
                                       "frac": round(mfma_tflops / PEAK_SPLIT_TFLOPS, 4),
                                       "note": "algorithmic FLOPs of the family / event time against 2500/3 TFLOP/s "
                                               "(dense 16-bit MFMA peak / 3 products per multiply)"}},
            "kernels": kern,
            "whole_step": {"alg_tflops": round(value / world * ALG_FLOP_PER_CLIP / 1e12, 2),
                           "alg_hbm_gbs_fp32": round(step_gbs, 1),
                           "hbm_frac_of_8TBs": round(step_gbs / PEAK_HBM_GBS, 4)},
            "step_trace": step_trace,
            "step_issue": ("one C call per step (md_plan_train_step via src/_step.py::FusedTrainStep)" if fused is not None else
                           "composed from Python (model() / loss / backward() / optimizer.step())"),
        }
        if rccl_messages is not None:
            out["rccl_messages_per_step"] = {"count": len(rccl_messages), "bytes": int(sum(b for _, b in rccl_messages)),
                                             "list": [[t, int(b)] for t, b in rccl_messages]}
            print("bench | collectives of one step (rank 0): " + ", ".join("%s %d B" % (t, b) for t, b in rccl_messages), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)                    # whatever the teardown prints does not follow the JSON line
    if dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
