"""GPU: two ranks (sharing the one card) run the real R(2+1)D trunk; gradients are exchanged stage by stage through
the executor's segment hook.  Every rank must end with the mean of the per-shard gradients (derived DP oracle)."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from tests import dp_workers

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_trunk_stage_hook_allreduce_two_ranks():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(dp_workers.gpu_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"), weights_only=True)
        r1 = torch.load(os.path.join(d, "rank1.pt"), weights_only=True)
    for k in r0["local"]:
        mean = (r0["local"][k] + r1["local"][k]) / 2
        scale = float(mean.abs().max()) + 1e-12
        assert float((r0["reduced"][k] - mean).abs().max()) / scale < 1e-5, k
        assert torch.equal(r0["reduced"][k], r1["reduced"][k]), k
    # sync-free step: the non-finite batch (NaN on rank 1 only) was skipped on both ranks, the next one applied on both
    for r in (r0, r1):
        assert r["ok_bad"] != 1.0 and r["ok_good"] == 1.0
        assert r["skipped_unchanged"] and r["changed"] >= len(r["after"]) - 1     # (linear.0.bias: zero gradient in front of BatchNorm1d, moved by weight decay only)
        assert r["steps"] == [1]                      # the skipped step does not advance Adam's step count
    for a, b in zip(r0["after"], r1["after"]):
        assert torch.equal(a, b)                      # replicas identical after the applied step


@pytest.mark.gpu
def test_bench_data_parallel_path_runs_on_rccl_with_one_rank():
    """The N>1 leg of bench.py (RCCL process group, broadcast, GradAllReducer with its side-stream ordering, dp_train_step with the
    finite flag in the last bucket and the device-side skip) on the real `nccl` backend -- with ONE rank, which is all a one-GPU box
    can host: every collective is a real RCCL call on this GPU.  The step time must be that of the single-process step (the
    all-reduces are no-ops in data volume) and the JSON line well formed."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MD_BENCH_FORCE_DP="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29519",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "3", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:6]          # ONE JSON line on stdout (RCCL's version banner must not be there)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "forced" in d["config"]["parallelism"] and d["value"] > 0
    assert d["ms_per_step"] < 12.0, d["ms_per_step"]
