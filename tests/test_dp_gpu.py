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
