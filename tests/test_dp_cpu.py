"""CPU, world_size 2 over gloo: the data-parallel gradient exchange of src/distributed.py.

The oracle for data parallelism is derived (the reference's DP sketch never exchanges gradients): after a
step every rank must hold the MEAN over ranks of the per-rank local gradients, identical parameters, and the
non-finite-loss decision must be collective."""
import os
import socket
import tempfile

import torch
import torch.multiprocessing as mp

from tests import dp_workers


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_grad_allreduce_world2_gloo():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(dp_workers.cpu_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"), weights_only=True)
        r1 = torch.load(os.path.join(d, "rank1.pt"), weights_only=True)
    assert r0["ok"] and r1["ok"]
    for l0, l1, g0, g1 in zip(r0["local"], r1["local"], r0["reduced"], r1["reduced"]):
        mean = (l0 + l1) / 2
        assert torch.allclose(g0, mean, rtol=1e-6, atol=1e-7)
        assert torch.equal(g0, g1)                       # every rank ends with the same reduced gradient
    for p0, p1 in zip(r0["params"], r1["params"]):
        assert torch.equal(p0, p1)                       # broadcast + identical update keep replicas in sync
    assert r0["finite_all"] is False and r1["finite_all"] is False      # one NaN rank stops every rank
