"""Re-sampling: the oracle restatement against torch.multinomial itself (CPU), and the device-side,
rank-sharded index stream of src.utils.sampler.ImbalancedDatasetSampler.device_indices against the reference recipe
(index-exact: src/utils/sampler.py:29-32)."""
import numpy as np
import pytest
import torch

from oracle import sampler as osa


class _DS:
    def __init__(self, labels):
        self.labels = labels

    def __len__(self):
        return len(self.labels)


@pytest.mark.parametrize("seed,n0,n1,ns", [(3, 5, 95, 100), (11, 1, 999, 4000), (0, 400, 600, 1000), (5, 7, 7, 3)])
def test_oracle_stream_equals_torch_multinomial(seed, n0, n1, ns):
    labels = [0] * n0 + [1] * n1
    w = osa.class_weights(labels)
    torch.manual_seed(seed)
    ref = torch.multinomial(torch.DoubleTensor(w), ns, replacement=True).numpy()
    torch.manual_seed(seed)
    got = osa.resampled_indices(w, ns)
    assert np.array_equal(got, ref)
    parts = [osa.shard(got, r, 4) for r in range(4)]
    merged = np.empty(ns, dtype=got.dtype)
    for r in range(4):
        merged[r::4] = parts[r]
    assert np.array_equal(merged, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 8])
def test_device_sharded_stream_is_index_exact(world):
    from src.utils.sampler import ImbalancedDatasetSampler
    rng = np.random.default_rng(1)
    labels = (rng.random(5000) < 0.04).astype(int).tolist()            # ~4 % disruptive, as the KSTAR shots
    ds = _DS(labels)
    sub = list(range(17, 4800, 3))                                      # the sampler's `indices` argument
    for indices, ns in ((None, None), (sub, 2500)):
        s = ImbalancedDatasetSampler(ds, indices=indices, num_samples=ns)
        torch.manual_seed(123)
        ref = list(iter(s))                                             # the reference recipe: one multinomial per epoch
        shards = []
        for r in range(world):
            torch.manual_seed(123)                                      # every rank holds the same generator state
            shards.append(s.device_indices("cuda:0", rank=r, world_size=world))
        torch.cuda.synchronize()
        assert all(t.is_cuda and t.dtype == torch.int64 for t in shards)
        for r in range(world):
            assert shards[r].cpu().tolist() == ref[r::world], (world, r)
        # classes re-balanced, and a second epoch continues the generator stream exactly as the reference would
        lab = torch.tensor(labels, device="cuda:0")[torch.cat(shards)]
        assert abs(float(lab.float().mean()) - 0.5) < 0.05
        ref2 = list(iter(s))
        torch.manual_seed(123)
        s.device_indices("cuda:0"); second = s.device_indices("cuda:0")
        assert second.cpu().tolist() == ref2
