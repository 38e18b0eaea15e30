"""Oracle (test infrastructure): NumPy restatement of the reference's ImbalancedDatasetSampler index stream.

Reference: src/utils/sampler.py:5-35 -- weights 1 / count(label) as a float64 tensor, one
``torch.multinomial(weights, num_samples, replacement=True)`` per epoch.  ATen's CPU kernel for that call
(aten/src/ATen/native/cpu/MultinomialKernel.cpp, torch 2.10) normalises a sequential running sum into a cumulative
distribution (last entry forced to 1) and, per draw, takes a uniform double from the generator and returns the leftmost
category whose cumulative probability is >= it.  ``torch.rand(n, dtype=float64, generator=g)`` draws from the same
uniform_real_distribution<double> serially, so it yields exactly the variates multinomial would consume
(tests/test_sampler.py pins this against torch.multinomial itself)."""
import numpy as np
import torch


def class_weights(labels) -> np.ndarray:
    labels = np.asarray(labels)
    _, inverse, counts = np.unique(labels, return_inverse=True, return_counts=True)
    return 1.0 / counts[inverse].astype(np.float64)


def resampled_indices(weights: np.ndarray, num_samples: int, generator: torch.Generator = None) -> np.ndarray:
    cum = np.cumsum(np.asarray(weights, dtype=np.float64))
    cum = cum / cum[-1]
    cum[-1] = 1.0
    u = torch.rand(num_samples, dtype=torch.float64, generator=generator).numpy()
    return np.searchsorted(cum, u, side="left")


def shard(stream: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Rank r's share of a per-epoch index stream: r, r + W, ... (DistributedSampler's partition, src/distributed.py:21)."""
    return stream[rank::world]
