"""Re-sampling for class imbalance: host-side counterpart of the reference's ``src/utils/sampler.py`` (same class name,
constructor and attributes, so ``train_*`` scripts that build ``ImbalancedDatasetSampler(dataset)`` keep working).

Every index is drawn with replacement with probability proportional to 1 / (number of samples sharing its label), which is
what the reference does (:5-35); the weight vector is a float64 tensor and the draw is a single ``torch.multinomial`` call on
it, so for the same torch RNG state the index stream is identical.  The per-label counts are taken with ``numpy.unique``
instead of a Python dictionary pass.
"""
from typing import Callable, Iterator, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.sampler import Sampler


class ImbalancedDatasetSampler(Sampler):
    def __init__(self, dataset: Dataset, indices: Optional[Sequence[int]] = None, num_samples: Optional[int] = None,
                 callback_get_label: Optional[Callable] = None):
        self.callback_get_label = callback_get_label
        self.indices = indices if indices is not None else list(range(len(dataset)))
        self.num_samples = num_samples if num_samples is not None else len(self.indices)
        labels = np.asarray([self._get_label(dataset, i) for i in self.indices])
        _, inverse, counts = np.unique(labels, return_inverse=True, return_counts=True)
        self.weights = torch.from_numpy(1.0 / counts[inverse].astype(np.float64))

    def _get_label(self, dataset, idx):
        return dataset.labels[idx]

    def __iter__(self) -> Iterator[int]:
        drawn = torch.multinomial(self.weights, self.num_samples, replacement=True)
        return iter([self.indices[k] for k in drawn.tolist()])

    def __len__(self) -> int:
        return self.num_samples

    # ------------------------------------------------------------------ device-side, rank-sharded form
    def cumulative_distribution(self) -> torch.Tensor:
        """Normalised cumulative distribution of the weights, float64, built exactly as ATen's CPU multinomial builds it
        (sequential running sum, divided by the total, last entry forced to 1)."""
        cum = np.cumsum(self.weights.numpy())
        cum = cum / cum[-1]
        cum[-1] = 1.0
        return torch.from_numpy(cum)

    def device_indices(self, device, rank: int = 0, world_size: int = 1, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """This rank's share of one epoch's resampled index stream as an int64 tensor ON `device`: draws rank, rank + world_size,
        ... of the stream ``torch.multinomial(self.weights, num_samples, replacement=True)`` would produce from the same CPU
        generator state (default generator unless given) -- index-exact, all ranks must hold the same generator state.  The
        uniform variates come from the host generator (a serial Mersenne-twister stream, num_samples doubles); the search runs
        in ``md_multinomial_shard``.  Labels or data gathered with these indices never leave the device."""
        import ctypes as C
        from .. import _native as N
        if not torch.device(device).type == "cuda":
            raise RuntimeError("ImbalancedDatasetSampler.device_indices: a CUDA device is required (no CPU fallback)")
        u = torch.rand(self.num_samples, dtype=torch.float64, generator=generator)       # the draws multinomial would make
        L = N.lib()
        n_out = L.md_multinomial_shard_count(self.num_samples, rank, world_size)
        out = torch.empty(n_out, dtype=torch.int64, device=device)
        if not hasattr(self, "_dev_tables") or self._dev_tables[0] != torch.device(device):
            imap = torch.as_tensor(self.indices, dtype=torch.int64)
            self._dev_tables = (torch.device(device), self.cumulative_distribution().to(device), imap.to(device))
        _, cum, imap = self._dev_tables
        ud = u.to(device, non_blocking=True)
        N.check(L.md_multinomial_shard(C.c_void_p(cum.data_ptr()), cum.numel(), C.c_void_p(ud.data_ptr()), self.num_samples,
                                       rank, world_size, C.c_void_p(imap.data_ptr()), C.c_void_p(out.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "md_multinomial_shard")
        return out
