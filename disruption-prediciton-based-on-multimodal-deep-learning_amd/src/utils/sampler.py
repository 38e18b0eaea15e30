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
