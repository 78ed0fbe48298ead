"""Index samplers - drop-ins for kod.data.samplers (kod/data/samplers.py:17-138; `_target_`s of
kod/configs/data/class_aware.yaml and repeat_factor.yaml).  Same constructors (a DatasetInfo-shaped object), same
torch RNG streams: RandomCycleSampler / ClassAwareSampler draw from the global torch generator in the reference's
construction order, RepeatFactorSampler from a torch.Generator seeded with 2023.  Pure host index logic; pinned to
the reference's own outputs by tests/golden/samplers.npz."""
from __future__ import annotations

import math
from typing import Iterator, Optional, Sequence

import torch
from torch.utils.data import Sampler, WeightedRandomSampler

from .filter import filter_dataset


class RandomCycleSampler:
    """Endless iterator over `data` in shuffled passes (kod/data/samplers.py:17-38).  RNG protocol of the reference: one
    `torch.randperm(len(data))` when the object is built and one more each time a pass is used up - drawn lazily, by the
    `next()` that starts the new pass - from `generator` (None = torch's global generator)."""

    def __init__(self, data: Sequence[int], generator: Optional[torch.Generator] = None):
        self.data, self.generator = data, generator
        self._pass = self._draw_pass()
        self._served = 0                    # items handed out from the current pass

    def _draw_pass(self):
        return torch.randperm(len(self.data), generator=self.generator).tolist()

    def __len__(self) -> int:
        return len(self.data)

    def __iter__(self):
        return self

    def __next__(self) -> int:
        if self._served >= len(self._pass):
            self._pass, self._served = self._draw_pass(), 0
        item = self.data[self._pass[self._served]]
        self._served += 1
        return item


class ClassAwareSampler(Sampler):
    """samplers.py:41-77: uniform class -> next image of that class (cyclic reshuffle); one epoch = len(samples)
    draws; `sampler_indices` is the side channel DetectionDataset's mosaic reads (kod/data/detection.py:114-122)."""

    def __init__(self, dataset_info):
        self.dataset_info = dataset_info
        position = {s.id: i for i, s in enumerate(dataset_info.samples)}
        self.label_to_index = {name: i for i, name in enumerate(dataset_info.classes)}
        self.label_iter_list = RandomCycleSampler(list(self.label_to_index.values()))
        self.data_iter_dict = {}
        for name, i in self.label_to_index.items():
            members = filter_dataset(ds_info=dataset_info, new_name=name, classes_to_include=[name]).samples
            self.data_iter_dict[i] = RandomCycleSampler([position[s.id] for s in members])

    def __iter__(self) -> Iterator[int]:
        indices = []
        while len(indices) < len(self.dataset_info.samples):
            indices.append(next(self.data_iter_dict[next(self.label_iter_list)]))
        self.sampler_indices = indices
        return iter(indices)

    def __len__(self) -> int:
        return len(self.dataset_info.samples)


def class_repeat_factors(dataset_info, threshold: float = 1.0, use_sqrt: bool = True) -> dict:
    """samplers.py:88-109: f_c = instances_c / total; r_c = max(1, t / f_c), square-rooted if use_sqrt."""
    count = dataset_info.get_instance_count()
    total = sum(count.values())
    out = {}
    for k in dataset_info.classes:
        r = max(1.0, threshold / (count[k] / total))
        out[k] = math.sqrt(r) if use_sqrt else r
    return out


class RepeatFactorSampler(WeightedRandomSampler):
    """samplers.py:80-138: weighted sampling with replacement; image weight = mean (sum / (n + 1e-6)) or, with
    reduction == "max", maximum of its targets' class repeat factors; generator seed 2023."""

    def __init__(self, dataset_info, reduction: Optional[str] = None, threshold: float = 1.0, use_sqrt: bool = True):
        self.dataset_info = dataset_info
        rc = class_repeat_factors(dataset_info, threshold, use_sqrt)
        self.image_repeat_factors = []
        for sample in dataset_info.samples:
            total, biggest = 0.0, 0.0
            for t in sample.targets:
                total += rc[t.class_name]
                biggest = max(biggest, rc[t.class_name])
            self.image_repeat_factors.append(biggest if reduction == "max" else total / (len(sample.targets) + 1e-6))
        self.generator = torch.Generator()
        self.generator.manual_seed(2023)
        super().__init__(torch.tensor(self.image_repeat_factors), num_samples=len(dataset_info.samples),
                         replacement=True, generator=self.generator)
