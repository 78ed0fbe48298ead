"""Index samplers (host logic; mirror kod/data/samplers.py:17-138).  Pure index generation with torch RNG streams
identical to the reference: RandomCycleSampler / ClassAwareSampler use the global torch generator,
RepeatFactorSampler a torch.Generator seeded with 2023."""
from __future__ import annotations

import math
from typing import Iterator, Optional, Sequence

import torch
from torch.utils.data import Sampler, WeightedRandomSampler


class RandomCycleSampler(Sampler):
    """samplers.py:17-38: endless shuffled cycle over a list."""

    def __init__(self, data: Sequence[int], generator: Optional[torch.Generator] = None):
        self.data = list(data)
        self.length = len(self.data)
        self.generator = generator
        self.indices = torch.randperm(self.length, generator=generator)
        self.current_index = 0

    def __iter__(self):
        return self

    def __len__(self):
        return self.length

    def __next__(self):
        if self.current_index == self.length:
            self.indices = torch.randperm(self.length, generator=self.generator)
            self.current_index = 0
        idx = self.data[int(self.indices[self.current_index])]
        self.current_index += 1
        return idx


class ClassAwareSampler(Sampler):
    """samplers.py:41-77: uniform class -> next image of that class (cyclic reshuffle)."""

    def __init__(self, class_to_images: Sequence[Sequence[int]], num_samples: int):
        """class_to_images[c] = dataset indices of the images that contain class c (the reference derives it
        with filter_dataset); construction order = reference's RNG order (label cycle first, then classes)."""
        self.num_samples = num_samples
        self.label_iter_list = RandomCycleSampler(list(range(len(class_to_images))))
        self.data_iter_dict = {c: RandomCycleSampler(list(ix)) for c, ix in enumerate(class_to_images)}

    def __iter__(self) -> Iterator[int]:
        indices = []
        while len(indices) < self.num_samples:
            label_index = next(self.label_iter_list)
            indices.append(next(self.data_iter_dict[label_index]))
        self.sampler_indices = indices
        return iter(indices)

    def __len__(self):
        return self.num_samples


def image_repeat_factors(image_target_classes: Sequence[Sequence[int]], class_instance_count: Sequence[int],
                         threshold: float = 1.0, reduction: Optional[str] = None, use_sqrt: bool = True):
    """samplers.py:88-128: f_c = instances_c / total; r_c = max(1, t / f_c) (sqrt if use_sqrt); per image the
    mean over its TARGETS (sum / (n + 1e-6)) or, with reduction == "max", the maximum."""
    total = float(sum(class_instance_count))
    rc = []
    for n in class_instance_count:
        v = max(1.0, threshold / (n / total))
        rc.append(math.sqrt(v) if use_sqrt else v)
    out = []
    for cls in image_target_classes:
        s, mx = 0.0, 0.0
        for c in cls:
            s += rc[c]
            mx = max(mx, rc[c])
        out.append(mx if reduction == "max" else s / (len(cls) + 1e-6))
    return out


class RepeatFactorSampler(WeightedRandomSampler):
    """samplers.py:80-138: weighted sampling with replacement, generator seed 2023."""

    def __init__(self, repeat_factors: Sequence[float], seed: int = 2023):
        self.image_repeat_factors = list(repeat_factors)
        self.generator = torch.Generator()
        self.generator.manual_seed(seed)
        super().__init__(torch.tensor(self.image_repeat_factors), num_samples=len(self.image_repeat_factors),
                         replacement=True, generator=self.generator)
