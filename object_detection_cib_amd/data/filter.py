"""filter_dataset (mirror of kod/data/filter.py:10-45): keep the targets of the named classes and the samples that
still have at least one target."""
from __future__ import annotations

from .cache import DatasetInfo, SampleInfo


def filter_dataset(ds_info: DatasetInfo, new_name: str, classes_to_include: list) -> DatasetInfo:
    for c in classes_to_include:
        if c not in ds_info.classes:
            raise ValueError(f"{c} is not in the original dataset!")
    keep = set(classes_to_include)
    samples = []
    for s in ds_info.samples:
        tg = [t for t in s.targets if t.class_name in keep]
        if tg:
            samples.append(SampleInfo(id=s.id, image_path=s.image_path, image_metadata=s.image_metadata, targets=tg))
    return DatasetInfo(name=new_name, date=ds_info.date, classes=classes_to_include, samples=samples)
