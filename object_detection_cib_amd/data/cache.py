"""Dataset description types (schema mirrors of kod/data/cache.py:21-147): what the samplers and the data module
read.  Host-side NamedTuples only; (de)serialisation of the reference's pickled caches stays with the reference
(its pickles resolve `kod.data.cache.*` - alias this module under that name to load them)."""
from __future__ import annotations

from datetime import datetime
from typing import NamedTuple


class ImageMetadata(NamedTuple):
    width: int
    height: int
    num_channels: int
    mime_type: str
    size_bytes: int


class TargetInfo(NamedTuple):
    bounding_box: object          # kod.core.bbox.boxes.XYXYBoundingBox-like (x1, y1, x2, y2)
    class_name: str


class SampleInfo(NamedTuple):
    id: str
    image_path: str
    image_metadata: ImageMetadata
    targets: list


class DatasetInfo(NamedTuple):
    name: str
    date: datetime
    classes: list
    samples: list

    def subset(self, num_samples: int) -> "DatasetInfo":
        return DatasetInfo(self.name, self.date, self.classes[:num_samples], self.samples[:num_samples])

    def filter(self, new_name: str, classes_to_include: list) -> "DatasetInfo":
        from .filter import filter_dataset
        return filter_dataset(self, new_name, classes_to_include)

    def get_instance_count(self) -> dict:
        """cache.py:128-147: instances per class, keyed in `classes` order."""
        stats = {c: 0 for c in self.classes}
        for s in self.samples:
            for t in s.targets or ():
                stats[t.class_name] += 1
        return stats
