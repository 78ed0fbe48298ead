import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs on torch's intra-op pool: size it to the cores the cgroup grants (a GPU box shows 256 CPUs
    # but a 16-core quota; 128 spinning threads get the whole process throttled)
    import torch
    from object_detection_cib_amd._lib import cpu_share
    torch.set_num_threads(min(torch.get_num_threads(), cpu_share()))


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
