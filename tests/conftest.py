import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


# GPU tests that start further processes on the same card (two gloo ranks, a one-rank RCCL group, bench.py as a child, loader
# workers) run FIRST: behind the rest of the suite each of them took ~50 s instead of ~5 (the parent then holds a HIP context
# with tens of GB of cached allocations and every code object of the suite; measured 567 s for the whole suite against ~200 s
# this way), and a module's cached device memory is returned when the module is done.
_MULTI_PROCESS_FIRST = ("test_dist_gpu.py", "test_entrypoints_dp_gpu.py", "test_teacher_prefetch_gpu.py", "test_training_gpu.py",
                        "test_bench_gpu.py")


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        return _MULTI_PROCESS_FIRST.index(name) if name in _MULTI_PROCESS_FIRST else len(_MULTI_PROCESS_FIRST)
    items.sort(key=rank)            # stable: the order inside a module and among the other modules is unchanged


@pytest.fixture(scope="module", autouse=True)
def _release_cached_device_memory():
    yield
    import torch
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
