import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _cpu_threads():
    """The CPU oracle's many small ops get SLOWER beyond ~32 threads on the GPU boxes' many-core hosts (bench.py's cpu_baseline uses
    32: 2.4 s per full-size cfg2 graph against 5.5 s with every core): the parity tests' oracle forwards run on at most 32."""
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


@pytest.fixture(autouse=True)
def _release_device_memory(request):
    """GPU tests build 40-sample samplers whose captured hipGraphs keep a step's intermediates (25 - 45 GB) in the graphs' memory
    pool; the objects sit in reference cycles, so they are collected after every test - the pool (shared by the captured steps of
    the process, sampler.Sampler._graph_pool) is then reused by the next capture.  No torch.cuda.empty_cache() here: handing
    capture-time segments back to the driver is exactly what leaks them on this ROCm stack."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            log = os.environ.get("DDP_TEST_MEM_LOG")      # diagnostic: device memory still held after each test
            if log:
                torch.cuda.synchronize()
                free, total = torch.cuda.mem_get_info()
                with open(log, "a") as f:
                    f.write(f"{request.node.nodeid}\tused_GB={(total - free) / 1e9:.1f}\ttorch_reserved_GB={torch.cuda.memory_reserved() / 1e9:.1f}"
                            f"\ttorch_allocated_GB={torch.cuda.memory_allocated() / 1e9:.1f}\n")
