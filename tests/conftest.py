import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle legs of the GPU tests are small CPU problems (8x8 ... 64x64 latents): on the GPU box's 256 logical CPUs torch's default thread count
    # makes them SLOWER -- the container runs under a CPU quota of ~16 cores (round 6: test_multi_axis_bank_carry_over 51 s at 64 threads, 28 s at 16;
    # test_ddim_sample_multi_axis_vs_oracle 34 -> 17 s; profiles/r6_gpu_tests_threads.log).  Tests that want more set it.
    nt = os.environ.get("TCL_TEST_THREADS")
    if nt or (os.cpu_count() or 8) > 32:
        import torch
        torch.set_num_threads(int(nt or 16))


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    return load


# ---------------------------------------------------------------------------------------------------------------- GPU suite order and per-test ceiling
# the driver stops `pytest -m gpu` at 1 200 s; round 5's suite ran the slowest test (configs[0] end to end, ~430 s of CPU oracle) in front
# of every leaf parity test and the driver saw 32 of 152.  Order now: the launch test of the end-to-end run first (its oracle legs then run in the
# background on the host's cores, tests/e2e_jobs.py), leaf parity against the reference's goldens next (cheapest, most decisive), composites after
# them, the full-size sweeps late, the tests that collect the end-to-end legs last.
_GPU_FILE_ORDER = ["test_gpu_path2.py", "test_gpu_kernels.py", "test_gpu_unet.py", "test_gpu_vae.py", "test_gpu_memflow.py", "test_gpu_rmbg.py",
                   "test_gpu_skew.py", "test_gpu_rccl.py", "test_gpu_path2_dist.py", "test_gpu_denoise_loop.py", "test_gpu_config4.py", "test_gpu_run.py",
                   "test_gpu_e2e_dist.py", "test_gpu_bench_dist.py", "test_gpu_fullsize.py", "test_gpu_e2e.py"]
_CEILING_S = float(os.environ.get("TCL_TEST_CEILING", "150"))


def pytest_collection_modifyitems(session, config, items):
    def key(it):
        f = os.path.basename(str(it.fspath))
        if it.name.startswith("test_config1_launch"):
            return (-1, 0)
        if f in _GPU_FILE_ORDER:
            return (1 + _GPU_FILE_ORDER.index(f), 0)
        return (0, 0)                                   # CPU tests and anything new keep their place in front
    items.sort(key=key)                                 # stable: the order inside a file is the file's


@pytest.fixture(autouse=True)
def _per_test_ceiling(request):
    """Any GPU test that takes longer than TCL_TEST_CEILING (150 s) fails: the suite has to fit the driver's clock with margin (VERDICT r5 #1c)."""
    import time
    t0 = time.time()
    yield
    dt = time.time() - t0
    if request.node.get_closest_marker("gpu") is not None and dt > _CEILING_S:
        pytest.fail(f"{request.node.nodeid} took {dt:.0f} s > the {_CEILING_S:.0f} s per-test ceiling (tests/conftest.py)")
