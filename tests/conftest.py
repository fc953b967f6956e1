import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def pytest_sessionstart(session):
    """Build the native pieces if a fresh checkout has not done so yet (the .so files are
    git-ignored): the HIP library (hipcc cross-compiles gfx950 without a GPU) and the C oracle."""
    jobs = str(min(8, os.cpu_count() or 1))
    if not os.path.exists(os.path.join(ROOT, "vptq_amd", "libvptq_hip.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "vptq_amd", "csrc"), "-j", jobs],
                       check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libvptq_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True,
                       stdout=subprocess.DEVNULL)


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _seeded():
    """Every test starts from the same torch seeds (CPU and GPU): a test that draws inputs with
    torch.randn / randint sees the same numbers on every run and every box."""
    import torch
    torch.manual_seed(20260924)
    yield


@pytest.fixture
def folded_arithmetic():
    """The opt-in fast arithmetic for the duration of a test (`vptq_amd.set_arithmetic("folded")`): tests of the routes that
    exist in that mode only - the sliced layouts of the large-codebook formats, the one-pass batched-decode kernel, the
    folded forms of the canonical kernels through the MODULE.  The product default is the reference's roundings."""
    import vptq_amd
    before = vptq_amd.arithmetic()
    vptq_amd.set_arithmetic("folded")
    yield
    vptq_amd.set_arithmetic(before)


@pytest.fixture
def selective_arithmetic():
    """`vptq_amd.set_arithmetic("selective")` for the duration of a test (round 6): the folded form with the reference's roundings on
    the blocks of columns an activation dominates - VPTQ_GEMV_SELECTIVE where a kernel implements it (the persistent chain launch),
    the reference's roundings everywhere else."""
    import vptq_amd
    before = vptq_amd.arithmetic()
    vptq_amd.set_arithmetic("selective")
    yield
    vptq_amd.set_arithmetic(before)
