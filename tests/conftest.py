import os
import sys
from pathlib import Path

import pytest

# The restatement's OpenMP team, for TESTS only (bench.py times it in its own process with every core): the parity tests call it
# thousands of times on images of a few thousand pixels, where a 256-thread team (the GPU box) costs more in fork / spin than it
# computes — two morphology tests 0.73 s with the default team, 0.32 s with 16 threads — and four xdist workers with 256 spinning
# threads each turned an 18-minute suite out of what runs serially in a fraction of that (r03zz).  Set before libgomp loads.
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "kornia-rs_amd"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Files whose GPU tests were written after the last device run of everything else go here: `pytest -x` stops at the first
# failure, so they are collected AFTER the device-verified files and a surprise in a new test cannot hide the verdict on the
# suite that already ran on MI355X.  Empty since round 2: every file has run on the device (GPUTEST_r01.json, profiles/r02h).
DEVICE_RUN_PENDING = ()


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda it: Path(str(it.fspath)).name in DEVICE_RUN_PENDING)  # stable: keeps file / definition order


def _ensure_built():
    """Build the HIP library and the oracle if a fresh checkout has neither (cross-compiles on CPU)."""
    import subprocess

    if not (ROOT / "kornia-rs_amd" / "lib" / "libkornia_hip.so").exists():
        subprocess.check_call(["make", "-C", str(ROOT / "kornia-rs_amd"), "-j8"], stdout=subprocess.DEVNULL)
    if not (ROOT / "oracle" / "libkornia_oracle.so").exists():
        subprocess.check_call(["make", "-C", str(ROOT / "oracle")], stdout=subprocess.DEVNULL)


_ensure_built()


@pytest.fixture
def dev_option():
    """Force one of the library's alternate code paths for the duration of a test: `dev_option("pre_grid", 0)`
    (kh_debug_set_option, include/kornia_hip_testing.h).  Everything set is restored to the production choice afterwards.  The options
    are PER THREAD (kh_runtime.hip::g_dev_opts): they reroute the launches of the thread that set them — this test's main thread — and
    nothing launched from another thread (a ShardPool worker, a thread the test starts).  A test that needs a forced path on pool
    threads sets the option inside the worker (e.g. `pool._each(lambda g: set_option(...))`)."""
    from kornia_rs import _ffi
    touched = []

    def set_(name, value):
        _ffi.check(_ffi.lib.kh_debug_set_option(name.encode(), int(value)))
        touched.append(name)

    yield set_
    for name in touched:
        _ffi.lib.kh_debug_set_option(name.encode(), -1)


@pytest.fixture(scope="session")
def gpu_stream():
    """A non-default HIP stream on device 0; GPU tests fail (not skip) if the device or the
    native library is missing — a silent fallback must never look green."""
    import kornia_rs
    from kornia_rs import hip

    assert hip.is_available(), "no HIP device visible: -m gpu tests must run on the GPU box"
    hip.set_device(0)
    return hip.Stream.new(0)
