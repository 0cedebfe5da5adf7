#!/usr/bin/env python3
"""Run `-m gpu` tests on a GPU-less box against the HOST SIMULATOR build of the product (tests/hostsim/): the real
kernels and host entries compiled for x86, HIP threads as fibers.  A copy of the tree with that library in place of
libkornia_hip.so is made under a temporary directory and pytest runs there; nothing in the product changes.

    python scripts/hostsim_run.py tests/test_geom_gpu.py -x -q
    KH_HOSTSIM_SANITIZE=address python scripts/hostsim_run.py tests -q -n 16     # AddressSanitizer: out-of-bounds kernel accesses
    KH_HOSTSIM_DEVICES=2 python scripts/hostsim_run.py tests/test_multi_device_gpu.py tests/test_sharding_gpu.py   # two simulated GPUs
    KH_HOSTSIM_PREBUILT=/tmp/sim.so ...   # reuse a library built by tests/hostsim/build.py instead of building one per run
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))


def main():
    import build as hostsim_build
    args = sys.argv[1:] or ["tests", "-q"]
    with tempfile.TemporaryDirectory() as tmp:
        tree = os.path.join(tmp, "repo")
        shutil.copytree(ROOT, tree, ignore=shutil.ignore_patterns(".git", "gpurun_out", "profiles", "__pycache__", "build", ".pytest_cache"))
        lib = os.path.join(tree, "kornia-rs_amd", "lib", "libkornia_hip.so")
        prebuilt = os.environ.get("KH_HOSTSIM_PREBUILT")
        if prebuilt and os.path.exists(prebuilt):
            shutil.copyfile(prebuilt, lib)
        else:
            hostsim_build.build(lib)
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", KH_HOSTSIM="1")
        if os.environ.get("KH_HOSTSIM_SANITIZE") == "undefined":
            rt = subprocess.check_output([hostsim_build.CXX, "-print-file-name=libclang_rt.ubsan_standalone-x86_64.so"], text=True).strip()
            env["LD_PRELOAD"] = rt
            env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=0:log_path=/tmp/kh_ubsan:" + os.environ.get("UBSAN_OPTIONS", "")
        if os.environ.get("KH_HOSTSIM_SANITIZE") == "address":  # the sanitizer runtime must be loaded before python's allocator is used
            rt = subprocess.check_output([hostsim_build.CXX, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
            env["LD_PRELOAD"] = rt
            env["ASAN_OPTIONS"] = "detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0:halt_on_error=1:" + os.environ.get("ASAN_OPTIONS", "")
        return subprocess.run([sys.executable, "-m", "pytest", *args, "-m", "gpu", "-p", "no:cacheprovider"], cwd=tree, env=env).returncode


if __name__ == "__main__":
    sys.exit(main())
