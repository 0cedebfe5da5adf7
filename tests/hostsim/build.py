#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — builds the product's .hip sources (kernels and host entries, unmodified apart from the one
`extern __shared__` declaration) for x86 against tests/hostsim/hip/hip_runtime.h, producing a library with the same C ABI
whose "device" is a fiber simulator.  Used by tests/test_hostsim.py and scripts/hostsim_run.py to run the REAL kernel code
on a GPU-less box.  It is never the product: libkornia_hip.so is always hipcc-built for gfx950.

    python tests/hostsim/build.py OUT.so
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "kornia-rs_amd", "csrc")
CXX = "/opt/rocm/lib/llvm/bin/clang++"
FLAGS = ["-x", "c++", "-std=c++17", "-O1", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w", "-DKH_HOSTSIM=1",
         f"-I{HERE}", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(HERE, "**", "*.*"), recursive=True)) + \
        sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))


def build_cached(out, sanitize=None):
    """build(), keyed on the content of every source it reads: the CPU suite needs the simulator in two modules (tests/test_sharding_gloo.py,
    tests/test_zz_hostsim.py) and a build is 90 s of clang.  The cache lives under the system temp directory; a stale entry cannot be hit."""
    import hashlib
    import shutil
    h = hashlib.sha256((sanitize or os.environ.get("KH_HOSTSIM_SANITIZE") or "").encode())
    for f in _sources():
        if os.path.isfile(f) and not f.endswith((".so", ".o", ".pyc")):
            h.update(f.encode()); h.update(open(f, "rb").read())
    cache = os.path.join(tempfile.gettempdir(), f"kh_hostsim_{os.getuid()}")
    os.makedirs(cache, exist_ok=True)
    hit = os.path.join(cache, h.hexdigest()[:32] + ".so")
    if not os.path.exists(hit):
        tmp = hit + f".{os.getpid()}.tmp"
        build(tmp, sanitize)
        os.replace(tmp, hit)
        for old in sorted(glob.glob(os.path.join(cache, "*.so")), key=os.path.getmtime)[:-3]:   # keep the three newest builds
            os.remove(old)
    shutil.copyfile(hit, out)
    return out


def build(out, sanitize=None):
    """sanitize="address": AddressSanitizer build (KH_HOSTSIM_SANITIZE=address scripts/hostsim_run.py ...): device buffers are
    plain heap blocks here, so a kernel that reads or writes one byte past an image is reported with its source line."""
    sanitize = sanitize or os.environ.get("KH_HOSTSIM_SANITIZE")
    extra = []
    if sanitize == "address":
        extra = ["-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer", "-g"]
    elif sanitize == "undefined":  # signed overflow, bad shifts, misaligned typed accesses, out-of-range float -> int casts
        extra = ["-fsanitize=undefined,float-cast-overflow", "-fno-sanitize=vptr,function", "-shared-libsan", "-fno-omit-frame-pointer", "-g"]
    objs = []
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(HERE, "hostsim.cpp")]:
            text = open(src).read()
            # dynamic LDS: `extern __shared__ T name[];` has no host spelling -> a fixed 160 KiB static array (the LDS size)
            text = re.sub(r"extern\s+__shared__\s+(__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];",
                          lambda m: f"static {m.group(1) or ''}{m.group(2)} {m.group(3)}[160 * 1024 / sizeof({m.group(2)})];", text)
            patched = os.path.join(tmp, os.path.basename(src) + ".cpp")
            open(patched, "w").write(f'#line 1 "{src}"\n' + text)
            obj = os.path.join(tmp, os.path.basename(src) + ".o")
            objs.append(obj)
            procs.append((src, subprocess.Popen([CXX, *FLAGS, *extra, "-c", patched, "-o", obj], stderr=subprocess.PIPE, text=True)))
        for src, p in procs:
            _, err = p.communicate()
            if p.returncode:
                sys.exit(f"{src}:\n{err[-4000:]}")
        subprocess.check_call([CXX, "-shared", *extra, "-o", out, *objs, "-lm", "-lpthread"])
    return out


if __name__ == "__main__":
    print(build(sys.argv[1] if len(sys.argv) > 1 else "/tmp/libkornia_hip_hostsim.so"))
