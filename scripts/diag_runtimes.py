#!/usr/bin/env python3
"""Root-cause check of the round-1 "copy not complete at hipStreamSynchronize" finding (profiles/r01zy_sdma.log).

Hypothesis: not an SDMA bug but TWO HIP/HSA runtimes in one process.  libkornia_hip.so needs libamdhip64.so.7 (RUNPATH
/opt/rocm/lib); torch's libtorch_hip.so asks for "libamdhip64.so" by file name and finds the wheel's bundled copy, so
"kornia_rs first, torch later" maps both.  This script replays that order in fresh subprocesses, SDMA left at its default:

  system+torch   KORNIA_HIP_RUNTIME=system, kornia_rs, then torch on the device, then 48 MiB host<->device round trips
  auto+torch     default policy (kornia_rs preloads the wheel's runtime): same sequence
  system         no torch at all
  torch-first    torch, then kornia_rs (the order bench.py always used)

and prints, per case, the runtime images mapped (/proc/self/maps) and how many round trips came back with holes.

    /usr/local/graft/bin/gpurun --timeout 600 -- 'python scripts/diag_runtimes.py | tee gpurun_out/r02_runtimes.log'
"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

CHILD = r'''
import json, os, sys
sys.path.insert(0, os.path.join(%(root)r, "kornia-rs_amd"))
import numpy as np
order = %(order)r
torch = None
if order == "torch-first":
    import torch
    torch.zeros(1, device="cuda")
import kornia_rs
from kornia_rs import _ffi, hip, Tensor
from kornia_rs.hip import DeviceBuffer, PinnedBuffer
hip.set_device(0)
stream = hip.Stream.new(0)
n = 48 << 20
rng = np.random.default_rng(7)
src = rng.integers(1, 256, n, dtype=np.uint8)        # no zero bytes: a hole of zeros is unmistakable
# some device work through OUR runtime first (the failing test order had ~60 tests before the torch one)
for _ in range(3):
    b = DeviceBuffer.from_numpy(src, stream); assert np.array_equal(b.to_numpy(np.uint8, (n,)), src); b.free()
if order.endswith("+torch"):
    import torch
    x = torch.rand(97, 129, 3, device="cuda")            # initialises torch's runtime (the second one under "system")
    t = Tensor.zeros((64, 64), "float32", stream=stream)
    try:
        tt = torch.from_dlpack(t); tt += 1; torch.cuda.synchronize()
    except Exception as e:                                  # the guard may refuse the interop: that is a valid outcome
        print("interop:", type(e).__name__, str(e)[:120], file=sys.stderr)
    del x
bad_pageable = bad_pinned = 0
holes = []
pin = PinnedBuffer(n)
for rep in range(12):
    b = DeviceBuffer.from_numpy(src, stream)
    got = b.to_numpy(np.uint8, (n,))
    if not np.array_equal(got, src):
        bad_pageable += 1
        w = np.flatnonzero(got != src); holes.append(("pageable", int(w[0]), int(w[-1]), int(w.size)))
    # pinned: async h2d + async d2h into pinned, one sync at the end
    pin.view()[:] = src
    _ffi.check(_ffi.lib.kh_memcpy_h2d_async(b.ptr, pin.ptr, n, stream.cuda_stream_ptr))
    pin2 = PinnedBuffer(n)
    _ffi.check(_ffi.lib.kh_memcpy_d2h_async(pin2.ptr, b.ptr, n, stream.cuda_stream_ptr))
    stream.synchronize()
    got = pin2.view().copy()
    if not np.array_equal(got, src):
        bad_pinned += 1
        w = np.flatnonzero(got != src); holes.append(("pinned", int(w[0]), int(w[-1]), int(w.size)))
    pin2.free(); b.free()
print(json.dumps({"choice": _ffi.RUNTIME_CHOICE, "images": _ffi.mapped_hip_runtimes(), "bad_pageable": bad_pageable,
                  "bad_pinned": bad_pinned, "holes": holes[:4], "sdma_env": os.environ.get("HSA_ENABLE_SDMA")}))
'''


def run(name, order, env_extra):
    env = dict(os.environ)
    env.pop("HSA_ENABLE_SDMA", None)
    env.pop("KORNIA_HIP_RUNTIME", None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": str(ROOT), "order": order}], env=env, capture_output=True, text=True,
                       timeout=280)
    line = next((l for l in p.stdout.splitlines() if l.startswith("{")), None)
    print(f"== {name}: rc {p.returncode}")
    if line:
        j = json.loads(line)
        print(f"   runtime choice : {j['choice']}")
        for k, v in j["images"].items():
            print(f"   {k:17s}: {len(v)} image(s)  {v}")
        print(f"   round trips with holes: pageable {j['bad_pageable']}/12, pinned {j['bad_pinned']}/12   {j['holes']}")
    else:
        print("   no result line; stderr tail:", p.stderr[-600:])
    sys.stdout.flush()


if __name__ == "__main__":
    run("system+torch (two runtimes expected)", "system+torch", {"KORNIA_HIP_RUNTIME": "system"})
    run("auto+torch (one runtime expected)", "auto+torch", {})
    run("system, no torch", "system", {"KORNIA_HIP_RUNTIME": "system"})
    run("torch-first", "torch-first", {})
    run("system+torch, HSA_ENABLE_SDMA=0 (round-1 workaround)", "system+torch", {"KORNIA_HIP_RUNTIME": "system", "HSA_ENABLE_SDMA": "0"})
