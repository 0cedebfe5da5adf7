#!/bin/bash
# Dev diagnosis: kornia_rs (preloading the torch wheel's HIP runtime) -> device work -> import torch -> device work exits
# with SIGSEGV after the script body has finished (r02a: rc -11).  Where?
set -u
OUT=gpurun_out/${1:-r02b}; mkdir -p "$OUT"
cd "$(dirname "$0")/../.."
body='
import os, sys
sys.path.insert(0, "kornia-rs_amd")
import numpy as np
import kornia_rs
from kornia_rs import hip, _ffi
print("choice:", _ffi.RUNTIME_CHOICE, flush=True)
VAR = os.environ.get("VAR", "")
if "nodev" not in VAR:
    hip.set_device(0)
    st = hip.Stream.new(0)
    b = hip.DeviceBuffer.from_numpy(np.arange(1 << 20, dtype=np.uint8), st)
    assert b.to_numpy(np.uint8, (1 << 20,))[5] == 5
if "notorch" not in VAR:
    import torch
    if "torchdev" in VAR:
        x = torch.rand(16, device="cuda"); torch.cuda.synchronize(); del x
if "cleanup" in VAR:
    import gc
    del b, st
    gc.collect()
print("body done", flush=True)
if "osexit" in VAR:
    os._exit(0)
'
for v in "torchdev" "" "nodev,torchdev" "notorch" "torchdev,cleanup" "torchdev,osexit"; do
  echo "=== VAR=[$v] auto" | tee -a "$OUT/exit.log"
  VAR="$v" timeout 120 python -X faulthandler -c "$body" >> "$OUT/exit.log" 2>&1; echo "rc=$?" | tee -a "$OUT/exit.log"
done
echo "=== VAR=[torchdev] KORNIA_HIP_RUNTIME=system" | tee -a "$OUT/exit.log"
VAR="torchdev" KORNIA_HIP_RUNTIME=system timeout 120 python -X faulthandler -c "$body" >> "$OUT/exit.log" 2>&1; echo "rc=$?" | tee -a "$OUT/exit.log"
echo "=== gdb backtrace, VAR=[torchdev] auto" | tee -a "$OUT/exit.log"
printf '%s' "$body" > /tmp/exit_body.py
VAR="torchdev" timeout 300 /opt/rocm/bin/rocgdb -batch -ex run -ex bt -ex "info sharedlibrary" --args python /tmp/exit_body.py 2>&1 | tail -60 >> "$OUT/exit.log"
echo "=== pytest exit code with torch interop (conftest order)" | tee -a "$OUT/exit.log"
timeout 300 python -m pytest tests/test_host_api_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee -a "$OUT/exit.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a "$OUT/exit.log"
