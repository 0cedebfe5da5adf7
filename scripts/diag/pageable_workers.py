"""Dev (r04zt): pageable H2D ring against the number of host memcpy workers."""
import sys, importlib.util
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "kornia-rs_amd"))
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from kornia_rs import hip, preprocess
workers = int(sys.argv[1])
preprocess._Staging.copy_workers = workers
hip.set_device(0); st = hip.Stream.new(0)
class A: batch = 0
wl = bench.WORKLOADS["nv12_h2d_preprocess_pageable"](A); wl.setup(st)
ev = []
for k in range(20):
    a, b = hip.Event(timing=True), hip.Event(timing=True)
    a.record(st); wl.step(); b.record(st); ev.append((a, b))
st.synchronize()
ms = [a.elapsed_ms(b) for a, b in ev][2:]
print(f"workers {workers}: mean {sum(ms) / len(ms):.3f} ms  min {min(ms):.3f}  max {max(ms):.3f}")
