"""Dev (r04zy): does the time of a 1R + 1W streaming kernel depend on where its output sits relative to its input?  The 4K gaussian
(25.5 GB in, 25.5 GB out) took 8.7 .. 9.9 ms across runs on one box.  Same source buffer, destination at several byte offsets inside
one larger allocation, interleaved, event-timed."""
import sys, importlib.util
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "kornia-rs_amd"))
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from kornia_rs import hip
from kornia_rs.hip import DeviceBuffer, lib, check

hip.set_device(0); st = hip.Stream.new(0)
class A: batch = 0
name = sys.argv[1] if len(sys.argv) > 1 else "gaussian_4k"
wl = bench.WORKLOADS[name](A); wl.setup(st)
nbytes = wl.dst.nbytes
big = DeviceBuffer(nbytes + (64 << 20), st, zeroed=False)
orig = wl.dst
class View:
    def __init__(self, ptr): self.ptr = ptr; self.data_ptr = ptr; self.nbytes = nbytes
def timed(reps=6):
    for _ in range(3): wl.step()
    st.synchronize()
    e0, e1 = hip.Event(timing=True), hip.Event(timing=True)
    e0.record(st)
    for _ in range(reps): wl.step()
    e1.record(st); st.synchronize()
    return e0.elapsed_ms(e1) / reps
print(f"src 0x{wl.src.ptr:x}  own dst 0x{orig.ptr:x}  big 0x{big.ptr:x}")
offs = [0, 256, 4096, 65536, (1 << 20) + 4096, (2 << 20), (3 << 20) + 12288, (32 << 20) + 64 * 1024]
for rnd in range(3):
    wl.dst = orig; base = timed()
    row = [f"own {base:.3f}"]
    for o in offs:
        wl.dst = View(big.ptr + o); row.append(f"+{o}: {timed():.3f}")
    print(f"round {rnd}: " + "  ".join(row))
