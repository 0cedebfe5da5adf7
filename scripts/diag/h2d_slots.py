"""Dev (r04zh): why does every other step of the H2D workload take 6.6 ms instead of 3.5 when it runs after the four NV12 / YUYV
workloads of the default bench line?  Times the DMA itself (events on the copy stream) per step, with the slot's device pointer."""
import importlib.util, sys, gc, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
sys.path.insert(0, str(ROOT / "kornia-rs_amd"))
from kornia_rs import hip, preprocess
from kornia_rs.hip import lib

class A: batch = 0
hip.set_device(0); st = hip.Stream.new(0)
mode = sys.argv[3] if len(sys.argv) > 3 else ""
first = None
if "first" in mode:
    first = bench.WORKLOADS["nv12_h2d_preprocess"](A); first.setup(st); st.synchronize()
before = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "none" else []
keep = []
for name in before:
    wl = bench.WORKLOADS[name](A); wl.setup(st)
    for _ in range(3): wl.step()
    st.synchronize()
    if len(sys.argv) > 2 and sys.argv[2] == "keep": keep.append(wl)
    else: del wl; gc.collect()
wl = first or bench.WORKLOADS["nv12_h2d_preprocess"](A)
if "depth3" in mode:
    preprocess._Staging.DEPTH = 3
if "hipmalloc" in mode:   # slot buffers straight from hipMalloc instead of the stream-ordered pool
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    class RawBuf:
        def __init__(self, nbytes, stream=None, zeroed=False):
            p = ctypes.c_void_p(); assert rt.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0
            self.ptr, self.nbytes, self.stream = p.value, nbytes, stream
        data_ptr = property(lambda self: self.ptr)
    preprocess.DeviceBuffer = RawBuf
nsteps = 30 if "long" in mode else 12
orig = lib.kh_memcpy_h2d_async
log = []
def timed_memcpy(dst, src, n, s):
    e0, e1 = hip.Event(timing=True), hip.Event(timing=True)
    lib.kh_event_record(e0._handle, s); t0 = time.perf_counter(); rc = orig(dst, src, n, s); t1 = time.perf_counter(); lib.kh_event_record(e1._handle, s)
    log.append((e0, e1, int(dst), int(src), int(n), (t1 - t0) * 1e3))
    return rc
lib.kh_memcpy_h2d_async = timed_memcpy
(wl.setup(st) if first is None else None); log.clear()
ev = []
for k in range(nsteps):
    a, b = hip.Event(timing=True), hip.Event(timing=True)
    a.record(st); wl.step(); b.record(st); ev.append((a, b))
st.synchronize()
for k, ((a, b), (e0, e1, dst, src, n, host_ms)) in enumerate(zip(ev, log)):
    print(f"step {k:2d}: compute-stream {a.elapsed_ms(b):6.3f} ms   DMA {e0.elapsed_ms(e1):6.3f} ms  host call {host_ms:6.3f} ms  dst 0x{dst:x} (mod 2MiB {dst % (2 << 20)})  src 0x{src:x}  {n} B")
