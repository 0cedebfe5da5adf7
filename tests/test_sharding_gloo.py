"""The N>1 path on CPU: two gloo ranks shard a batch, run their slice, agree on the aggregate.
No data-path collective exists; only the barrier + max/sum reductions bench.py uses."""
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import os, sys, time
sys.path.insert(0, os.path.join(sys.argv[1], "kornia-rs_amd"))
import torch, torch.distributed as dist
from kornia_rs import sharding
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lo, hi = sharding.shard_range(1025, rank, world)
dist.barrier()
t0 = time.perf_counter()
time.sleep(0.05 * (rank + 1))          # rank 1 is the straggler
elapsed = time.perf_counter() - t0
units, slowest = sharding.aggregate_throughput(float(hi - lo), elapsed, dist)
spans = [None] * world
dist.all_gather_object(spans, (lo, hi))
if rank == 0:
    assert spans == [(0, 513), (513, 1025)], spans
    assert units == 1025.0 and slowest >= 0.1, (units, slowest)
    print("OK", units, round(slowest, 3))

# BASELINE configs[4] (undistort + warp_perspective, 2048 images over the node): every rank derives ITS slice from the same
# planner the in-process sharder uses, the shared operands (camera model, homography) are replicated by value — identical bytes
# on every rank, no broadcast — and the per-rank work adds up to the whole batch with nothing done twice.
import hashlib, struct
N5, W5, H5 = 2048, 3840, 2160
mine = sharding.plan(N5, world)[rank]
assert mine == sharding.shard_range(N5, rank, world)
intr = (577.48583984375 * 3.0, 652.8748779296875 * 3.0, 577.48583984375 * 2.7, 386.1428833007812 * 2.7)
dist8 = (1.7547749280929563, 0.0097926277667284, -0.027250492945313457, 2.1092164516448975, 0.462927520275116,
         -0.08215277642011642, -0.00005535508171073161, 0.00003768636770639569)
hm = [1.03, 0.05, -3.0 * W5 / 129.0, -0.02, 0.97, 4.0 * H5 / 97.0, 2.0 / (H5 * W5), 1.5 / (W5 * H5), 1.0]
digest = hashlib.sha256(struct.pack("<12d", *intr, *dist8) + struct.pack("<9f", *hm)).hexdigest()
work = [None] * world
dist.all_gather_object(work, (mine, digest, (mine[1] - mine[0]) * W5 * H5))
if rank == 0:
    spans5 = [w[0] for w in work]
    assert spans5 == sharding.plan(N5, world) and spans5[0][0] == 0 and spans5[-1][1] == N5
    assert all(a[1] == b[0] for a, b in zip(spans5, spans5[1:]))            # contiguous, no overlap, no gap
    assert len({w[1] for w in work}) == 1                                    # replicated operands agree bit for bit
    assert sum(w[2] for w in work) == N5 * W5 * H5                           # the whole batch, once
    print("C5 OK", spans5)
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script), str(ROOT)],
        env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK 1025.0" in out.stdout
    assert "C5 OK [(0, 1024), (1024, 2048)]" in out.stdout


# ---- the N > 1 path executing REAL kernels on every rank --------------------------------------------------------
# Each gloo rank loads the HOST-SIMULATOR build of the product (tests/hostsim: the same .hip sources compiled for x86, every HIP
# thread a fiber; selected with KORNIA_HIP_LIB — test infrastructure, the shipped library is always the gfx950 build) and runs ITS
# slice of the batch through the product API: Preprocessor.run_raw_batch for the north star, ShardedImgproc.undistort_warp for
# BASELINE configs[4].  Results are all_gather'ed and rank 0 compares the whole batch with the oracle, frame by frame.

KERNEL_WORKER = r'''
import os, sys
root = sys.argv[1]
sys.path[:0] = [os.path.join(root, "kornia-rs_amd"), os.path.join(root, "tests")]
import numpy as np
import torch, torch.distributed as dist
import oracle_ffi as O
from kornia_rs import Preprocessor, Tensor, hip, sharding
from kornia_rs.hip import DeviceBuffer

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
local_rank = int(os.environ.get("LOCAL_RANK", rank))
dev = local_rank if local_rank < hip.device_count() else 0     # a simulated process owns one device; on a node, rank r owns GPU r
hip.set_device(dev)
stream = hip.Stream.new(dev)
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

# -- north star: 7 NV12 frames of 64x34, same-size (the identity kernel) and a 40x24 letterbox (the generic kernel)
N, W, H = 7, 64, 34
fb = W * H * 3 // 2
base = O.pattern_u8(fb + 31 * N)
lo, hi = sharding.shard_range(N, rank, world)
mine = np.stack([base[31 * k: 31 * k + fb] for k in range(lo, hi)])
src = DeviceBuffer.from_numpy(mine.reshape(-1), stream)
outs = {}
for (ow, oh, mode) in [(W, H, "stretch"), (40, 24, "letterbox")]:
    pre = Preprocessor(mode=mode, format="nv12", mean=MEAN, std=STD, stream=stream)
    dst = Tensor.uninit((hi - lo, 3, oh, ow), "float32", stream)
    pre.run_raw_batch(src, W, H, dst, frame_stride=fb)
    outs[mode] = dst.numpy_raw()
parts = [None] * world
dist.all_gather_object(parts, ((lo, hi), outs))
if rank == 0:
    spans = [p[0] for p in parts]
    assert spans == sharding.plan(N, world), spans
    for (ow, oh, mode) in [(W, H, "stretch"), (40, 24, "letterbox")]:
        got = np.concatenate([p[1][mode] for p in parts], axis=0)
        assert got.shape == (N, 3, oh, ow), got.shape
        for k in range(N):
            want = O.preprocess(base[31 * k: 31 * k + fb], W, H, ow, oh, fmt="nv12", mode=mode, mean=MEAN, std=STD)[0]
            assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), (mode, k)
    print("NV12 sharded kernels OK", spans)

# -- BASELINE configs[4]: undistort (remap with Brown-Conrady maps) then warp_perspective, 5 images of 96x64x3 f32
M, IW, IH = 5, 96, 64
imgs = [np.roll(O.pattern_f32(IW * IH * 3 + k), -k)[: IW * IH * 3].reshape(IH, IW, 3).copy() for k in range(M)]
intr = (577.48583984375 * IW / 1280, 652.8748779296875 * IW / 1280, 577.48583984375 * IH / 800, 386.1428833007812 * IH / 800)
dist8 = (1.7547749280929563, 0.0097926277667284, -0.027250492945313457, 2.1092164516448975, 0.462927520275116,
         -0.08215277642011642, -0.00005535508171073161, 0.00003768636770639569)
hm = [1.03, 0.05, -3.0 * IW / 129.0, -0.02, 0.97, 4.0 * IH / 97.0, 2.0 / (IH * IW), 1.5 / (IW * IH), 1.0]
lo5, hi5 = sharding.plan(M, world)[rank]
sp = sharding.ShardedImgproc([dev])                       # this rank's pool: one device, one stream, one worker thread
batch = sp.scatter(imgs[lo5:hi5])
res = sp.undistort_warp(batch, intr, dist8, hm).numpy()
parts5 = [None] * world
dist.all_gather_object(parts5, ((lo5, hi5), res))
if rank == 0:
    got = [im for p in parts5 for im in p[1]]
    assert len(got) == M
    mx, my = O.correction_map(intr, dist8, IW, IH)
    for k in range(M):
        want = O.warp_perspective(O.remap(imgs[k], mx, my), hm, IW, IH)
        assert np.array_equal(np.asarray(got[k]).view(np.uint32), want.view(np.uint32)), k
    print("C5 sharded kernels OK", [p[0] for p in parts5])
dist.destroy_process_group()
'''


def _hostsim_lib(tmp_path_factory):
    sys.path.insert(0, str(ROOT / "tests" / "hostsim"))
    import build as hostsim_build
    out = tmp_path_factory.mktemp("hostsim") / "libkornia_hip_hostsim.so"
    hostsim_build.build_cached(str(out))
    return out


import pytest  # noqa: E402


@pytest.fixture(scope="module")
def hostsim_lib(tmp_path_factory):
    return _hostsim_lib(tmp_path_factory)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_run_their_slices_through_the_real_kernels(tmp_path, hostsim_lib):
    script = tmp_path / "kernel_worker.py"
    script.write_text(KERNEL_WORKER)
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KORNIA_HIP_LIB=str(hostsim_lib), KH_HOSTSIM="1",
               KH_HOSTSIM_DEVICES="2", OMP_NUM_THREADS="4")   # two simulated GPUs: rank 1 selects device 1, as on a node
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script), str(ROOT)],
        env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "NV12 sharded kernels OK [(0, 4), (4, 7)]" in out.stdout, out.stdout
    assert "C5 sharded kernels OK [(0, 3), (3, 5)]" in out.stdout, out.stdout


def test_bench_two_ranks_produce_one_valid_line(hostsim_lib):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one rank per device), on the simulator: ONE JSON
    line from rank 0, n_gpus 2, value = the units of BOTH ranks over the slowest rank's time."""
    import json
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KORNIA_HIP_LIB=str(hostsim_lib), KH_HOSTSIM="1",
               KH_HOSTSIM_DEVICES="2", OMP_NUM_THREADS="4")   # two simulated GPUs: rank 1 selects device 1, as on a node
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4"],
        env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak" and j["unit"] == "Mpixels/s"
    assert j["config"]["workload"] == "nv12_1080p_to_chw_f32_b4" and j["config"]["batch_per_gpu"] == 4
    per_step_mpx = 2 * 4 * 1920 * 1080 / 1e6                       # both ranks' frames
    assert abs(j["value"] - per_step_mpx / (j["ms_per_step"] / 1e3)) <= 0.06     # `value` is rounded to 0.1 Mpx/s; the simulator manages ~12
    assert "cpu_baseline" not in j                                   # rank-0, N = 1 only
    assert j["roofline"]["alg_bytes_per_launch"] == 4 * (1920 * 1080 * 3 // 2 + 12 * 1920 * 1080)
