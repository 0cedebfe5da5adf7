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
