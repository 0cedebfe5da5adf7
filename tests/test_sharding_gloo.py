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
