"""The process-per-GPU launcher of bench.py on REAL hardware, as far as a one-GPU box can take it (no 8-GPU node was
available to any round, `SCALE_r0*.json`: skipped).

* ``force``: the RCCL process group at world size 1 — ``init_process_group("nccl", device_id=...)``, ``barrier(device_ids=...)``,
  ``all_reduce(MAX)`` of the elapsed time on a device tensor, ``destroy_process_group`` — i.e. every collective call the N > 1 path
  makes, through the RCCL library of the box, under ``torch.distributed.run`` exactly as the driver launches it.
* ``share``: two ranks launched by ``torch.distributed.run`` that both drive GPU 0 (gloo group: RCCL refuses two ranks on one device):
  two bench processes run the real kernel at the same time behind a common barrier and rank 0 prints ONE line whose value is both
  ranks' pixels over the slower rank's time.

The same launcher with two simulated devices and gloo runs on CPU in tests/test_sharding_gloo.py."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(nproc, mode, batch, extra=()):
    port = _free_port()
    env = dict(os.environ, KORNIA_BENCH_DIST_TEST=mode, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "2", "--batch", str(batch),
           "--also", "none", "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.skipif(os.environ.get("KH_HOSTSIM") == "1", reason="needs torch to see a real device (RCCL)")
def test_rccl_collectives_of_the_launcher_at_world_size_one():
    j = _launch(1, "force", 16)
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["scaling"] == "weak" and j["unit"] == "Mpixels/s"
    assert j["config"]["workload"] == "nv12_1080p_to_chw_f32_b16"
    assert abs(j["value"] - 16 * 1920 * 1080 / 1e6 / (j["ms_per_step"] / 1e3)) <= 2e-3 * j["value"]   # ms_per_step is printed to 0.1 us
    assert 0.2 < j["roofline"]["frac"] < 1.0            # a real launch: 16 frames are far too few to reach the N = 1024 rate, but not zero


@pytest.mark.skipif(os.environ.get("KH_HOSTSIM") == "1", reason="two OS processes sharing one real GPU")
def test_two_ranks_share_one_gpu_behind_a_common_barrier():
    j = _launch(2, "share", 64)
    assert j["n_gpus"] == 2 and j["config"]["batch_per_gpu"] == 64 and j["config"]["workload"] == "nv12_1080p_to_chw_f32_b64"
    both = 2 * 64 * 1920 * 1080 / 1e6                  # weak scaling: `value` counts BOTH ranks' frames over the slower rank's time
    assert abs(j["value"] - both / (j["ms_per_step"] / 1e3)) <= 2e-3 * j["value"]
    assert "cpu_baseline" not in j
