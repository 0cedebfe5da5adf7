"""Static lint of every gfx950 kernel (no GPU): no scratch, no VGPR spills, no FLAT memory instructions — scripts/kernel_resources.py
cross-compiles csrc/ and reads the compiler's resource remarks and the ISA.  Each of the three has cost a GPU visit before."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_no_scratch_no_spills_no_flat_memory_instructions():
    out = subprocess.run([sys.executable, str(ROOT / "scripts" / "kernel_resources.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    head = out.stdout.splitlines()[:4]
    m = re.search(r"kernels with scratch: (\d+), with VGPR spills: (\d+)", head[1])
    assert m and m.group(1) == "0" and m.group(2) == "0", head[1]
    m = re.search(r"kernels with flat memory instructions: (\d+)", head[2])
    assert m and m.group(1) == "0", "\n".join(l for l in out.stdout.splitlines() if l.startswith("#"))
    assert int(re.match(r"# (\d+) kernels", head[0]).group(1)) > 300


def test_no_getenv_in_the_product_sources():
    """Static half of the rule: no source of the shipped library calls getenv."""
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent / "kornia-rs_amd" / "csrc"
    hits = [f"{p.name}:{i + 1}" for p in sorted(root.iterdir()) for i, line in enumerate(p.read_text().splitlines())
            if "getenv(" in line and not line.lstrip().startswith("//")]
    assert not hits, hits
