"""Host-side rules added in round 2 that need no device: one HIP runtime per process (the root cause of round 1's incomplete
copies), stream workspaces, the memory-domain vocabulary, the sharder's argument checks."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import kornia_rs
from kornia_rs import _ffi
first = _ffi.mapped_hip_runtimes()
import torch
after = _ffi.mapped_hip_runtimes()
try:
    _ffi.assert_single_runtime(); guard = "ok"
except _ffi.MultipleHipRuntimes as e:
    guard = "refused"
buf = C = None
import ctypes
b = ctypes.create_string_buffer(4096)
n = _ffi.lib.kh_hip_runtime_images(b, 4096)
print(json.dumps({"choice": _ffi.RUNTIME_CHOICE, "first": first, "after": after, "guard": guard, "c_count": n, "c_paths": b.value.decode().split()}))
'''


def _child(env_extra):
    env = {k: v for k, v in os.environ.items() if k != "KORNIA_HIP_RUNTIME"}
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD % str(ROOT / "kornia-rs_amd")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    return json.loads(next(l for l in r.stdout.splitlines() if l.startswith("{")))


def test_default_policy_keeps_one_runtime_whatever_the_import_order():
    pytest.importorskip("torch")
    j = _child({})
    assert j["choice"].startswith(("torch bundle", "already mapped"))
    assert len(j["first"]["libamdhip64"]) == 1 and j["after"] == j["first"]            # importing torch later maps nothing new
    assert len(j["after"]["libhsa-runtime64"]) == 1
    assert j["guard"] == "ok" and j["c_count"] == 1 and j["c_paths"] == j["after"]["libamdhip64"]


def test_system_runtime_plus_torch_is_detected_and_refused():
    """KORNIA_HIP_RUNTIME=system reproduces round 1's load order: this library binds to /opt/rocm's runtime, then torch maps its bundled
    copy beside it.  Both the Python guard and the C entry see two images."""
    pytest.importorskip("torch")
    j = _child({"KORNIA_HIP_RUNTIME": "system"})
    assert j["choice"].startswith("system")
    assert len(j["first"]["libamdhip64"]) == 1
    if len(j["after"]["libamdhip64"]) == 1:
        pytest.skip("this torch build shares the system HIP runtime")
    assert len(j["after"]["libamdhip64"]) == 2 and len(j["after"]["libhsa-runtime64"]) == 2
    assert j["guard"] == "refused" and j["c_count"] == 2


def test_workspace_entries_validate_without_a_device():
    from kornia_rs import _ffi
    lib = _ffi.lib
    assert lib.kh_stream_set_workspace(None, None, 16) == _ffi.KH_ERR_INVALID_ARG
    assert "both" in _ffi.last_error()
    assert lib.kh_stream_set_workspace(None, 4096, 0) == _ffi.KH_ERR_INVALID_ARG
    assert lib.kh_stream_set_workspace(None, None, 0) == _ffi.KH_OK          # unregistering nothing is fine
    assert lib.kh_last_workspace_bytes(None) == _ffi.KH_ERR_INVALID_ARG
    n = C.c_size_t(123)
    assert lib.kh_last_workspace_bytes(C.byref(n)) == _ffi.KH_OK
    lib.kh_dlpack_noop_deleter(None)                                         # callable, does nothing


def test_memory_domain_vocabulary():
    import numpy as np
    from kornia_rs import Image, Tensor
    from kornia_rs.tensor import MemoryDomain as D
    assert D.is_host_accessible(D.HOST) and D.is_host_accessible(D.UNIFIED) and not D.is_host_accessible(D.DEVICE)
    assert D.is_device_accessible(D.DEVICE) and D.is_device_accessible(D.UNIFIED) and not D.is_device_accessible(D.HOST)
    t = Tensor.from_numpy(np.zeros((2, 3), np.float32))
    assert t.domain == D.HOST and not t.is_device and not t.is_unified and t.is_host_accessible and not t.is_pinned
    img = Image.from_numpy(np.zeros((4, 5, 3), np.uint8))
    assert img.domain == D.HOST and not img.is_unified
    with pytest.raises(Exception):
        img.to_hip_unified(None)  # no device here: the allocation fails loudly, there is no host stand-in


def test_sharder_rejects_bad_device_lists_without_a_device():
    from kornia_rs import hip
    from kornia_rs.sharding import ShardedPreprocessor, shard_range
    with pytest.raises(ValueError, match="at least one"):
        ShardedPreprocessor([])
    if hip.device_count() == 0:
        with pytest.raises(ValueError, match="out of range"):
            ShardedPreprocessor([0], format="nv12")
    # the planner: contiguous, balanced, covers everything exactly once
    for n, g in [(0, 3), (1, 4), (7, 3), (1024, 8), (1025, 8)]:
        spans = [shard_range(n, r, g) for r in range(g)]
        assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_elf_names_of_the_library_and_the_bundle():
    sys.path.insert(0, str(ROOT / "kornia-rs_amd"))
    from kornia_rs import _ffi
    names = _ffi.elf_dynamic_names(_ffi.LIB_PATH)
    assert any(n.startswith("libamdhip64.so") for n in names["needed"])
    assert _ffi.elf_dynamic_names(__file__) == {"soname": None, "needed": []}  # not an ELF: empty, no exception


LOAD_CHILD = r'''
import sys, os, ctypes
sys.path.insert(0, %r)
mode = os.environ["KH_TEST_MODE"]
if mode == "two_runtimes":           # a second HIP runtime image is already in the process when the package loads
    import importlib.util, pathlib
    ctypes.CDLL("/opt/rocm/lib/libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
    cand = pathlib.Path(importlib.util.find_spec("torch").origin).resolve().parent / "lib" / "libamdhip64.so"
    ctypes.CDLL(str(cand), mode=ctypes.RTLD_GLOBAL)
try:
    if mode == "soname_mismatch":    # the bundle claims another ROCm major: it must not be preloaded
        import importlib.util, types
        origin = os.path.join(sys.path[0], "kornia_rs", "_ffi.py")   # by path: importing the package would load the library
        src = open(origin).read().replace('have = elf_dynamic_names(cand)["soname"]', 'have = "libamdhip64.so.6"')
        mod = types.ModuleType("kornia_rs._ffi"); mod.__file__ = origin
        exec(compile(src, origin, "exec"), mod.__dict__)
        print("CHOICE", mod.RUNTIME_CHOICE); print("MAPPED", len(mod.mapped_hip_runtimes()["libamdhip64"]))
    else:
        from kornia_rs import _ffi
        print("LOADED", _ffi.RUNTIME_CHOICE)
except Exception as e:
    print("RAISED", type(e).__name__, str(e)[:200])
'''


def _load_child(mode, extra=None):
    env = {k: v for k, v in os.environ.items() if not k.startswith("KORNIA_HIP_RUNTIME")}
    env.update({"KH_TEST_MODE": mode, **(extra or {})})
    r = subprocess.run([sys.executable, "-c", LOAD_CHILD % str(ROOT / "kornia-rs_amd")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_loading_into_a_two_runtime_process_fails_loudly():
    """the single-runtime check ran only on DLPack hand-overs; a process that already maps two images (or ends up with
    two through the preload) now fails at load — or warns with KORNIA_HIP_RUNTIME_CHECK=warn."""
    pytest.importorskip("torch")
    if not Path("/opt/rocm/lib/libamdhip64.so").exists():
        pytest.skip("no system HIP runtime to map beside torch's")
    out = _load_child("two_runtimes")
    if "LOADED" in out:
        pytest.skip("this torch build shares the system HIP runtime")
    assert "RAISED MultipleHipRuntimes" in out, out
    out = _load_child("two_runtimes", {"KORNIA_HIP_RUNTIME_CHECK": "warn"})
    assert "LOADED already mapped" in out, out


def test_a_bundle_with_another_soname_is_not_preloaded():
    pytest.importorskip("torch")
    out = _load_child("soname_mismatch")
    assert "CHOICE system (torch bundle" in out and "MAPPED 1" in out, out


def test_hip_versions_are_reported():
    """kh_hip_versions: what the library was compiled against and what the bound runtime reports — the skew check behind the torch-bundle
    preload (a torch wheel can bundle an older ROCm than the one the library was built with)."""
    from kornia_rs import _ffi
    build, run = _ffi.hip_versions()
    assert build[0] >= 6 and len(build) == 3 and len(run) == 3        # built with ROCm 6+ (this image: 7.2)
    assert _ffi.lib.kh_hip_versions(None, None) == _ffi.KH_ERR_INVALID_ARG
