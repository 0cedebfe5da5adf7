#!/usr/bin/env python3
"""Dry-run the `-m gpu` tests on a GPU-less box against a MOCK of libkornia_hip.so, to catch host-glue mistakes
(wrong attribute / argument count / ctypes type / import) in tests and in the Python mirror before a GPU run is spent.

The mock is generated from kornia_rs/_ffi.py::SIGNATURES: "device" memory is host malloc, copies are memcpy, streams /
events / graphs are dummy handles, host-only helpers (matrix inversion, kernel taps, ...) forward to the REAL library,
and every compute entry runs the REAL library's argument validation (which stops with KH_ERR_HIP once it reaches the missing
device — reported as KH_OK) WITHOUT computing anything.  So value assertions are expected to fail; what must
not happen is any other exception.  Nothing here touches the product: a copy of the tree with the mock in place of the
library is made under a temporary directory and pytest runs there.

    python scripts/glue_dryrun.py [pytest args...]      # default: tests -m gpu
    python scripts/glue_dryrun.py --no-asserts [...]     # python -O: assert statements are compiled away, so every test
                                                         # body runs to its END (not just to its first value check)
Exit status 1 if a test died of anything but an AssertionError (or, with --no-asserts, a `pytest.raises` that the mock
cannot satisfy: "DID NOT RAISE").
"""
import ctypes as C
import os
import re
import shutil
import subprocess
import sys
import tempfile
import xml.etree.ElementTree as ET

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kornia-rs_amd"))

# host-only entries: forwarded to the real library (they never touch the HIP runtime)
FORWARD = {"kh_invert_affine_transform", "kh_get_rotation_matrix2d", "kh_invert_homography", "kh_box_blur_kernel_1d",
           "kh_gaussian_kernel_1d", "kh_gaussian_resolve", "kh_quantize_kernel_256", "kh_morph_kernel", "kh_pixel_mapping_coeffs",
           "kh_debug_fast_quot", "kh_preprocess_variant", "kh_version", "kh_last_error", "kh_fused_pipeline_build",
           "kh_fused_pipeline_describe", "kh_fused_pipeline_destroy"}

SPECIAL = {
    "kh_device_count": "{ if (a0) *(int32_t*)a0 = 1; return 0; }",
    "kh_get_device": "{ if (a0) *(int32_t*)a0 = 0; return 0; }",
    "kh_device_info": "{ if (a1 && a2) { strncpy((char*)a1, \"mock gfx950\", a2 - 1); ((char*)a1)[a2 - 1] = 0; } if (a3) *(int32_t*)a3 = 256; if (a4) *(uint64_t*)a4 = 288ull << 30; return 0; }",
    "kh_stream_create": "{ *(void**)a0 = malloc(8); return 0; }",
    "kh_stream_destroy": "{ free(a0); return 0; }",
    "kh_event_create": "{ *(void**)a0 = malloc(8); return 0; }",
    "kh_event_destroy": "{ free(a0); return 0; }",
    "kh_event_elapsed_ms": "{ if (a2) *(float*)a2 = 1.0f; return 0; }",
    "kh_malloc_async": "{ *(void**)a0 = a2 ? calloc(a1 ? a1 : 1, 1) : malloc(a1 ? a1 : 1); return *(void**)a0 ? 0 : -2; }",
    "kh_free_async": "{ free(a0); return 0; }",
    "kh_host_alloc": "{ *(void**)a0 = malloc(a1 ? a1 : 1); return 0; }",
    "kh_host_free": "{ free(a0); return 0; }",
    "kh_malloc_managed": "{ *(void**)a0 = calloc(a1 ? a1 : 1, 1); return 0; }",
    "kh_free": "{ free(a0); return 0; }",
    "kh_memcpy_h2d_async": "{ if (a2) memcpy(a0, a1, a2); return 0; }",
    "kh_memcpy_d2h_async": "{ if (a2) memcpy(a0, a1, a2); return 0; }",
    "kh_memcpy_d2d_async": "{ if (a2) memmove(a0, a1, a2); return 0; }",
    "kh_memset_async": "{ if (a2) memset(a0, a1, a2); return 0; }",
    "kh_pointer_domain": "{ if (a1) *(int32_t*)a1 = 1; if (a2) *(int32_t*)a2 = 0; return 0; }",
    "kh_mem_get_info": "{ if (a0) *(uint64_t*)a0 = 200ull << 30; if (a1) *(uint64_t*)a1 = 288ull << 30; return 0; }",
    "kh_graph_capture_begin": "{ return a0 ? 0 : -1; }",
    "kh_graph_launch": "{ return a0 ? 0 : -1; }",
    "kh_graph_capture_end": "{ *(void**)a1 = malloc(8); return 0; }",
    "kh_graph_destroy": "{ free(a0); return 0; }",
    "kh_find_min_max_f32": "{ if (a3) *(float*)a3 = 0.0f; if (a4) *(float*)a4 = 1.0f; return 0; }",
}


def ctype(t):
    if t is None:
        return "void"
    if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
        return "void*"
    return {C.c_int32: "int32_t", C.c_int64: "int64_t", C.c_uint32: "uint32_t", C.c_uint64: "uint64_t", C.c_size_t: "size_t",
            C.c_float: "float", C.c_double: "double", C.c_uint8: "uint8_t"}[t]


def generate(signatures, real_lib):
    out = ["#include <dlfcn.h>", "#include <stdint.h>", "#include <stdlib.h>", "#include <string.h>", "",
           f"static void* real(const char* name) {{ static void* h; if (!h) h = dlopen(\"{real_lib}\", RTLD_NOW | RTLD_LOCAL); return dlsym(h, name); }}", ""]
    for name, (res, args) in signatures.items():
        rt = ctype(res)
        params = ", ".join(f"{ctype(a)} a{i}" for i, a in enumerate(args)) or "void"
        if name in FORWARD:
            types = ", ".join(ctype(a) for a in args) or "void"
            call = f"(({rt} (*)({types}))real(\"{name}\"))({', '.join(f'a{i}' for i in range(len(args)))})"
            body = "{ " + ("" if rt == "void" else "return ") + call + "; }"
        elif name in SPECIAL:
            body = SPECIAL[name]
        elif rt == "int32_t":
            # compute entry: let the REAL library validate the arguments (it fails with KH_ERR_HIP = -2 only once it reaches
            # the device, i.e. after validation passed) and report success in its place — nothing is computed
            types = ", ".join(ctype(a) for a in args) or "void"
            call = f"((int32_t (*)({types}))real(\"{name}\"))({', '.join(f'a{i}' for i in range(len(args)))})"
            body = "{ int32_t rc = " + call + "; return rc == -2 ? 0 : rc; }"
        else:
            body = "{ " + ("" if rt == "void" else "return 0;") + " }"
        out.append(f"{rt} {name}({params}) {body}")
    return "\n".join(out) + "\n"


BENCH_SNIPPET = r"""
import importlib.util, sys
sys.path.insert(0, "kornia-rs_amd")
spec = importlib.util.spec_from_file_location("bench", "bench.py")
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from kornia_rs import hip
class Args: batch = 2
stream = hip.Stream.new(0)
for name, make in bench.WORKLOADS.items():
    wl = make(Args)
    wl.setup(stream)
    wl.step(); wl.step()
    stream.synchronize()
    d = wl.describe()
    assert d["workload"] == wl.name and wl.alg_bytes_per_launch > 0 and wl.units_per_step > 0, name
    print("ok", name, flush=True)
"""


def mock_tree(tmp):
    from kornia_rs import _ffi
    tree = os.path.join(tmp, "repo")
    shutil.copytree(ROOT, tree, ignore=shutil.ignore_patterns(".git", "gpurun_out", "profiles", "__pycache__", "build", ".pytest_cache"))
    real = os.path.join(tmp, "libkornia_hip_real.so")
    shutil.copy(_ffi.LIB_PATH, real)
    src = os.path.join(tmp, "mock.c")
    open(src, "w").write(generate(_ffi.SIGNATURES, real))
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-w", src, "-o", os.path.join(tree, "kornia-rs_amd", "lib", "libkornia_hip.so"), "-ldl"])
    return tree


def bench_dryrun():
    """Every bench workload: setup / step / describe at batch 2 against the mock (no torch, no timing)."""
    with tempfile.TemporaryDirectory() as tmp:
        tree = mock_tree(tmp)
        r = subprocess.run([sys.executable, "-c", BENCH_SNIPPET], cwd=tree, capture_output=True, text=True)
    print(r.stdout, end="")
    if r.returncode:
        print(r.stderr[-2000:])
    return r.returncode


def cpp_dryrun():
    """The C++ mirror's device legs against the mock: value checks print FAIL (expected), exceptions print THROW (glue bugs)."""
    status = 0
    with tempfile.TemporaryDirectory() as tmp:
        tree = mock_tree(tmp)
        lib = os.path.join(tree, "kornia-rs_amd", "lib")
        for name in ("host_mirror_test", "host_mirror_ops_test"):
            exe = os.path.join(tmp, name)
            subprocess.check_call(["g++", "-std=c++17", "-O1", f"-I{os.path.join(tree, 'include')}", os.path.join(tree, "tests", "cpp", name + ".cpp"),
                                   "-o", exe, f"-L{lib}", "-lkornia_hip", f"-Wl,-rpath,{lib}", "-ldl"])
            r = subprocess.run([exe, "gpu"], capture_output=True, text=True)
            lines = r.stdout.splitlines()
            thrown = [ln for ln in lines if ln.startswith("THROW")] + ([f"crashed with status {r.returncode}: {r.stderr[-300:]}"] if r.returncode not in (0, 1) else [])
            print(f"{name}: {sum(ln.startswith('FAIL') for ln in lines)} value failures (expected), {len(thrown)} exceptions; {lines[-1] if lines else ''}")
            for ln in thrown:
                print("  GLUE ", ln)
            status |= bool(thrown)
    return status


def main():
    from kornia_rs import _ffi
    if "--bench" in sys.argv[1:]:
        return bench_dryrun()
    if "--cpp" in sys.argv[1:]:
        return cpp_dryrun()
    no_asserts = "--no-asserts" in sys.argv[1:]
    args = [a for a in sys.argv[1:] if a != "--no-asserts"] or ["tests", "-m", "gpu"]
    with tempfile.TemporaryDirectory() as tmp:
        tree = os.path.join(tmp, "repo")
        shutil.copytree(ROOT, tree, ignore=shutil.ignore_patterns(".git", "gpurun_out", "profiles", "__pycache__", "build", ".pytest_cache"))
        real = os.path.join(tmp, "libkornia_hip_real.so")
        shutil.copy(_ffi.LIB_PATH, real)
        src = os.path.join(tmp, "mock.c")
        open(src, "w").write(generate(_ffi.SIGNATURES, real))
        mock = os.path.join(tree, "kornia-rs_amd", "lib", "libkornia_hip.so")
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-w", src, "-o", mock, "-ldl"])
        xml = os.path.join(tmp, "report.xml")
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        if no_asserts:
            env["PYTHONOPTIMIZE"] = "1"
        subprocess.run([sys.executable, "-m", "pytest", *args, "-q", "-p", "no:cacheprovider", f"--junitxml={xml}", "--tb=short",
                        "--maxfail=100000", *(["--assert=plain"] if no_asserts else [])], cwd=tree, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cases = list(ET.parse(xml).getroot().iter("testcase"))
    bad, value, passed = [], 0, 0
    for c in cases:
        problems = [e for e in c if e.tag in ("failure", "error")]
        if not problems:
            passed += 1
            continue
        text = (problems[0].get("message") or "") + "\n" + (problems[0].text or "")
        last = [ln for ln in text.strip().splitlines() if ln.startswith("E ")]
        if "DID NOT RAISE" in text:
            value += 1
        elif problems[0].tag == "failure" and re.search(r"AssertionError|^E\s+assert ", text, re.M) and "Error:" not in " ".join(
                ln for ln in last if "AssertionError" not in ln and not ln.startswith("E   assert") and not ln.startswith("E    ")):
            value += 1
            if os.environ.get("GLUE_VERBOSE"):
                print(f"  value {c.get('name')}: {(last[0] if last else '')[:150]}")
        else:
            bad.append((c.get("classname", "") + "::" + c.get("name", ""), (last[0] if last else text.strip()[:200])))
    print(f"{len(cases)} tests against the mock: {passed} passed, {value} failed on VALUES (expected: the mock computes nothing), "
          f"{len(bad)} died of something else")
    for name, why in bad:
        print(f"  GLUE  {name}\n        {why}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
