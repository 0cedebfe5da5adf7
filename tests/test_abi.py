"""The C-ABI library loads on a CPU-only box and exports exactly what include/kornia_hip.h
declares; argument validation happens before any device call.  No GPU, no compute."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "kornia_hip.h"
TEST_HEADER = ROOT / "include" / "kornia_hip_testing.h"   # the two test hooks: exported, but not part of the boundary a host binds


def declared_symbols(boundary_only: bool = False):
    text = HEADER.read_text() + ("" if boundary_only else TEST_HEADER.read_text())
    return sorted(set(re.findall(r"KH_API\s+[\w\s\*]+?\b(kh_\w+)\s*\(", text)))


def test_test_hooks_are_not_part_of_the_boundary():
    """kh_debug_* live in kornia_hip_testing.h only: not in the boundary header, not in the generated Rust sys crate, and the
    product's host layer never calls them (tests and bench.py's A/B switch do)."""
    hooks = sorted(set(declared_symbols()) - set(declared_symbols(boundary_only=True)))
    assert hooks == ["kh_debug_fast_quot", "kh_debug_set_option"]
    assert "kh_debug_" not in re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    assert "kh_debug_" not in (ROOT / "integration" / "kornia-hip-sys" / "src" / "lib.rs").read_text()
    for path in (ROOT / "kornia-rs_amd" / "kornia_rs").glob("*.py"):
        if path.name != "_ffi.py":
            assert "kh_debug_" not in path.read_text(), f"{path} calls a test hook"


def test_header_declares_symbols():
    syms = declared_symbols()
    assert "kh_preprocess_to_chw" in syms and "kh_malloc_async" in syms and len(syms) >= 30


def test_library_exports_every_declared_symbol():
    from kornia_rs import _ffi
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(_ffi.LIB_PATH)], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in declared_symbols() if s not in exported]
    assert not missing, f"declared in kornia_hip.h but not exported: {missing}"
    extra = sorted(s for s in exported if s.startswith("kh_") and s not in declared_symbols())
    assert not extra, f"exported but not declared in kornia_hip.h: {extra}"


def test_python_binding_covers_every_declared_symbol():
    from kornia_rs import _ffi
    assert sorted(_ffi.SIGNATURES) == declared_symbols()
    assert _ffi.version().startswith("kornia-hip")


def test_library_embeds_gfx950_code_object():
    from kornia_rs import _ffi
    blob = _ffi.LIB_PATH.read_bytes()
    assert b"gfx950" in blob and b"preprocess_nv12_identity" in blob


def test_product_package_never_touches_the_oracle():
    """The shipped path must not import, link or name the oracle (CPU fallback = void parity)."""
    for path in (ROOT / "kornia-rs_amd").rglob("*"):
        if path.suffix in {".py", ".hip", ".h", ".hpp", ".cpp"} or path.name == "Makefile":
            text = path.read_text(errors="ignore")
            assert "oracle" not in text.lower(), f"{path} mentions the oracle"
    from kornia_rs import _ffi
    needed = subprocess.check_output(["objdump", "-p", str(_ffi.LIB_PATH)], text=True)
    assert "oracle" not in needed


def _params(**over):
    from kornia_rs import _ffi
    p = _ffi.PreprocessParams()
    p.scale_x = p.scale_y = 1.0
    p.src_w, p.src_h, p.src_pitch, p.src_bpp, p.fmt = 8, 6, 8, 1, _ffi.KH_FMT_NV12
    p.dst_w, p.dst_h = 8, 6
    p.sampling, p.out_dtype, p.nframes = _ffi.KH_SAMPLE_BILINEAR, _ffi.KH_OUT_F32, 1
    p.dst_frame_stride = 3 * 48
    for k, v in over.items():
        setattr(p, k, v)
    return p


@pytest.mark.parametrize("over,code", [
    (dict(src_h=5), -1),                 # NV12 odd height (SourceFormat::dims_ok)
    (dict(fmt=4, src_w=7, src_pitch=14, src_bpp=2), -1),  # YUYV odd width
    (dict(fmt=9), -1),
    (dict(sampling=3), -1),
    (dict(out_dtype=2), -1),
    (dict(dst_w=0), -1),
    (dict(fmt=0, src_bpp=2, src_pitch=16), -1),   # interleaved needs bpp 3|4
    (dict(fmt=0, src_bpp=3, src_pitch=10), -6),   # pitch shorter than a row
    (dict(dst_w=40000, dst_h=40000), -4),         # 32-bit index guard
    (dict(nframes=70000), -4),
    (dict(nframes=-1), -1),
])
def test_preprocess_validation_happens_before_launch(over, code):
    from kornia_rs import _ffi
    p = _params(**over)
    rc = _ffi.lib.kh_preprocess_to_chw(None, C.c_void_p(64), C.c_void_p(64), C.byref(p))
    assert rc == code, _ffi.last_error()
    assert _ffi.last_error()
    assert _ffi.lib.kh_preprocess_variant(C.byref(p)) is None


def test_preprocess_null_and_empty():
    from kornia_rs import _ffi
    p = _params()
    assert _ffi.lib.kh_preprocess_to_chw(None, None, None, C.byref(p)) == -1
    assert _ffi.lib.kh_preprocess_to_chw(None, None, None, None) == -1
    p.nframes = 0  # empty batch: nothing to do, no device needed
    assert _ffi.lib.kh_preprocess_to_chw(None, None, None, C.byref(p)) == 0
    assert _ffi.lib.kh_preprocess_variant(C.byref(_params())) == b"nv12_identity"
    # scale 1 into a smaller grid: every bilinear tap sits on a whole source pixel -> the generic kernel's one-tap form
    assert _ffi.lib.kh_preprocess_variant(C.byref(_params(dst_w=7, dst_h=5))) == b"generic_bilinear_on_grid"
    assert _ffi.lib.kh_preprocess_variant(C.byref(_params(dst_w=7, dst_h=5, scale_x=0.7))) == b"generic"
    assert _ffi.lib.kh_preprocess_variant(C.byref(_params(dst_w=7, dst_h=5, sampling=_ffi.KH_SAMPLE_NEAREST))) == b"generic"


def test_preprocess_list_validation():
    """kh_preprocess_to_chw_list: the frame list is checked on the host before anything is launched; an empty batch needs nothing."""
    from kornia_rs import _ffi
    p = _params(nframes=3)
    frames = _ffi.pointer_array([64, 0, 192])
    assert _ffi.lib.kh_preprocess_to_chw_list(None, frames, C.c_void_p(64), C.byref(p)) == -1
    assert "list index 1" in _ffi.last_error()
    assert _ffi.lib.kh_preprocess_to_chw_list(None, None, C.c_void_p(64), C.byref(p)) == -1
    assert "null frame list" in _ffi.last_error()
    assert _ffi.lib.kh_preprocess_to_chw_list(None, frames, None, C.byref(p)) == -1
    assert _ffi.lib.kh_preprocess_to_chw_list(None, frames, C.c_void_p(64), None) == -1
    p.nframes = 0
    assert _ffi.lib.kh_preprocess_to_chw_list(None, None, None, C.byref(p)) == 0


def test_no_device_fails_loudly_not_silently():
    """On a box without a GPU the runtime entry points return KH_ERR_HIP with a message — the
    product never computes on the CPU instead."""
    from kornia_rs import _ffi, hip
    if hip.is_available():
        pytest.skip("a GPU is present")
    ptr = C.c_void_p(0)
    rc = _ffi.lib.kh_malloc_async(C.byref(ptr), 1024, 1, None)
    assert rc == _ffi.KH_ERR_HIP and "HIP error" in _ffi.last_error()
    with pytest.raises(_ffi.KorniaHipError):
        hip.DeviceBuffer(16)


def test_tile_id_division_is_exact():
    """kh_common.h::FastDiv (multiply-shift division of block ids by launch constants): exact for every
    divisor up to 4096 plus large / awkward ones, at all multiples' boundaries and the top of the id range."""
    import numpy as np
    from kornia_rs import _ffi
    q = _ffi.lib.kh_debug_fast_quot
    rng = np.random.default_rng(0)
    divisors = list(range(1, 4097)) + [4800, 8640, 69120, 65535, 65537, 1 << 20, (1 << 20) + 1, 3 * 5 * 7 * 11 * 13 * 17, (1 << 30) - 1,
                                       (1 << 30) + 1, 0x7fefffff]
    top = 0x7ff00000
    for d in divisors:
        ns = {0, 1, d - 1, d, d + 1, top - 1, top, top // d * d, top // d * d - 1}
        ns.update(int(k) * d + o for k in rng.integers(0, top // d + 1, 6) for o in (-1, 0, 1))
        for n in ns:
            if 0 <= n <= top:
                assert q(n, d) == n // d, (n, d)


def test_rust_sys_crate_is_in_step_with_the_header():
    """integration/kornia-hip-sys/src/lib.rs is generated from include/kornia_hip.h (no rustc in this image, so it
    is kept correct mechanically): regenerating must reproduce the committed file, and every exported symbol, status
    code and #[repr(C)] field must be bound."""
    import importlib.util
    from kornia_rs import _ffi
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", ROOT / "scripts" / "gen_rust_ffi.py")
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = (ROOT / "integration" / "kornia-hip-sys" / "src" / "lib.rs").read_text()
    assert gen.generate() == committed, "run scripts/gen_rust_ffi.py"
    bound = set(re.findall(r"pub fn (kh_\w+)\(", committed))
    assert bound == set(declared_symbols(boundary_only=True)) == set(_ffi.SIGNATURES) - {"kh_debug_fast_quot", "kh_debug_set_option"}
    for const in ("KH_OK", "KH_ERR_HIP", "KH_ERR_SINGULAR", "KH_FMT_NV12", "KH_INTERP_LANCZOS", "KH_FUSE_WRITE_CHW_F32"):
        assert re.search(rf"pub const {const}: i32 = -?\d+;", committed), const
    fields = re.search(r"pub struct kh_preprocess_params \{(.*?)\}", committed, re.S).group(1)
    assert [f.split(":")[0].strip().replace("pub ", "") for f in fields.strip().splitlines()] == \
        [name for name, _ in _ffi.PreprocessParams._fields_]


def test_header_is_plain_c_and_a_c_consumer_links(tmp_path):
    """include/kornia_hip.h is the boundary a cgo / JNI / Rust binding compiles against: it must be valid C99 (-pedantic), and a
    C program linked with libkornia_hip.so must be able to call the host-only entries and read the error text."""
    import subprocess
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "abi_consumer"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{root / 'include'}", str(root / "tests" / "c" / "abi_consumer.c"),
           "-o", str(exe), f"-L{root / 'kornia-rs_amd' / 'lib'}", "-lkornia_hip", f"-Wl,-rpath,{root / 'kornia-rs_amd' / 'lib'}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c-abi ok" in r.stdout, r.stdout + r.stderr
