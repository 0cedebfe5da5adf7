"""convert_yuyv_to_rgb_u8 (P/color/yuv/mod.rs:319-480; device: P/cuda/color/video.rs:128-190).  CPU: the restatement
against an independent numpy form and known answers; the PRODUCT's per-pixel source (kh_video_modes.h, the file the
gfx950 kernel compiles) built for the host and swept over all 2^24 (Y, U, V) triples of every mode."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_ffi as O

ROOT = Path(__file__).resolve().parent.parent
COEF = {"bt601_full": (1436, 352, 731, 1815), "bt709_full": (1612, 192, 479, 1900), "bt601_limited": (1634, 401, 832, 2066)}


def numpy_decode(buf, w, h, mode):
    q = buf.reshape(h, w // 2, 4).astype(np.int64)
    y = np.stack([q[:, :, 0], q[:, :, 2]], -1).reshape(h, w)
    u = np.repeat(q[:, :, 1], 2, axis=1) - 128
    v = np.repeat(q[:, :, 3], 2, axis=1) - 128
    if mode == "bt601_limited":
        y = ((y - 16) * 1192 + 512) >> 10
    rv, gu, gv, bu = COEF[mode]
    r = y + ((rv * v + 512) >> 10)
    g = y - ((gu * u + gv * v + 512) >> 10)
    b = y + ((bu * u + 512) >> 10)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("mode", sorted(O.YUV_MODE))
def test_restatement_against_numpy_and_known_answers(mode):
    w, h = 38, 7
    buf = O.pattern_u8(w * h * 2)
    assert np.array_equal(O.yuyv_to_rgb_mode(buf, w, h, mode), numpy_decode(buf, w, h, mode))
    grey = O.yuyv_to_rgb_mode(np.array([128, 128, 128, 128], np.uint8), 2, 1, mode)  # neutral chroma
    assert grey.reshape(-1).tolist() == ([128] * 6 if mode != "bt601_limited" else [130] * 6)
    black = O.yuyv_to_rgb_mode(np.array([16, 128, 235, 128], np.uint8), 2, 1, "bt601_limited").reshape(-1).tolist()
    assert black == [0, 0, 0, 255, 255, 255]  # limited range: 16 -> 0, 235 -> 255
    # odd width: the last pixel of every row is left untouched (rows are walked in 6-byte chunks, mod.rs:374-376)
    odd = O.yuyv_to_rgb_mode(O.pattern_u8(5 * 3 * 2), 5, 3, mode, fill=77)
    assert (odd[:, 4, :] == 77).all() and np.array_equal(odd[:, :4, :].reshape(3, 2, 6), np.stack(
        [numpy_decode(O.pattern_u8(5 * 3 * 2)[r * 10: r * 10 + 8], 4, 1, mode).reshape(2, 6) for r in range(3)]))


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("vm") / "libvideo_modes_host.so"
    cmd = ["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-Wall", "-Wextra", "-Werror", f"-I{ROOT / 'kornia-rs_amd' / 'csrc'}",
           str(ROOT / "tests" / "cpp" / "video_modes_host.cpp"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = C.CDLL(str(out))
    lib.host_yuv_mode_table.argtypes = [C.c_int, np.ctypeslib.ndpointer(np.uint8, flags="C")]
    return lib


@pytest.mark.parametrize("mode", sorted(O.YUV_MODE))
def test_product_pixel_source_equals_the_restatement_for_every_input(host_lib, mode):
    table = np.empty((256, 256, 256, 3), np.uint8)
    assert host_lib.host_yuv_mode_table(O.YUV_MODE[mode], table.reshape(-1)) == 0
    # the same 2^24 triples as a YUYV image: one row per (y, u), pairs (Y0 = Y1 = y, U = u, V = v)
    yy, uu, vv = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    buf = np.stack([yy, uu, yy, vv], -1).reshape(-1)
    want = O.yuyv_to_rgb_mode(buf, 512, 256 * 256, mode).reshape(256, 256, 256, 2, 3)
    assert np.array_equal(want[..., 0, :], want[..., 1, :])
    assert np.array_equal(table, want[..., 0, :])
    assert host_lib.host_yuv_mode_table(3, table.reshape(-1)) == -1
