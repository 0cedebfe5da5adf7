"""GPU parity for the pointer-list batch forms (round 6): N separately allocated operands in ceil(N / 128) launches (f32 geometry
and filters, `kh_*_f32_list`) or ceil(N / 256) launches (fused preprocess, `kh_preprocess_to_chw_list`) — the operands the
reference's per-image operators and `Preprocessor::run_raw_batch(frames: &[&CudaSlice<u8>], ..)` (P/preprocess.rs:1258-1282) are
handed.  Every image of a list launch must equal the CPU oracle's result for that image bit for bit, wherever its buffer lies
(odd spacing, misaligned bases, more images than one launch carries)."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import assert_same_bits, fptr

pytestmark = pytest.mark.gpu
IMAGENET = dict(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))


def _raw(fmt, w, h, k):
    n = {"rgb": 3 * w * h, "gray": w * h, "nv12": w * h * 3 // 2, "yuyv": 2 * w * h}[fmt]
    return np.roll(O.pattern_u8(n + 31 * k), -31 * k)[:n].copy()


class _View:
    """A frame inside a larger device allocation (what `_ptr_len` accepts through the CUDA array interface)."""

    def copy_from_host(self, raw):
        self._buf.copy_from_host(raw, offset=self._off)

    def __init__(self, buf, offset, nbytes):
        self._buf, self.ptr, self._off = buf, buf.ptr + offset, offset
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (buf.ptr + offset, False), "version": 3}


def _scattered_frames(stream, raws, misalign=0):
    """Frames at UNEQUALLY spaced bases, whatever the allocator's granularity (on the device the stream-ordered pool rounds small
    allocations to one size, so separate allocations alone come out equally spaced): two arenas, frame k in arena k % 2 at an
    offset whose gaps vary (multiples of 64 bytes, + `misalign` for the alignment tests)."""
    from kornia_rs.hip import DeviceBuffer
    fb = raws[0].size
    slot = (fb + 63) // 64 * 64 + 64 * 5
    arenas = [DeviceBuffer(slot * (len(raws) // 2 + 2) + 64, stream, zeroed=False) for _ in range(2)]
    out, cursor = [], [0, 0]
    for k, raw in enumerate(raws):
        a = k % 2
        off = cursor[a] + 64 * ((k * 7) % 5) + misalign
        arenas[a].copy_from_host(raw, offset=off)
        out.append(_View(arenas[a], off, raw.size))
        cursor[a] = (off + fb + 63) // 64 * 64
    return out


@pytest.mark.parametrize("fmt,w,h,dw,dh,mode,sampling", [
    ("nv12", 64, 32, 64, 32, "stretch", "bilinear"),      # the north-star kernel (identity), list form
    ("nv12", 64, 32, 40, 40, "letterbox", "bilinear"),    # generic four-tap
    ("nv12", 96, 48, 32, 32, "letterbox", "bilinear"),    # on-grid quads (scale 1/3)
    ("yuyv", 64, 32, 40, 24, "stretch", "nearest"),
    ("rgb", 46, 34, 31, 27, "letterbox", "bilinear"),
    ("gray", 22, 18, 57, 41, "letterbox", "lanczos"),
])
@pytest.mark.parametrize("f16", [False, True])
def test_run_raw_batch_list_matches_oracle(gpu_stream, fmt, w, h, dw, dh, mode, sampling, f16):
    from kornia_rs import Preprocessor, Tensor
    n = 7
    raws = [_raw(fmt, w, h, k) for k in range(n)]
    pre = Preprocessor(mode=mode, format=fmt, sampling=sampling, f16=f16, stream=gpu_stream, **IMAGENET)
    dst = Tensor.uninit((n, 3, dh, dw), "float16" if f16 else "float32", gpu_stream)
    frames = _scattered_frames(gpu_stream, raws)
    ptrs = [f.ptr for f in frames]
    assert len({b - a for a, b in zip(ptrs, ptrs[1:])}) > 1, "the test needs unequally spaced frames"
    pre.run_raw_batch(frames, w, h, dst)
    got = dst.numpy_raw()
    for k in range(n):
        want = O.preprocess(raws[k], w, h, dw, dh, fmt=fmt, mode=mode, sampling=sampling, f16=f16, **IMAGENET)
        g = got[k].view(np.uint16) if f16 else got[k]
        assert_same_bits(g.reshape(want.shape[-3:]), want.reshape(want.shape[-3:]), f"{fmt} list frame {k}")


def test_run_raw_batch_list_crosses_the_launch_slices(gpu_stream):
    """600 separately allocated NV12 frames: three launches of 256 / 256 / 88 frame bases; every frame equals the oracle and the
    equally spaced form."""
    from kornia_rs import Preprocessor, Tensor
    from kornia_rs.hip import DeviceBuffer
    w, h, n = 32, 16, 600
    raws = [_raw("nv12", w, h, k) for k in range(n)]
    pre = Preprocessor(mode="stretch", format="nv12", stream=gpu_stream, **IMAGENET)
    dst = Tensor.uninit((n, 3, h, w), "float32", gpu_stream)
    pre.run_raw_batch(_scattered_frames(gpu_stream, raws), w, h, dst)
    got = dst.numpy()
    packed = DeviceBuffer.from_numpy(np.concatenate(raws), gpu_stream)
    dst2 = Tensor.uninit((n, 3, h, w), "float32", gpu_stream)
    pre.run_raw_batch(packed, w, h, dst2, frame_stride=raws[0].size)
    assert_same_bits(got, dst2.numpy(), "list vs equally spaced")
    for k in (0, 1, 255, 256, 257, 511, 512, 599):
        want = O.preprocess(raws[k], w, h, w, h, fmt="nv12", mode="stretch", **IMAGENET)
        assert_same_bits(got[k], want.reshape(3, h, w), f"frame {k}")


@pytest.mark.parametrize("misalign", [1, 2, 6])
def test_run_raw_batch_list_misaligned_frames(gpu_stream, misalign):
    """Frame bases that are not multiples of 4 (2) bytes: the identity kernel's dword loads (the generic kernel's paired chroma loads)
    must not be used; the result is still the oracle's."""
    from kornia_rs import Preprocessor, Tensor
    w, h, n = 64, 32, 5
    raws = [_raw("nv12", w, h, k) for k in range(n)]
    for (dw, dh, mode) in [(w, h, "stretch"), (40, 40, "letterbox")]:
        pre = Preprocessor(mode=mode, format="nv12", stream=gpu_stream, **IMAGENET)
        dst = Tensor.uninit((n, 3, dh, dw), "float32", gpu_stream)
        pre.run_raw_batch(_scattered_frames(gpu_stream, raws, misalign), w, h, dst)
        got = dst.numpy()
        for k in range(n):
            want = O.preprocess(raws[k], w, h, dw, dh, fmt="nv12", mode=mode, **IMAGENET)
            assert_same_bits(got[k], want.reshape(3, dh, dw), f"misalign {misalign} frame {k} {mode}")


def test_list_launch_is_capturable(gpu_stream):
    """Nothing is allocated or uploaded by a list launch (the bases travel in the kernel arguments): it records into a graph, and
    the replay reads the frames' CURRENT contents."""
    from kornia_rs import Preprocessor, Tensor, hip
    w, h, n = 64, 32, 9
    raws = [_raw("nv12", w, h, k) for k in range(n)]
    frames = _scattered_frames(gpu_stream, raws)
    pre = Preprocessor(mode="stretch", format="nv12", stream=gpu_stream, **IMAGENET)
    dst = Tensor.uninit((n, 3, h, w), "float32", gpu_stream)
    g = hip.Graph.capture(lambda: pre.run_raw_batch(frames, w, h, dst), retain=[frames, dst], stream=gpu_stream)
    raws2 = [_raw("nv12", w, h, k + 100) for k in range(n)]
    for f, r in zip(frames, raws2):
        f.copy_from_host(r)
    g.replay()
    got = dst.numpy()
    for k in range(n):
        want = O.preprocess(raws2[k], w, h, w, h, fmt="nv12", mode="stretch", **IMAGENET)
        assert_same_bits(got[k], want.reshape(3, h, w), f"replayed frame {k}")


# ---- imgproc.*_batch ---------------------------------------------------------------------------------------------------------------

def _img(w, h, c, seed):
    return np.roll(O.pattern_f32(w * h * c + seed), -seed)[: w * h * c].reshape(h, w, c).copy()


def _images(stream, arrs):
    """Device Images at unequally spaced bases (see _scattered_frames): views into two arenas with varying gaps."""
    from kornia_rs import Image, Tensor
    from kornia_rs.hip import DeviceBuffer
    nb = arrs[0].nbytes
    slot = (nb + 255) // 256 * 256 + 256 * 7
    arenas = [DeviceBuffer(slot * (len(arrs) // 2 + 2), stream, zeroed=False) for _ in range(2)]
    imgs, cursor = [], [0, 0]
    for k, a in enumerate(arrs):
        ar = k % 2
        off = cursor[ar] + 256 * ((k * 5) % 7)
        arenas[ar].copy_from_host(a.reshape(-1), offset=off)
        imgs.append(Image(Tensor(a.shape, "float32", device_ptr=arenas[ar].ptr + off, device=stream.device, stream=stream, keepalive=arenas[ar])))
        cursor[ar] = (off + nb + 255) // 256 * 256
    ptrs = [im.data_ptr for im in imgs]
    assert len(imgs) < 3 or len({b - a for a, b in zip(ptrs, ptrs[1:])}) > 1
    return imgs, arenas


@pytest.mark.parametrize("mode", ["nearest", "bilinear", "bicubic", "lanczos"])
@pytest.mark.parametrize("c", [1, 3, 4])
def test_resize_batch_matches_oracle(gpu_stream, mode, c):
    from kornia_rs import imgproc
    n, (sw, sh, dw, dh) = 5, (63, 41, 30, 22)
    arrs = [_img(sw, sh, c, 31 * k) for k in range(n)]
    imgs, _keep = _images(gpu_stream, arrs)
    outs = imgproc.resize_batch(imgs, (dh, dw), mode)
    for k in range(n):
        assert_same_bits(outs[k].numpy(), O.resize(arrs[k], dw, dh, mode), f"resize_batch {mode} c{c} image {k}")
        assert_same_bits(outs[k].numpy(), imgproc.resize(imgs[k], (dh, dw), mode).numpy(), "batch vs single call")


def test_resize_batch_crosses_the_launch_slices(gpu_stream):
    """300 images: launches of 128 / 128 / 44 (src, dst) pairs; into caller-provided destinations."""
    from kornia_rs import Image, imgproc
    n, (sw, sh, dw, dh) = 300, (40, 24, 17, 11)
    arrs = [_img(sw, sh, 3, 31 * k) for k in range(n)]
    imgs, _keep = _images(gpu_stream, arrs)
    outs = [Image.uninit(dw, dh, 3, "float32", gpu_stream) for _ in range(n)]
    back = imgproc.resize_batch(imgs, None, "bilinear", outs=outs)
    assert all(a is b for a, b in zip(back, outs))
    for k in (0, 127, 128, 129, 255, 256, 299):
        assert_same_bits(outs[k].numpy(), O.resize(arrs[k], dw, dh, "bilinear"), f"image {k}")


def test_image_batch_is_validated_once_and_reusable(gpu_stream):
    """imgproc.ImageBatch: the pointer arrays built once; repeated calls on the same batches (new contents) give the oracle's bits."""
    from kornia_rs import Image, imgproc
    n, (sw, sh, dw, dh) = 5, (40, 24, 17, 11)
    arrs = [_img(sw, sh, 3, 31 * k) for k in range(n)]
    imgs, arenas = _images(gpu_stream, arrs)
    sb = imgproc.ImageBatch(imgs)
    ob = imgproc.ImageBatch([Image.uninit(dw, dh, 3, "float32", gpu_stream) for _ in range(n)])
    assert (sb.width, sb.height, sb.channels, len(sb.streams)) == (sw, sh, 3, 1)
    for rnd in range(2):
        back = imgproc.resize_batch(sb, None, "bicubic", outs=ob)
        assert back is ob
        for k in range(n):
            assert_same_bits(ob[k].numpy(), O.resize(arrs[k], dw, dh, "bicubic"), f"round {rnd} image {k}")
        arrs = [_img(sw, sh, 3, 31 * k + 7) for k in range(n)]      # rewrite the sources in place: the batch reads current contents
        for im, a in zip(imgs, arrs):
            from kornia_rs import hip
            hip.h2d(im.data_ptr, a.reshape(-1), gpu_stream)


def test_image_batch_calls_are_capturable(gpu_stream):
    """The *_batch forms allocate nothing when the destinations are given (pointer arrays in the kernel arguments): a chain of them
    records into one graph; the replay reads the sources' current contents."""
    from kornia_rs import Image, hip, imgproc
    n, (sw, sh, dw, dh) = 5, (64, 40, 24, 16)
    arrs = [_img(sw, sh, 3, 31 * k) for k in range(n)]
    imgs, arenas = _images(gpu_stream, arrs)
    sb = imgproc.ImageBatch(imgs)
    mid = imgproc.ImageBatch([Image.uninit(dw, dh, 3, "float32", gpu_stream) for _ in range(n)])
    out = imgproc.ImageBatch([Image.uninit(dw, dh, 3, "float32", gpu_stream) for _ in range(n)])

    def chain():
        imgproc.resize_batch(sb, None, "bilinear", outs=mid)
        imgproc.gaussian_blur_batch(mid, (3, 3), (0.8, 0.8), outs=out)

    g = hip.Graph.capture(chain, retain=[sb, mid, out, arenas], stream=gpu_stream)
    arrs = [_img(sw, sh, 3, 31 * k + 5) for k in range(n)]
    for im, a in zip(imgs, arrs):
        hip.h2d(im.data_ptr, a.reshape(-1), gpu_stream)
    g.replay()
    for k in range(n):
        want = O.gaussian_blur(O.resize(arrs[k], dw, dh, "bilinear"), (3, 3), (0.8, 0.8))
        assert_same_bits(out[k].numpy(), want, f"replayed image {k}")


def test_warp_and_remap_batches_match_oracle(gpu_stream):
    from kornia_rs import Image, imgproc
    n, (w, h) = 6, (96, 64)   # remap: 6 = one full group of four images + a partial one
    arrs = [_img(w, h, 3, 31 * k) for k in range(n)]
    imgs, _keep = _images(gpu_stream, arrs)
    m = imgproc.get_rotation_matrix2d((w / 2, h / 2), 12.0, 0.9)
    outs = imgproc.warp_affine_batch(imgs, m, (h, w), "bilinear")
    for k in range(n):
        assert_same_bits(outs[k].numpy(), O.warp_affine(arrs[k], m, w, h, "bilinear"), f"warp_affine_batch {k}")
    hm = [1.0, 0.05, 3.0, 0.02, 0.95, -2.0, 1e-4, 2e-4, 1.0]
    outs = imgproc.warp_perspective_batch(imgs, hm, (h, w), "bicubic")
    for k in range(n):
        assert_same_bits(outs[k].numpy(), O.warp_perspective(arrs[k], hm, w, h, "bicubic"), f"warp_perspective_batch {k}")
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    mx, my = (xs * 0.9 + 2.5).astype(np.float32), (ys * 1.05 - 1.25).astype(np.float32)
    dmx, dmy = Image.from_numpy(mx[..., None]).to_hip(gpu_stream), Image.from_numpy(my[..., None]).to_hip(gpu_stream)
    outs = imgproc.remap_batch(imgs, dmx, dmy, "bilinear")
    for k in range(n):
        assert_same_bits(outs[k].numpy(), O.remap(arrs[k], mx, my, "bilinear"), f"remap_batch {k}")


def test_filter_batches_match_oracle(gpu_stream):
    from kornia_rs import imgproc
    n = 4
    # 344 x 3 floats per row: >= 1024 columns and a multiple of four -> the four-columns-per-lane rolling kernel; 70 x 3 -> one column
    for (w, h) in [(344, 40), (70, 33)]:
        arrs = [_img(w, h, 3, 31 * k) for k in range(n)]
        imgs, _keep = _images(gpu_stream, arrs)
        outs = imgproc.gaussian_blur_batch(imgs, (7, 7), (1.5, 1.5))
        for k in range(n):
            assert_same_bits(outs[k].numpy(), O.gaussian_blur(arrs[k], (7, 7), (1.5, 1.5)), f"gaussian_blur_batch {w}x{h} {k}")
        outs = imgproc.box_blur_batch(imgs, (5, 5))
        kx = O.box_kernel_1d(5)
        for k in range(n):
            assert_same_bits(outs[k].numpy(), O.separable_filter(arrs[k], kx, kx), f"box_blur_batch {k}")
        outs = imgproc.sobel_batch(imgs, 3)
        for k in range(n):
            assert_same_bits(outs[k].numpy(), O.gradient_magnitude(arrs[k], 0, 3), f"sobel_batch {k}")
        outs = imgproc.separable_filter_batch(imgs, [0.25, 0.5, 0.25], [0.1, 0.2, 0.4, 0.2, 0.1])   # unequal taps: the LDS-tile kernel
        for k in range(n):
            assert_same_bits(outs[k].numpy(), O.separable_filter(arrs[k], np.array([0.25, 0.5, 0.25], np.float32),
                                                                 np.array([0.1, 0.2, 0.4, 0.2, 0.1], np.float32)), f"separable_filter_batch {k}")


def test_batch_operand_errors_are_typed(gpu_stream):
    from kornia_rs import Image, imgproc
    from kornia_rs.image import ImageError
    a = Image.from_numpy(_img(8, 6, 3, 0)).to_hip(gpu_stream)
    b = Image.from_numpy(_img(9, 6, 3, 0)).to_hip(gpu_stream)
    with pytest.raises(ImageError, match="every image of a batch must be"):
        imgproc.resize_batch([a, b], (4, 4))
    with pytest.raises(ImageError, match="must all be on the host or all on the device"):
        imgproc.resize_batch([a, Image.from_numpy(_img(8, 6, 3, 0))], (4, 4))
    with pytest.raises(ImageError, match="empty batch"):
        imgproc.resize_batch([], (4, 4))
    with pytest.raises(ImageError, match="destinations"):
        imgproc.resize_batch([a, a], None, outs=[Image.uninit(4, 4, 3, "float32", gpu_stream)])
    with pytest.raises(ImageError, match="determinant|singular"):
        imgproc.warp_perspective_batch([a], [0.0] * 9, (6, 8))


def test_list_abi_direct(gpu_stream):
    """The C entry itself: host arrays of device pointers, one call; a NULL in the list is refused before any launch."""
    from kornia_rs import _ffi
    from kornia_rs.hip import DeviceBuffer
    n, (sw, sh, dw, dh) = 3, (33, 21, 12, 9)
    arrs = [_img(sw, sh, 3, 31 * k) for k in range(n)]
    srcs = [DeviceBuffer.from_numpy(a.reshape(-1), gpu_stream) for a in arrs]
    dsts = [DeviceBuffer(dw * dh * 3 * 4, gpu_stream, zeroed=False) for _ in range(n)]
    sp, dp = _ffi.pointer_array([s.ptr for s in srcs]), _ffi.pointer_array([d.ptr for d in dsts])
    _ffi.check(_ffi.lib.kh_resize_f32_list(gpu_stream.cuda_stream_ptr, sp, dp, n, sw, sh, dw, dh, 3, O.MODE["bilinear"], 0))
    for k in range(n):
        assert_same_bits(dsts[k].to_numpy(np.float32, (dh, dw, 3)), O.resize(arrs[k], dw, dh, "bilinear"), f"image {k}")
    bad = _ffi.pointer_array([srcs[0].ptr, 0, srcs[2].ptr])
    assert _ffi.lib.kh_resize_f32_list(gpu_stream.cuda_stream_ptr, bad, dp, n, sw, sh, dw, dh, 3, 1, 0) == _ffi.KH_ERR_INVALID_ARG
    assert "list index 1" in _ffi.last_error()
