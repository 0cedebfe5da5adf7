"""The host mirror on a real device: Image residency, imgproc dispatch end-to-end, cross-stream
fences, DLPack / __cuda_array_interface__ zero-copy with torch-ROCm (kornia-py/tests/
test_image_device.py, test_dlpack.py, test_torch_zero_copy.py, test_cuda.py)."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def test_image_to_hip_roundtrip_and_ops(gpu_stream):
    from kornia_rs import Image, imgproc
    rgb = O.pattern_u8(258 * 195 * 3).reshape(195, 258, 3)  # config[0] shape, on the device path
    dev = Image.from_numpy(rgb).to_hip(gpu_stream)
    assert dev.is_device and dev.device == "cuda:0" and dev.to_cuda(gpu_stream) is dev
    assert np.array_equal(dev.numpy(), rgb) and not dev.numpy().flags.writeable
    gray = imgproc.gray_from_rgb(dev)
    assert gray.is_device and gray.shape == (195, 258, 1)
    assert np.array_equal(gray.numpy().reshape(-1), O.color_map("gray_from_rgb_u8", rgb, 1))
    back = gray.cpu()
    assert not back.is_device and np.array_equal(back.numpy(), gray.numpy())

    f = Image.from_numpy(O.pattern_f32(129 * 97 * 3).reshape(97, 129, 3)).to_hip(gpu_stream)
    small = imgproc.resize(f, (48, 64), "bilinear")
    assert small.shape == (48, 64, 3)
    assert np.array_equal(small.numpy(), O.resize(f.numpy(), 64, 48))
    blur = imgproc.gaussian_blur(f, (7, 7), (1.5, 1.5))
    assert np.array_equal(blur.numpy(), O.gaussian_blur(f.numpy(), (7, 7), (1.5, 1.5)))
    hm = [1.03, 0.05, -3.0, -0.02, 0.97, 4.0, 2.0 / (97.0 * 129.0), 1.5 / (129.0 * 97.0), 1.0]
    warped = imgproc.warp_perspective(f, hm, (97, 129), "bilinear")
    assert np.array_equal(warped.numpy(), O.warp_perspective(f.numpy(), hm, 129, 97))
    mx, my = imgproc.generate_correction_map_polynomial((300.0, 300.0, 64.0, 48.0), (0.1, 0.01, 0, 0, 0, 0, 1e-4, 1e-4),
                                                        (129, 97), gpu_stream)
    und = imgproc.remap(f, mx, my)
    assert np.array_equal(und.numpy(), O.remap(f.numpy(), mx.numpy()[:, :, 0], my.numpy()[:, :, 0]))
    lo, hi = imgproc.find_min_max(f)
    assert (lo, hi) == (float(f.numpy().min()), float(f.numpy().max()))
    norm = imgproc.normalize_mean_std(f, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    want = ((f.numpy() - np.array((0.485, 0.456, 0.406), np.float32)) / np.array((0.229, 0.224, 0.225), np.float32)).astype(np.float32)
    assert np.array_equal(norm.numpy(), want)
    assert np.array_equal(imgproc.horizontal_flip(f).numpy(), f.numpy()[:, ::-1])
    assert np.array_equal(imgproc.crop(f, 3, 5, 40, 30).numpy(), f.numpy()[5:35, 3:43])
    raw = O.pattern_u8(64 * 32 * 3 // 2)
    from kornia_rs.hip import DeviceBuffer
    dec = imgproc.rgb_from_nv12(DeviceBuffer.from_numpy(raw, gpu_stream), 64, 32)
    assert np.array_equal(dec.numpy(), O.rgb_from_nv12(raw, 64, 32))
    enc = imgproc.nv12_from_rgb(dec)
    assert np.array_equal(enc.numpy(), O.nv12_from_rgb(dec.numpy()))


def test_cross_stream_destination_is_fenced(gpu_stream):
    """dst allocated (and zero-filled) on another stream: the dispatch fences it into the source's stream before the launch
    AND fences the launch stream back into dst's stream afterwards (DeviceExec::for_streams + run, P/cuda/dispatch.rs:50-82).
    The launch stream is kept busy with 4K blurs and nothing here synchronises it: ``dst.numpy()`` drains only dst's own
    stream, so without the fence back it would read the memset's zeros."""
    from kornia_rs import Image, Preprocessor, Stream, Tensor, imgproc
    from kornia_rs.hip import DeviceBuffer
    other = Stream.new(0)
    src = Image.from_numpy(O.pattern_f32(300 * 200 * 3).reshape(200, 300, 3)).to_hip(gpu_stream)
    want = O.gaussian_blur(src.numpy(), (5, 5), (1.0, 1.0))
    import os
    bw, bh, busy = (3840, 2160, 6) if not os.environ.get("KH_HOSTSIM") else (320, 180, 1)  # the fiber simulator has no asynchrony to hide
    big = Image.from_numpy(O.pattern_f32(bh * bw * 3).reshape(bh, bw, 3)).to_hip(gpu_stream)
    big_dst = Image.uninit(bw, bh, 3, "float32", gpu_stream)
    raw = O.pattern_u8(64 * 32 * 3 // 2)
    dev_raw = DeviceBuffer.from_numpy(raw, gpu_stream)
    pre = Preprocessor(mode="stretch", format="nv12", stream=gpu_stream)
    want_pre = O.preprocess(raw, 64, 32, 64, 32, fmt="nv12", mode="stretch")
    for _ in range(4):
        for _ in range(busy):
            imgproc.gaussian_blur(big, (7, 7), (1.5, 1.5), dst=big_dst)  # a few ms of queued work ahead of the op under test
        dst = Image.zeros(300, 200, 3, "float32", stream=other)  # memset queued on `other`
        imgproc.gaussian_blur(src, (5, 5), (1.0, 1.0), dst=dst)
        t = Tensor.zeros((1, 3, 32, 64), "float32", stream=other)
        pre.run_raw(dev_raw, 64, 32, t)
        assert np.array_equal(dst.numpy(), want)
        assert np.array_equal(t.numpy().view(np.uint32), want_pre.view(np.uint32))
    gpu_stream.synchronize()


def test_torch_rocm_zero_copy_interop(gpu_stream):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        import os
        if os.environ.get("KH_HOSTSIM") == "1":
            pytest.skip("host simulator: torch has no device to share a tensor with")
        pytest.fail("torch sees no HIP device on the GPU box")
    from kornia_rs import Image, Preprocessor, Stream, Tensor, dlpack, imgproc
    # 1. our device tensor -> torch (DLPack, kDLROCM) : same pointer, no copy
    raw = O.pattern_u8(64 * 32 * 3 // 2)
    pre = Preprocessor(mode="stretch", format="nv12", stream=gpu_stream)
    out = pre.run(raw, 64, 32, 32, 64)
    assert out.__dlpack_device__() == (dlpack.kDLROCM, 0)
    tt = torch.from_dlpack(out)
    assert tt.is_cuda and tt.data_ptr() == out.data_ptr and tuple(tt.shape) == (1, 3, 32, 64)
    torch.cuda.synchronize()
    assert np.array_equal(tt.cpu().numpy(), O.preprocess(raw, 64, 32, 64, 32, fmt="nv12", mode="stretch"))
    # 2. __cuda_array_interface__ consumer
    t2 = torch.as_tensor(out, device="cuda")
    assert t2.data_ptr() == out.data_ptr
    # 3. torch tensor -> our Image (zero-copy alias on torch's stream) -> device op -> back to torch
    x = torch.rand(97, 129, 3, device="cuda", dtype=torch.float32)
    ts = Stream.from_cuda_stream(torch.cuda.current_stream())
    img = Image.from_dlpack(x, stream=ts)
    assert img.is_device and img.data_ptr == x.data_ptr() and img.shape == (97, 129, 3)
    g = imgproc.gray_from_rgb(img)
    torch.cuda.synchronize()
    want = O.color_map("gray_from_rgb_f32", x.cpu().numpy(), 1).reshape(97, 129, 1)
    assert np.array_equal(g.numpy(), want)
    # keepalive: dropping the torch name must not free memory we still alias
    ptr = x.data_ptr()
    del x
    assert img.data_ptr == ptr and np.array_equal(imgproc.gray_from_rgb(img).numpy(), want)


def test_typed_video_buffers_decode_on_device(gpu_stream):
    import oracle_ffi as O
    from kornia_rs import color_spaces as cs, imgproc
    w, h = 64, 32
    raw = O.pattern_u8(w * h * 3 // 2)
    for cls, layout in ((cs.Nv12, 0), (cs.Nv21, 1), (cs.I420, 2), (cs.Yv12, 3)):
        got = imgproc.rgb_from_video(cls(w, h, raw).to_hip(gpu_stream)).cpu().numpy()
        assert np.array_equal(got, O.rgb_from_nv12(raw, w, h, layout))
    raw2 = O.pattern_u8(w * h * 2)
    for cls, layout in ((cs.Yuyv8, 0), (cs.Uyvy8, 1), (cs.Yvyu8, 2)):
        buf = cls(w, h, raw2).to_hip(gpu_stream)
        assert buf.is_device and np.array_equal(buf.cpu().as_slice(), raw2)
        got = imgproc.rgb_from_video(buf).cpu().numpy()
        assert np.array_equal(got, O.rgb_from_yuyv(raw2, w, h, layout))


def test_hip_allocators_place_tensors(gpu_stream):
    """a3 / a4: the HIP allocators behind `TensorAllocator` — device (zeroed / uninit), pinned host, unified — and the handle each
    tensor keeps (T/allocator.rs:73-144, T/cuda.rs:214-262, 355-380, 440-511)."""
    import kornia_rs as K
    from kornia_rs.allocator import HipAllocator, HipUnifiedAllocator, Layout, PinnedAllocator
    dev = HipAllocator(gpu_stream)
    r = dev.allocate(Layout(4096, 16))
    assert r.domain == "device" and r.len_bytes() == 4096 and r.as_ptr() % 16 == 0 and r.stream is gpu_stream
    t = K.Tensor.zeros_in((3, 5), "float32", dev)
    assert t.is_device and t.alloc is dev and not t.cpu().numpy().any()
    assert isinstance(K.Tensor.zeros((2, 2), "uint8", gpu_stream).alloc, HipAllocator)
    u = K.Tensor.uninit((2, 2), "uint8", gpu_stream)
    assert isinstance(u.alloc, HipAllocator) and not u.alloc.zeroed
    p = K.Tensor.zeros_in((16,), "uint8", PinnedAllocator())
    assert not p.is_device and p.is_pinned and not p.numpy().any()
    m = K.Tensor.zeros_in((4, 4), "float32", HipUnifiedAllocator(gpu_stream))
    assert m.is_unified and m.is_host_accessible and not m.numpy().any()
    assert isinstance(K.Tensor.from_numpy(np.ones(4, np.float32)).to_hip(gpu_stream).alloc, HipAllocator)
