"""The three memory domains of the tensor model on the device: Host, Device and Unified
(managed memory: ``MemoryDomain::Unified`` T/resource.rs:19-60, ``CudaUnifiedAllocator`` T/cuda.rs:440-511,
``Image::{zeros_cuda_unified, to_cuda_unified, zeros_pinned}`` I/cuda.rs:53-221), plus the DLPack device codes they
export under (T/dlpack.rs:76-84)."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def test_unified_image_is_host_and_device_accessible(gpu_stream):
    from kornia_rs import Image, dlpack, hip, imgproc
    from kornia_rs.tensor import MemoryDomain
    w, h = 129, 97
    img = Image.zeros_hip_unified(w, h, 3, "uint8", gpu_stream)
    assert img.domain == MemoryDomain.UNIFIED and img.is_unified and img.is_device  # dispatch treats it as device-resident
    assert img.device == "cuda:0" and img.stream is gpu_stream
    assert hip.pointer_domain(img.data_ptr) == (2, 0)           # KH_DOMAIN_UNIFIED on device 0
    assert img.__dlpack_device__() == (dlpack.kDLCUDAManaged, 0)
    view = img.numpy()                                           # the managed bytes themselves: no copy, writable
    assert view.ctypes.data == img.data_ptr and view.flags.writeable and not view.any()
    src = O.pattern_u8(w * h * 3).reshape(h, w, 3)
    view[...] = src                                              # HOST write ...
    assert np.array_equal(img.as_slice(), src.reshape(-1))
    gray = imgproc.gray_from_rgb(img)                            # ... visible to a KERNEL without any upload
    assert not gray.is_unified and gray.is_device
    assert np.array_equal(gray.numpy(), O.gray_from_rgb_u8(src))
    # unified destination: the kernel's result is readable through the host view after the stream drains (numpy() does)
    out = Image.zeros_cuda_unified(w, h, 1, "uint8", gpu_stream)
    assert imgproc.gray_from_rgb(img, dst=out) is out
    assert np.array_equal(out.numpy(), O.gray_from_rgb_u8(src))
    # device source -> unified destination and back: both count as device operands, no MixedResidency
    dev = Image.from_numpy(src).to_hip(gpu_stream)
    out2 = Image.zeros_hip_unified(w, h, 1, "uint8", gpu_stream)
    imgproc.gray_from_rgb(dev, dst=out2)
    assert np.array_equal(out2.numpy(), O.gray_from_rgb_u8(src))


def test_to_hip_unified_and_cpu_copy(gpu_stream):
    from kornia_rs import Image, ImageError, imgproc
    src = O.pattern_f32(64 * 48 * 3).reshape(48, 64, 3)
    host = Image.from_numpy(src)
    uni = host.to_hip_unified(gpu_stream)
    assert uni.is_unified and np.array_equal(uni.numpy(), src) and uni.data_ptr != host.data_ptr
    assert Image.to_cuda_unified is Image.to_hip_unified
    with pytest.raises(ImageError):
        uni.to_hip_unified(gpu_stream)                           # only host images are uploaded
    got = imgproc.gaussian_blur(uni, (5, 5), (1.0, 1.0))
    assert np.array_equal(got.numpy(), O.gaussian_blur(src, (5, 5), (1.0, 1.0)))
    back = uni.cpu()                                             # an owned host copy, detached from the managed bytes
    assert not back.is_device and np.array_equal(back.numpy(), src)
    uni.numpy()[0, 0, 0] = 7.0
    assert back.numpy()[0, 0, 0] == src[0, 0, 0]
    with pytest.raises(ImageError) as e:                         # host + unified is still a mixed pair
        imgproc.gaussian_blur(host, (5, 5), (1.0, 1.0), dst=uni)
    assert e.value.kind == "MixedResidency"


def test_unified_tensor_as_preprocess_destination(gpu_stream):
    from kornia_rs import Preprocessor, Tensor
    from kornia_rs.hip import DeviceBuffer
    raw = O.pattern_u8(64 * 32 * 3 // 2)
    dst = Tensor.zeros_unified((1, 3, 32, 64), "float32", gpu_stream)
    assert dst.is_device and dst.is_unified and dst.is_host_accessible
    Preprocessor(mode="stretch", format="nv12", stream=gpu_stream).run_raw(DeviceBuffer.from_numpy(raw, gpu_stream), 64, 32, dst)
    want = O.preprocess(raw, 64, 32, 64, 32, fmt="nv12", mode="stretch")
    assert np.array_equal(dst.numpy().view(np.uint32), want.view(np.uint32))


def test_pinned_host_image(gpu_stream):
    from kornia_rs import Image, dlpack, hip
    pin = Image.zeros_pinned(40, 30, 3, "uint8")
    assert not pin.is_device and pin.domain == "host" and pin.tensor.is_pinned
    assert hip.pointer_domain(pin.data_ptr)[0] == 3              # KH_DOMAIN_HOST_PINNED
    assert pin.__dlpack_device__() == (dlpack.kDLROCMHost, 0)
    src = O.pattern_u8(40 * 30 * 3).reshape(30, 40, 3)
    pin.numpy()[...] = src
    dev = pin.to_hip(gpu_stream)
    assert dev.is_device and np.array_equal(dev.numpy(), src)
    back = Image.from_dlpack(pin)                                # kDLROCMHost imports as a host image, zero-copy
    assert not back.is_device and back.data_ptr == pin.data_ptr and back.tensor.is_pinned


def test_unified_dlpack_round_trips(gpu_stream):
    from kornia_rs import Tensor, dlpack
    t = Tensor.zeros_unified((5, 7), "float32", gpu_stream)
    t.numpy()[...] = np.arange(35, dtype=np.float32).reshape(5, 7)
    again = Tensor.from_dlpack(t)                                # kDLCUDAManaged -> Unified, same bytes
    assert again.is_unified and again.data_ptr == t.data_ptr and np.array_equal(again.numpy(), t.numpy())
    dev = Tensor.from_dlpack(_Capsule(t.__dlpack__(dl_device=(dlpack.kDLROCM, 0)), (dlpack.kDLROCM, 0)), stream=gpu_stream)
    assert dev.is_device and not dev.is_unified and dev.data_ptr == t.data_ptr
    host = Tensor.from_dlpack(_Capsule(t.__dlpack__(dl_device=(dlpack.kDLCPU, 0)), (dlpack.kDLCPU, 0)))
    assert not host.is_device and host.data_ptr == t.data_ptr and host.numpy()[4, 6] == 34.0
    with pytest.raises(BufferError):
        t.__dlpack__(dl_device=(dlpack.kDLROCM, 5))
    with pytest.raises(BufferError):
        Tensor.zeros((2, 2), "float32", stream=gpu_stream).__dlpack__(dl_device=(dlpack.kDLCPU, 0))  # export never copies


def test_unified_torch_consumer(gpu_stream):
    """torch-ROCm has no managed DLPack code; it asks for the device view of the same allocation."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        import os
        if os.environ.get("KH_HOSTSIM") == "1":
            pytest.skip("host simulator: torch has no device to share a tensor with")
        pytest.fail("torch sees no HIP device on the GPU box")
    from kornia_rs import Tensor, dlpack
    t = Tensor.zeros_unified((4, 6), "float32", gpu_stream)
    t.numpy()[...] = 2.0
    tt = torch.from_dlpack(_Capsule(t.__dlpack__(dl_device=(dlpack.kDLROCM, 0)), (dlpack.kDLROCM, 0)))
    assert tt.is_cuda and tt.data_ptr() == t.data_ptr
    tt.mul_(3.0)
    torch.cuda.synchronize()
    assert float(t.numpy()[3, 5]) == 6.0                         # the device write is visible through the host view


class _Capsule:
    """A minimal exporter handing out one prepared capsule (lets a test pick the dl_device it was made for)."""

    def __init__(self, capsule, device):
        self._capsule, self._device = capsule, device

    def __dlpack__(self, **kw):
        return self._capsule

    def __dlpack_device__(self):
        return self._device
