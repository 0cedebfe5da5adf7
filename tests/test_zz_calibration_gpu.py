"""``kornia_rs.calibration`` on a device: typed camera parameters -> correction maps -> remap, against the CPU
restatement (P/calibration/distortion.rs:135-152, examples/undistort).  Sorted last on purpose: it chains entry
points the earlier files already pin individually."""
from dataclasses import astuple

import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def test_undistort_image_matches_maps_then_remap(gpu_stream):
    from kornia_rs import Image, ImageError, calibration, imgproc
    intr = calibration.CameraIntrinsic(300.0, 300.0, 64.0, 48.0)
    dist = calibration.PolynomialDistortion(k1=0.1, k2=0.01, p1=1e-4, p2=1e-4)
    host = O.pattern_f32(129 * 97 * 3).reshape(97, 129, 3)
    src = Image.from_numpy(host).to_hip(gpu_stream)
    wx, wy = O.correction_map(astuple(intr), astuple(dist), 129, 97)
    want = O.remap(host, wx, wy)
    got = calibration.undistort_image(src, intr, dist)
    assert got.is_device and np.array_equal(got.numpy(), want)
    # cached maps + nearest, and the u8 twin through the same maps
    maps = calibration.generate_correction_map_polynomial(intr, dist, (129, 97), gpu_stream)
    assert np.array_equal(maps[0].numpy()[:, :, 0], wx) and np.array_equal(maps[1].numpy()[:, :, 0], wy)
    near = calibration.undistort_image(src, intr, dist, "nearest", maps=maps)
    assert np.array_equal(near.numpy(), O.remap(host, wx, wy, "nearest"))
    rgb = O.pattern_u8(129 * 97 * 3).reshape(97, 129, 3)
    und8 = calibration.undistort_image(Image.from_numpy(rgb).to_hip(gpu_stream), intr, dist, maps=maps)
    assert np.array_equal(und8.numpy(), O.remap_u8(rgb, wx, wy))
    with pytest.raises(ImageError) as e:
        calibration.undistort_image(Image.from_numpy(host), intr, dist)
    assert e.value.kind == "HostPathUnavailable"
    assert imgproc.crop is imgproc.crop_image
