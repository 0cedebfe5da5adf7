"""Camera calibration parameters and the Brown-Conrady point model (host side).

Mirrors ``kornia_imgproc::calibration`` (crates/kornia-imgproc/src/calibration/mod.rs:12-40,
calibration/distortion.rs:23-152): typed ``CameraIntrinsic`` / ``PolynomialDistortion``, the f64 per-point
model (``distort_point_polynomial``; once-per-camera host arithmetic) and the device map builder
``generate_correction_map_polynomial`` whose maps feed ``imgproc.remap``; ``undistort_image`` chains the two.
"""
from dataclasses import astuple, dataclass
from typing import Optional, Tuple

from . import imgproc
from .hip import Stream
from .image import Image, ImageError


@dataclass(frozen=True)
class CameraIntrinsic:
    """Pinhole intrinsics (calibration/mod.rs:12-21)."""
    fx: float
    fy: float
    cx: float
    cy: float


@dataclass(frozen=True)
class PolynomialDistortion:
    """Radial k1..k6 (rational model) and tangential p1, p2 (calibration/distortion.rs:23-40); all default 0."""
    k1: float = 0.0
    k2: float = 0.0
    k3: float = 0.0
    k4: float = 0.0
    k5: float = 0.0
    k6: float = 0.0
    p1: float = 0.0
    p2: float = 0.0


def distort_point_polynomial(x: float, y: float, intrinsic: CameraIntrinsic,
                             distortion: PolynomialDistortion) -> Tuple[float, float]:
    """Distorted pixel position of the undistorted pixel ``(x, y)`` — f64, same evaluation order as
    calibration/distortion.rs:68-110 (Python floats are IEEE binary64, no contraction)."""
    fx, fy, cx, cy = (float(v) for v in astuple(intrinsic))
    k1, k2, k3, k4, k5, k6, p1, p2 = (float(v) for v in astuple(distortion))
    x = (float(x) - cx) / fx
    y = (float(y) - cy) / fy
    r2 = x * x + y * y
    r4 = r2 * r2
    r6 = r4 * r2
    kr = (1.0 + k1 * r2 + k2 * r4 + k3 * r6) / (1.0 + k4 * r2 + k5 * r4 + k6 * r6)
    x_2 = 2.0 * x
    y_2 = 2.0 * y
    xy_2 = x_2 * y
    xd = x * kr + xy_2 * p1 + p2 * (r2 + x_2 * x)
    yd = y * kr + p1 * (r2 + y_2 * y) + xy_2 * p2
    return fx * xd + cx, fy * yd + cy


def generate_correction_map_polynomial(intrinsic: CameraIntrinsic, distortion: PolynomialDistortion,
                                       size: Tuple[int, int], stream: Stream) -> Tuple[Image, Image]:
    """Device ``(map_x, map_y)`` for ``size = (width, height)`` (calibration/distortion.rs:135-152; the extrinsic /
    new-intrinsic arguments of the reference are unused there and are not taken here)."""
    return imgproc.generate_correction_map_polynomial(astuple(intrinsic), astuple(distortion), size, stream)


def undistort_image(src: Image, intrinsic: CameraIntrinsic, distortion: PolynomialDistortion,
                    interpolation: str = "bilinear", dst: Optional[Image] = None,
                    maps: Optional[Tuple[Image, Image]] = None) -> Image:
    """Undistort a device image: correction maps (built on the image's stream unless ``maps`` carries the cached
    pair) followed by ``remap`` — the ``examples/undistort`` flow."""
    if not src.is_device:
        raise ImageError("HostPathUnavailable",
                                 "undistort_image: host image — this build provides the HIP device backend only")
    if maps is None:
        maps = generate_correction_map_polynomial(intrinsic, distortion, (src.width, src.height), src.stream)
    return imgproc.remap(src, maps[0], maps[1], interpolation, dst)
