"""``kornia_rs`` — host-side mirror of the reference's Python package for the imgproc hot path,
running on the MI355X HIP backend (``libkornia_hip.so``) instead of CUDA.

Only the hot-path surface exists here: ``Image`` / ``Tensor`` with host/device
residency, ``Stream``, the ``imgproc`` free functions and the fused ``Preprocessor``.
Importing the package loads the shared library and fails loudly if it is missing.
"""
from . import _ffi
from . import hip
from .hip import IMAGENET_MEAN, IMAGENET_STD, Stream
from . import allocator
from .allocator import CpuAllocator, HipAllocator, HipUnifiedAllocator, Layout, PinnedAllocator, TensorAllocator, TensorAllocatorError, host_alloc
from .tensor import Tensor
from .image import Image, ImageError
from .preprocess import Normalize, Preprocessor, PreprocessorBuilder, PreprocessError, ResizeMode, SourceFormat
from . import imgproc
from . import fusion
from . import color_spaces
from . import calibration
from . import colormap
from .colormap import ColormapType
from .color_spaces import ColorSpace
from . import sharding
from . import rust_api

cuda = hip  # reference module name (kornia_rs.cuda.Stream); same objects, HIP underneath

__version__ = "0.1.0"
__all__ = [
    "IMAGENET_MEAN", "IMAGENET_STD", "Stream", "Tensor", "Image", "ImageError", "Preprocessor", "PreprocessorBuilder", "Normalize",
    "PreprocessError", "ResizeMode", "SourceFormat", "imgproc", "rust_api", "fusion", "color_spaces", "ColorSpace", "calibration", "colormap", "ColormapType", "hip", "cuda", "sharding",
    "allocator", "TensorAllocator", "TensorAllocatorError", "CpuAllocator", "PinnedAllocator", "HipAllocator", "HipUnifiedAllocator", "Layout", "host_alloc",
]
