"""Named colour maps — ``ColormapType`` / ``ColormapLut`` of crates/kornia-imgproc/src/color/colormap.rs:49-100.

All 21 reference names parse (``ColormapType.from_name`` is case-insensitive like the reference's) and all 21 tables are bundled
in ``data/colormaps.npy``: nineteen are rebuilt from their public definitions by ``scripts/gen_colormaps.py``; parula and
deepgreen have no public closed form and are shipped as the constant 3 x 256 byte tables they are (the same bytes as
``colormap_luts.rs``; every table's SHA-256 is checked against the reference's, ``tests/golden/colormaps/``).
``apply_colormap`` also takes any caller-provided 3 x 256 table.
"""
import enum
import json
import os
from typing import Optional, Union

import numpy as np

from .image import ImageError

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


class ColormapType(enum.Enum):
    AUTUMN = "autumn"; BONE = "bone"; JET = "jet"; WINTER = "winter"; RAINBOW = "rainbow"; OCEAN = "ocean"
    SUMMER = "summer"; SPRING = "spring"; COOL = "cool"; HSV = "hsv"; PINK = "pink"; HOT = "hot"; PARULA = "parula"
    MAGMA = "magma"; INFERNO = "inferno"; PLASMA = "plasma"; VIRIDIS = "viridis"; CIVIDIS = "cividis"
    TWILIGHT = "twilight"; TURBO = "turbo"; DEEPGREEN = "deepgreen"

    @staticmethod
    def from_name(name: str) -> Optional["ColormapType"]:  # colormap.rs:78-84
        try:
            return ColormapType(str(name).lower())
        except ValueError:
            return None


_tables = None


def _load():
    global _tables
    if _tables is None:
        with open(os.path.join(_DATA, "colormaps.json")) as f:
            names = json.load(f)
        data = np.load(os.path.join(_DATA, "colormaps.npy"))
        if data.shape != (len(names), 3, 256) or data.dtype != np.uint8:
            raise ImageError("ImageDataNotInitialized", "colormaps.npy does not match its index")
        _tables = {n: data[i] for i, n in enumerate(names)}
    return _tables


def bundled() -> list:
    return sorted(_load())


def lut(colormap: Union[str, ColormapType]) -> np.ndarray:
    """The (3, 256) uint8 table ``r[256], g[256], b[256]`` of a named map (``ColormapType::lut``, colormap.rs:86-93)."""
    kind = colormap if isinstance(colormap, ColormapType) else ColormapType.from_name(colormap)
    if kind is None:
        raise ImageError("InvalidArgument", f"unknown colormap {colormap!r}; names: {', '.join(k.value for k in ColormapType)}")
    table = _load().get(kind.value)
    if table is None:
        raise ImageError("InvalidArgument", f"colormap {kind.value!r}: its table is not bundled in this build (bundled: "
                                            f"{', '.join(bundled())}); pass the 3x256 table itself to apply_colormap")
    return table
