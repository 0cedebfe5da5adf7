#!/usr/bin/env python3
"""Builds kornia_rs/data/colormaps.npy — the named 256-entry RGB colour maps that can be DERIVED here.

The reference bundles 21 OpenCV tables (crates/kornia-imgproc/src/color/colormap.rs:49-73).  Nothing is copied
from it: the tables below are rebuilt from their public definitions and only COMPARED with the reference when it is
mounted (and their SHA-256 digests are committed under tests/golden/colormaps/ so the CPU tests pin them anywhere):

* autumn / spring / cool / winter — OpenCV's `linear_colormap` construction (imgproc/src/colormap.cpp): 11 control
  points 0.1 apart, f32 `linspace` + `interp1`, `convertTo(CV_8U, 255)` (round half to even);
* magma / inferno / plasma / viridis / cividis / turbo — the published 256-entry float tables (matplotlib ships them
  verbatim), `round(255 * v)`;
* summer / hsv / bone / pink / hot / rainbow / ocean — 64 Octave control points (`summer(64)`, `hsv(64)`, ... from the
  piecewise-linear definitions in Octave's scripts/image/*.m) through the same `linear_colormap` interpolation;
* jet — 256 control points of min(4x - a, -4x + b) clipped to [0, 1] (matplotlib's `jet` segments), x = i / 255;
* twilight — matplotlib's 510-entry `twilight` table as control points through the same interpolation.

The other two (parula, deepgreen) are literal tables with no public closed form; `apply_colormap` names them in
its error and still takes any caller-provided 3x256 table.
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def cv_linspace(x0, x1, n):
    step = f32((f32(x1) - f32(x0)) / f32(n - 1))
    return np.array([f32(f32(x0) + f32(f32(i) * step)) for i in range(n)], f32)


def cv_interp1(X, Y, XI):
    Y = np.asarray(Y, f32)
    out = np.zeros(len(XI), f32)
    for i, xi in enumerate(XI):
        low, high = 0, len(X) - 1
        if xi < X[low]:
            high = 1
        if xi > X[high]:
            low = high - 1
        while high - low > 1:
            c = low + ((high - low) >> 1)
            if xi > X[c]:
                low = c
            else:
                high = c
        out[i] = f32(Y[low] + f32(f32(f32(xi - X[low]) * f32(Y[high] - Y[low])) / f32(X[high] - X[low])))
    return out


def cv_linear_colormap(r, g, b):
    X, XI = cv_linspace(0, 1, len(r)), cv_linspace(0, 1, 256)
    return np.stack([np.rint(cv_interp1(X, c, XI) * f32(255.0)).astype(np.uint8) for c in (r, g, b)])


def build():
    t = [i / 10 for i in range(11)]
    one, zero = [1.0] * 11, [0.0] * 11
    maps = {
        "autumn": cv_linear_colormap(one, t, zero),
        "spring": cv_linear_colormap(one, t, t[::-1]),
        "cool": cv_linear_colormap(t, t[::-1], one),
        "winter": cv_linear_colormap(zero, t, [1.0 - 0.05 * i for i in range(11)]),
    }
    # summer / hsv — OpenCV interpolates 64 Octave control points: summer(64) = [x, 0.5 + x/2, 0.4], hsv(64) = hsv2rgb(linspace(0, 1, 64), 1, 1)
    import colorsys
    x64 = [k / 63.0 for k in range(64)]
    maps["summer"] = cv_linear_colormap(x64, [0.5 + v / 2 for v in x64], [0.4] * 64)
    hsv = [colorsys.hsv_to_rgb(v % 1.0, 1.0, 1.0) for v in x64]
    maps["hsv"] = cv_linear_colormap([c[0] for c in hsv], [c[1] for c in hsv], [c[2] for c in hsv])
    # Octave's piecewise-linear definitions on x = linspace(0, 1, 64)
    x = np.linspace(0, 1, 64)
    w = np.where
    maps["rainbow"] = cv_linear_colormap(w(x < 2 / 5, 1, w(x < 3 / 5, -5 * x + 3, w(x < 4 / 5, 0, 10 / 3 * x - 8 / 3))),
                                         w(x < 2 / 5, 5 / 2 * x, w(x < 3 / 5, 1, w(x < 4 / 5, -5 * x + 4, 0))),
                                         w(x < 3 / 5, 0, w(x < 4 / 5, 5 * x - 3, 1)))
    maps["bone"] = cv_linear_colormap(w(x < 3 / 4, 7 / 8 * x, 11 / 8 * x - 3 / 8),
                                      w(x < 3 / 8, 7 / 8 * x, w(x < 3 / 4, 29 / 24 * x - 1 / 8, 7 / 8 * x + 1 / 8)),
                                      w(x < 3 / 8, 29 / 24 * x, 7 / 8 * x + 1 / 8))
    maps["pink"] = cv_linear_colormap(np.sqrt(w(x < 3 / 8, 14 / 9 * x, 2 / 3 * x + 1 / 3)),
                                      np.sqrt(w(x < 3 / 8, 2 / 3 * x, w(x < 3 / 4, 14 / 9 * x - 1 / 3, 2 / 3 * x + 1 / 3))),
                                      np.sqrt(w(x < 3 / 4, 2 / 3 * x, 2 * x - 1)))
    maps["hot"] = cv_linear_colormap(w(x < 2 / 5, 5 / 2 * x, 1.0), w(x < 2 / 5, 0, w(x < 4 / 5, 5 / 2 * x - 1, 1.0)), w(x < 4 / 5, 0, 5 * x - 4))
    cutin = 64 // 3
    ramp = lambda step: np.arange(0, 63 + 1e-9, step)
    pad = lambda v: np.concatenate([np.zeros(64 - len(v)), v]) / 63
    maps["ocean"] = cv_linear_colormap(pad(ramp(63 / cutin)), pad(ramp(63 / (2 * cutin))), np.arange(64) / 63)
    x = np.arange(256) / 255.0
    clip = lambda v: np.clip(v, 0, 1)
    maps["jet"] = cv_linear_colormap(clip(np.minimum(4 * x - 1.5, -4 * x + 4.5)), clip(np.minimum(4 * x - 0.5, -4 * x + 3.5)),
                                     clip(np.minimum(4 * x + 0.5, -4 * x + 2.5)))
    from matplotlib import colormaps
    tw = np.array(colormaps["twilight"].colors)
    maps["twilight"] = cv_linear_colormap(tw[:, 0], tw[:, 1], tw[:, 2])
    for name in ("magma", "inferno", "plasma", "viridis", "cividis", "turbo"):
        maps[name] = np.rint(colormaps[name](x)[:, :3] * 255.0).astype(np.uint8).T.copy()
    return maps


def reference_tables():
    path = "/root/reference/crates/kornia-imgproc/src/color/colormap_luts.rs"
    if not os.path.exists(path):
        return None
    out = {}
    for m in re.finditer(r"static (\w+)_LUT: ColormapLut = ColormapLut \{(.*?)\n\};", open(path).read(), re.S):
        chans = [np.array([int(v) for v in re.findall(r"\d+", arr)], np.uint8) for arr in re.findall(r"[rgb]: \[(.*?)\]", m.group(2), re.S)]
        out[m.group(1).lower()] = np.stack(chans)
    return out


def main():
    maps = build()
    ref = reference_tables()
    digests = {}
    if ref is not None:
        for name, lut in maps.items():
            if not np.array_equal(lut, ref[name]):
                sys.exit(f"{name}: derived table differs from the reference's")
        digests = {name: hashlib.sha256(t.tobytes()).hexdigest() for name, t in sorted(ref.items())}
        with open(os.path.join(ROOT, "tests", "golden", "colormaps", "reference_sha256.json"), "w") as f:
            json.dump(digests, f, indent=1, sort_keys=True)
    names = sorted(maps)
    np.save(os.path.join(ROOT, "kornia-rs_amd", "kornia_rs", "data", "colormaps.npy"), np.stack([maps[n] for n in names]))
    with open(os.path.join(ROOT, "kornia-rs_amd", "kornia_rs", "data", "colormaps.json"), "w") as f:
        json.dump(names, f)
    print(f"{len(names)} tables written ({', '.join(names)}); reference check: {'passed' if ref is not None else 'skipped'}")


if __name__ == "__main__":
    main()
