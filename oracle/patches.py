"""Oracle (test infrastructure): the multi-scale patch pyramid the reference's data loaders cut around a
sample point, restated from source (GDAL / OGR / cv2 are absent here).

Reference: MyUtils1.py (= MyUtils2.py:286-437 for inference)
  get_scales                          :130-156   windows [inner, object, object+interval, object+2*interval],
                                                 factors[i] = window[i] / configs.scales[i], configs.scales = [32,64,128,1] (config.py:32)
  geo -> pixel                        :67-73     XPixel = int(abs((gt[0]-XGeo)/gt[1]) + 1), YLine likewise (note the +1)
  calculate_left_top_point_and_size   :219-223   top-left = int(mid - L/2)  (Python int(): truncation toward zero)
  cut_image                           :162-200   window clipped to the raster, zero-padded into uint8 [bands, L, L]
  resize_data                         :202-216   per band cv2.resize(band, (t, t), INTER_AREA) on uint8, stack, /255.0 -> float32

PARITY UNPINNED for the resize: OpenCV is not installed, the reference pins no version and has no test that
fixes its output.  This build therefore DEFINES the resize as the exact area average

    out[oy, ox] = round_half_even( sum_{iy, ix} ov_y(oy, iy) * ov_x(ox, ix) * in[iy, ix] / L^2 )

where, in units of 1/t of an input pixel, input pixel i covers [i*t, (i+1)*t) and output pixel o covers
[o*L, (o+1)*L); ov = length of the intersection (an integer), so sum_i ov(o, i) = L.  The same rule is used for
L > t (box filter, the case INTER_AREA is meant for), L == t (identity) and L < t (a footprint inside one or two
input pixels).  KNOWN DIFFERENCES from OpenCV's uint8 INTER_AREA, as far as its published algorithm goes (cv::resize,
imgproc/resize.cpp): (1) the exact 2x reduction rounds half UP there ((a+b+c+d+2)>>2), other integer ratios multiply an integer
sum by a float reciprocal and round half to even, non-integer ratios accumulate float weights -- so ties and near-ties can differ by
one LSB from the exact rational rounding used here; (2) for L < t (enlarging) OpenCV does not area-average at all: it switches to a
bilinear interpolation with area-style coordinates, which this spec does not imitate.  Windows with L < t occur (inner ~ 16..31 px
into the 32 px target); a checkpoint trained on OpenCV-made patches therefore sees slightly different inputs for those windows.
Pinned by the reference itself: L == t (identity) and everything AROUND the resize -- tests/golden/sweep.npz.  Everything is integer arithmetic, hence bit-exact between this oracle and the GPU kernel
(dm_patch_pyramid).  Everything else above (windows, truncation, clipping, zero padding, /255) is restated 1:1.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

CONFIG_SCALES = (32, 64, 128, 1)     # config.py:32 `scales`


def get_scales(inner: int, obj: int) -> Tuple[List[int], List[float]]:
    """MyUtils1.py:130-156."""
    interval = int(obj - inner)
    windows = [inner, obj, obj + interval, obj + 2 * interval]
    factors = [windows[i] * 1.0 / CONFIG_SCALES[i] for i in range(4)]
    return windows, factors


def geo_to_pixel(gt: Sequence[float], x_geo: float, y_geo: float) -> Tuple[int, int]:
    """MyUtils1.py:67-73 (GDAL geotransform gt)."""
    return int(abs((gt[0] - x_geo) / gt[1]) + 1), int(abs((gt[3] - y_geo) / gt[5]) + 1)


def top_left(mid_x: int, mid_y: int, L: int) -> Tuple[int, int]:
    """MyUtils1.py:219-223: int() truncates toward zero."""
    return int(mid_x - L / 2), int(mid_y - L / 2)


def cut_image(img: np.ndarray, x0: int, y0: int, L: int) -> np.ndarray:
    """MyUtils1.py:162-200: uint8 [bands, L, L], zero outside the raster."""
    bands, H, W = img.shape
    dst = np.zeros((bands, L, L), dtype=np.uint8)
    xs, xe = max(0, x0), min(W, x0 + L)
    ys, ye = max(0, y0), min(H, y0 + L)
    if xe > xs and ye > ys:
        dst[:, ys - y0:ye - y0, xs - x0:xe - x0] = img[:, ys:ye, xs:xe]
    return dst


def overlap_matrix(L: int, t: int) -> np.ndarray:
    """ov[o, i] (int64 [t, L]): overlap of output pixel o with input pixel i in units of 1/t input pixels."""
    o = np.arange(t, dtype=np.int64)[:, None]
    i = np.arange(L, dtype=np.int64)[None, :]
    lo = np.maximum(o * L, i * t)
    hi = np.minimum((o + 1) * L, (i + 1) * t)
    return np.maximum(hi - lo, 0)


def area_resize_u8(band: np.ndarray, t: int) -> np.ndarray:
    """uint8 [L, L] -> uint8 [t, t] by the exact area average defined in the module docstring."""
    L = band.shape[0]
    ov = overlap_matrix(L, t)
    num = ov @ band.astype(np.int64) @ ov.T                  # exact
    den = np.int64(L) * np.int64(L)
    q, r = num // den, num % den
    up = (2 * r > den) | ((2 * r == den) & (q % 2 == 1))     # round half to even
    return (q + up).astype(np.uint8)


def patch_pyramid(img: np.ndarray, x: int, y: int, windows: Sequence[int], targets: Sequence[int] = CONFIG_SCALES) -> List[np.ndarray]:
    """float32 [bands, t_i, t_i] per scale: crop (zero-padded) -> exact-area resize on uint8 -> /255.0."""
    out = []
    for L, t in zip(windows, targets):
        x0, y0 = top_left(x, y, L)
        win = cut_image(img, x0, y0, L)
        res = np.stack([area_resize_u8(win[b], t) for b in range(win.shape[0])])
        out.append(res.astype(np.float32) / 255.0)
    return out
