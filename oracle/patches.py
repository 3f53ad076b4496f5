"""Oracle (test infrastructure): the multi-scale patch pyramid the reference's data loaders cut around a
sample point, restated from source (GDAL / OGR / cv2 are absent here).

Reference: MyUtils1.py (= MyUtils2.py:286-437 for inference)
  get_scales                          :130-156   windows [inner, object, object+interval, object+2*interval],
                                                 factors[i] = window[i] / configs.scales[i], configs.scales = [32,64,128,1] (config.py:32)
  geo -> pixel                        :67-73     XPixel = int(abs((gt[0]-XGeo)/gt[1]) + 1), YLine likewise (note the +1)
  calculate_left_top_point_and_size   :219-223   top-left = int(mid - L/2)  (Python int(): truncation toward zero)
  cut_image                           :162-200   window clipped to the raster, zero-padded into uint8 [bands, L, L]
  resize_data                         :202-216   per band cv2.resize(band, (t, t), INTER_AREA) on uint8, stack, /255.0 -> float32

PARITY UNPINNED for the resize: OpenCV is not installed here (and nothing may be installed: no network), the reference pins no
version and has no test that fixes its output.  `cv_resize_area_u8` restates the published algorithm of
cv::resize(..., INTER_AREA) for single-channel uint8 (OpenCV 4.x, modules/imgproc/src/resize.cpp, x86-64 baseline build: plain C++
float arithmetic without FMA contraction), branch by branch:

  * scale = L / t an integer (`is_area_fast`, resizeAreaFast_): 1 -> copy; 2 -> (a + b + c + d + 2) >> 2; k > 2 -> integer sum of the
    k x k block times float(1 / k^2), cvRound (round half to even);
  * L > t, not an integer ratio (computeResizeAreaTab + ResizeArea_Invoker<uchar, float>): per axis a table of (source index, float
    weight) -- a leading fractional cell (sx1 - fsx1) / cellWidth when it exceeds 1e-3, whole cells 1 / cellWidth, a trailing fractional
    cell -- with fsx1 = dx * scale in double and scale = 1 / ((double) t / L); a source row is folded along x in float in table order
    (buf += S * alpha), rows are folded in float (sum = beta * buf, then sum += beta * buf), cvRound at the end;
  * L < t (INTER_AREA is not implemented for enlarging: the bilinear code runs with area-style coordinates, `area_mode`):
    sx = floor(dx * scale), fx = (float)((dx + 1) - (sx + 1) * inv_scale), fx <= 0 ? 0 : fx - floor(fx), fx = 0 at the last source
    column; coefficients cvRound((1 - fx) * 2048), cvRound(fx * 2048) (INTER_RESIZE_COEF_BITS = 11); horizontal pass in int32
    (S[sx] * a0 + S[sx + 1] * a1, or S[sx] * 2048 where sx + 1 is outside), vertical pass
    (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 with the second row clipped to the last one.

It is the DEFAULT rule of the accelerated path since round 3 ("opencv"); it is unpinned but faithful: no fixture produced by cv2 itself
exists, so a one-LSB difference to a particular OpenCV build (SIMD kernels, HAL backends, FMA contraction on other targets) cannot
be excluded.  The round-1/2 rule stays available as "exact_area" (`area_resize_u8`): the exact rational area average

    out[oy, ox] = round_half_even( sum_{iy, ix} ov_y(oy, iy) * ov_x(ox, ix) * in[iy, ix] / L^2 )

where, in units of 1/t of an input pixel, input pixel i covers [i*t, (i+1)*t) and output pixel o covers [o*L, (o+1)*L); ov = length
of the intersection (an integer).  It differs from OpenCV in ties (2x rounds half up there), in the float weights of non-integer
ratios and, fundamentally, for L < t, where OpenCV interpolates bilinearly.
Pinned by the reference itself: L == t (identity under both rules) and everything AROUND the resize -- tests/golden/sweep.npz.
Both rules are bit-exact between this oracle and the GPU kernel (dm_patch_pyramid): "exact_area" is integer arithmetic, "opencv"
is float32 arithmetic in a fixed order with explicit roundings (no FMA contraction on either side).
Everything else above (windows, truncation, clipping, zero padding, /255) is restated 1:1.
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


INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def _cv_round(x) -> np.ndarray:
    """cvRound on float32 / float64 values: round half to even (SSE cvtss2si / lrint under the default rounding mode)."""
    return np.rint(x).astype(np.int64)


def cv_area_tab(ssize: int, dsize: int):
    """computeResizeAreaTab (resize.cpp): list over dx of [(si, float32 alpha), ...] in table order."""
    inv_scale = np.float64(dsize) / np.float64(ssize)
    scale = np.float64(1.0) / inv_scale
    tab = []
    for dx in range(dsize):
        fsx1 = np.float64(dx) * scale
        fsx2 = fsx1 + scale
        cell = min(scale, np.float64(ssize) - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        ent = []
        if sx1 - fsx1 > 1e-3:
            ent.append((sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            ent.append((sx, np.float32(np.float64(1.0) / cell)))
        if fsx2 - sx2 > 1e-3:
            ent.append((sx2, np.float32(min(min(fsx2 - sx2, np.float64(1.0)), cell) / cell)))
        tab.append(ent)
    return tab


def cv_linear_coeffs(ssize: int, dsize: int):
    """The `area_mode` branch of cv::resize's coefficient loop for ksize = 2, uint8 (fixed point).  Returns (ofs, a0, a1, dmax):
    source index, the two 11-bit coefficients per destination index, and the first destination index whose right neighbour lies outside
    (xmax; the horizontal pass uses S[ofs] * 2048 from there on)."""
    inv_scale = np.float64(dsize) / np.float64(ssize)
    scale = np.float64(1.0) / inv_scale
    ofs = np.zeros(dsize, np.int64); a0 = np.zeros(dsize, np.int64); a1 = np.zeros(dsize, np.int64)
    dmax = dsize
    for d in range(dsize):
        sx = int(np.floor(np.float64(d) * scale))
        fx = np.float32(np.float64(d + 1) - np.float64(sx + 1) * inv_scale)
        fx = np.float32(0.0) if fx <= 0 else np.float32(fx - np.float32(np.floor(fx)))
        if sx + 1 >= ssize:
            dmax = min(dmax, d)
            if sx >= ssize - 1:
                fx, sx = np.float32(0.0), ssize - 1
        ofs[d] = sx
        a0[d] = _cv_round(np.float32(np.float32(1.0) - fx) * np.float32(INTER_RESIZE_COEF_SCALE))
        a1[d] = _cv_round(fx * np.float32(INTER_RESIZE_COEF_SCALE))
    return ofs, a0, a1, dmax


def cv_is_area_fast(L: int, t: int) -> bool:
    """cv::resize's `is_area_fast` for a square L -> t shrink: scale = 1. / ((double)t / L) is compared with its rounded value
    against DBL_EPSILON.  For most integer ratios k = L / t the double quotient is exactly k; for some (k = 49, 93, 98, 99, 103, ...:
    1 / (1 / k) != k in binary64) it is k +- 1e-14, the test fails and OpenCV takes the float area-table path instead of the integer
    block mean -- e.g. L = 1568, t = 32."""
    if L % t != 0:
        return False
    scale = np.float64(1.0) / (np.float64(t) / np.float64(L))
    return bool(abs(scale - np.rint(scale)) < np.finfo(np.float64).eps)


def cv_resize_area_u8(band: np.ndarray, t: int) -> np.ndarray:
    """uint8 [L, L] -> uint8 [t, t] as cv2.resize(band, (t, t), interpolation=cv2.INTER_AREA) computes it (module docstring;
    reference call site MyUtils1.py:202-216)."""
    L = band.shape[0]
    assert band.shape == (L, L) and band.dtype == np.uint8
    src = band.astype(np.int64)
    if cv_is_area_fast(L, t):
        k = L // t
        if k == 1:
            return band.copy()
        blocks = src.reshape(t, k, t, k).sum(axis=(1, 3))
        if k == 2:
            return ((blocks + 2) >> 2).astype(np.uint8)
        scale = np.float32(1.0) / np.float32(k * k)
        return np.clip(_cv_round(blocks.astype(np.float32) * scale), 0, 255).astype(np.uint8)
    if L > t:                                                       # true area interpolation, float
        tab = cv_area_tab(L, t)
        srcf = band.astype(np.float32)
        hbuf = np.zeros((L, t), np.float32)                         # every source row folded along x, in table order
        for dx, ent in enumerate(tab):
            acc = np.zeros(L, np.float32)
            for si, al in ent:
                acc = (acc + srcf[:, si] * al).astype(np.float32)
            hbuf[:, dx] = acc
        out = np.zeros((t, t), np.uint8)
        for dy, ent in enumerate(tab):
            acc = None
            for si, be in ent:
                term = (be * hbuf[si]).astype(np.float32)
                acc = term if acc is None else (acc + term).astype(np.float32)
            out[dy] = np.clip(_cv_round(acc), 0, 255).astype(np.uint8)
        return out
    ofs, a0, a1, dmax = cv_linear_coeffs(L, t)                      # enlarging: bilinear with area-style coordinates, 11-bit fixed point
    right = np.minimum(ofs + 1, L - 1)
    h = src[:, ofs] * a0[None, :] + src[:, right] * a1[None, :]
    h[:, dmax:] = src[:, ofs[dmax:]] * INTER_RESIZE_COEF_SCALE
    sy, b0, b1 = cv_linear_rows(L, t)                               # rows: no border special case, indices clipped instead
    r0, r1 = np.minimum(sy, L - 1), np.minimum(sy + 1, L - 1)
    v = (((b0[:, None] * (h[r0] >> 4)) >> 16) + ((b1[:, None] * (h[r1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def cv_linear_rows(ssize: int, dsize: int):
    """Row coefficients of the enlarging branch: as `cv_linear_coeffs` but WITHOUT the last-column special case (the row loop of
    cv::resize does not zero fy at the border; the row index is clipped instead)."""
    inv_scale = np.float64(dsize) / np.float64(ssize)
    scale = np.float64(1.0) / inv_scale
    b0 = np.zeros(dsize, np.int64); b1 = np.zeros(dsize, np.int64); idx = np.zeros(dsize, np.int64)
    for d in range(dsize):
        sy = int(np.floor(np.float64(d) * scale))
        fy = np.float32(np.float64(d + 1) - np.float64(sy + 1) * inv_scale)
        fy = np.float32(0.0) if fy <= 0 else np.float32(fy - np.float32(np.floor(fy)))
        b0[d] = _cv_round(np.float32(np.float32(1.0) - fy) * np.float32(INTER_RESIZE_COEF_SCALE))
        b1[d] = _cv_round(fy * np.float32(INTER_RESIZE_COEF_SCALE))
        idx[d] = sy
    return idx, b0, b1


RESIZE_RULES = {"opencv": cv_resize_area_u8, "exact_area": area_resize_u8}


def patch_pyramid(img: np.ndarray, x: int, y: int, windows: Sequence[int], targets: Sequence[int] = CONFIG_SCALES,
                  resize: str = "opencv") -> List[np.ndarray]:
    """float32 [bands, t_i, t_i] per scale: crop (zero-padded) -> resize on uint8 (rule `resize`) -> /255.0."""
    fn = RESIZE_RULES[resize]
    out = []
    for L, t in zip(windows, targets):
        x0, y0 = top_left(x, y, L)
        win = cut_image(img, x0, y0, L)
        res = np.stack([fn(win[b], t) for b in range(win.shape[0])])
        out.append(res.astype(np.float32) / 255.0)
    return out
