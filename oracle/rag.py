"""Oracle (test infrastructure): region-adjacency graph and per-superpixel statistics from a label raster, numpy.

NOT reference behaviour: the reference reads the edge list (lines.shp `LEFT_FID`/`RIGHT_FID`, MyUtils2.py:155-193) and the
15 designed attributes (MyUtils1.py:79-114: area, peri, len, width, smooth, std0-2, mean0-2, shapeness, compact, bright,
border) from shapefiles produced by external GIS software, whose formulas it does not contain.  SURVEY 8f rank 2 asks for
an on-device replacement with a spec of the build's own; this file IS that spec:

  4-neighbourhood on the pixel grid; ids outside [0, S) are ignored.
  edge (a < b)      : some pixel of a is a 4-neighbour of a pixel of b; weight = number of such pixel edges
  area              : pixel count
  peri              : pixel edges whose other side is a different id (inner) or outside the raster (border)
  len / width       : larger / smaller side of the bounding box, in pixels
  smooth            : peri / (2 (bw + bh))                       (perimeter over bounding-box perimeter)
  mean_b / std_b    : per band b < 3, population statistics from exact integer sums: var = E[x^2] - E[x]^2, clamped at 0
  shapeness         : peri / (4 sqrt(area))                      (shape index)
  compact           : area / (bw * bh)
  bright            : mean of the band means
  border            : inner perimeter (pixel edges shared with other superpixels)
All floating point is IEEE double (+, -, *, /, sqrt) rounded once to float32 at the end.
"""
from __future__ import annotations

import numpy as np


def rag_edges(labels: np.ndarray, S: int):
    """(edges int32 [E,2] sorted by (a,b), weights int32 [E])."""
    L = labels.astype(np.int64)
    pairs = []
    for a, b in ((L[:, :-1], L[:, 1:]), (L[:-1, :], L[1:, :])):
        a, b = a.reshape(-1), b.reshape(-1)
        ok = (a != b) & (a >= 0) & (b >= 0) & (a < S) & (b < S)
        lo, hi = np.minimum(a[ok], b[ok]), np.maximum(a[ok], b[ok])
        pairs.append(lo * S + hi)
    keys, counts = np.unique(np.concatenate(pairs), return_counts=True)
    return np.stack((keys // S, keys % S), 1).astype(np.int32), counts.astype(np.int32)


def label_stats(labels: np.ndarray, tile: np.ndarray, S: int):
    L = labels.astype(np.int64)
    H, W = L.shape
    ok = (L >= 0) & (L < S)
    flat = L[ok]
    nb = min(tile.shape[0], 3)
    count = np.bincount(flat, minlength=S).astype(np.int64)
    sums = np.zeros((S, nb), dtype=np.int64)
    sumsq = np.zeros((S, nb), dtype=np.int64)
    for b in range(nb):
        v = tile[b].astype(np.int64)[ok]
        sums[:, b] = np.bincount(flat, weights=v.astype(np.float64), minlength=S).astype(np.int64)      # exact below 2^53
        sumsq[:, b] = np.bincount(flat, weights=(v * v).astype(np.float64), minlength=S).astype(np.int64)
    ys, xs = np.nonzero(ok)
    bbox = np.empty((S, 4), dtype=np.int32)
    bbox[:, 0] = bbox[:, 1] = np.iinfo(np.int32).max
    bbox[:, 2] = bbox[:, 3] = -1
    np.minimum.at(bbox[:, 0], flat, xs.astype(np.int32)); np.minimum.at(bbox[:, 1], flat, ys.astype(np.int32))
    np.maximum.at(bbox[:, 2], flat, xs.astype(np.int32)); np.maximum.at(bbox[:, 3], flat, ys.astype(np.int32))
    pad = np.full((H + 2, W + 2), -2, dtype=np.int64)            # -2 = outside the raster
    pad[1:-1, 1:-1] = L
    peri = np.zeros((S, 2), dtype=np.int64)
    for dy, dx in ((0, -1), (0, 1), (-1, 0), (1, 0)):
        nb_ = pad[1 + dy:H + 1 + dy, 1 + dx:W + 1 + dx]
        peri[:, 0] += np.bincount(L[ok & (nb_ != L) & (nb_ != -2)], minlength=S)
        peri[:, 1] += np.bincount(L[ok & (nb_ == -2)], minlength=S)
    return {"count": count, "sum": sums, "sumsq": sumsq, "bbox": bbox, "peri": peri}


def designed_features(st) -> np.ndarray:
    S = st["count"].shape[0]
    nb = st["sum"].shape[1]
    f = np.zeros((S, 15), dtype=np.float32)
    for s in range(S):
        c = int(st["count"][s])
        if c == 0:
            continue
        area = np.float64(c)
        pin, pbd = np.float64(st["peri"][s, 0]), np.float64(st["peri"][s, 1])
        per = pin + pbd
        bw = np.float64(int(st["bbox"][s, 2]) - int(st["bbox"][s, 0]) + 1)
        bh = np.float64(int(st["bbox"][s, 3]) - int(st["bbox"][s, 1]) + 1)
        mean, sd = [np.float64(0)] * 3, [np.float64(0)] * 3
        for b in range(nb):
            m = np.float64(st["sum"][s, b]) / area
            var = np.float64(st["sumsq"][s, b]) / area - m * m
            mean[b], sd[b] = m, np.sqrt(var if var > 0 else np.float64(0))
        f[s] = [area, per, max(bw, bh), min(bw, bh), per / (np.float64(2) * (bw + bh)), sd[0], sd[1], sd[2], mean[0], mean[1], mean[2],
                per / (np.float64(4) * np.sqrt(area)), area / (bw * bh), (mean[0] + mean[1] + mean[2]) / np.float64(max(nb, 1)), pin]
    return f
