"""Region-adjacency graph and designed features from a label raster, on the device.

The reference consumes a RAG edge list and 15 per-superpixel attributes that external GIS software wrote into
shapefiles (`lines.shp` LEFT_FID / RIGHT_FID, MyUtils2.py:155-193; attribute order MyUtils1.py:79-114).  This module
derives both from the segmentation's label raster and the image tile with three HIP kernels (csrc/dm_rag.hip), which
makes the ExtractFeatures pipeline self-contained on the GPU.  Definitions: oracle/rag.py (the build's own spec).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import _lib
from .ops import _need_cuda, _stream, check

FEATURE_NAMES = ("area", "peri", "len", "width", "smooth", "std0", "std1", "std2", "mean0", "mean1", "mean2",
                 "shapeness", "compact", "bright", "border")          # MyUtils1.py:79-114


def label_stats(labels: torch.Tensor, tile: torch.Tensor, n_labels: int) -> Dict[str, torch.Tensor]:
    """Exact integer statistics per superpixel: count, per-band sum / sum of squares (first three bands), bounding box,
    perimeter (shared with other labels / on the raster border)."""
    _need_cuda(labels, tile)
    if labels.dtype != torch.int32 or tile.dtype != torch.uint8 or tile.dim() != 3 or labels.shape != tile.shape[1:]:
        raise ValueError("labels must be int32 [H,W] and tile uint8 [bands,H,W] over the same raster")
    labels, tile = labels.contiguous(), tile.contiguous()
    bands, H, W = tile.shape
    nb, dev = min(bands, 3), labels.device
    out = {"count": torch.empty(n_labels, dtype=torch.int64, device=dev),
           "sum": torch.empty((n_labels, nb), dtype=torch.int64, device=dev),
           "sumsq": torch.empty((n_labels, nb), dtype=torch.int64, device=dev),
           "bbox": torch.empty((n_labels, 4), dtype=torch.int32, device=dev),
           "peri": torch.empty((n_labels, 2), dtype=torch.int64, device=dev)}
    check(_lib.lib().dm_label_stats(labels.data_ptr(), tile.data_ptr(), bands, H, W, n_labels, out["count"].data_ptr(),
                                    out["sum"].data_ptr(), out["sumsq"].data_ptr(), out["bbox"].data_ptr(), out["peri"].data_ptr(),
                                    _stream()), "dm_label_stats")
    out["bands"] = nb
    return out


def designed_features(stats: Dict[str, torch.Tensor]) -> torch.Tensor:
    """float32 [S,15] in the reference's attribute order (FEATURE_NAMES)."""
    S = stats["count"].numel()
    feat = torch.empty((S, 15), dtype=torch.float32, device=stats["count"].device)
    check(_lib.lib().dm_label_features(stats["count"].data_ptr(), stats["sum"].data_ptr(), stats["sumsq"].data_ptr(),
                                       stats["bbox"].data_ptr(), stats["peri"].data_ptr(), S, stats["bands"], feat.data_ptr(), _stream()),
          "dm_label_features")
    return feat


def rag_edges(labels: torch.Tensor, n_labels: int, max_edges: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """(edges int32 [E,2] with a < b, sorted by (a, b); shared boundary length int32 [E] in pixel edges)."""
    _need_cuda(labels)
    if labels.dtype != torch.int32 or labels.dim() != 2:
        raise ValueError("labels must be int32 [H,W]")
    labels = labels.contiguous()
    H, W = labels.shape
    dev = labels.device
    max_edges = max_edges or max(1024, 8 * n_labels)          # planar graph: E <= 3 S - 6; 8 S leaves room for raster artefacts
    log2 = max(10, (4 * max_edges - 1).bit_length())          # load factor <= 1/4
    keys = torch.empty(1 << log2, dtype=torch.int64, device=dev)
    cnts = torch.empty(1 << log2, dtype=torch.int32, device=dev)
    ek = torch.empty(max_edges, dtype=torch.int64, device=dev)
    ec = torch.empty(max_edges, dtype=torch.int32, device=dev)
    meta = torch.empty(2, dtype=torch.int32, device=dev)
    check(_lib.lib().dm_rag_edges(labels.data_ptr(), H, W, n_labels, keys.data_ptr(), cnts.data_ptr(), log2, ek.data_ptr(), ec.data_ptr(),
                                  max_edges, meta.data_ptr(), meta[1:].data_ptr(), _stream()), "dm_rag_edges")
    n, overflow = (int(v) for v in meta.tolist())
    if overflow or n > max_edges:
        raise RuntimeError(f"RAG has more than max_edges={max_edges} edges (found {n}, table overflow={bool(overflow)}); pass a larger max_edges")
    order = torch.argsort(ek[:n])                             # canonical order; keys are unique
    k = ek[:n][order]
    edges = torch.stack((k // n_labels, k % n_labels), 1).to(torch.int32)
    return edges, ec[:n][order]


def points_to_csr(labels: torch.Tensor, xy: torch.Tensor, n_labels: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """CSR membership (ptr int32 [S+1], idx int32 [P]) of sample points (x, y) in superpixels, the form
    rag_similarity_sweep consumes (the reference's space-separated `PointID` strings, ExtractFeatures.py:175-179).
    Points keep their order inside a superpixel."""
    lab = labels[xy[:, 1].long(), xy[:, 0].long()].long()
    order = torch.argsort(lab, stable=True)
    counts = torch.bincount(lab, minlength=n_labels)
    ptr = torch.zeros(n_labels + 1, dtype=torch.int64, device=labels.device)
    ptr[1:] = torch.cumsum(counts, 0)
    return ptr.to(torch.int32), order.to(torch.int32)
