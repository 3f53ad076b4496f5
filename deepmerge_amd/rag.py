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


def merge_components(edges: torch.Tensor, merge: torch.Tensor, n_labels: int, max_rounds: int = 64) -> torch.Tensor:
    """Connected components of the superpixel graph restricted to the edges flagged `merge` (the step after the sweep; the
    reference hands this to external GIS tooling).  Returns int32 [S]: the smallest superpixel id of each one's component."""
    _need_cuda(edges, merge)
    edges = edges.to(torch.int32).contiguous()
    m8 = merge.to(torch.uint8).contiguous()
    if edges.numel() and int(edges.max()) >= n_labels:
        raise ValueError(f"merge_components: edge endpoint {int(edges.max())} >= n_labels {n_labels}")
    if edges.numel() and bool(((edges < 0).any(1) & (m8 != 0)).any()):
        raise ValueError("merge_components: an edge with a -1 ('no polygon') endpoint is flagged for merging")
    dev = edges.device
    parent = torch.empty(n_labels, dtype=torch.int32, device=dev)
    changed = torch.zeros(1, dtype=torch.int32, device=dev)
    for r in range(max_rounds):
        check(_lib.lib().dm_merge_round(edges.data_ptr(), m8.data_ptr(), edges.shape[0], n_labels, parent.data_ptr(), changed.data_ptr(),
                                        int(r == 0), _stream()), "dm_merge_round")
        if int(changed.item()) == 0:
            return parent
    raise RuntimeError(f"merge_components did not converge in {max_rounds} rounds")


def merge_partition(ptr: torch.Tensor, idx: torch.Tensor, edges: torch.Tensor, root: torch.Tensor):
    """Apply a component labelling: merged CSR point lists (members in ascending old id, their points in the old order) and
    the edge list between the merged regions (self edges dropped, duplicates folded, sorted).
    Returns (new_id int32 [S] dense 0..C-1 in order of the component's smallest member, ptr', idx', edges')."""
    S = root.numel()
    root = root.long()
    is_root = root == torch.arange(S, device=root.device)
    dense = torch.cumsum(is_root.to(torch.int64), 0) - 1               # id of a root among the roots
    new_id = dense[root]
    C = int(is_root.sum())
    counts = (ptr[1:] - ptr[:-1]).long()
    owner = torch.repeat_interleave(new_id, counts)                    # new region of every entry of idx (old CSR order)
    order = torch.argsort(owner, stable=True)
    new_idx = idx.long()[order].to(torch.int32)
    new_ptr = torch.zeros(C + 1, dtype=torch.int64, device=root.device)
    new_ptr[1:] = torch.cumsum(torch.bincount(owner, minlength=C), 0)
    live = (edges >= 0).all(1)                                         # -1 = "no polygon" (MyUtils2.py:184-186): never relabelled
    edges = edges[live]
    if edges.numel() and int(edges.max()) >= S:
        raise ValueError(f"edge endpoint {int(edges.max())} is not a superpixel id (S = {S})")
    a, b = new_id[edges[:, 0].long()], new_id[edges[:, 1].long()]
    keep = a != b
    lo, hi = torch.minimum(a[keep], b[keep]), torch.maximum(a[keep], b[keep])
    keys = torch.unique(lo * C + hi)
    new_edges = torch.stack((keys // C, keys % C), 1).to(torch.int32)
    return new_id.to(torch.int32), new_ptr.to(torch.int32), new_idx, new_edges
