"""MI355X-native counterpart of the reference's `ExtractFeatures.py` inference sweep.

Reference flow (ExtractFeatures.py:45-86, :150-225): embed every sample point in batches of 2000
(eval forward), append the [P,100] float32 rows to an HDF5 dataset, then for each region-adjacency
edge gather the point rows of both polygons, mean-pool them and write the Euclidean distance as
`simi`.  GDAL rasters / shapefiles are out of scope (tensors in, tensors out); the HDF5 feature store exists as a writer /
reader of the same layout (deepmerge_amd/h5store.py, save_h5 / ReadFeatures below).  This module keeps the
features resident in HBM and runs the sweep as two kernels:
    dm_segment_mean      per-polygon mean over its sample points (CSR: ptr[S+1], idx[P])
    dm_edge_similarity   per-edge simi + merge = simi < margin
The arithmetic order of both is pinned (oracle/sweep_strict.c), so `merge` is bit-exact.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops


class FeatureIO:
    """`FeatureIO(net, checkpoint_path)` as at ExtractFeatures.py:27-43: eval mode, weights frozen."""

    def __init__(self, net: torch.nn.Module, checkpoint_path: Optional[str] = None, device: str = "cuda:0"):
        self.net = net
        if checkpoint_path is not None:
            state = torch.load(checkpoint_path, map_location="cpu")
            self.net.load_state_dict(state["net"])          # same checkpoint dict layout as Train_SMT.py:325-331
        self.net.to(device).eval()
        for p in self.net.parameters():
            p.requires_grad = False
        self.device = device
        self.features: Optional[torch.Tensor] = None

    @torch.no_grad()
    def extract_features(self, patches: Sequence[torch.Tensor], designed: torch.Tensor, batch_size: int = 2000) -> torch.Tensor:
        """Embed P sample points (per-scale patch stacks [P, C, s, s] and designed features [P,1,19]) in
        point order, `batch_size` at a time (ExtractFeatures.py:45, :58-79); returns / keeps F [P,100] fp32."""
        P = designed.shape[0]
        out = torch.empty((P, 100), dtype=torch.float32, device=self.device)
        S = len(self.net.input_image_scales)                  # the 4th 1x1 patch is dropped (:68-70)
        for s in range(0, P, batch_size):
            e = min(P, s + batch_size)
            x = [patches[i][s:e].to(self.device) for i in range(S)]
            out[s:e] = self.net(x, designed[s:e].to(self.device))
        self.features = out
        return out

    @torch.no_grad()
    def extract_features_from_tile(self, tile: torch.Tensor, points_xy: torch.Tensor, inner: torch.Tensor, obj: torch.Tensor,
                                   region_features: torch.Tensor, batch_size: int = 2000, geotransform=None,
                                   process_group=None) -> torch.Tensor:
        """`extract_features(image_path, point_path, h5_file_path, batch_size)` (ExtractFeatures.py:45-86) with the raster and the
        point table already in memory: `tile` uint8 [bands, H, W] on the GPU (what GDAL's ReadAsArray returns), one row per
        sample point in FID order -- pixel position (or geo coordinates when `geotransform` is given: the reference's own
        conversion with its +1, MyUtils1.py:67-73), the `inner` / `object` window fields and the 15 designed attributes.  The
        per-point window arithmetic, crop, resize and patch-embed operand layout run on the device (patches.point_batch_cols);
        returns / keeps F [P, 100] fp32, rows in point order, as the HDF5 `dataset` would hold them.
        The band resize follows `self.resize` ("opencv" by default: cv2.resize(..., INTER_AREA) as MyUtils1.py:202-216 calls it, restated
        branch by branch; set `FeatureIO.resize = "exact_area"` for the exact rational area average of earlier rounds)."""
        from .patches import geo_to_pixel, point_batch_cols
        import torch.distributed as dist
        if geotransform is not None:
            points_xy = geo_to_pixel(geotransform, points_xy[:, 0], points_xy[:, 1])
        world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        if world > 1:
            # points are independent -> contiguous equal shards per rank (the tile is replicated: 64 MiB), no exchange
            # during the encode, ONE all-gather of the [P, 100] rows (24 MB for a 4096^2 tile) before the sweep
            rank = dist.get_rank(process_group)
            P = points_xy.shape[0]
            per = (P + world - 1) // world
            lo, hi = min(P, rank * per), min(P, (rank + 1) * per)
            local = self._extract_local(tile, points_xy[lo:hi], inner[lo:hi], obj[lo:hi], region_features[lo:hi], batch_size) \
                if hi > lo else torch.empty((0, 100), dtype=torch.float32, device=self.device)
            padded = torch.zeros((per, 100), dtype=torch.float32, device=self.device)
            padded[:hi - lo] = local
            parts = [torch.empty_like(padded) for _ in range(world)]
            dist.all_gather(parts, padded, group=process_group)
            self.features = torch.cat(parts)[:P].contiguous()
            return self.features
        return self._extract_local(tile, points_xy, inner, obj, region_features, batch_size)

    @torch.no_grad()
    def _extract_local(self, tile, points_xy, inner, obj, region_features, batch_size):
        from .patches import point_batch_cols
        P = points_xy.shape[0]
        scales = list(self.net.input_image_scales)
        grid = self.net.cube_size[1]
        dtype = ops.act_dtype(getattr(self.net, "numerics", "bf16"))
        out = torch.empty((P, 100), dtype=torch.float32, device=self.device)
        xy = points_xy.to(self.device).to(torch.int32)
        feats = region_features.to(self.device)
        for s in range(0, P, batch_size):
            e = min(P, s + batch_size)
            patches, designed = point_batch_cols(tile, xy[s:e], inner[s:e], obj[s:e], feats[s:e], scales=scales, grid=grid, dtype=dtype,
                                                 resize=getattr(self, "resize", "opencv"))
            out[s:e] = self.net(patches, designed)
        self.features = out
        return out

    # -- the HDF5 feature store of the reference (ExtractFeatures.py:88-117), without h5py: deepmerge_amd/h5store.py ----------
    def save_h5(self, h5_file_path: str, batch_size: int = 2000) -> int:
        """Write the resident features as the reference's `dataset` ([P, 100] float32, first dimension unlimited, chunked),
        appended `batch_size` rows at a time like the upstream loop.  Returns the number of rows written."""
        from .h5store import H5FeatureWriter
        if self.features is None:
            raise RuntimeError("no features to save: run extract_features / extract_features_from_tile first")
        F = self.features
        with H5FeatureWriter(h5_file_path, width=F.shape[1]) as w:
            for s in range(0, F.shape[0], batch_size):
                w.append(F[s:s + batch_size].float().cpu().numpy())
        return int(F.shape[0])

    def ReadFeatures(self, h5_file_path: str):
        """Open a feature store for GetFeaturesByID (ExtractFeatures.py:103-107)."""
        from .h5store import H5FeatureReader
        self.Close()
        self.h5py_file = H5FeatureReader(h5_file_path)
        self.dataset = self.h5py_file

    def Close(self):
        if getattr(self, "h5py_file", None) is not None:
            self.h5py_file.close()
            self.h5py_file = None
            self.dataset = None

    def GetFeaturesByID(self, idx: int):
        """Row `idx` of the opened store (numpy float32 [100], as h5py returns it) or, without one, of the resident tensor."""
        if getattr(self, "dataset", None) is not None:
            if idx >= len(self.dataset):
                raise IndexError("index error!")
            return self.dataset[idx]
        if self.features is None or idx >= self.features.shape[0]:
            raise IndexError("index error!")
        return self.features[idx]


def rag_similarity_sweep(features: torch.Tensor, ptr: torch.Tensor, idx: torch.Tensor, edges: torch.Tensor,
                         margin: float = 1.0):
    """The per-edge loop of `test_for_shp` (ExtractFeatures.py:164-219) for ALL edges at once.

    features [P,D] fp32, ptr int32 [S+1], idx int32 [P'] (polygon -> its PointID list), edges int32 [E,2]
    (LEFT_FID, RIGHT_FID; -1 = no polygon, skipped as at MyUtils2.py:184-186 -> simi NaN, merge False).
    Returns (pooled [S,D], simi [E], merge [E] bool)."""
    pooled = ops.segment_mean(features.contiguous(), ptr.to(torch.int32).contiguous(), idx.to(torch.int32).contiguous())
    simi, merge = ops.edge_similarity(pooled, edges.to(torch.int32).contiguous(), margin)
    return pooled, simi, merge.bool()


def near_margin_count(simi: torch.Tensor, margin: float = 1.0, band: float = 1e-4) -> int:
    """Edges whose similarity lies within `band` of the margin: the only merge decisions that can differ between two encoders
    whose features agree to ~1e-4 (reported next to the bit-exact `merge` comparison).  NaN (skipped edges) never counts."""
    return int(((simi - margin).abs() < band).sum())
