"""MI355X-native counterpart of the reference's `ExtractFeatures.py` inference sweep.

Reference flow (ExtractFeatures.py:45-86, :150-225): embed every sample point in batches of 2000
(eval forward), append the [P,100] float32 rows to an HDF5 dataset, then for each region-adjacency
edge gather the point rows of both polygons, mean-pool them and write the Euclidean distance as
`simi`.  Storage (HDF5 / shapefile fields, GDAL rasters) is out of scope; this module keeps the
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

    def GetFeaturesByID(self, idx: int) -> torch.Tensor:
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
