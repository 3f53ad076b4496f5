"""GPU patch pyramid for the pair encoder: the per-point window arithmetic of the reference's loaders
(MyUtils1.py:60-77, :116-223; MyUtils2.py:286-437) with the crop + resize on the device (dm_patch_pyramid).

The reference does this per item on one host thread (GDAL ReadAsArray + per-band cv2.resize); here one launch
per scale serves a whole batch of sample points straight from a tile resident in HBM.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from . import ops

CONFIG_SCALES = (32, 64, 128, 1)      # config.py:32; the 4th (1x1) patch is produced upstream but dropped by v3 (ExtractFeatures.py:68-70)


def get_scales(inner: torch.Tensor, obj: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """MyUtils1.py:130-156, vectorised: int tensors [P] -> windows int32 [P,4], factors float32 [P,4]."""
    inner, obj = inner.to(torch.int32), obj.to(torch.int32)
    interval = obj - inner
    windows = torch.stack((inner, obj, obj + interval, obj + 2 * interval), dim=1)
    factors = windows.to(torch.float32) / torch.tensor(CONFIG_SCALES, dtype=torch.float32, device=windows.device)
    return windows, factors


def geo_to_pixel(gt: Sequence[float], x_geo: torch.Tensor, y_geo: torch.Tensor) -> torch.Tensor:
    """MyUtils1.py:67-73: int(abs((gt[0]-X)/gt[1]) + 1), int(abs((gt[3]-Y)/gt[5]) + 1) -> int32 [P,2]."""
    px = (torch.abs((gt[0] - x_geo.double()) / gt[1]) + 1).to(torch.int32)
    py = (torch.abs((gt[3] - y_geo.double()) / gt[5]) + 1).to(torch.int32)
    return torch.stack((px, py), dim=1)


def point_batch(tile: torch.Tensor, xy: torch.Tensor, inner: torch.Tensor, obj: torch.Tensor, region_features: torch.Tensor,
                scales: Sequence[int] = (32, 64, 128), resize: str = "opencv") -> Tuple[List[torch.Tensor], torch.Tensor]:
    """The model inputs for P sample points: per-scale patches float32 [P, bands, s, s] (device) and designed
    features [P, 1, 19] = 15 region features || 4 scale factors (MyUtils1.py:60-77).
    tile: uint8 [bands,H,W] on the GPU; xy int [P,2] pixel coordinates; inner/obj int [P]; region_features [P,15]."""
    windows, factors = get_scales(inner, obj)
    windows = windows.to(tile.device)
    patches = [ops.patch_pyramid(tile, xy, windows[:, i].contiguous(), int(t), resize=resize) for i, t in enumerate(scales)]
    designed = torch.cat((region_features.to(torch.float32), factors.to(region_features.device)), dim=1).unsqueeze(1)
    return patches, designed


def point_batch_cols(tile: torch.Tensor, xy: torch.Tensor, inner: torch.Tensor, obj: torch.Tensor, region_features: torch.Tensor,
                     scales: Sequence[int] = (32, 64, 128), grid: int = 8, dtype: torch.dtype = torch.bfloat16, resize: str = "opencv"):
    """point_batch with the gather FUSED into the patch-embed operand load: per scale an ops.PatchCols
    (bf16 im2col rows written by the gather kernel itself) instead of the fp32 [P, bands, s, s] tensor + dm_patchify.
    Bytes per point and scale: window in (L^2 * bands, uint8) + s^2 * bands * 2 out -- against + s^2 * bands * (4 + 4 + 2) for the
    unfused chain.  The model consumes the list exactly like image tensors; results are bit-identical."""
    windows, factors = get_scales(inner, obj)
    windows = windows.to(tile.device)
    patches = [ops.patch_pyramid_cols(tile, xy, windows[:, i].contiguous(), int(t), grid, dtype, resize=resize) for i, t in enumerate(scales)]
    designed = torch.cat((region_features.to(torch.float32), factors.to(region_features.device)), dim=1).unsqueeze(1)
    return patches, designed
