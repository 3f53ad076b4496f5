"""Fuzz of the training feed's gather (dm_pair_batch_gather: several output-row bands per (sample, band), window rows staged per band) against the
single-tile entry points (dm_patch_pyramid / dm_patch_pyramid_cols, one block per (sample, band), whole window staged): random tile stacks, window
sides from 1 to the bound (integer ratios, shrinks, enlargements), points far outside the raster, power-of-two and other targets, both resize rules.
Bit for bit.      python tools/fuzz_feed.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from deepmerge_amd import ops
DEV = "cuda:0"


def run(rounds=60, seed=7, verbose=True):
  rng = np.random.default_rng(seed)
  bad = 0
  for it in range(rounds):
      T_, bands = int(rng.integers(1, 4)), int(rng.integers(1, 5))
      H, W = int(rng.integers(40, 400)), int(rng.integers(40, 400))
      tiles = torch.from_numpy(rng.integers(0, 256, size=(T_, bands, H, W), dtype=np.uint8)).to(DEV)
      P = int(rng.integers(1, 24))
      target = int(rng.choice([8, 16, 24, 32, 40, 64, 96, 128, 200, 256]))
      grid = int(rng.choice([g for g in (1, 2, 4, 8) if target % g == 0]))
      mw = int(rng.integers(max(2, target // 4), 385))
      inner = rng.integers(1, mw + 1, P).astype(np.int32)
      obj = inner.copy()                                              # scale index 0 uses `inner` as the window side
      inner[:3] = [1, mw, min(mw, target)][:min(3, P)][:len(inner[:3])] if P >= 3 else inner[:3]
      if P >= 6:
          inner[3] = min(mw, 2 * target); inner[4] = min(mw, max(1, target // 2)); inner[5] = min(mw, 3 * target)
      obj = inner.copy()
      xy = np.stack([rng.integers(-30, W + 30, P), rng.integers(-30, H + 30, P)], 1).astype(np.int32)
      tid = rng.integers(0, T_, P).astype(np.int32)
      rule = str(rng.choice(["opencv", "exact_area"]))
      mv = lambda a: torch.from_numpy(a).to(DEV)
      for rows in (False, True):
          dt = torch.bfloat16 if (rows and rng.integers(0, 2)) else torch.float32
          ps = target // grid
          out = torch.empty((P * grid * grid, bands * ps * ps) if rows else (P, bands, target, target), dtype=dt if rows else torch.float32, device=DEV)
          ops.pair_batch_gather(tiles, mv(tid), mv(xy), mv(inner), mv(obj), 0, target, mw, out, grid=grid if rows else 0, resize=rule)
          for t in range(T_):
              sel = np.nonzero(tid == t)[0]
              if sel.size == 0:
                  continue
              if rows:
                  ref = ops.patch_pyramid_cols(tiles[t], mv(xy[sel]), mv(inner[sel]), target, grid, dt, max_window=mw, resize=rule).cols.view(len(sel), grid * grid, -1)
                  got = out.view(P, grid * grid, -1)[mv(sel.astype(np.int64))]
              else:
                  ref = ops.patch_pyramid(tiles[t], mv(xy[sel]), mv(inner[sel]), target, max_window=mw, resize=rule)
                  got = out[mv(sel.astype(np.int64))]
              if not torch.equal(got.view(torch.uint8), ref.view(torch.uint8)):
                  bad += 1
                  d = (got.float() - ref.float()).abs()
                  print(f"MISMATCH round {it}: target {target} grid {grid} rows {rows} rule {rule} mw {mw} tile {t}: {int((d > 0).sum())} elements, max {float(d.max())}; windows {inner[sel][:8]}")
  if verbose:
    print(f"{rounds} rounds, {bad} mismatching (tile, form) comparisons")
  return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(os.environ.get("SEED", 7))) else 0)
