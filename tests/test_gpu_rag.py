"""GPU: deepmerge_amd.rag (csrc/dm_rag.hip) against oracle/rag.py -- bit-exact integers, features exact to float32."""
import numpy as np
import pytest
import torch

from oracle import rag as OR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def superpixels(H, W, cell, seed):
    """Jittered-grid Voronoi labels (irregular, spatially coherent regions)."""
    rng = np.random.default_rng(seed)
    gy, gx = (H + cell - 1) // cell, (W + cell - 1) // cell
    cy = (np.arange(gy)[:, None] + rng.uniform(0.2, 0.8, (gy, gx))) * cell
    cx = (np.arange(gx)[None, :] + rng.uniform(0.2, 0.8, (gy, gx))) * cell
    yy, xx = np.mgrid[0:H, 0:W]
    best = np.full((H, W), np.inf); lab = np.zeros((H, W), np.int32)
    by, bx = yy // cell, xx // cell
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            ny, nx = np.clip(by + dy, 0, gy - 1), np.clip(bx + dx, 0, gx - 1)
            d = (yy - cy[ny, nx]) ** 2 + (xx - cx[ny, nx]) ** 2
            upd = d < best
            best[upd] = d[upd]; lab[upd] = (ny * gx + nx)[upd]
    return lab, gy * gx


@pytest.mark.parametrize("H,W,cell,bands", [(64, 80, 9, 3), (257, 301, 13, 4), (1024, 1024, 29, 1), (33, 17, 40, 2)])
def test_rag_and_features_match_oracle(H, W, cell, bands):
    from deepmerge_amd import rag
    lab, S = superpixels(H, W, cell, H + W)
    S += 3                                                   # a few ids that never occur
    rng = np.random.default_rng(1)
    tile = rng.integers(0, 256, (bands, H, W), dtype=np.uint8)
    want_e, want_w = OR.rag_edges(lab, S)
    want_st = OR.label_stats(lab, tile, S)
    want_f = OR.designed_features(want_st)
    tl, tt = torch.from_numpy(lab).to(DEV), torch.from_numpy(tile).to(DEV)
    edges, w = rag.rag_edges(tl, S)
    assert np.array_equal(edges.cpu().numpy(), want_e) and np.array_equal(w.cpu().numpy(), want_w)
    st = rag.label_stats(tl, tt, S)
    for k in ("count", "sum", "sumsq", "bbox", "peri"):
        assert np.array_equal(st[k].cpu().numpy(), want_st[k]), k
    f = rag.designed_features(st).cpu().numpy()
    assert np.array_equal(f, want_f)
    edges2, w2 = rag.rag_edges(tl, S)                        # insertion order varies; the result must not
    assert torch.equal(edges, edges2) and torch.equal(w, w2)


def test_rag_rejects_small_capacity_and_bad_input():
    from deepmerge_amd import rag
    lab, S = superpixels(128, 128, 5, 3)
    tl = torch.from_numpy(lab).to(DEV)
    with pytest.raises(RuntimeError):
        rag.rag_edges(tl, S, max_edges=16)
    with pytest.raises(ValueError):
        rag.rag_edges(tl.long(), S)
    with pytest.raises(ValueError):
        rag.label_stats(tl, torch.zeros((3, 64, 64), dtype=torch.uint8, device=DEV), S)


def test_points_to_csr_feeds_the_sweep():
    from deepmerge_amd import rag
    from deepmerge_amd.ExtractFeatures import rag_similarity_sweep
    lab, S = superpixels(96, 96, 12, 5)
    tl = torch.from_numpy(lab).to(DEV)
    ys, xs = np.mgrid[2:96:4, 2:96:4]
    xy = torch.from_numpy(np.stack((xs.reshape(-1), ys.reshape(-1)), 1).astype(np.int32)).to(DEV)
    ptr, idx = rag.points_to_csr(tl, xy, S)
    assert int(ptr[-1]) == xy.shape[0]
    member = lab[ys.reshape(-1), xs.reshape(-1)]
    for s in (0, S // 2, S - 1):
        got = idx[int(ptr[s]):int(ptr[s + 1])].cpu().numpy()
        assert np.array_equal(got, np.nonzero(member == s)[0])
    edges, _ = rag.rag_edges(tl, S)
    feats = torch.randn((xy.shape[0], 100), device=DEV)
    pooled, simi, merge = rag_similarity_sweep(feats, ptr, idx, edges)
    assert pooled.shape == (S, 100) and simi.shape[0] == edges.shape[0] == merge.shape[0]


@pytest.mark.parametrize("S,E,frac", [(50, 120, 0.3), (5000, 20000, 0.2), (20000, 60000, 0.05), (300, 0, 0.5)])
def test_merge_components_matches_scipy(S, E, frac):
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from deepmerge_amd import rag
    rng = np.random.default_rng(S + E)
    edges = rng.integers(0, S, size=(max(E, 1), 2)).astype(np.int32)[:E].reshape(E, 2)
    merge = (rng.random(E) < frac)
    ea, eb = edges[merge, 0], edges[merge, 1]
    _, comp = connected_components(coo_matrix((np.ones(len(ea)), (ea, eb)), shape=(S, S)), directed=False)
    want = np.zeros(S, dtype=np.int64)
    first = {}
    for s in range(S):
        first.setdefault(comp[s], s)
        want[s] = first[comp[s]]                           # smallest member id of the component
    te = torch.from_numpy(edges).to(DEV) if E else torch.zeros((0, 2), dtype=torch.int32, device=DEV)
    tm = torch.from_numpy(merge).to(DEV) if E else torch.zeros((0,), dtype=torch.bool, device=DEV)
    got = rag.merge_components(te, tm, S)
    assert np.array_equal(got.cpu().numpy().astype(np.int64), want)


def test_merge_loop_on_a_raster():
    """sweep -> merge -> re-pool -> re-score until nothing merges: every round's partition stays consistent."""
    from deepmerge_amd import rag
    from deepmerge_amd.ExtractFeatures import rag_similarity_sweep
    lab, S = superpixels(128, 128, 8, 11)
    tl = torch.from_numpy(lab).to(DEV)
    ys, xs = np.mgrid[1:128:2, 1:128:2]
    xy = torch.from_numpy(np.stack((xs.reshape(-1), ys.reshape(-1)), 1).astype(np.int32)).to(DEV)
    ptr, idx = rag.points_to_csr(tl, xy, S)
    edges, _ = rag.rag_edges(tl, S)
    g = torch.Generator(device=DEV); g.manual_seed(0)
    group = torch.randint(0, 12, (S,), device=DEV, generator=g)          # superpixels of one group embed alike
    centers = torch.randn((12, 100), device=DEV, generator=g) * 2.0
    member = tl[xy[:, 1].long(), xy[:, 0].long()].long()
    feats = centers[group[member]] + 0.01 * torch.randn((xy.shape[0], 100), device=DEV, generator=g)
    n_regions, rounds = S, 0
    while True:
        pooled, simi, merge = rag_similarity_sweep(feats, ptr, idx, edges, margin=1.0)
        if not bool(merge.any()):
            break
        root = rag.merge_components(edges, merge, n_regions)
        new_id, ptr, idx, edges = rag.merge_partition(ptr, idx, edges, root)
        assert int(ptr[-1]) == xy.shape[0] and sorted(idx.cpu().tolist()) == list(range(xy.shape[0]))
        assert int(new_id.max()) + 1 == ptr.numel() - 1 < n_regions
        n_regions = ptr.numel() - 1
        rounds += 1
        assert rounds < 20
    assert rounds >= 1 and n_regions < S
    # what is left are regions of different groups only: no remaining edge joins two regions that embed alike
    assert float(simi.min()) >= 1.0 if simi.numel() else True
