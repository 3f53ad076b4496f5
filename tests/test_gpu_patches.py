"""GPU: dm_patch_pyramid against the oracle, bit for bit under both resize rules ("opencv": cv::resize INTER_AREA restated, float32 in a
fixed order; "exact_area": integer arithmetic) + host window helpers."""
import numpy as np
import pytest
import torch

from oracle import patches as OP

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rule", ["opencv", "exact_area"])
def test_patch_pyramid_bit_exact_vs_oracle(rule):
    from deepmerge_amd import ops
    resize = OP.RESIZE_RULES[rule]
    rng = np.random.default_rng(1)
    bands, H, W = 4, 300, 340
    img = rng.integers(0, 256, size=(bands, H, W), dtype=np.uint8)
    P = 40
    xy = np.stack([rng.integers(-5, W + 5, P), rng.integers(-5, H + 5, P)], 1).astype(np.int32)
    xy[:4] = [[0, 0], [W - 1, H - 1], [1, H - 1], [W // 2, 0]]                       # corners / edges
    tile = torch.from_numpy(img).to(DEV)
    for t, lo, hi in ((32, 7, 90), (64, 20, 130), (128, 40, 200), (16, 1, 40), (4, 150, 250), (2, 150, 250)):
        wins = rng.integers(lo, hi, P).astype(np.int32)
        if t <= 4:                                # integer ratios whose DOUBLE quotient is not the integer (cv_is_area_fast): table path
            wins[8:12] = [49 * t, 93 * t if 93 * t <= 384 else 98 * t // 2, 98 * t if 98 * t <= 384 else 50 * t, 48 * t]
        wins[0], wins[1] = t, 2 * t                                                   # identity and exact 2x box
        wins[2], wins[3], wins[4], wins[5] = min(3 * t, 384), (4 * t + 2) // 3, (8 * t) // 5, max(1, (3 * t) // 4)    # 1/3, ~3/4, 5/8 shrinks; 4/3 enlarging
        wins[6], wins[7] = max(1, t // 2), min(4 * t, 384)                                      # exact 2x enlarging (replication under "opencv"), 1/4
        got = ops.patch_pyramid(tile, torch.from_numpy(xy).to(DEV), torch.from_numpy(wins).to(DEV), t, resize=rule).cpu().numpy()
        for p in range(P):
            x0, y0 = OP.top_left(int(xy[p, 0]), int(xy[p, 1]), int(wins[p]))
            win = OP.cut_image(img, x0, y0, int(wins[p]))
            want = np.stack([resize(win[b], t) for b in range(bands)]).astype(np.float32) / 255.0
            assert np.array_equal(got[p].view(np.uint32), want.view(np.uint32)), (rule, t, p, wins[p])


def test_point_batch_matches_reference_contract():
    from deepmerge_amd.patches import geo_to_pixel, get_scales, point_batch
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(3, 256, 256), dtype=np.uint8)
    P = 16
    xy = rng.integers(0, 256, (P, 2)).astype(np.int32)
    inner = rng.integers(16, 64, P).astype(np.int32)
    obj = inner + rng.integers(8, 48, P).astype(np.int32)
    feats = rng.normal(size=(P, 15)).astype(np.float32)
    patches, designed = point_batch(torch.from_numpy(img).to(DEV), torch.from_numpy(xy).to(DEV), torch.from_numpy(inner),
                                    torch.from_numpy(obj), torch.from_numpy(feats).to(DEV))
    assert [tuple(p.shape) for p in patches] == [(P, 3, 32, 32), (P, 3, 64, 64), (P, 3, 128, 128)]
    assert designed.shape == (P, 1, 19)
    for p in range(P):
        w, f = OP.get_scales(int(inner[p]), int(obj[p]))
        want = OP.patch_pyramid(img, int(xy[p, 0]), int(xy[p, 1]), w[:3], (32, 64, 128))
        for i in range(3):
            assert np.array_equal(patches[i][p].cpu().numpy(), want[i])
        np.testing.assert_allclose(designed[p, 0, 15:].cpu().numpy(), np.array(f, np.float32), rtol=1e-7)
        np.testing.assert_array_equal(designed[p, 0, :15].cpu().numpy(), feats[p])
    gt = (100.0, 0.5, 0, 900.0, 0, -0.5)
    px = geo_to_pixel(gt, torch.tensor([110.2, 100.0]), torch.tensor([880.3, 900.0]))
    assert px.tolist() == [list(OP.geo_to_pixel(gt, 110.2, 880.3)), [1, 1]]


def test_patch_pyramid_full_tile_properties():
    """BASELINE configs[3] scale: 4096x4096x4 tile, ~60k points.  Size-independent checks: a constant tile gives constant
    patches; windows fully outside give zeros; the kernel is deterministic; sampled points match the oracle."""
    from deepmerge_amd import ops
    rng = np.random.default_rng(3)
    bands, H, W = 4, 4096, 4096
    tile = torch.randint(0, 256, (bands, H, W), dtype=torch.uint8, device=DEV)
    P = 59643
    xy = torch.stack([torch.randint(0, W, (P,)), torch.randint(0, H, (P,))], 1).to(torch.int32).to(DEV)
    wins = torch.randint(24, 97, (P,), dtype=torch.int32, device=DEV)
    a = ops.patch_pyramid(tile, xy, wins, 64, max_window=96)
    b = ops.patch_pyramid(tile, xy, wins, 64, max_window=96)
    assert torch.equal(a, b)
    img = tile.cpu().numpy()
    for p in rng.integers(0, P, 12):
        x0, y0 = OP.top_left(int(xy[p, 0]), int(xy[p, 1]), int(wins[p]))
        win = OP.cut_image(img, x0, y0, int(wins[p]))
        want = np.stack([OP.cv_resize_area_u8(win[c], 64) for c in range(bands)]).astype(np.float32) / 255.0
        assert np.array_equal(a[p].cpu().numpy(), want)
    const = torch.full((bands, 512, 512), 200, dtype=torch.uint8, device=DEV)
    inside = torch.tensor([[256, 256]], dtype=torch.int32, device=DEV)
    for rule in ("opencv", "exact_area"):                # a constant tile stays constant through every branch: shrink, integer ratio, enlarge
        for L in (77, 96, 64, 32, 20, 16):
            out = ops.patch_pyramid(const, inside, torch.tensor([L], dtype=torch.int32, device=DEV), 32, resize=rule)
            assert torch.equal(out, torch.full_like(out, 200.0 / 255.0)), (rule, L)
    far = torch.tensor([[5000, -4000]], dtype=torch.int32, device=DEV)
    assert float(ops.patch_pyramid(const, far, torch.tensor([50], dtype=torch.int32, device=DEV), 32).abs().max()) == 0.0
    with pytest.raises(ValueError):
        ops.patch_pyramid(const, inside, torch.tensor([500], dtype=torch.int32, device=DEV), 32)


def test_identity_windows_match_reference_crops():
    """tests/golden/sweep.npz `ident/*`: the reference's calculate_left_top_point_and_size + cut_image for windows whose
    length equals the target (the resize is then the identity under any rule), including windows hanging over every border."""
    from deepmerge_amd import ops
    from util import load_fx
    fx = load_fx("sweep.npz")
    tile = torch.from_numpy(fx["crop/img"]).to(DEV)
    for k, (mx, my, L) in enumerate(fx["ident/mid_len"]):
        xy = torch.tensor([[int(mx), int(my)]], dtype=torch.int32, device=DEV)
        got = ops.patch_pyramid(tile, xy, torch.tensor([int(L)], dtype=torch.int32, device=DEV), int(L))[0].cpu().numpy()
        assert np.array_equal(got, fx[f"ident/out{k}"].astype(np.float32) / 255.0), k


def test_point_chain_matches_reference_fixture():
    """get_all_features through the device helpers: geo -> pixel (+1), windows, factors, designed-feature layout."""
    from deepmerge_amd.patches import geo_to_pixel, get_scales
    from util import load_fx
    fx = load_fx("sweep.npz")
    gt = fx["crop/gt"].tolist()
    for k in range(int(fx["point/n"])):
        inner, obj, gx, gy = fx[f"point/{k}/inner_object_xy"]
        windows, factors = get_scales(torch.tensor([int(inner)]), torch.tensor([int(obj)]))
        assert np.array_equal(windows.numpy().astype(np.float32), fx[f"point/{k}/scales"])
        want = fx[f"point/{k}/designed"]
        assert np.array_equal(factors.numpy(), want[:, 15:])
        px = geo_to_pixel(gt, torch.tensor([gx], dtype=torch.float64), torch.tensor([gy], dtype=torch.float64))[0].tolist()
        # the pixel the reference centred its crops on: recover it from the identity between its crop and cut_image at that pixel
        img = fx["crop/img"]
        for i, L in enumerate(windows[0].tolist()):
            x0, y0 = OP.top_left(px[0], px[1], L)
            assert np.array_equal(OP.cut_image(img, x0, y0, L), fx[f"point/{k}/crop{i}"])


@pytest.mark.parametrize("rule", ["opencv", "exact_area"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fused_gather_rows_equal_pyramid_then_patchify(dtype, rule):
    """dm_patch_pyramid_cols == dm_patch_pyramid -> dm_patchify, bit for bit (windows over every border, 4 bands, all four targets)."""
    from deepmerge_amd import ops
    rng = np.random.default_rng(9)
    bands, H, W, P = 4, 300, 340, 24
    tile = torch.from_numpy(rng.integers(0, 256, size=(bands, H, W), dtype=np.uint8)).to(DEV)
    xy = torch.from_numpy(np.stack([rng.integers(-5, W + 5, P), rng.integers(-5, H + 5, P)], 1).astype(np.int32)).to(DEV)
    for t, lo, hi in ((32, 7, 90), (64, 20, 130), (128, 40, 200), (256, 60, 300)):
        wins = torch.from_numpy(rng.integers(lo, hi, P).astype(np.int32)).to(DEV)
        wins[0] = t
        planar = ops.patch_pyramid(tile, xy, wins, t, resize=rule)
        want = ops.patchify(planar, t // 8, dtype)
        got = ops.patch_pyramid_cols(tile, xy, wins, t, 8, dtype, resize=rule)
        assert got.cols.shape == want.shape == (P * 64, bands * (t // 8) ** 2) and got.shape == (P, bands, t, t)
        assert torch.equal(got.cols.view(torch.int16 if dtype == torch.bfloat16 else torch.int32), want.view(torch.int16 if dtype == torch.bfloat16 else torch.int32)), t


def test_extract_from_tile_equals_image_path():
    """FeatureIO.extract_features_from_tile (fused gather) == gather to images then the ordinary eval forward, bit for bit, and
    feeds the sweep; geo coordinates go through the reference's own pixel conversion."""
    from deepmerge_amd.ExtractFeatures import FeatureIO, rag_similarity_sweep
    from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
    from deepmerge_amd.patches import point_batch
    torch.manual_seed(0)
    rng = np.random.default_rng(3)
    bands = 4
    tile = torch.from_numpy(rng.integers(0, 256, size=(bands, 400, 400), dtype=np.uint8)).to(DEV)
    net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=[32, 64, 128], depth=[1, 1, 1], in_c=bands, numerics="bf16")
    fio = FeatureIO(net, None, DEV)
    P = 50
    xy = torch.from_numpy(rng.integers(1, 400, (P, 2)).astype(np.int32))
    inner = torch.from_numpy(rng.integers(16, 64, P).astype(np.int32)); obj = inner + torch.from_numpy(rng.integers(8, 48, P).astype(np.int32))
    feats = torch.from_numpy(np.exp(rng.uniform(-2, 3, (P, 15))).astype(np.float32))
    F = fio.extract_features_from_tile(tile, xy, inner, obj, feats, batch_size=16)
    patches, designed = point_batch(tile, xy.to(DEV), inner, obj, feats.to(DEV))
    F_img = fio.extract_features(patches, designed, batch_size=16)
    assert F.shape == (P, 100) and torch.equal(F, F_img)
    assert torch.equal(fio.GetFeaturesByID(7), F[7])
    gt = (1000.0, 0.5, 0.0, 5000.0, 0.0, -0.5)
    geo = torch.stack([gt[0] + (xy[:, 0].double() - 0.75) * gt[1], gt[3] + (xy[:, 1].double() - 0.75) * gt[5]], 1)   # int(|d|/res + 1) == xy
    assert torch.equal(fio.extract_features_from_tile(tile, geo, inner, obj, feats, batch_size=16, geotransform=gt), F)   # (same batching: the GEMM tile choice depends on M)
    ptr = torch.arange(0, P + 1, 5, dtype=torch.int32, device=DEV); idx = torch.arange(P, dtype=torch.int32, device=DEV)
    edges = torch.tensor([[0, 1], [1, 2], [-1, 3], [8, 9]], dtype=torch.int32, device=DEV)
    pooled, simi, merge = rag_similarity_sweep(F, ptr, idx, edges, 1.0)
    assert pooled.shape == (10, 100) and bool(torch.isnan(simi[2])) and not bool(merge[2])


def _feed_table(rng, T, size, B):
    from deepmerge_amd.feed import PairTable
    P = 2 * B
    inner = rng.integers(16, 65, P).astype(np.int32)
    obj = (inner + rng.integers(8, 49, P)).astype(np.int32)
    tid = rng.integers(0, T, P).astype(np.int32)
    xy = rng.integers(-4, size + 4, (P, 2)).astype(np.int32)            # some windows hang over the raster edge (zero padding)
    feats = rng.normal(size=(P, 15)).astype(np.float32)
    flag = (np.arange(B) % 2).astype(np.int64)
    mv = lambda a: torch.from_numpy(a).to(DEV)
    return PairTable(mv(tid), mv(xy), mv(inner), mv(obj), mv(feats), mv(flag)), (tid, xy, inner, obj, feats, flag)


@pytest.mark.parametrize("rows,numerics", [(True, "bf16"), (True, "fp32"), (False, "bf16")])
def test_pair_feed_equals_point_batch_per_tile(rows, numerics):
    """The sync-free training feed (deepmerge_amd/feed.py, dm_pair_batch_gather: tile id as a table column, window arithmetic and the
    designed rows on the device) produces, bit for bit, what `point_batch` / `point_batch_cols` produce tile by tile from host-side
    windows (the feed of rounds 1-4)."""
    from deepmerge_amd import ops
    from deepmerge_amd.feed import PairFeed
    from deepmerge_amd.patches import point_batch, point_batch_cols
    rng = np.random.default_rng(11)
    T, bands, size, B = 3, 4, 320, 9
    scales = [32, 64, 128, 256]
    tiles = torch.from_numpy(rng.integers(0, 256, size=(T, bands, size, size), dtype=np.uint8)).to(DEV)
    table, (tid, xy, inner, obj, feats, flag) = _feed_table(rng, T, size, B)
    feed = PairFeed(tiles, scales, B, [64, 112, 160, 208], rows=rows, numerics=numerics)
    left, ld, right, rd, fl = feed.fill(table)
    feed.check()
    assert torch.equal(fl.cpu(), torch.from_numpy(flag).float())
    dtype = ops.act_dtype(numerics)
    for t in range(T):
        sel = np.nonzero(tid == t)[0]
        if sel.size == 0:
            continue
        args = (tiles[t], torch.from_numpy(xy[sel]).to(DEV), torch.from_numpy(inner[sel]), torch.from_numpy(obj[sel]), torch.from_numpy(feats[sel]).to(DEV))
        want, wd = (point_batch_cols(*args, scales=scales, dtype=dtype) if rows else point_batch(*args, scales=scales))
        for i, s in enumerate(scales):
            for k, p in enumerate(sel):
                got = (left[i][p:p + 1] if p < B else right[i][p - B:p - B + 1])
                ref = want[i][k:k + 1]
                if rows:
                    got, ref = got.cols, ref.cols
                assert got.dtype == ref.dtype and torch.equal(got.view(torch.uint8), ref.reshape(got.shape).view(torch.uint8)), (t, s, p)
        for k, p in enumerate(sel):
            got = ld[p] if p < B else rd[p - B]
            assert torch.equal(got.view(torch.int32), wd[k].view(torch.int32))


def test_pair_feed_flags_out_of_range_samples_without_a_sync():
    from deepmerge_amd.feed import PairFeed
    rng = np.random.default_rng(12)
    T, bands, size, B = 2, 3, 200, 4
    tiles = torch.from_numpy(rng.integers(0, 256, size=(T, bands, size, size), dtype=np.uint8)).to(DEV)
    table, _ = _feed_table(rng, T, size, B)
    feed = PairFeed(tiles, [32, 64, 128], B, 160, rows=False)
    feed.fill(table); feed.check()
    table.tile_id[3] = T                      # no such tile
    left, *_ = feed.fill(table)
    assert float(left[0][3].abs().max()) == 0.0 and float(left[0][2].abs().max()) > 0.0
    with pytest.raises(ValueError):
        feed.check()
    feed.fill(_feed_table(rng, T, size, B)[0]); feed.check()          # the flag was cleared
    small = PairFeed(tiles, [32, 64, 128], B, 40, rows=False)         # bound below the table's windows
    small.fill(table)
    with pytest.raises(ValueError):
        small.check()


def test_trainer_steps_on_feed_rows_equal_steps_on_patch_tensors():
    """PairTrainer fed by PairFeed rows (graph replay, the feed writing straight into the captured step's inputs) == the same steps on
    fp32 patch tensors from the per-tile feed: same loss every step, same weights after three steps, bit for bit."""
    from deepmerge_amd.feed import PairFeed
    from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
    from deepmerge_amd.patches import point_batch
    from deepmerge_amd.trainer import PairTrainer
    rng = np.random.default_rng(13)
    T, bands, size, B = 3, 3, 256, 4
    scales = [32, 64, 128]
    tiles = torch.from_numpy(rng.integers(0, 256, size=(T, bands, size, size), dtype=np.uint8)).to(DEV)
    tables = [_feed_table(rng, T, size, B) for _ in range(4)]
    losses, weights = {}, {}
    for kind in ("feed", "tensors"):
        torch.manual_seed(3)
        net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=list(scales), depth=[1, 1, 1], in_c=bands, numerics="bf16").to(DEV)
        tr = PairTrainer(net, margin=1.0, lr=1e-3)
        tr.enable_graph(warmup=1)
        feed = PairFeed(tiles, scales, B, [64, 112, 160], numerics="bf16", trainer=tr) if kind == "feed" else None
        out = []
        for table, (tid, xy, inner, obj, feats, flag) in tables:
            if feed is not None:
                batch = feed.fill(table)
            else:
                P = 2 * B
                patches = [torch.empty((P, bands, s, s), device=DEV) for s in scales]
                designed = torch.empty((P, 1, 19), device=DEV)
                for t in range(T):
                    sel = np.nonzero(tid == t)[0]
                    if sel.size == 0:
                        continue
                    p, d = point_batch(tiles[t], torch.from_numpy(xy[sel]).to(DEV), torch.from_numpy(inner[sel]), torch.from_numpy(obj[sel]),
                                       torch.from_numpy(feats[sel]).to(DEV), scales=scales)
                    idx = torch.from_numpy(sel).to(DEV)
                    for i in range(len(scales)):
                        patches[i][idx] = p[i]
                    designed[idx] = d
                batch = ([t[:B] for t in patches], designed[:B], [t[B:] for t in patches], designed[B:], torch.from_numpy(flag).to(DEV))
            out.append(float(tr.step(*batch)))
        if feed is not None:
            feed.check()
            assert feed._bound and tr.graph_error is None
        losses[kind], weights[kind] = out, tr.fp.flat.clone()
    assert losses["feed"] == losses["tensors"]
    assert torch.equal(weights["feed"], weights["tensors"])


def test_pair_feed_fuzz_against_the_single_tile_gathers():
    """tools/fuzz_feed.py, a short run: random tile stacks, window sides 1 .. bound, points outside the raster, power-of-two and other
    targets, both resize rules, patches and rows -- the banded multi-tile gather equals the single-tile entry points bit for bit."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_feed
    assert fuzz_feed.run(rounds=25, seed=11, verbose=False) == 0
