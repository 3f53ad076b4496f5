"""CPU: the patch-pyramid oracle (oracle/patches.py) on known answers.  The window arithmetic is restated from the
reference; the resize is this build's exact-area spec (OpenCV parity unpinned) and is pinned here by hand-computed cases."""
import numpy as np

from oracle import patches as OP


def test_get_scales_and_geo_to_pixel_known_answers():
    w, f = OP.get_scales(24, 40)
    assert w == [24, 40, 56, 72]
    assert f == [24 / 32, 40 / 64, 56 / 128, 72 / 1]
    assert OP.geo_to_pixel((100.0, 0.5, 0, 900.0, 0, -0.5), 110.2, 880.3) == (21, 40)     # int(20.4+1), int(39.4+1)


def test_top_left_truncates_toward_zero():
    assert OP.top_left(10, 10, 8) == (6, 6)
    assert OP.top_left(10, 10, 7) == (6, 6)          # 10 - 3.5 = 6.5 -> 6
    assert OP.top_left(2, 1, 7) == (-1, -2)          # -1.5 -> -1 (toward zero), -2.5 -> -2
    assert OP.top_left(0, 0, 1) == (0, 0)            # -0.5 -> 0


def test_cut_image_zero_pads_outside_raster():
    img = np.arange(2 * 4 * 5, dtype=np.uint8).reshape(2, 4, 5) + 1
    w = OP.cut_image(img, -1, 2, 4)
    assert w.shape == (2, 4, 4)
    assert (w[:, :, 0] == 0).all() and (w[:, 2:, :] == 0).all()
    assert np.array_equal(w[:, :2, 1:], img[:, 2:4, 0:3])
    assert (OP.cut_image(img, 10, 10, 3) == 0).all()


def test_area_resize_known_answers():
    a = np.arange(16, dtype=np.uint8).reshape(4, 4) * 10
    assert np.array_equal(OP.area_resize_u8(a, 4), a)                                  # identity
    box = OP.area_resize_u8(a, 2)                                                      # exact 2x2 box means
    assert np.array_equal(box, np.array([[25, 45], [105, 125]], np.uint8))
    # round half to even: mean of (0,1,0,0)=0.25->0 ; (1,1,1,0)=0.75->1 ; (1,1,0,0)=0.5->0 ; (3,3,2,2)=2.5->2 ; (3,4,3,4)=3.5->4
    t = np.array([[0, 1, 1, 1, 1, 1], [0, 0, 1, 0, 0, 0], [3, 3, 3, 4, 0, 0], [2, 2, 3, 4, 0, 0]], np.uint8)
    t = np.vstack([t, np.zeros((2, 6), np.uint8)])
    r = OP.area_resize_u8(t, 3)
    assert r[0, 0] == 0 and r[0, 1] == 1 and r[0, 2] == 0 and r[1, 0] == 2 and r[1, 1] == 4
    # non-integer factor 3 -> 2: output pixel 0 covers input [0, 1.5): weights (1, .5)/1.5 per axis
    b = np.array([[90, 0, 0], [0, 0, 0], [0, 0, 0]], np.uint8)
    assert OP.area_resize_u8(b, 2)[0, 0] == 40                                         # 90 * (1/1.5)^2 = 40
    # up-scaling 2 -> 4: footprints of half a pixel -> pixel replication
    c = np.array([[10, 20], [30, 40]], np.uint8)
    assert np.array_equal(OP.area_resize_u8(c, 4), np.kron(c, np.ones((2, 2), np.uint8)))
    # up-scaling 2 -> 3: middle output pixel straddles both inputs equally
    assert OP.area_resize_u8(c, 3)[0].tolist() == [10, 15, 20]
    ov = OP.overlap_matrix(7, 3)
    assert (ov.sum(1) == 7).all() and (ov.sum(0) == 3).all()


def test_patch_pyramid_shapes_and_range():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(4, 50, 60), dtype=np.uint8)
    out = OP.patch_pyramid(img, 3, 48, [24, 40, 56], (32, 64, 128))
    assert [o.shape for o in out] == [(4, 32, 32), (4, 64, 64), (4, 128, 128)]
    assert all(o.dtype == np.float32 and 0 <= o.min() and o.max() <= 1 for o in out)
    assert all(np.array_equal(np.round(o * 255), o * 255) for o in out)                # values on the uint8/255 grid


# ---- pinned against the reference's own MyUtils1 functions (tests/golden/sweep.npz, make_golden.py gen_sweep) --------------
def _fx():
    from util import load_fx
    return load_fx("sweep.npz")


def test_window_arithmetic_matches_reference():
    fx = _fx()
    for (inner, obj), sc, fc in zip(fx["win/inner_object"], fx["win/scales"], fx["win/factors"]):
        w, f = OP.get_scales(int(inner), int(obj))
        assert w == sc.tolist() and f == fc.tolist()
    for (mx, my, L), lt in zip(fx["win/mid_len"], fx["win/left_top"]):
        assert OP.top_left(int(mx), int(my), int(L)) == (int(lt[0]), int(lt[1])) and int(lt[2]) == int(lt[3]) == int(L)


def test_cut_image_matches_reference():
    fx = _fx()
    img = fx["crop/img"]
    for k, (x0, y0, L) in enumerate(fx["crop/boxes"]):
        assert np.array_equal(OP.cut_image(img, int(x0), int(y0), int(L)), fx[f"crop/out{k}"]), k
    for k, (mx, my, L) in enumerate(fx["ident/mid_len"]):
        x0, y0 = OP.top_left(int(mx), int(my), int(L))
        want = fx[f"ident/out{k}"]
        assert np.array_equal(OP.cut_image(img, x0, y0, int(L)), want), k
        # L == target: every resize rule is the identity there, so the whole pyramid stage is pinned for such windows
        assert np.array_equal(OP.patch_pyramid(img, int(mx), int(my), [int(L)], (int(L),))[0], want.astype(np.float32) / 255.0)


def test_point_chain_matches_reference():
    """get_all_features (MyUtils1.py:60-77): designed-feature order (D1), factors (D2), geo -> pixel with its +1 (D3),
    windows and zero-padded crops (D4, before the unpinned resize)."""
    fx = _fx()
    img, gt = fx["crop/img"], fx["crop/gt"]
    for k in range(int(fx["point/n"])):
        inner, obj, gx, gy = fx[f"point/{k}/inner_object_xy"]
        w, f = OP.get_scales(int(inner), int(obj))
        assert np.array_equal(np.array(w, np.float32)[None], fx[f"point/{k}/scales"])
        designed = np.concatenate([fx[f"point/{k}/fields"], f]).astype(np.float32)[None]
        assert np.array_equal(designed, fx[f"point/{k}/designed"])
        px, py = OP.geo_to_pixel(gt, float(gx), float(gy))
        for i, L in enumerate(w):
            x0, y0 = OP.top_left(px, py, L)
            assert np.array_equal(OP.cut_image(img, x0, y0, L), fx[f"point/{k}/crop{i}"]), (k, i)


def test_cv_inter_area_branches_known_answers():
    """`cv_resize_area_u8` = cv::resize(..., INTER_AREA) on uint8 restated (reference call site MyUtils1.py:202-216; no cv2 here, so the
    anchors are values worked out by hand from the published algorithm, one per branch)."""
    a = np.arange(16, dtype=np.uint8).reshape(4, 4)
    assert np.array_equal(OP.cv_resize_area_u8(a, 4), a)                                # scale 1: copy
    q = np.array([[1, 2], [3, 4]], np.uint8)
    assert OP.cv_resize_area_u8(q, 1)[0, 0] == 3 and OP.area_resize_u8(q, 1)[0, 0] == 2   # 2x: (10 + 2) >> 2 rounds the tie UP; exact area: half to even
    n = np.zeros((3, 3), np.uint8); n[1, 1] = 13
    assert OP.cv_resize_area_u8(n, 1)[0, 0] == 1                                         # 3x: 13 * float(1/9) = 1.44 -> 1
    n[1, 1] = 14
    assert OP.cv_resize_area_u8(n, 1)[0, 0] == 2                                         # 14 / 9 = 1.56 -> 2
    e = np.array([[10, 20], [30, 40]], np.uint8)
    assert np.array_equal(OP.cv_resize_area_u8(e, 4), np.kron(e, np.ones((2, 2), np.uint8)))   # exact 2x enlarging: every fx is 0 -> replication
    # 3 -> 2 (scale 1.5): tables [(0, 2/3), (1, 1/3)] and [(1, 1/3), (2, 2/3)]
    tab = OP.cv_area_tab(3, 2)
    assert [[si for si, _ in ent] for ent in tab] == [[0, 1], [1, 2]]
    np.testing.assert_allclose([[float(al) for _, al in ent] for ent in tab], [[2 / 3, 1 / 3], [1 / 3, 2 / 3]], rtol=1e-6)
    b = np.zeros((3, 3), np.uint8); b[1, 1] = 90
    assert OP.cv_resize_area_u8(b, 2).tolist() == [[10, 10], [10, 10]]                   # 90 * (1/3)^2
    # 3 -> 4 (enlarging, scale 0.75): sx = [0, 0, 1, 2], fx = [0, 2/3, 1/3, 0] -> coefficients of 2048
    ofs, a0, a1, dmax = OP.cv_linear_coeffs(3, 4)
    assert ofs.tolist() == [0, 0, 1, 2] and a1.tolist() == [0, 1365, 683, 0] and a0.tolist() == [2048, 683, 1365, 2048] and dmax == 3
    row = np.array([[0, 90, 30]] * 3, np.uint8)
    assert OP.cv_resize_area_u8(row, 4)[0].tolist() == [0, 60, 70, 30]                   # (0*683 + 90*1365)/2048 = 59.99 -> 60; (90*1365 + 30*683)/2048 = 69.99 -> 70


def test_cv_inter_area_properties():
    """Size-independent properties over all three branches: a constant image stays constant; shrinking stays within one grey level
    of the exact rational area average (they differ in ties and float rounding only); enlarging is a convex combination."""
    rng = np.random.default_rng(11)
    for L, t in ((64, 32), (96, 32), (43, 32), (51, 32), (24, 32), (16, 32), (200, 128), (77, 64), (31, 32), (33, 32)):
        c = np.full((L, L), 137, np.uint8)
        assert np.array_equal(OP.cv_resize_area_u8(c, t), np.full((t, t), 137, np.uint8)), (L, t)
        img = rng.integers(0, 256, (L, L), dtype=np.uint8)
        got = OP.cv_resize_area_u8(img, t)
        if L >= t:
            assert np.abs(got.astype(int) - OP.area_resize_u8(img, t).astype(int)).max() <= 1, (L, t)
        else:
            assert got.min() >= img.min() and got.max() <= img.max()
    img = rng.integers(0, 256, (3, 40, 40), dtype=np.uint8)
    for rule in ("opencv", "exact_area"):
        out = OP.patch_pyramid(img, 20, 20, [20, 40, 56], (32, 32, 32), resize=rule)
        assert [o.shape for o in out] == [(3, 32, 32)] * 3 and all(o.dtype == np.float32 for o in out)



def test_cv_is_area_fast_follows_the_double_quotient():
    """cv::resize decides its integer fast path on scale = 1. / ((double)t / L) against DBL_EPSILON, not on L % t:
    for k = 49, 93, 98, ... the double quotient is k +- 1e-14 and the float table path runs (reference call site MyUtils1.py:202-216)."""
    assert all(OP.cv_is_area_fast(k * 32, 32) for k in (1, 2, 3, 4, 5, 7, 8, 12, 48, 50))
    slow = [k for k in range(1, 130) if not OP.cv_is_area_fast(k * 4, 4)]
    assert slow == [49, 93, 98, 99, 103, 105, 107, 117, 123]
    assert not OP.cv_is_area_fast(100, 32) and not OP.cv_is_area_fast(16, 32)
    # on such a window both paths are area means; the table path rounds float sums (no integer block mean): a constant stays constant,
    # a random image stays within one grey level of the exact rational mean, and a 196 x 196 delta image gives the table weights' product
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (196, 196), dtype=np.uint8)
    got = OP.cv_resize_area_u8(img, 4)
    assert np.abs(got.astype(int) - OP.area_resize_u8(img, 4).astype(int)).max() <= 1
    assert np.array_equal(OP.cv_resize_area_u8(np.full((196, 196), 201, np.uint8), 4), np.full((4, 4), 201, np.uint8))
    tab = OP.cv_area_tab(196, 4)
    assert [len(e) for e in tab] == [50, 50, 50, 49] or all(49 <= len(e) <= 51 for e in tab)      # k = 49 cells + the 1e-14 slivers
