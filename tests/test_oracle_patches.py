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
