"""CPU: the sweep oracle (numpy restatement + pinned-order C variant) on known answers and edge cases."""
import numpy as np
import pytest

from oracle import sweep as OS


def test_euclidean_distance_known_answers():
    X = np.array([[0, 0, 0], [3, 4, 0]], np.float32)
    Y = np.array([[0, 0, 0], [0, 0, 12], [3, 4, 0]], np.float32)
    D = OS.euclidean_distance(X, Y)
    assert D.dtype == np.float32 and D.shape == (2, 3)
    np.testing.assert_array_equal(D, np.array([[0, 12, 5], [5, 13, 0]], np.float32))
    # the expanded form clamps tiny negatives instead of producing NaN
    x = np.full((1, 100), 0.1, np.float32)
    assert OS.euclidean_distance(x, x.copy())[0, 0] == 0.0


def make_case(rng, S=50, D=100, kmax=5, E=200, dead=True):
    counts = rng.integers(1, kmax + 1, size=S)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    P = int(ptr[-1])
    idx = rng.permutation(P).astype(np.int32)
    F = (rng.normal(size=(P, D)) * 0.08).astype(np.float32)
    edges = rng.integers(0, S, size=(E, 2)).astype(np.int32)
    same = edges[:, 0] == edges[:, 1]          # a RAG never joins a polygon to itself
    edges[same, 1] = (edges[same, 0] + 1) % S
    if dead:
        edges[::17, 0] = -1
        edges[5::23, 1] = -1
    return F, ptr, idx, edges


def test_sweep_numpy_vs_strict_order():
    rng = np.random.default_rng(0)
    F, ptr, idx, edges = make_case(rng)
    simi = OS.edge_similarity(F, ptr, idx, edges)
    pooled = OS.pool_all(F, ptr, idx)
    np.testing.assert_array_equal(simi, OS.edge_similarity_from_pooled(pooled, edges))
    spooled = OS.strict_segment_mean(F, ptr, idx)
    np.testing.assert_array_equal(spooled, pooled)            # row-by-row accumulation == np.mean(axis=0)
    ssimi, smerge = OS.strict_edge_similarity(spooled, edges, 1.0)
    dead = (edges < 0).any(1)
    assert np.isnan(simi[dead]).all() and np.isnan(ssimi[dead]).all() and not smerge[dead].any()
    live = ~dead
    np.testing.assert_allclose(ssimi[live], simi[live], rtol=3e-6, atol=1e-6)   # np.dot (BLAS) vs pinned pairwise order
    merge = OS.merge_decisions(simi, 1.0)
    near = np.abs(simi - 1.0) < 1e-5
    assert np.array_equal(merge[live & ~near], smerge[live & ~near])
    assert 0 < merge[live].mean() < 1, "fixture must contain both merge and no-merge edges"


@pytest.mark.parametrize("D", [1, 7, 8, 9, 64, 100, 128, 129, 300])
def test_strict_pairwise_matches_numpy_sum(D):
    """The pinned order IS numpy's contiguous float32 pairwise sum: |x|^2 computed both ways is bit-equal."""
    rng = np.random.default_rng(D)
    pooled = rng.normal(size=(2, D)).astype(np.float32)
    pooled[1] = 0
    simi, _ = OS.strict_edge_similarity(pooled, np.array([[0, 1]], np.int32))
    want = np.sqrt(np.sum(pooled[0] ** 2))
    assert simi[0] == want


def test_self_distance_is_cancellation_noise_not_nan():
    """|x|^2 + |x|^2 - 2 x.x in float32 is 0 up to rounding; both orders clamp it and stay far below margin."""
    rng = np.random.default_rng(4)
    pooled = rng.normal(size=(3, 100)).astype(np.float32)
    e = np.array([[1, 1]], np.int32)
    assert OS.edge_similarity_from_pooled(pooled, e)[0] < 5e-3
    assert OS.strict_edge_similarity(pooled, e)[0][0] < 5e-3


def test_empty_and_single_point_polygons():
    F = np.arange(12, dtype=np.float32).reshape(3, 4)
    ptr = np.array([0, 0, 1, 3], np.int32)
    idx = np.array([2, 0, 1], np.int32)
    pooled = OS.strict_segment_mean(F, ptr, idx)
    np.testing.assert_array_equal(pooled[0], 0)
    np.testing.assert_array_equal(pooled[1], F[2])
    np.testing.assert_array_equal(pooled[2], (F[0] + F[1]) / np.float32(2))
