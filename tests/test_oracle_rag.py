"""CPU: oracle/rag.py (the build's own spec for the label-raster RAG and designed attributes) on
hand-checkable rasters."""
import numpy as np

from oracle import rag as OR

L = np.array([[0, 0, 1, 1],
              [0, 2, 2, 1],
              [3, 3, 2, 1]], dtype=np.int32)


def test_edges_and_weights_known_answer():
    edges, w = OR.rag_edges(L, 4)
    assert edges.tolist() == [[0, 1], [0, 2], [0, 3], [1, 2], [2, 3]]
    #  0-1: (0,1)|(0,2)                         -> 1
    #  0-2: (1,0)|(1,1), (0,1)/(1,1)            -> 2
    #  0-3: (1,0)/(2,0)                         -> 1
    #  1-2: (0,2)/(1,2), (1,2)|(1,3), (2,2)|(2,3) -> 3
    #  2-3: (1,1)/(2,1), (2,1)|(2,2)            -> 2
    assert w.tolist() == [1, 2, 1, 3, 2]


def test_stats_and_features_known_answer():
    tile = np.arange(2 * 12, dtype=np.uint8).reshape(2, 3, 4) * 3
    st = OR.label_stats(L, tile, 5)                         # id 4 never occurs
    assert st["count"].tolist() == [3, 4, 3, 2, 0]
    assert st["bbox"][0].tolist() == [0, 0, 1, 1] and st["bbox"][1].tolist() == [2, 0, 3, 2] and st["bbox"][4].tolist() == [2**31 - 1, 2**31 - 1, -1, -1]
    assert st["sum"][0].tolist() == [0 + 3 + 12, 36 + 39 + 48]
    # label 3 = pixels (2,0),(2,1): inner edges: up x2, right x1 = 3; border: left 1 + bottom 2 = 3
    assert st["peri"][3].tolist() == [3, 3]
    assert int(st["peri"][:, 0].sum()) == 2 * int(OR.rag_edges(L, 5)[1].sum())      # every shared pixel edge is seen from both sides
    f = OR.designed_features(st)
    assert f.shape == (5, 15) and not f[4].any()
    area, per, ln, wd = f[3, :4]
    assert (area, per, ln, wd) == (2.0, 6.0, 2.0, 1.0)
    assert abs(f[3, 4] - 6.0 / (2 * 3)) < 1e-7 and abs(f[3, 11] - 6.0 / (4 * np.sqrt(2.0))) < 1e-6 and f[3, 12] == 1.0 and f[3, 14] == 3.0
    m0 = (tile[0, 2, 0] + tile[0, 2, 1]) / 2.0
    assert abs(f[3, 8] - m0) < 1e-6 and abs(f[3, 5] - np.std([float(tile[0, 2, 0]), float(tile[0, 2, 1])])) < 1e-6
    assert f[3, 7] == 0.0 and f[3, 10] == 0.0            # only two bands


def test_ids_outside_range_are_ignored():
    L2 = L.copy(); L2[0, 0] = -1; L2[2, 3] = 9
    edges, w = OR.rag_edges(L2, 4)
    assert all(0 <= a < b < 4 for a, b in edges.tolist())
    st = OR.label_stats(L2, np.zeros((1, 3, 4), np.uint8), 4)
    assert st["count"].tolist() == [2, 3, 3, 2]
