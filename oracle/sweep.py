"""Oracle (test infrastructure): the ExtractFeatures region-adjacency similarity sweep.

Restatement of the reference's pure-numpy arithmetic, PINNED to the reference's own outputs
(tests/golden/sweep.npz, produced by tests/golden/make_golden.py::gen_sweep which imports the unmodified
ExtractFeatures.py and calls these very functions; checked by tests/test_oracle_sweep.py):
  Euclidean_distance        ExtractFeatures.py:119-147  (= Train_SMT.py:115-131 = MC_Lyu_2020 :228-237)
  per-edge loop body        ExtractFeatures.py:188-219  (gather point rows, np.mean(axis=0), distance, .max())
  RAG edge filtering        MyUtils2.py:177-192         (edges with LEFT_FID == -1 or RIGHT_FID == -1 skipped)
Storage (HDF5 rows, shapefile fields) is replaced by in-memory arrays: features F[P,100]
float32 in point-id order, polygon -> points as CSR (ptr[S+1], idx[*]), edges as int [E,2].

`merge := simi < margin` is the build's definition; the reference stops at
writing the float `simi`.
"""
import numpy as np


def euclidean_distance(X: np.ndarray, Y: np.ndarray) -> np.ndarray:
    """D[n,m] = sqrt(max(0, |x|^2 + |y|^2 - 2 x.y)) in the input dtype (float32 from the feature store)."""
    n, m = X.shape[0], Y.shape[0]
    X2 = np.sum(X ** 2, axis=1)
    Y2 = np.sum(Y ** 2, axis=1)
    D = np.tile(X2.reshape(n, 1), (1, m)) + np.tile(Y2.reshape(m, 1), (1, n)).T - 2 * np.dot(X, Y.T)
    D[D < 0] = 0
    return np.sqrt(D)


def pool_polygon(F: np.ndarray, ptr: np.ndarray, idx: np.ndarray, s: int) -> np.ndarray:
    """Mean feature of polygon s over its sample points: rows are stacked in PointID order and
    reduced with np.mean(axis=0) (ExtractFeatures.py:190-212)."""
    rows = F[idx[ptr[s]:ptr[s + 1]]]
    return np.mean(rows, axis=0)


def pool_all(F: np.ndarray, ptr: np.ndarray, idx: np.ndarray) -> np.ndarray:
    S = len(ptr) - 1
    out = np.zeros((S, F.shape[1]), dtype=F.dtype)
    for s in range(S):
        if ptr[s + 1] > ptr[s]:
            out[s] = pool_polygon(F, ptr, idx, s)
    return out


def edge_similarity(F: np.ndarray, ptr: np.ndarray, idx: np.ndarray, edges: np.ndarray) -> np.ndarray:
    """simi[e] for every RAG edge (L,R); edges touching -1 are skipped upstream and get NaN here."""
    E = edges.shape[0]
    simi = np.full((E,), np.nan, dtype=np.float32)
    for e in range(E):
        L, R = int(edges[e, 0]), int(edges[e, 1])
        if L == -1 or R == -1:
            continue
        a = pool_polygon(F, ptr, idx, L)[np.newaxis, :]
        b = pool_polygon(F, ptr, idx, R)[np.newaxis, :]
        simi[e] = euclidean_distance(a, b).max()
    return simi


def edge_similarity_from_pooled(pooled: np.ndarray, edges: np.ndarray) -> np.ndarray:
    """Same distance formula evaluated on already pooled [S,p] features, vectorised over edges.
    Operation order per edge is identical to euclidean_distance on 1xp inputs:
    fl(fl(|a|^2 + |b|^2) - fl(2*fl(a.b)))."""
    E = edges.shape[0]
    simi = np.full((E,), np.nan, dtype=np.float32)
    for e in range(E):
        L, R = int(edges[e, 0]), int(edges[e, 1])
        if L == -1 or R == -1:
            continue
        simi[e] = euclidean_distance(pooled[L][None, :], pooled[R][None, :]).max()
    return simi


def merge_decisions(simi: np.ndarray, margin: float = 1.0) -> np.ndarray:
    """merge[e] = simi[e] < margin (NaN -> False); margin from Train_SMT.py:380."""
    with np.errstate(invalid="ignore"):
        return np.less(simi, np.float32(margin))


# ---- pinned-order variant (oracle/sweep_strict.c) ----------------------------------------------
def _strict_lib():
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "liboracle_sweep.so")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
    lib = ctypes.CDLL(path)
    return lib, ctypes


def strict_segment_mean(F: np.ndarray, ptr: np.ndarray, idx: np.ndarray) -> np.ndarray:
    lib, C = _strict_lib()
    F = np.ascontiguousarray(F, np.float32); ptr = np.ascontiguousarray(ptr, np.int32); idx = np.ascontiguousarray(idx, np.int32)
    S, D = len(ptr) - 1, F.shape[1]
    out = np.zeros((S, D), np.float32)
    lib.dm_oracle_segment_mean(F.ctypes.data_as(C.c_void_p), ptr.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
                               out.ctypes.data_as(C.c_void_p), C.c_int32(S), C.c_int32(D))
    return out


def strict_edge_similarity(pooled: np.ndarray, edges: np.ndarray, margin: float = 1.0):
    lib, C = _strict_lib()
    pooled = np.ascontiguousarray(pooled, np.float32); edges = np.ascontiguousarray(edges, np.int32)
    E, D = edges.shape[0], pooled.shape[1]
    simi = np.zeros(E, np.float32); merge = np.zeros(E, np.uint8)
    lib.dm_oracle_edge_similarity(pooled.ctypes.data_as(C.c_void_p), edges.ctypes.data_as(C.c_void_p),
                                  simi.ctypes.data_as(C.c_void_p), merge.ctypes.data_as(C.c_void_p),
                                  C.c_int32(E), C.c_int32(D), C.c_float(margin))
    return simi, merge.astype(bool)
