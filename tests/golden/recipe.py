"""Deterministic tensor recipe shared by the golden-vector generator and the tests.

The reference's checkpoints (88 M parameters) are far too large to commit, and its
own initialisation depends on torch's global RNG stream and on module construction
order.  Fixtures therefore pin the *weights* by name: every tensor is a pure function
of (its state_dict key, its shape), generated here with numpy's PCG64 stream seeded by
crc32(key).  make_golden.py loads these weights into the unmodified reference modules;
the tests load the very same weights into the oracle restatement and into the HIP path.

Nothing here comes from the reference tree; it is fixture plumbing.
"""
import zlib

import numpy as np


def _rng(name: str, salt: int = 0) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(name.encode("utf-8")), salt])


def det_weight(name: str, shape) -> np.ndarray:
    """Deterministic float32 parameter for state_dict key `name`."""
    shape = tuple(int(s) for s in shape)
    r = _rng(name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "running_var":                       # BatchNorm2d statistics of the v4 / v5 aux heads
        w = r.uniform(0.5, 1.5, shape)
    elif leaf == "running_mean":
        w = r.normal(0.0, 0.1, shape)
    elif ".aux.1." in name:                         # BatchNorm2d affine
        w = (1.0 + 0.1 * r.normal(size=shape)) if leaf == "weight" else 0.05 * r.normal(size=shape)
    elif "relative_position_bias_table" in name:
        w = r.normal(0.0, 0.3, shape)
    elif leaf in ("cls_token", "dist_token") or "pos_embed" in leaf:
        w = r.normal(0.0, 0.05, shape)
    elif "norm" in name.rsplit(".", 2)[-2] if "." in name else False:
        # LayerNorm affine: weight around 1, bias around 0 (both non-trivial so their
        # gradients are exercised).
        w = (1.0 + 0.1 * r.normal(size=shape)) if leaf == "weight" else 0.05 * r.normal(size=shape)
    elif leaf == "bias":
        w = r.normal(0.0, 0.02, shape)
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        std = 0.03 if len(shape) == 2 else 1.0 / np.sqrt(fan_in)
        w = r.normal(0.0, std, shape)
    else:
        w = r.normal(0.0, 0.02, shape)
    return np.ascontiguousarray(w, dtype=np.float32)


def det_input(name: str, shape, kind: str = "normal") -> np.ndarray:
    """Deterministic float32 input tensor (images in [0,1], features ~N(0,1), ...)."""
    shape = tuple(int(s) for s in shape)
    r = _rng("input:" + name, 1)
    if kind == "normal":
        x = r.normal(0.0, 1.0, shape)
    elif kind == "unit":  # image-like, uint8 grid / 255 as the reference's loaders produce
        x = r.integers(0, 256, shape).astype(np.float64) / 255.0
    elif kind == "designed":
        # 15 physical region features + 4 scale factors: wide dynamic range.
        x = np.exp(r.uniform(np.log(1e-2), np.log(1e3), shape))
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(x, dtype=np.float32)


def sample_index(name: str, numel: int, k: int = 2048) -> np.ndarray:
    """Which flat elements of tensor `name` a fixture pins."""
    if numel <= k:
        return np.arange(numel, dtype=np.int64)
    return np.sort(_rng("sample:" + name, 2).choice(numel, size=k, replace=False)).astype(np.int64)


def summarize(name: str, t: np.ndarray, k: int = 2048) -> dict:
    """Compact, order-independent pin of a tensor: shape, sum, L2 and k sampled entries."""
    t = np.asarray(t)
    flat = t.reshape(-1).astype(np.float64)
    idx = sample_index(name, flat.size, k)
    return {
        name + "/shape": np.asarray(t.shape, dtype=np.int64),
        name + "/sum": np.float64(flat.sum()),
        name + "/l2": np.float64(np.sqrt((flat * flat).sum())),
        name + "/vals": t.reshape(-1)[idx].astype(np.float32),
    }


def check_summary(name: str, got: np.ndarray, fx, rtol: float, k: int = 2048, max_rtol: float = None, atol: float = 0.0):
    """Assert `got` matches the pinned summary (SURVEY 8d parity gate form):
      * relative L2 error over the sampled entries <= rtol,
      * max-abs error / max-abs(expected) over the sampled entries <= max_rtol (default 10*rtol),
      * the whole tensor's L2 norm within rtol of the pinned one.
    `atol` is an absolute per-element floor for tensors whose expected value is pure cancellation
    noise (e.g. the shared output bias of a Siamese pair: its two gradient halves cancel exactly).
    """
    got = np.asarray(got)
    shape = tuple(int(s) for s in fx[name + "/shape"])
    assert tuple(got.shape) == shape, f"{name}: shape {got.shape} != {shape}"
    flat = got.reshape(-1).astype(np.float64)
    assert np.isfinite(flat).all(), f"{name}: non-finite values"
    idx = sample_index(name, flat.size, k)
    want = fx[name + "/vals"].astype(np.float64)
    max_rtol = 10.0 * rtol if max_rtol is None else max_rtol
    err = flat[idx] - want
    l2w = float(np.sqrt((want * want).sum()))
    l2e = float(np.sqrt((err * err).sum()))
    scale = float(np.abs(want).max())
    assert l2e <= rtol * l2w + atol * np.sqrt(idx.size) + 1e-30, f"{name}: rel-L2 error {l2e / max(l2w, 1e-300):.3e} > {rtol}"
    assert np.abs(err).max() <= max_rtol * scale + atol + 1e-30, (
        f"{name}: max-abs error {np.abs(err).max():.3e} vs scale {scale:.3e} (> {max_rtol})")
    l2 = float(fx[name + "/l2"])
    got_l2 = float(np.sqrt((flat * flat).sum()))
    assert abs(got_l2 - l2) <= rtol * max(l2, 1e-30) + atol * np.sqrt(flat.size) + 1e-30, f"{name}: l2 {got_l2} != {l2}"


def summary_error(name: str, got: np.ndarray, fx, k: int = 2048):
    """(rel-L2, max-abs/scale) of `got` against the pinned sample -- for reporting drift."""
    flat = np.asarray(got).reshape(-1).astype(np.float64)
    idx = sample_index(name, flat.size, k)
    want = fx[name + "/vals"].astype(np.float64)
    err = flat[idx] - want
    return (float(np.sqrt((err * err).sum()) / max(np.sqrt((want * want).sum()), 1e-300)),
            float(np.abs(err).max() / max(np.abs(want).max(), 1e-300)))


def worst_gradient(prefix: str, named_grads, fx, k: int = 2048, floor: float = 1e-8):
    """(name, rel-L2) of the gradient furthest from its pinned sample, over the tensors whose fixture sample has an l2 norm above
    `floor` -- exactly-zero gradients (the k-bias of an attention block, say) have no relative error and are skipped."""
    worst = ("(none)", 0.0)
    for n, g in named_grads:
        if g is None:
            continue
        want = fx[prefix + n + "/vals"].astype(np.float64)
        if np.sqrt((want * want).sum()) < floor:
            continue
        e = summary_error(prefix + n, g, fx, k=k)[0]
        if e > worst[1]:
            worst = (n, e)
    return worst

