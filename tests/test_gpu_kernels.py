"""GPU: kernel-level parity through the C-ABI (deepmerge_amd.ops -> libdeepmerge_hip.so).

GEMM / attention layouts are first checked with small-integer data, for which bf16 and fp32 MFMA
results are EXACT, so any lane/fragment/swizzle mistake shows up as a bit mismatch; then with random
data against the CPU oracle (numpy / torch fp32) at the tolerances of SURVEY 8d.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from deepmerge_amd import ops
    return ops


def _ints(rng, shape, lo=-3, hi=4):
    return torch.from_numpy(rng.integers(lo, hi, size=shape).astype(np.float32))


DT = {"fp32": torch.float32, "bf16": torch.bfloat16}


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (512, 512, 256), (256, 384, 192), (200, 136, 72), (16, 768, 768), (1000, 100, 3840), (64, 2304, 768)])
def test_gemm_exact_integers(mode, layout, M, N, K):
    ops = _ops()
    from deepmerge_amd._lib import DM_NN, DM_NT, DM_TN
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = _ints(rng, (M, K))
    b = _ints(rng, (N, K))
    want = a.double() @ b.double().T
    dt = DT[mode]
    if mode == "bf16" and layout != "NT" and N % 8 != 0:
        pytest.skip("bf16 n-contiguous operands need N % 8 == 0 (the model's N = 100 head runs in fp32)")
    if layout == "NT":
        A, B_, lay = a, b, DM_NT
    elif layout == "NN":
        A, B_, lay = a, b.T.contiguous(), DM_NN
    else:
        A, B_, lay = a.T.contiguous(), b.T.contiguous(), DM_TN
    A, B_ = A.to(DEV).to(dt), B_.to(DEV).to(dt)
    Cc = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(lay, A, B_, Cc, M, N, K, lda=A.shape[1], ldb=B_.shape[1], ldc=N)
    got = Cc.cpu().double()
    assert torch.equal(got, want), f"max diff {(got - want).abs().max()}"


@pytest.fixture
def ring_env():
    """Route every legal bf16 NT product to the two-workgroups-per-CU ring kernel (dm_gemm_ring.hip) for one test."""
    import os
    old = {k: os.environ.get(k) for k in ("DM_GEMM_RING", "DM_GEMM_RING_WM")}

    def set_(wm):
        os.environ["DM_GEMM_RING"] = "2"
        os.environ["DM_GEMM_RING_WM"] = str(wm)
    yield set_
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.fixture
def w4_env():
    """Route every legal bf16 NT / NN product to the 4-wave persistent kernel (dm_gemm_w4.hip) for one test."""
    import os
    old = os.environ.get("DM_GEMM_W4")
    os.environ["DM_GEMM_W4"] = "2"
    yield
    if old is None:
        os.environ.pop("DM_GEMM_W4", None)
    else:
        os.environ["DM_GEMM_W4"] = old


@pytest.mark.parametrize("layout", ["NT", "NN"])
@pytest.mark.parametrize("M,N,K", [(256, 192, 128), (512, 384, 256), (1000, 192, 768), (16, 768, 3072), (4096, 2304, 768),
                                   (66000, 192, 128), (33000, 384, 256), (16384, 768, 128)])
def test_gemm_w4_exact_integers(w4_env, layout, M, N, K):
    """Exact small-integer products through the 4-wave persistent kernel (register-staged operands, tiles chained into one K-step
    sequence): one to several tiles per workgroup, ragged M, 2 .. 48 K steps, both layouts of B."""
    _persistent_exact_integers(layout, M, N, K)


@pytest.mark.parametrize("Mrows,N,K,acc", [(16384, 768, 768, True), (4096, 2304, 768, False), (16384, 3072, 768, True), (256, 256, 192, True),
                                          (1280, 768, 3072, False)])
def test_gemm_w4_wgrad(Mrows, N, K, acc):
    """TN through the 4-wave kernel (DM_GEMM_W4_TN=2: every legal product): both operands m-contiguous (transposed LDS reads), one
    (tile, K slice) per workgroup with a shorter last slice, slab reduction in slice order; unsplit products write the gradient
    directly (with and without accumulation).  Exact in small integers, run-to-run identical."""
    import os
    ops = _ops()
    from deepmerge_amd._lib import DM_TN
    old = os.environ.get("DM_GEMM_W4_TN")
    os.environ["DM_GEMM_W4_TN"] = "2"
    try:
        rng = np.random.default_rng(Mrows + N + K)
        dy = _ints(rng, (Mrows, N), -1, 2)
        x = _ints(rng, (Mrows, K), -1, 2)
        g0 = torch.from_numpy(rng.integers(-5, 6, size=(N, K)).astype(np.float32))
        want = (g0.double() if acc else 0) + dy.double().T @ x.double()
        dyd, xd = dy.to(DEV).to(torch.bfloat16), x.to(DEV).to(torch.bfloat16)
        G = g0.clone().to(DEV)
        ops.gemm(DM_TN, dyd, xd, G, N, K, Mrows, lda=N, ldb=K, ldc=K, accumulate=acc)
        assert torch.equal(G.cpu().double(), want), f"max diff {(G.cpu().double() - want).abs().max()}"
        G2 = g0.clone().to(DEV)
        ops.gemm(DM_TN, dyd, xd, G2, N, K, Mrows, lda=N, ldb=K, ldc=K, accumulate=acc)
        assert torch.equal(G, G2)
        # with the bias gradient (column sums of dy) riding along
        G3, cs = g0.clone().to(DEV), torch.ones(N, device=DEV)
        ops.gemm(DM_TN, dyd, xd, G3, N, K, Mrows, lda=N, ldb=K, ldc=K, accumulate=acc, colsum_out=cs, colsum_accumulate=True)
        assert torch.equal(G3, G) and torch.equal(cs.cpu().double(), 1.0 + dy.double().sum(0))
    finally:
        if old is None:
            os.environ.pop("DM_GEMM_W4_TN", None)
        else:
            os.environ["DM_GEMM_W4_TN"] = old


def test_gemm_w4_epilogues(w4_env):
    test_gemm_epilogues("bf16", M=512, N=384)


def _persistent_exact_integers(layout, M, N, K):
    """Exact small-integer products through a persistent kernel: one to several tiles per workgroup (the flattened K-step
    sequence crosses tile boundaries), ragged M, 2 .. 48 K steps, both operand layouts of B; fp32 and bf16 outputs."""
    ops = _ops()
    from deepmerge_amd._lib import DM_NN, DM_NT
    rng = np.random.default_rng(M + 3 * N + 7 * K)
    a = _ints(rng, (M, K))
    b = _ints(rng, (N, K) if layout == "NT" else (K, N))
    want = a.double() @ (b.double().T if layout == "NT" else b.double())
    A, B_ = a.to(DEV).to(torch.bfloat16), b.to(DEV).to(torch.bfloat16)
    lay, ldb = (DM_NT, K) if layout == "NT" else (DM_NN, N)
    Cc = torch.full((M + 3, N), float("nan"), device=DEV)
    ops.gemm(lay, A, B_, Cc, M, N, K, lda=K, ldb=ldb, ldc=N)
    assert torch.equal(Cc[:M].cpu().double(), want), f"max diff {(Cc[:M].cpu().double() - want).abs().max()}"
    assert torch.isnan(Cc[M:]).all(), "rows past M must not be written"
    Ch = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(lay, A, B_, Ch, M, N, K, lda=K, ldb=ldb, ldc=N)
    assert torch.equal(Ch.cpu().double(), want.to(torch.bfloat16).double())


@pytest.mark.parametrize("wm", [8, 4])
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (512, 384, 96), (1000, 136, 768), (16, 768, 3072), (4096, 2304, 768), (300, 8, 160), (257, 264, 2304)])
def test_gemm_ring_exact_integers(ring_env, wm, M, N, K):
    """Exact small-integer products through the ring kernel: ragged M (zero-filled by the DMA descriptor, masked stores),
    N tails (N % 8 == 0), 2 .. 96 K steps, both tile heights; bf16 and fp32 outputs."""
    ops = _ops()
    from deepmerge_amd._lib import DM_NT
    ring_env(wm)
    rng = np.random.default_rng(M + 3 * N + 7 * K + wm)
    a, b = _ints(rng, (M, K)), _ints(rng, (N, K))
    want = a.double() @ b.double().T
    A, B_ = a.to(DEV).to(torch.bfloat16), b.to(DEV).to(torch.bfloat16)
    Cc = torch.full((M + 3, N), float("nan"), device=DEV)
    ops.gemm(DM_NT, A, B_, Cc, M, N, K, lda=K, ldb=K, ldc=N)
    assert torch.equal(Cc[:M].cpu().double(), want), f"max diff {(Cc[:M].cpu().double() - want).abs().max()}"
    assert torch.isnan(Cc[M:]).all(), "rows past M must not be written"
    Ch = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(DM_NT, A, B_, Ch, M, N, K, lda=K, ldb=K, ldc=N)
    assert torch.equal(Ch.cpu().double(), want.to(torch.bfloat16).double())


@pytest.mark.parametrize("wm", [8, 4])
def test_gemm_ring_epilogues(ring_env, wm):
    ring_env(wm)
    test_gemm_epilogues("bf16")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_gemm_epilogues(mode, M=192, N=256):
    """bias, GELU (+saved pre-activation), DGELU, residual, accumulate, bf16 output, grouped rows."""
    ops = _ops()
    from deepmerge_amd._lib import DM_EPI_DGELU, DM_EPI_GELU, DM_NN, DM_NT
    rng = np.random.default_rng(5)
    K = 128
    dt = DT[mode]
    a, b = _ints(rng, (M, K), -2, 3), _ints(rng, (N, K), -2, 3)
    bias = torch.from_numpy(rng.normal(size=N).astype(np.float32))
    res = torch.from_numpy(rng.normal(size=(M, N)).astype(np.float32))
    base = (a.double() @ b.double().T) * 0.03125
    a = a * 0.03125   # exact in bf16
    A, B_ = a.to(DEV).to(dt), b.to(DEV).to(dt)
    # bias + residual, fp32 out
    out = torch.empty((M, N), device=DEV)
    ops.gemm(DM_NT, A, B_, out, M, N, K, lda=K, ldb=K, ldc=N, bias=bias.to(DEV), residual=res.to(DEV))
    np.testing.assert_allclose(out.cpu().double().numpy(), (base + bias.double() + res.double()).numpy(), rtol=0, atol=2e-6)
    # GELU with saved pre-activation
    pre = torch.empty((M, N), device=DEV, dtype=dt)
    h = torch.empty((M, N), device=DEV, dtype=dt)
    ops.gemm(DM_NT, A, B_, h, M, N, K, lda=K, ldb=K, ldc=N, bias=bias.to(DEV), epilogue=DM_EPI_GELU, aux=pre, ldaux=N)
    u = (base + bias.double())
    tol = 1e-5 if mode == "fp32" else 1.6e-2
    np.testing.assert_allclose(pre.float().cpu().double().numpy(), u.numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(h.float().cpu().double().numpy(), torch.nn.functional.gelu(u).numpy(), rtol=tol, atol=tol)
    # DGELU epilogue: out = (A B^T) * gelu'(aux)
    aux = torch.from_numpy(rng.normal(size=(M, N)).astype(np.float32))
    out2 = torch.empty((M, N), device=DEV)
    ops.gemm(DM_NT, A, B_, out2, M, N, K, lda=K, ldb=K, ldc=N, epilogue=DM_EPI_DGELU, aux=aux.to(DEV), ldaux=N)
    x = aux.double().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    np.testing.assert_allclose(out2.cpu().double().numpy(), (base * x.grad).numpy(), rtol=1e-5, atol=1e-5 if mode == "fp32" else 1e-4)
    # GELU with the derivative saved instead of the pre-activation, and the matching multiply epilogue
    from deepmerge_amd._lib import DM_EPI_GELU_GRAD, DM_EPI_MUL
    gp = torch.empty((M, N), device=DEV, dtype=dt)
    h2 = torch.empty((M, N), device=DEV, dtype=dt)
    ops.gemm(DM_NT, A, B_, h2, M, N, K, lda=K, ldb=K, ldc=N, bias=bias.to(DEV), epilogue=DM_EPI_GELU_GRAD, aux=gp, ldaux=N)
    uu = u.clone().requires_grad_(True)
    torch.nn.functional.gelu(uu).sum().backward()
    np.testing.assert_allclose(h2.float().cpu().double().numpy(), torch.nn.functional.gelu(u).numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(gp.float().cpu().double().numpy(), uu.grad.numpy(), rtol=tol, atol=tol)
    out4 = torch.empty((M, N), device=DEV)
    ops.gemm(DM_NT, A, B_, out4, M, N, K, lda=K, ldb=K, ldc=N, epilogue=DM_EPI_MUL, aux=aux.to(DEV), ldaux=N)
    np.testing.assert_allclose(out4.cpu().double().numpy(), (base * aux.double()).numpy(), rtol=1e-6, atol=1e-6)
    # accumulate
    out3 = res.clone().to(DEV)
    ops.gemm(DM_NT, A, B_, out3, M, N, K, lda=K, ldb=K, ldc=N, accumulate=True)
    np.testing.assert_allclose(out3.cpu().double().numpy(), (base + res.double()).numpy(), rtol=0, atol=2e-6)
    # grouped rows: groups of 64 rows land in slices of a [3, 100, N] cube at token offset 36
    cube = torch.zeros((M // 64, 100, N), device=DEV)
    ops.gemm(DM_NT, A, B_, cube[:, 36:], M, N, K, lda=K, ldb=K, ldc=N, rows_per_group=64, group_stride=100 * N)
    np.testing.assert_allclose(cube[:, 36:].cpu().double().numpy().reshape(M, N), base.numpy(), rtol=0, atol=2e-6)
    assert float(cube[:, :36].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("Mrows,N,K", [(16384, 768, 768), (4096, 2304, 768), (777 * 8, 768, 3072)])
def test_gemm_wgrad_split_k(mode, Mrows, N, K):
    """TN with automatic split-K + deterministic slab reduction, accumulate into an existing gradient."""
    ops = _ops()
    from deepmerge_amd._lib import DM_TN
    rng = np.random.default_rng(11)
    dt = DT[mode]
    dy = _ints(rng, (Mrows, N), -1, 2)
    x = _ints(rng, (Mrows, K), -1, 2)
    g0 = torch.from_numpy(rng.integers(-5, 6, size=(N, K)).astype(np.float32))
    want = g0.double() + dy.double().T @ x.double()
    G = g0.clone().to(DEV)
    ops.gemm(DM_TN, dy.to(DEV).to(dt), x.to(DEV).to(dt), G, N, K, Mrows, lda=N, ldb=K, ldc=K, accumulate=True)
    assert torch.equal(G.cpu().double(), want)
    G2 = g0.clone().to(DEV)
    ops.gemm(DM_TN, dy.to(DEV).to(dt), x.to(DEV).to(dt), G2, N, K, Mrows, lda=N, ldb=K, ldc=K, accumulate=True)
    assert torch.equal(G, G2), "split-K reduction must be run-to-run deterministic"


def test_gemm_operands_beyond_2gib():
    """Eval batches (ExtractFeatures.py: 2000 points x 192 tokens x hidden 3072) give operands > 2 GiB; the tile
    descriptors are re-based per panel / per stage, so every layout must still be exact there."""
    ops = _ops()
    from deepmerge_amd._lib import DM_NN, DM_NT, DM_TN
    g = torch.Generator(device=DEV); g.manual_seed(5)
    R, W, S = 360000, 3072, 64                                    # R*W*2 B = 2.2 GB
    big = torch.randint(-1, 2, (R, W), device=DEV, generator=g).to(torch.bfloat16)
    small_w = torch.randint(-1, 2, (S, W), device=DEV, generator=g).to(torch.bfloat16)
    small_r = torch.randint(-1, 2, (R, S), device=DEV, generator=g).to(torch.bfloat16)
    assert big.numel() * 2 > (1 << 31)
    bigf = big.float()
    # NT, k-contiguous big A: C[R,S] = big @ small_w^T
    out = torch.empty((R, S), device=DEV)
    ops.gemm(DM_NT, big, small_w, out, R, S, W, lda=W, ldb=W, ldc=S)
    assert torch.equal(out, bigf @ small_w.float().T)
    # NN, big output (> 2 GiB fp32) : C[R,W] = small_r @ small_w
    out2 = torch.empty((R, W), device=DEV)
    ops.gemm(DM_NN, small_r, small_w, out2, R, W, S, lda=S, ldb=W, ldc=W)
    assert torch.equal(out2, small_r.float() @ small_w.float())
    del out2
    # NN, m-contiguous big B with K = R rows: C[S,W] = small_r^T(as [S,R] row-major) @ big
    a_t = small_r.T.contiguous()
    out3 = torch.empty((S, W), device=DEV)
    ops.gemm(DM_NN, a_t, big, out3, S, W, R, lda=R, ldb=W, ldc=W)
    want3 = a_t.float() @ bigf
    assert torch.equal(out3, want3)
    # TN, both operands m-contiguous, K = R: C[W,S] = big^T @ small_r
    out4 = torch.empty((W, S), device=DEV)
    ops.gemm(DM_TN, big, small_r, out4, W, S, R, lda=W, ldb=S, ldc=S)
    assert torch.equal(out4, want3.T.contiguous())


def test_gemm_generic_fp32_odd_shapes():
    ops = _ops()
    from deepmerge_amd._lib import DM_NN, DM_NT, DM_TN
    rng = np.random.default_rng(3)
    M, N, K = 37, 768, 19
    a = torch.from_numpy(rng.normal(size=(M, K)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(N, K)).astype(np.float32))
    bias = torch.from_numpy(rng.normal(size=N).astype(np.float32))
    out = torch.empty((M, N), device=DEV)
    ops.gemm(DM_NT, a.to(DEV), b.to(DEV), out, M, N, K, lda=K, ldb=K, ldc=N, bias=bias.to(DEV))
    np.testing.assert_allclose(out.cpu().numpy(), (a.double() @ b.double().T + bias.double()).numpy(), rtol=1e-5, atol=1e-5)
    dy = torch.from_numpy(rng.normal(size=(M, N)).astype(np.float32))
    dx = torch.empty((M, K), device=DEV)
    ops.gemm(DM_NN, dy.to(DEV), b.to(DEV), dx, M, K, N, lda=N, ldb=K, ldc=K)
    np.testing.assert_allclose(dx.cpu().numpy(), (dy.double() @ b.double()).numpy(), rtol=1e-4, atol=1e-4)
    dw = torch.empty((N, K), device=DEV)
    ops.gemm(DM_TN, dy.to(DEV), a.to(DEV), dw, N, K, M, lda=N, ldb=K, ldc=K)
    np.testing.assert_allclose(dw.cpu().numpy(), (dy.double().T @ a.double()).numpy(), rtol=1e-4, atol=1e-4)


def test_gemm_rejects_bad_arguments():
    ops = _ops()
    from deepmerge_amd._lib import DM_NT
    a = torch.zeros((8, 19), device=DEV, dtype=torch.bfloat16)
    b = torch.zeros((16, 19), device=DEV, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.gemm(DM_NT, a, b, torch.empty((8, 16), device=DEV), 8, 16, 19, lda=19, ldb=19, ldc=16)   # K % 8 != 0 in bf16
    with pytest.raises(ValueError):
        ops.gemm(DM_NT, a, b, torch.empty((8, 16), device=DEV), 0, 16, 16)
    with pytest.raises(RuntimeError):
        ops.gemm(DM_NT, a.cpu(), b.cpu(), torch.empty((8, 16)), 8, 16, 16)


# ---------------------------------------------------------------------------------------------
def _attn_ref(qkv, bias, scale):
    """fp64 reference of the fused core on CPU.  qkv [B,N,3,H,D] double; bias [H,N,N] or None."""
    q, k, v = qkv[:, :, 0].permute(0, 2, 1, 3), qkv[:, :, 1].permute(0, 2, 1, 3), qkv[:, :, 2].permute(0, 2, 1, 3)
    s = (q * scale) @ k.transpose(-1, -2)
    if bias is not None:
        s = s + bias[None]
    p = torch.softmax(s, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(qkv.shape[0], qkv.shape[1], -1)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("N,with_bias", [(256, True), (192, True), (128, False), (197, False), (198, False), (193, True), (160, True)])
def test_attention_pipelined_kernels(N, with_bias):
    """B * H >= 96 routes bf16 attention with 128 <= N <= 256 to the persistent LDS-DMA kernels (dm_attention_pipe.hip):
    exact tilings, masked (ragged) N without a bias (vit_model.py), and ragged N with a bias (forward pipelined, backward
    on the generic kernels)."""
    test_attention_forward_backward("bf16", N, with_bias, B=8, H=12)


@pytest.mark.parametrize("B,H,N,with_bias", [(9, 12, 256, True), (36, 3, 197, False), (11, 12, 128, True), (70, 12, 256, True), (13, 10, 192, True), (50, 12, 197, False),
                                             (29, 11, 224, False)])
def test_attention_pipelined_workgroup_order(B, H, N, with_bias):
    """The XCD-aware workgroup order of the pipelined kernels (pipe_coords: ids L, L + 8, ... are the row blocks of one
    (head, sample chunk)) with group counts that are NOT multiples of 8 (the surplus workgroups of the rounded-up grid must
    leave), uneven sample chunks, one and two row blocks.  The 8-wave kernels deal (head, sample) units out as 256 contiguous runs:
    the last two cases give runs of 2-3 / 1-2 units that cross head boundaries."""
    test_attention_forward_backward("bf16", N, with_bias, B=B, H=H)


@pytest.mark.parametrize("N,with_bias", [(256, True), (197, False), (160, True), (224, False)])
def test_attention_q32_deferred_maximum(N, with_bias):
    """dm_attention_q32.hip keeps the FIRST 32-key tile's row maximum as the softmax reference and rescales l / O only when a later
    tile exceeds it by more than 2^16.  Keys whose norm grows with their index force that path several times per row (scores of
    later tiles are far above the first tile's); the result must still be the exact softmax (fp64 reference, bf16 tolerance).
    Reference semantics: softmax over the whole row, nets/ShfitScaleFormer.py:127-131 / vit_model.py:125-127."""
    ops = _ops()
    rng = np.random.default_rng(N)
    B, H, D = 8, 12, 64
    qkv = rng.normal(size=(B, N, 3, H, D)).astype(np.float32)
    qkv[:, :, 1] *= (1.0 + np.arange(N, dtype=np.float32) / 12.0)[None, :, None, None]      # |k| grows 20x from the first to the last key
    qkv = torch.from_numpy(qkv).to(torch.bfloat16)
    bias = torch.from_numpy(rng.normal(size=(H, N, N)).astype(np.float32)) if with_bias else None
    o_ref, lse_ref = _attn_ref(qkv.double(), None if bias is None else bias.double(), 0.125)
    spread = (qkv[:, :, 0].double().permute(0, 2, 1, 3) @ qkv[:, :, 1].double().permute(0, 2, 3, 1)) * 0.125 * 1.4427
    assert ((spread[..., 32:].amax(-1) - spread[..., :32].amax(-1)) > 16).float().mean() > 0.5        # most rows do take the rescale path
    out, lse = ops.attention_fwd(qkv.to(DEV), None if bias is None else bias.to(DEV), B, N, H, D, 0.125)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref.numpy(), rtol=2e-2, atol=2e-2)
    err = (out.float().cpu().double() - o_ref).abs().max().item()
    assert err < 3e-2, f"forward max err {err}"


@pytest.mark.parametrize("scales,B,H", [(4, 8, 12), (3, 10, 12), (4, 33, 3), (4, 50, 12), (3, 43, 7)])
def test_attention_relpos_table_in_kernel(scales, B, H):
    """dm_attention_fwd_relpos forms the bias of a (scales, 8, 8) token cube from the table inside the kernel (table in LDS, x axis
    reversed, two floats per read; the last cases make a workgroup's run of (head, sample) units cross into the next head, where
    the table is reloaded); it must equal dm_attention_fwd on the gathered dense bias (same kernel family, same summation
    order: differences only from where bias / scale is rounded) and the fp64 reference.  Index rule: nets/ShfitScaleFormer.py:139-156."""
    from oracle import s2former as O
    ops = _ops()
    cube = (scales, 8, 8)
    N, D = 64 * scales, 64
    rng = np.random.default_rng(100 + scales)
    qkv = torch.from_numpy(rng.normal(size=(B, N, 3, H, D)).astype(np.float32)).to(torch.bfloat16)
    n_bins = (2 * scales - 1) * 225
    table = torch.from_numpy(rng.normal(size=(n_bins, H)).astype(np.float32))
    index = torch.from_numpy(O.relpos_index(cube).astype(np.int32))
    assert ops.relpos_inkernel(B, N, H, D, cube, torch.bfloat16)
    assert not ops.relpos_inkernel(B, N, H, D, (scales, 4, 16), torch.bfloat16) and not ops.relpos_inkernel(B, N, H, D, cube, torch.float32)
    bias = ops.relpos_bias_gather(table.to(DEV), index.to(DEV), N)
    o_dense, lse_dense = ops.attention_fwd(qkv.to(DEV), bias, B, N, H, D, 0.125)
    out, lse = ops.attention_fwd_relpos(qkv.to(DEV), table.to(DEV), cube, B, N, H, D, 0.125)
    o_ref, lse_ref = _attn_ref(qkv.double(), bias.cpu().double(), 0.125)
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref.numpy(), rtol=2e-2, atol=2e-2)
    assert (out.float().cpu().double() - o_ref).abs().max().item() < 3e-2
    assert (out.float() - o_dense.float()).abs().max().item() <= 2 ** -6          # one bf16 step of an O(1) output at most
    np.testing.assert_allclose(lse.cpu().numpy(), lse_dense.cpu().numpy(), rtol=0, atol=1e-4)
    with pytest.raises(ValueError):
        ops.attention_fwd_relpos(qkv.to(DEV), table[:-1].to(DEV), cube, B, N, H, D, 0.125)
    # backward: both passes read the table in LDS (dQ 8 waves; dK / dV 4 waves, which also accumulates the table-gradient slab on the
    # matrix pipe); no dense rows are passed
    dout = torch.from_numpy(rng.normal(size=(B, N, H * D)).astype(np.float32)).to(torch.bfloat16).to(DEV)
    bias, bias_t = ops.relpos_bias_gather(table.to(DEV), index.to(DEV), N, transposed=True)
    d_dense, slab_d, info_d = ops.attention_bwd(qkv.to(DEV), bias, o_dense, dout, lse_dense, B, N, H, D, 0.125, index.to(DEV), n_bins, bias_t=bias_t)
    d_tab, slab_t, info_t = ops.attention_bwd(qkv.to(DEV), None, o_dense, dout, lse_dense, B, N, H, D, 0.125, index.to(DEV), n_bins,
                                              table=table.to(DEV), cube=cube)
    dt_dense = torch.empty((n_bins, H), device=DEV); dt_tab = torch.empty((n_bins, H), device=DEV)
    ops.relpos_bias_scatter(slab_d, dt_dense, B, H, info_d, n_bins)
    ops.relpos_bias_scatter(slab_t, dt_tab, B, H, info_t, n_bins)
    q64 = qkv.double().requires_grad_(True)
    t64 = table.double().requires_grad_(True)
    b64 = t64[index.long().reshape(-1)].reshape(N, N, H).permute(2, 0, 1)
    o64, _ = _attn_ref(q64, b64, 0.125)
    o64.backward(dout.cpu().double())
    scale_g = q64.grad.abs().max().item()
    assert (d_tab.float().cpu().double() - q64.grad).abs().max().item() < 2e-2 * scale_g
    assert (d_tab.float() - d_dense.float()).abs().max().item() < 1e-2 * scale_g
    scale_t = t64.grad.abs().max().item()
    err_tab = (dt_tab.cpu().double() - t64.grad).abs().max().item() / scale_t
    err_dense = (dt_dense.cpu().double() - t64.grad).abs().max().item() / scale_t
    print(f"table gradient max error / max: table-in-LDS kernels {err_tab:.2e}, dense-row kernels {err_dense:.2e}")
    assert err_tab < 1e-2 and err_dense < 1e-2
    d_tab2, slab_t2, info_t2 = ops.attention_bwd(qkv.to(DEV), None, o_dense, dout, lse_dense, B, N, H, D, 0.125, index.to(DEV), n_bins,
                                                 table=table.to(DEV), cube=cube)
    dt_tab2 = torch.empty((n_bins, H), device=DEV)
    ops.relpos_bias_scatter(slab_t2, dt_tab2, B, H, info_t2, n_bins)
    assert torch.equal(d_tab, d_tab2) and torch.equal(dt_tab, dt_tab2)          # run-to-run deterministic (no atomics anywhere)


@pytest.mark.parametrize("N,scales,B,H", [(256, 4, 8, 12), (192, 3, 9, 12), (197, 0, 10, 12), (160, 0, 7, 5), (224, 0, 3, 4), (256, 0, 5, 3),
                                         (256, 4, 24, 12), (192, 3, 30, 12)])
def test_attention_split_forward_backward(N, scales, B, H):
    """dm_attention_split_fwd (the bf16x3 mode's attention): fp32 tensors, every product a split-bf16 triple on the matrix pipe, the
    bias from the table in LDS (scales > 0) or none (ragged N allowed; the last two cases have more (head, sample) units than CUs: the
    8-wave forward / dQ kernels then walk runs of units that cross head boundaries and reload the table).  Against fp64: the error must be at the fp32 kernels' level
    (1e-5-ish), nowhere near bf16's 1e-2.  Reference: nets/ShfitScaleFormer.py:119-133 / vit_model.py:119-133 in fp32."""
    from oracle import s2former as O
    ops = _ops()
    D = 64
    rng = np.random.default_rng(7 * N + scales)
    qkv = torch.from_numpy(rng.normal(size=(B, N, 3, H, D)).astype(np.float32))
    cube = (scales, 8, 8) if scales else None
    table = bias64 = None
    if scales:
        n_bins = (2 * scales - 1) * 225
        table = torch.from_numpy(rng.normal(size=(n_bins, H)).astype(np.float32))
        index = torch.from_numpy(O.relpos_index(cube).astype(np.int64))
        bias64 = table.double()[index.reshape(-1)].reshape(N, N, H).permute(2, 0, 1)
    assert ops.attention_split_ok(B, N, H, D, cube)
    o_ref, lse_ref = _attn_ref(qkv.double(), bias64, 0.125)
    out, lse, hi, lo = ops.attention_fwd_split(qkv.to(DEV), None if table is None else table.to(DEV), cube, B, N, H, D, 0.125)
    # the images: hi is the bf16 rounding, hi + lo recovers x to 2^-17 relative
    assert torch.equal(hi.cpu(), qkv.to(torch.bfloat16))
    assert ((hi.float() + lo.float()).cpu() - qkv).abs().max().item() <= 2.0 ** -16 * qkv.abs().max().item()
    err_o = (out.cpu().double() - o_ref).abs().max().item()
    err_l = (lse.cpu().double() - lse_ref).abs().max().item()
    print(f"split-bf16 attention N={N}: max |out - fp64| {err_o:.2e}, max |lse - fp64| {err_l:.2e}")
    assert err_o < 5e-5 and err_l < 5e-5
    o32, lse32 = ops.attention_fwd(qkv.to(DEV), None if bias64 is None else bias64.float().contiguous().to(DEV), B, N, H, D, 0.125)
    assert (out - o32).abs().max().item() < 5e-5
    # backward: dqkv (and the table gradient through the slab) against fp64 autograd
    dout = torch.from_numpy(rng.normal(size=(B, N, H * D)).astype(np.float32))
    q64 = qkv.double().requires_grad_(True)
    t64 = None if table is None else table.double().requires_grad_(True)
    b64 = None if table is None else t64[index.reshape(-1)].reshape(N, N, H).permute(2, 0, 1)
    o64, _ = _attn_ref(q64, b64, 0.125)
    o64.backward(dout.double())
    idx32 = None if table is None else index.to(torch.int32).to(DEV)
    dqkv, slab, info = ops.attention_bwd_split(hi, lo, None if table is None else table.to(DEV), cube, out, dout.to(DEV), lse, B, N, H, D, 0.125,
                                               idx32, 0 if table is None else n_bins)
    g = q64.grad
    err_g = (dqkv.cpu().double() - g).abs().max().item() / g.abs().max().item()
    print(f"split-bf16 attention backward N={N}: max |dqkv - fp64| / max |dqkv| = {err_g:.2e}")
    assert err_g < 2e-5
    # ABI 4, plane pairs on both sides: a pre-split qkv (what the qkv product writes with DM_BF16_PAIR) gives the same forward, and
    # the backward writes dqkv as the pair that splitting the fp32 result would give -- neither pass exists any more in the model
    if (B * N) % 8 == 0:
        qp = ops.split_planes(qkv.to(DEV).view(B * N, 3 * H * D))
        out2, lse2, hi2, lo2 = ops.attention_fwd_split(qp, None if table is None else table.to(DEV), cube, B, N, H, D, 0.125)
        assert torch.equal(out2, out) and torch.equal(lse2, lse) and torch.equal(hi2.reshape(hi.shape), hi) and torch.equal(lo2.reshape(lo.shape), lo)
        out3, _, _, _, op = ops.attention_fwd_split(qp, None if table is None else table.to(DEV), cube, B, N, H, D, 0.125, out_pair=True)
        assert torch.equal(out3, out) and torch.equal(op.t, ops.split_planes(out.view(B * N, H * D)).t)
        dq_pair, slab2, _ = ops.attention_bwd_split(hi, lo, None if table is None else table.to(DEV), cube, out, dout.to(DEV), lse, B, N, H, D, 0.125,
                                                    idx32, 0 if table is None else n_bins, pair=True)
        assert torch.equal(dq_pair.t, ops.split_planes(dqkv.view(B * N, 3 * H * D)).t)
        assert slab is None or torch.equal(slab2, slab)
    if table is not None:
        dt = torch.empty((n_bins, H), device=DEV)
        ops.relpos_bias_scatter(slab, dt, B, H, info, n_bins)
        err_t = (dt.cpu().double() - t64.grad).abs().max().item() / t64.grad.abs().max().item()
        print(f"   table gradient: max error / max = {err_t:.2e}")
        assert err_t < 2e-5


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("N,with_bias", [(12, True), (16, True), (48, True), (64, True), (192, True), (256, True), (197, False), (198, False), (100, True), (37, True), (250, True)])
def test_attention_forward_backward(mode, N, with_bias, B=3, H=4):
    ops = _ops()
    rng = np.random.default_rng(N)
    D = 64
    dt = DT[mode]
    qkv = torch.from_numpy(rng.normal(size=(B, N, 3, H, D)).astype(np.float32))
    qkv = qkv.to(dt).float()                     # operands exactly representable in the mode's dtype
    n_bins = 157
    table = torch.from_numpy(rng.normal(size=(n_bins, H)).astype(np.float32))
    index = torch.from_numpy(rng.integers(0, n_bins, size=(N, N)).astype(np.int32))
    dout = torch.from_numpy(rng.normal(size=(B, N, H * D)).astype(np.float32)).to(dt).float()
    scale = 0.125

    q64 = qkv.double().requires_grad_(True)
    t64 = table.double().requires_grad_(True)
    bias64 = t64[index.long().reshape(-1)].reshape(N, N, H).permute(2, 0, 1) if with_bias else None
    o_ref, lse_ref = _attn_ref(q64, bias64, scale)
    (o_ref * dout.double()).sum().backward()

    qd = qkv.to(DEV).to(dt)
    bias = bias_t = None
    if with_bias:
        bias, bias_t = ops.relpos_bias_gather(table.to(DEV), index.to(DEV), N, transposed=True)
        np.testing.assert_array_equal(bias.cpu().numpy(), bias64.detach().float().numpy())
        np.testing.assert_array_equal(bias_t.cpu().numpy(), bias64.detach().float().transpose(1, 2).numpy())
        if N in (48, 192):
            bias_t = None           # exercise the strided fallback of the key-major kernel too
    out, lse = ops.attention_fwd(qd, bias, B, N, H, D, scale)
    tol = 2e-5 if mode == "fp32" else 2e-2
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref.detach().numpy(), rtol=1e-4 if mode == "fp32" else 2e-2, atol=1e-4 if mode == "fp32" else 2e-2)
    err = (out.float().cpu().double() - o_ref.detach()).abs().max().item()
    assert err < tol, f"forward max err {err}"

    dqkv, slab, rows = ops.attention_bwd(qd, bias, out, dout.to(DEV).to(dt), lse, B, N, H, D, scale,
                                         index.to(DEV) if with_bias else None, n_bins if with_bias else 0, bias_t=bias_t)
    gq = q64.grad
    scale_ref = gq.abs().max().item()
    err = (dqkv.float().cpu().double() - gq).abs().max().item()
    assert err < (5e-5 if mode == "fp32" else 4e-2) * max(1.0, scale_ref), f"dqkv max err {err} (scale {scale_ref})"
    if with_bias:
        dtable = torch.empty((n_bins, H), device=DEV)
        ops.relpos_bias_scatter(slab, dtable, B, H, rows, n_bins)
        gt = t64.grad
        err = (dtable.cpu().double() - gt).abs().max().item()
        assert err < (1e-4 if mode == "fp32" else 6e-2) * max(1.0, gt.abs().max().item()), f"dtable max err {err}"


def test_partial_reduce_batch_equals_separate_reductions():
    """dm_layernorm_bwd_partials + ONE dm_partial_reduce_batch launch for many LayerNorm backward passes == dm_layernorm_bwd's own
    reduction, bit for bit (store and accumulate jobs, more than 32 jobs = two launches, different widths); then the deferred path of
    ops.layernorm_bwd inside a real backward pass (queued, flushed by the engine's end-of-backward callback)."""
    import ctypes as C
    ops = _ops()
    from deepmerge_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(21)
    jobs, keep = [], []
    want = []
    for i in range(35):
        rows, cols = int(rng.integers(1, 5000)), [768, 128, 1024, 64][i % 4]
        x = torch.from_numpy(rng.normal(size=(rows, cols)).astype(np.float32)).to(DEV)
        dy = torch.from_numpy(rng.normal(size=(rows, cols)).astype(np.float32)).to(DEV)
        if i % 3 == 0:
            dy = dy.to(torch.bfloat16)
        g = torch.from_numpy(rng.normal(size=cols).astype(np.float32)).to(DEV)
        _y, mean, rstd = ops.layernorm_fwd(x, g, torch.zeros_like(g), 1e-6, torch.float32)
        acc = bool(i % 2)
        dg0 = torch.from_numpy(rng.normal(size=cols).astype(np.float32)).to(DEV)
        db0 = torch.from_numpy(rng.normal(size=cols).astype(np.float32)).to(DEV)
        dg_ref, db_ref = dg0.clone(), db0.clone()
        dx_ref, _, _ = ops.layernorm_bwd(dy, x, g, mean, rstd, dgamma=dg_ref, dbeta=db_ref, accumulate=acc)
        dx = torch.empty_like(x)
        part = torch.empty(lib.dm_layernorm_bwd_partial_floats(cols), dtype=torch.float32, device=DEV)
        n_part = C.c_int32(0)
        _lib.check(lib.dm_layernorm_bwd_partials(dy.data_ptr(), ops._dt(dy), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None,
                                                 dx.data_ptr(), None, part.data_ptr(), rows, cols, C.byref(n_part), ops._stream()), "partials")
        assert torch.equal(dx, dx_ref) and n_part.value > 0
        dg, db = dg0.clone(), db0.clone()
        jobs.append((part.data_ptr(), dg.data_ptr(), db.data_ptr(), n_part.value, 2 * cols, cols, int(acc)))
        keep.append((part, dg, db))
        want.append((dg_ref, db_ref))
    items = (_lib.DmReduceItem * len(jobs))()
    for i, f in enumerate(jobs):
        items[i].partial, items[i].out0, items[i].out1, items[i].nrows, items[i].width, items[i].split, items[i].accumulate = f
    _lib.check(lib.dm_partial_reduce_batch(items, len(jobs), ops._stream()), "batch")
    for (part, dg, db), (dg_ref, db_ref) in zip(keep, want):
        assert torch.equal(dg, dg_ref) and torch.equal(db, db_ref)
    with pytest.raises(ValueError):
        _lib.check(lib.dm_partial_reduce_batch(items, 0, ops._stream()), "batch")

    # the deferred path inside a backward pass
    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, gy):
            assert ops._in_backward()
            Probe.out = ops.layernorm_bwd(gy.contiguous(), Probe.x, Probe.g, Probe.mean, Probe.rstd, dgamma=Probe.dg, dbeta=Probe.db, accumulate=False, defer=True)
            Probe.pending = len(ops._pending_reduce)
            return gy
    rows, cols = 777, 768
    Probe.x = torch.from_numpy(rng.normal(size=(rows, cols)).astype(np.float32)).to(DEV)
    Probe.g = torch.ones(cols, device=DEV)
    _y, Probe.mean, Probe.rstd = ops.layernorm_fwd(Probe.x, Probe.g, torch.zeros_like(Probe.g), 1e-6, torch.float32)
    Probe.dg, Probe.db = torch.full((cols,), float("nan"), device=DEV), torch.full((cols,), float("nan"), device=DEV)
    inp = torch.from_numpy(rng.normal(size=(rows, cols)).astype(np.float32)).to(DEV).requires_grad_(True)
    Probe.apply(inp).sum().backward()
    assert Probe.pending == 1 and not ops._pending_reduce          # queued inside the pass, flushed at its end
    dg_ref, db_ref = torch.empty(cols, device=DEV), torch.empty(cols, device=DEV)
    ops.layernorm_bwd(torch.ones_like(Probe.x), Probe.x, Probe.g, Probe.mean, Probe.rstd, dgamma=dg_ref, dbeta=db_ref)
    assert torch.equal(Probe.dg, dg_ref) and torch.equal(Probe.db, db_ref)


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(1, 768), (37, 768), (4096, 768), (50, 128), (9, 1024),
                                       (1, 1280), (514, 1280), (5000, 1028), (33, 8192)])      # > 1024 columns: dm_rows_wide.hip
def test_layernorm(rows, cols, out_dtype):
    ops = _ops()
    rng = np.random.default_rng(rows + cols)
    x = torch.from_numpy(rng.normal(2.0, 3.0, size=(rows, cols)).astype(np.float32))
    g = torch.from_numpy(rng.normal(1.0, 0.2, size=cols).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=cols).astype(np.float32))
    dy = torch.from_numpy(rng.normal(size=(rows, cols)).astype(np.float32)).to(out_dtype).float()
    dres = torch.from_numpy(rng.normal(size=(rows, cols)).astype(np.float32))
    xr, gr, br = x.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (cols,), gr, br, 1e-5)
    (yr * dy.double()).sum().backward()
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, out_dtype)
    tol = 2e-5 if out_dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(y.float().cpu().numpy(), yr.detach().numpy(), rtol=tol, atol=tol)
    g0 = torch.from_numpy(rng.normal(size=cols).astype(np.float32))
    dgam, dbet = g0.clone().to(DEV), g0.clone().to(DEV)
    dx, _, _ = ops.layernorm_bwd(dy.to(DEV).to(out_dtype), x.to(DEV), g.to(DEV), mean, rstd, dres=dres.to(DEV),
                                 dgamma=dgam, dbeta=dbet, accumulate=True)
    np.testing.assert_allclose(dx.cpu().numpy(), (xr.grad + dres.double()).numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(dgam.cpu().numpy(), (gr.grad + g0.double()).numpy(), rtol=2e-4, atol=2e-4 * max(1, rows) ** 0.5)
    np.testing.assert_allclose(dbet.cpu().numpy(), (br.grad + g0.double()).numpy(), rtol=2e-4, atol=2e-4 * max(1, rows) ** 0.5)


def test_pooling_patchify_colsum_cast():
    ops = _ops()
    from oracle import s2former as O
    rng = np.random.default_rng(9)
    B, S, side, Cc = 3, 4, 8, 768
    x = torch.from_numpy(rng.normal(size=(B, S * side * side, Cc)).astype(np.float32))
    xd = x.to(DEV).requires_grad_(True)
    y = ops.TokenPoolFn.apply(xd, S, side)
    want = O.token_pool2x2(x.double(), S, side)
    np.testing.assert_allclose(y.detach().cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
    go = torch.from_numpy(rng.normal(size=tuple(y.shape)).astype(np.float32))
    y.backward(go.to(DEV))
    xr = x.double().requires_grad_(True)
    (O.token_pool2x2(xr, S, side) * go.double()).sum().backward()
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-6, atol=1e-6)
    # group mean
    z = torch.from_numpy(rng.normal(size=(B, S * 4, Cc)).astype(np.float32))
    zd = z.to(DEV).requires_grad_(True)
    gm = ops.GroupMeanFn.apply(zd, 4)
    np.testing.assert_allclose(gm.detach().cpu().numpy(), z.reshape(B * S, 4, Cc).mean(1).numpy(), rtol=1e-6, atol=1e-6)
    gm.sum().backward()
    np.testing.assert_allclose(zd.grad.cpu().numpy(), np.full(z.shape, 0.25, np.float32))
    # patchify == unfold with (c, dy, dx) column order
    img = torch.from_numpy(rng.normal(size=(2, 4, 64, 64)).astype(np.float32))
    for p in (4, 8, 16, 32, 1, 2):
        cols = ops.patchify(img.to(DEV), p, torch.float32).cpu()
        want = torch.nn.functional.unfold(img, kernel_size=p, stride=p).transpose(1, 2).reshape(-1, 4 * p * p)
        assert torch.equal(cols, want), p
    img14 = torch.from_numpy(rng.normal(size=(3, 3, 56, 56)).astype(np.float32))      # ViT-H/14 patch side (element-wise kernel)
    for p, dt in ((14, torch.float32), (7, torch.bfloat16)):
        cols = ops.patchify(img14.to(DEV), p, dt).cpu()
        assert torch.equal(cols, torch.nn.functional.unfold(img14, kernel_size=p, stride=p).transpose(1, 2).reshape(-1, 3 * p * p).to(dt)), p
    colsb = ops.patchify(img.to(DEV), 8, torch.bfloat16).cpu()
    assert torch.equal(colsb, torch.nn.functional.unfold(img, kernel_size=8, stride=8).transpose(1, 2).reshape(-1, 256).bfloat16())
    # colsum
    X = torch.from_numpy(rng.normal(size=(1234, 768)).astype(np.float32))
    out = torch.ones(768, device=DEV)
    ops.colsum(X.to(DEV), out, accumulate=True)
    np.testing.assert_allclose(out.cpu().numpy(), 1.0 + X.double().sum(0).numpy(), rtol=1e-4, atol=1e-4)
    outb = torch.empty(768, device=DEV)
    ops.colsum(X.to(DEV).bfloat16(), outb)
    np.testing.assert_allclose(outb.cpu().numpy(), X.bfloat16().double().sum(0).numpy(), rtol=1e-4, atol=1e-3)
    # cast
    w = torch.from_numpy(rng.normal(size=(1001,)).astype(np.float32))
    assert torch.equal(ops.cast(w.to(DEV), torch.bfloat16).cpu(), w.bfloat16())


def test_contrastive_loss_and_adam():
    ops = _ops()
    from oracle import adam as OA
    from oracle import losses as OL
    from util import load_fx
    fx = load_fx("ops_s2former.npz")
    a = torch.from_numpy(fx["loss/a"]); b = torch.from_numpy(fx["loss/b"]); flag = torch.from_numpy(fx["loss/flag"])
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    loss = ops.ContrastiveLossFn.apply(ad, bd, flag.to(DEV), 1.0)
    (loss * 3.0).backward()
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    lr_ = OL.contrastive_loss(ar, br, flag, 1.0)
    (lr_ * 3.0).backward()
    assert abs(loss.item() - float(fx["loss_i64/value"])) < 1e-6 * abs(float(fx["loss_i64/value"])) + 1e-7
    np.testing.assert_allclose(ad.grad.cpu().numpy(), ar.grad.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-5, atol=1e-7)
    # Adam: 3 steps against the oracle (itself pinned to torch.optim.Adam by the golden fixtures)
    rng = np.random.default_rng(2)
    n = 100003
    p0 = torch.from_numpy(rng.normal(size=n).astype(np.float32))
    pr, mr, vr = p0.clone(), torch.zeros(n), torch.zeros(n)
    pd, md, vd = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    lp = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    for step in range(1, 4):
        g = torch.from_numpy((rng.normal(size=n) * 10.0 ** rng.uniform(-6, 0, size=n)).astype(np.float32))
        OA.adam_step(pr, g * 0.5, mr, vr, step, lr=1e-3)
        ops.adam_step(pd, g.to(DEV), md, vd, step, lr=1e-3, grad_scale=0.5, param_lp=lp)
    np.testing.assert_allclose(pd.cpu().numpy(), pr.numpy(), rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(md.cpu().numpy(), mr.numpy(), rtol=2e-6, atol=2e-8)
    np.testing.assert_allclose(vd.cpu().numpy(), vr.numpy(), rtol=4e-6, atol=1e-16)
    assert torch.equal(lp.cpu(), pd.cpu().bfloat16())


@pytest.mark.parametrize("unit", [True, False])
def test_stacked_pair_loss_equals_the_two_tensor_loss(unit):
    """The loss on the two halves of one [2B, D] matrix (ops.split_halves, what the Siamese encoders return): one gradient matrix,
    bit-identical to the two-tensor form; started with ops.unit_grad the stored gradients come back as they are."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    f = torch.randn(2 * 33, 100, generator=g)
    flag = (torch.arange(33) % 2).to(DEV)
    fd = f.to(DEV).requires_grad_(True)
    h = fd * 1.0                                           # (a non-leaf, as the encoder's output is)
    a, b = ops.split_halves(h)
    assert ops.stacked_halves(a, b) is h and ops.stacked_halves(b, a) is None and ops.stacked_halves(h[:33], h[33:]) is None
    loss = ops.contrastive_loss(a, b, flag, 1.0)
    assert type(loss.grad_fn).__name__ == "ContrastivePairLossFnBackward"
    if unit:
        loss.backward(ops.unit_grad(DEV))
    else:
        (loss * 0.7).backward()
    ad, bd = f[:33].to(DEV).requires_grad_(True), f[33:].to(DEV).requires_grad_(True)
    ref = ops.ContrastiveLossFn.apply(ad, bd, flag, 1.0)
    (ref * (1.0 if unit else 0.7)).backward()
    assert torch.equal(loss.detach(), ref.detach())
    assert torch.equal(fd.grad[:33], ad.grad) and torch.equal(fd.grad[33:], bd.grad)


@pytest.mark.parametrize("B,K", [(4, 11), (37, 100), (256, 1000)])
def test_cross_entropy_matches_torch(B, K):
    """dm_cross_entropy (MultiLoss / ClassLoss, Losses.py:52-53, :83-84) against torch's CPU float64 CrossEntropyLoss."""
    ops = _ops()
    rng = np.random.default_rng(B * 1000 + K)
    x = torch.from_numpy(rng.normal(0, 3, size=(B, K)).astype(np.float32))
    ti = torch.from_numpy(rng.integers(0, K, size=B))
    tp = torch.softmax(torch.from_numpy(rng.normal(size=(B, K)).astype(np.float32)), 1)
    for tgt in (ti, tp):
        xr = x.double().requires_grad_(True)
        want = torch.nn.functional.cross_entropy(xr, tgt.double() if tgt.dtype.is_floating_point else tgt)
        want.backward()
        xg = x.to(DEV).requires_grad_(True)
        got = ops.CrossEntropyFn.apply(xg, tgt.to(DEV))
        (got * 1.5).backward()
        assert abs(got.item() - want.item()) <= 2e-6 * abs(want.item())
        np.testing.assert_allclose(xg.grad.cpu().numpy(), 1.5 * xr.grad.numpy(), rtol=2e-5, atol=1e-8)
    with pytest.raises(IndexError):
        ops.CrossEntropyFn.apply(x.to(DEV), torch.full((B,), K, dtype=torch.int64, device=DEV))


@pytest.mark.parametrize("M,N,K", [(4096, 3072, 768), (4000, 3000, 704), (16384, 768, 64), (3968, 3080, 1088)])
def test_gemm256_pipeline_exact_and_epilogues(M, N, K):
    """Large bf16 forward products route to the 256x256 LDS-DMA pipeline (dm_gemm256.hip): exact on integer data,
    ragged M / N edges zero-filled by the descriptor, every fused epilogue."""
    ops = _ops()
    from deepmerge_amd._lib import DM_EPI_GELU, DM_NT
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K)
    a = torch.randint(-2, 3, (M, K), device=DEV, generator=g).to(torch.bfloat16)
    b = torch.randint(-2, 3, (N, K), device=DEV, generator=g).to(torch.bfloat16)
    ref = a.float() @ b.float().T
    out = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(DM_NT, a, b, out, M, N, K, lda=K, ldb=K, ldc=N)
    assert torch.equal(out, ref.to(torch.bfloat16))
    # bias + fp32 residual + fp32 output (proj / fc2 form)
    bias = torch.randint(-3, 4, (N,), device=DEV, generator=g).float()
    res = torch.randint(-5, 6, (M, N), device=DEV, generator=g).float()
    out32 = torch.empty((M, N), device=DEV)
    ops.gemm(DM_NT, a, b, out32, M, N, K, lda=K, ldb=K, ldc=N, bias=bias, residual=res)
    assert torch.equal(out32, ref + bias + res)
    # bias + GELU with the pre-activation saved (fc1 form)
    a2 = (a.float() * 0.125).to(torch.bfloat16)
    pre = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    h = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(DM_NT, a2, b, h, M, N, K, lda=K, ldb=K, ldc=N, bias=bias, epilogue=DM_EPI_GELU, aux=pre, ldaux=N)
    want_pre = ref * 0.125 + bias
    assert torch.equal(pre, want_pre.to(torch.bfloat16))
    want_h = torch.nn.functional.gelu(want_pre.double()).float()
    assert float((h.float() - want_h).abs().max()) <= 2.0 ** -7 * float(want_h.abs().max())


@pytest.mark.parametrize("layout", ["NT", "NN"])
@pytest.mark.parametrize("M,N,K", [(4096, 768, 3072), (1024, 768, 3072), (4096, 768, 2304), (1000, 776, 1600), (4096, 768, 4096)])
def test_gemm_forward_k_slices_exact_and_epilogues(layout, M, N, K):
    """Round 4: forward / dgrad products with a long contraction and a small output (the 4096- / 1024-token stages' fc2 forward
    nets/ShfitScaleFormer.py:55, fc1 / qkv dgrad, the patch embed :28-37) run as K slices: fp32 partial tiles + splitk_epilogue_kernel.
    Exact on integer data for every fused epilogue, ragged M / N, grouped rows; run-to-run identical; the plan reports its workspace."""
    ops = _ops()
    from deepmerge_amd import _lib
    from deepmerge_amd._lib import DM_EPI_GELU_GRAD, DM_EPI_MUL, DM_NN, DM_NT
    lay = DM_NT if layout == "NT" else DM_NN
    assert _lib.lib().dm_gemm_workspace_bytes(lay, M, N, K) >= 2 * M * N * 4          # a slab of >= 2 slices: the split is planned
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K)
    a = torch.randint(-2, 3, (M, K), device=DEV, generator=g).to(torch.bfloat16)
    if layout == "NT":
        b = torch.randint(-2, 3, (N, K), device=DEV, generator=g).to(torch.bfloat16)
        ref, ldb = a.float() @ b.float().T, K
    else:
        b = torch.randint(-2, 3, (K, N), device=DEV, generator=g).to(torch.bfloat16)
        ref, ldb = a.float() @ b.float(), N
    outs = []
    for _ in range(2):
        out = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
        ops.gemm(lay, a, b, out, M, N, K, lda=K, ldb=ldb, ldc=N)
        outs.append(out)
    assert torch.equal(outs[0], ref.to(torch.bfloat16)) and torch.equal(outs[0], outs[1])
    bias = torch.randint(-3, 4, (N,), device=DEV, generator=g).float()
    res = torch.randint(-5, 6, (M, N), device=DEV, generator=g).float()
    out32 = torch.empty((M, N), device=DEV)
    ops.gemm(lay, a, b, out32, M, N, K, lda=K, ldb=ldb, ldc=N, bias=bias, residual=res)      # fc2 forward form
    assert torch.equal(out32, ref + bias + res)
    aux = torch.randint(-2, 3, (M, N), device=DEV, generator=g).to(torch.bfloat16)
    outm = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(lay, a, b, outm, M, N, K, lda=K, ldb=ldb, ldc=N, epilogue=DM_EPI_MUL, aux=aux, ldaux=N)      # dgrad x saved GELU' form
    assert torch.equal(outm, (ref * aux.float()).to(torch.bfloat16))
    a2 = (a.float() * 0.125).to(torch.bfloat16)
    d = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    h = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(lay, a2, b, h, M, N, K, lda=K, ldb=ldb, ldc=N, bias=bias, epilogue=DM_EPI_GELU_GRAD, aux=d, ldaux=N)
    pre = (ref * 0.125 + bias).double()
    want_h = torch.nn.functional.gelu(pre).float()
    want_d = (0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5).float()
    assert float((h.float() - want_h).abs().max()) <= 2.0 ** -7 * float(want_h.abs().max())
    assert float((d.float() - want_d).abs().max()) <= 2.0 ** -7
    if M % 64 == 0 and layout == "NT":          # grouped rows: tokens written at their offset of a wider cube (patch-embed form, :869-882)
        T, S = 64, 3
        cube = torch.zeros((M // T, S * T, N), device=DEV)
        ops.gemm(lay, a, b, cube.view(-1)[T * N:], M, N, K, lda=K, ldb=ldb, ldc=N, bias=bias, rows_per_group=T, group_stride=S * T * N)
        assert torch.equal(cube[:, T:2 * T, :].reshape(M, N), ref + bias) and float(cube[:, :T].abs().max()) == 0 and float(cube[:, 2 * T:].abs().max()) == 0


@pytest.mark.parametrize("layout,M,N,K", [("NT", 16384, 3072, 768), ("NN", 16384, 3072, 768), ("NT", 9000, 2304, 768), ("NN", 8200, 768, 2304),
                                          ("NT", 16384, 768, 128)])
def test_gemm256_persistent_pipeline_exact_and_epilogues(layout, M, N, K, monkeypatch):
    """Round 4: the 256x256 LDS-DMA pipeline as a PERSISTENT kernel (several tiles per workgroup, the next tile's first K tiles staged
    under the current tile's last ones, epilogue between two K tiles through a wave-private swizzled block): exact on integer data
    for the fused epilogues of the 16384-token stage (nets/ShfitScaleFormer.py:53-56, :119, :134), ragged M, workgroups with
    different tile counts, run-to-run identical."""
    ops = _ops()
    from deepmerge_amd._lib import DM_EPI_GELU_GRAD, DM_EPI_MUL, DM_NN, DM_NT
    monkeypatch.setenv("DM_GEMM_256", "2")            # every legal product on the pipeline ...
    monkeypatch.setenv("DM_GEMM_256P", "1")           # ... in its persistent form where tiles > CUs (off by default: no faster)
    monkeypatch.setenv("DM_GEMM_W4", "0"); monkeypatch.setenv("DM_GEMM_RING", "0")
    lay = DM_NT if layout == "NT" else DM_NN
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K)
    a = torch.randint(-2, 3, (M, K), device=DEV, generator=g).to(torch.bfloat16)
    if layout == "NT":
        b = torch.randint(-2, 3, (N, K), device=DEV, generator=g).to(torch.bfloat16)
        ref, ldb = a.float() @ b.float().T, K
    else:
        b = torch.randint(-2, 3, (K, N), device=DEV, generator=g).to(torch.bfloat16)
        ref, ldb = a.float() @ b.float(), N
    outs = []
    for _ in range(2):
        out = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
        ops.gemm(lay, a, b, out, M, N, K, lda=K, ldb=ldb, ldc=N)
        outs.append(out)
    assert torch.equal(outs[0], ref.to(torch.bfloat16)) and torch.equal(outs[0], outs[1])
    bias = torch.randint(-3, 4, (N,), device=DEV, generator=g).float()
    res = torch.randint(-5, 6, (M, N), device=DEV, generator=g).float()
    out32 = torch.empty((M, N), device=DEV)
    ops.gemm(lay, a, b, out32, M, N, K, lda=K, ldb=ldb, ldc=N, bias=bias, residual=res)
    assert torch.equal(out32, ref + bias + res)
    outb = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(lay, a, b, outb, M, N, K, lda=K, ldb=ldb, ldc=N, bias=bias)
    assert torch.equal(outb, (ref + bias).to(torch.bfloat16))
    aux = torch.randint(-2, 3, (M, N), device=DEV, generator=g).to(torch.bfloat16)
    outm = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(lay, a, b, outm, M, N, K, lda=K, ldb=ldb, ldc=N, epilogue=DM_EPI_MUL, aux=aux, ldaux=N)
    assert torch.equal(outm, (ref * aux.float()).to(torch.bfloat16))
    a2 = (a.float() * 0.125).to(torch.bfloat16)
    d = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    h = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(lay, a2, b, h, M, N, K, lda=K, ldb=ldb, ldc=N, bias=bias, epilogue=DM_EPI_GELU_GRAD, aux=d, ldaux=N)
    pre = (ref * 0.125 + bias).double()
    want_h = torch.nn.functional.gelu(pre).float()
    want_d = (0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5).float()
    assert float((h.float() - want_h).abs().max()) <= 2.0 ** -7 * float(want_h.abs().max())
    assert float((d.float() - want_d).abs().max()) <= 2.0 ** -7


def test_gemm256_pipeline_wgrad_and_dgrad_layouts():
    """The m-contiguous images of the 256x256 pipeline (hardware-transposed LDS reads): TN with split-K + accumulate
    (wgrad of a 16384-token stage) and NN with many tiles, exact on integer data and run-to-run identical."""
    ops = _ops()
    from deepmerge_amd._lib import DM_NN, DM_TN
    g = torch.Generator(device=DEV); g.manual_seed(9)
    K, M, N = 16384, 776, 3072                                   # dy [K, M], x [K, N]; M is ragged against the 256 tile
    dy = torch.randint(-1, 2, (K, M), device=DEV, generator=g).to(torch.bfloat16)
    x = torch.randint(-1, 2, (K, N), device=DEV, generator=g).to(torch.bfloat16)
    g0 = torch.randint(-4, 5, (M, N), device=DEV, generator=g).float()
    want = g0 + dy.float().T @ x.float()
    outs = []
    for _ in range(2):
        G = g0.clone()
        ops.gemm(DM_TN, dy, x, G, M, N, K, lda=M, ldb=N, ldc=N, accumulate=True)
        outs.append(G)
    assert torch.equal(outs[0], want) and torch.equal(outs[0], outs[1])
    Mr, Kd, Nd = 16384, 256, 4096                                # dgrad form: dx[Mr, Nd] = dy[Mr, Kd] @ W[Kd, Nd]
    a = torch.randint(-2, 3, (Mr, Kd), device=DEV, generator=g).to(torch.bfloat16)
    w = torch.randint(-2, 3, (Kd, Nd), device=DEV, generator=g).to(torch.bfloat16)
    out = torch.empty((Mr, Nd), device=DEV, dtype=torch.bfloat16)
    ops.gemm(DM_NN, a, w, out, Mr, Nd, Kd, lda=Kd, ldb=Nd, ldc=Nd)
    assert torch.equal(out, (a.float() @ w.float()).to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K", [(768, 3072, 16384), (2304, 768, 16384), (776, 3072, 16384), (768, 768, 4096), (104, 768, 640),
                                   (768, 768, 1024), (3072, 768, 1024), (768, 256, 4096), (768, 64, 4096), (40, 24, 72), (200, 136, 1000)])
def test_gemm_wgrad_with_fused_column_sums(M, N, K):
    """colsum_a: the bias gradient rides on the wgrad (ones-fragment MFMAs on the 256x256, 4-wave and 64x64 kernels, the column-sum
    kernels otherwise); exact on integer data, accumulate honoured separately for dW and db.  The second row of shapes takes the
    64x64 tiles (1024-token stage, patch embeds, ragged M / N / K tails)."""
    ops = _ops()
    from deepmerge_amd._lib import DM_TN
    g = torch.Generator(device=DEV); g.manual_seed(M + N)
    dy = torch.randint(-1, 2, (K, M), device=DEV, generator=g).to(torch.bfloat16)
    x = torch.randint(-1, 2, (K, N), device=DEV, generator=g).to(torch.bfloat16)
    db0 = torch.randint(-3, 4, (M,), device=DEV, generator=g).float()
    for acc_b in (False, True):
        dw = torch.empty((M, N), device=DEV)
        db = db0.clone()
        ops.gemm(DM_TN, dy, x, dw, M, N, K, lda=M, ldb=N, ldc=N, colsum_out=db, colsum_accumulate=acc_b)
        assert torch.equal(dw, dy.float().T @ x.float())
        want = dy.float().sum(0) + (db0 if acc_b else 0)
        assert torch.equal(db, want)
    with pytest.raises(ValueError):
        ops.gemm(DM_TN, dy, x, dw, M, N, K, lda=M, ldb=N, ldc=N, colsum_out=torch.empty(M - 1, device=DEV))


@pytest.mark.parametrize("T,acc", [(1024, False), (4096, False), (4096, True), (1152, True), (8192, False), (16384, True), (9216, False), (50432, False)])
def test_gemm_grouped_block_weight_gradients(T, acc, monkeypatch, sk=False):
    if sk:
        monkeypatch.setenv("DM_GEMM_GROUPED", "3")
    """dm_gemm_grouped: the four weight gradients of a block (dW = dy^T x for qkv / proj / fc1 / fc2, nets/ShfitScaleFormer.py:35, :119,
    :134 under autograd) in ONE launch of the 4-wave kernel -- 144 tiles, one K slice each, up to 12288 tokens; the separate calls
    beyond (16384 = the headline's stage 0, 50432 = ViT-B/16 at 128 pairs) -- the bias gradients from the same launch.  Exact on integer
    data (so the order of the fp32 additions does not matter): equal to the separate dm_gemm calls bit for bit, with `accumulate` /
    `colsum_accumulate` honoured.  (test_gemm_grouped_stream_k_*: the same groups through the stream-K form.)"""
    ops = _ops()
    from deepmerge_amd._lib import DM_TN
    g = torch.Generator(device=DEV); g.manual_seed(T + int(acc))
    C, H = 768, 3072
    calls, want = [], []
    for i, (m, n) in enumerate([(C, H), (H, C), (C, C), (3 * C, C)]):          # the order BlockFn.backward issues them in
        dy = torch.randint(-1, 2, (T, m), device=DEV, generator=g).to(torch.bfloat16)
        x = torch.randint(-1, 2, (T, n), device=DEV, generator=g).to(torch.bfloat16)
        dw0 = torch.randint(-3, 4, (m, n), device=DEV, generator=g).float()
        db0 = torch.randint(-3, 4, (m,), device=DEV, generator=g).float()
        dw, db = dw0.clone(), db0.clone()
        calls.append(((DM_TN, dy, x, dw, m, n, T), dict(lda=m, ldb=n, ldc=n, accumulate=acc, colsum_out=db, colsum_accumulate=acc)))
        want.append((dy.float().T @ x.float() + (dw0 if acc else 0), dy.float().sum(0) + (db0 if acc else 0), dw, db))
    ops.gemm_grouped(calls)
    for k, (w_dw, w_db, dw, db) in enumerate(want):
        assert torch.equal(dw, w_dw), f"product {k}: weight gradient"
        assert torch.equal(db, w_db), f"product {k}: bias gradient"


@pytest.mark.parametrize("T,acc,sk", [(1024, False, False), (4096, True, False), (8192, False, True)])
def test_gemm_grouped_plane_pair_weight_gradients(T, acc, sk, monkeypatch):
    """The grouped launch on hi / lo plane pairs (the "bf16x3" products of the tolerance mode): dW = hi^T hi + hi^T lo + lo^T hi and
    db = colsum(hi) + colsum(lo), exact on data whose pieces, products and partial sums are exactly representable.  sk: the stream-K form."""
    if sk:
        monkeypatch.setenv("DM_GEMM_GROUPED", "3")
    ops = _ops()
    from deepmerge_amd._lib import DM_TN
    g = torch.Generator(device=DEV); g.manual_seed(3 * T + int(acc))
    C, H = 768, 3072

    def mat(shape):      # +-(i + j / 1024), i in {1, 2}: hi = +-i, lo = +-j / 1024; every product kept and every partial sum fits 24 bits
        sign = torch.randint(0, 2, shape, device=DEV, generator=g).float() * 2 - 1
        return sign * (torch.randint(1, 3, shape, device=DEV, generator=g).float() + torch.randint(0, 4, shape, device=DEV, generator=g).float() / 1024)
    calls, want = [], []
    for (m, n) in [(C, H), (H, C), (C, C), (3 * C, C)]:
        dy, x = ops.split_planes(mat((T, m))), ops.split_planes(mat((T, n)))
        dw0 = torch.randint(-3, 4, (m, n), device=DEV, generator=g).float()
        db0 = torch.randint(-3, 4, (m,), device=DEV, generator=g).float()
        dw, db = dw0.clone(), db0.clone()
        calls.append(((DM_TN, dy, x, dw, m, n, T), dict(lda=m, ldb=n, ldc=n, accumulate=acc, colsum_out=db, colsum_accumulate=acc)))
        ah, al, bh, bl = dy.t[0].double(), dy.t[1].double(), x.t[0].double(), x.t[1].double()
        w = (ah.T @ bh + ah.T @ bl + al.T @ bh).float()
        want.append((w + (dw0 if acc else 0), (ah.sum(0) + al.sum(0)).float() + (db0 if acc else 0), dw, db))
    ops.gemm_grouped(calls)
    for k, (w_dw, w_db, dw, db) in enumerate(want):
        assert torch.equal(dw, w_dw), f"product {k}: weight gradient"
        assert torch.equal(db, w_db), f"product {k}: bias gradient"


@pytest.mark.parametrize("T,acc", [(4096, False), (8192, False), (16384, True), (50432, False)])
def test_gemm_grouped_stream_k_block_weight_gradients(T, acc, monkeypatch):
    """The block's four weight gradients through the stream-K form (DM_GEMM_GROUPED=3: equal runs of K steps per workgroup across tile
    boundaries, partial pieces summed in workgroup order by one fix-up launch): exact."""
    test_gemm_grouped_block_weight_gradients(T, acc, monkeypatch, sk=True)


@pytest.mark.parametrize("shapes,T", [([(768, 768)], 16384), ([(768, 768), (2304, 768)], 9216), ([(256, 192), (3072, 768), (768, 3072)], 12288),
                                      ([(3072, 3072), (768, 768)], 8192)])
def test_gemm_grouped_stream_k_uneven_groups(shapes, T, monkeypatch):
    """The stream-K form on groups whose products differ in tile count (1 .. 192 tiles: the workgroups are dealt in proportion to the K
    steps), a single product, and a product with more tiles than its share of workgroups (pieces longer than a tile: three slots)."""
    ops = _ops()
    from deepmerge_amd._lib import DM_TN
    monkeypatch.setenv("DM_GEMM_GROUPED", "3")
    g = torch.Generator(device=DEV); g.manual_seed(T + len(shapes))
    calls, want = [], []
    for (m, n) in shapes:
        dy = torch.randint(-1, 2, (T, m), device=DEV, generator=g).to(torch.bfloat16)
        x = torch.randint(-1, 2, (T, n), device=DEV, generator=g).to(torch.bfloat16)
        dw, db = torch.empty((m, n), device=DEV), torch.empty((m,), device=DEV)
        calls.append(((DM_TN, dy, x, dw, m, n, T), dict(lda=m, ldb=n, ldc=n, colsum_out=db)))
        want.append((dy.float().T @ x.float(), dy.float().sum(0), dw, db))
    ops.gemm_grouped(calls)
    for k, (w_dw, w_db, dw, db) in enumerate(want):
        assert torch.equal(dw, w_dw), f"product {k}: weight gradient"
        assert torch.equal(db, w_db), f"product {k}: bias gradient"


@pytest.mark.parametrize("shapes,T,acc,force", [([(768, 768), (2304, 768)], 16384, False, False), ([(768, 768), (2304, 768)], 50432, True, False),
                                                  ([(768, 3072), (3072, 768), (768, 768), (2304, 768)], 4096, False, True)])
def test_gemm_grouped_sliced_form(shapes, T, acc, force, monkeypatch):
    """dm_gemm_grouped's third form: the products of the group share ONE launch with the same K slices -- the proj gradient (12 tiles) next
    to the qkv gradient (36) of a 16384-token block: 5 slices on 240 workgroups instead of 16 + 7 on two launches -- partial tiles to each
    product's own slab, summed in slice order by its own reduction.  Exact on integer data; column sums and `accumulate` honoured."""
    ops = _ops()
    from deepmerge_amd._lib import DM_TN
    if force:
        monkeypatch.setenv("DM_GEMM_GROUPED", "4")
    g = torch.Generator(device=DEV); g.manual_seed(T + len(shapes) + int(acc))
    calls, want = [], []
    for (m, n) in shapes:
        dy = torch.randint(-1, 2, (T, m), device=DEV, generator=g).to(torch.bfloat16)
        x = torch.randint(-1, 2, (T, n), device=DEV, generator=g).to(torch.bfloat16)
        dw0 = torch.randint(-3, 4, (m, n), device=DEV, generator=g).float()
        db0 = torch.randint(-3, 4, (m,), device=DEV, generator=g).float()
        dw, db = dw0.clone(), db0.clone()
        calls.append(((DM_TN, dy, x, dw, m, n, T), dict(lda=m, ldb=n, ldc=n, accumulate=acc, colsum_out=db, colsum_accumulate=acc)))
        want.append((dy.float().T @ x.float() + (dw0 if acc else 0), dy.float().sum(0) + (db0 if acc else 0), dw, db))
    ops.gemm_grouped(calls)
    for k, (w_dw, w_db, dw, db) in enumerate(want):
        assert torch.equal(dw, w_dw), f"product {k}: weight gradient"
        assert torch.equal(db, w_db), f"product {k}: bias gradient"


def test_gemm_grouped_falls_back_to_separate_calls():
    """Groups the one-launch form does not describe -- a ragged member, a single product, a forward product, mixed `accumulate`, fp32
    operands -- run as the separate calls would; errors of a member surface as dm_gemm's."""
    ops = _ops()
    from deepmerge_amd._lib import DM_NT, DM_TN
    g = torch.Generator(device=DEV); g.manual_seed(11)

    def tn(m, n, K, acc=False, dt=torch.bfloat16):
        dy = torch.randint(-1, 2, (K, m), device=DEV, generator=g).to(dt)
        x = torch.randint(-1, 2, (K, n), device=DEV, generator=g).to(dt)
        dw0 = torch.randint(-3, 4, (m, n), device=DEV, generator=g).float()
        dw = dw0.clone()
        return ((DM_TN, dy, x, dw, m, n, K), dict(lda=m, ldb=n, ldc=n, accumulate=acc)), dy.float().T @ x.float() + (dw0 if acc else 0), dw
    groups = [[tn(768, 768, 1024), tn(104, 768, 640)],                       # ragged member
              [tn(768, 768, 1024)],                                          # one product
              [tn(768, 768, 1024, acc=True), tn(768, 3072, 1024)],           # mixed accumulate
              [tn(768, 768, 1024, dt=torch.float32), tn(256, 192, 256, dt=torch.float32)],      # fp32 operands
              [tn(768, 3072, 16384, acc=True), tn(3072, 768, 16384)]]       # long contraction: the rule keeps the sliced launches
    for grp in groups:
        ops.gemm_grouped([c for c, _, _ in grp])
        for _, want, dw in grp:
            assert torch.equal(dw, want)
    a = torch.randint(-2, 3, (512, 768), device=DEV, generator=g).to(torch.bfloat16)
    w = torch.randint(-2, 3, (768, 768), device=DEV, generator=g).to(torch.bfloat16)
    out = torch.empty((512, 768), device=DEV, dtype=torch.bfloat16)
    (c, want, dw) = tn(768, 768, 1024)
    ops.gemm_grouped([((DM_NT, a, w, out, 512, 768, 768), dict(lda=768, ldb=768, ldc=768)), c])
    assert torch.equal(out, (a.float() @ w.float().T).to(torch.bfloat16)) and torch.equal(dw, want)
    with pytest.raises(ValueError):
        ops.gemm_grouped([c, ((DM_TN, a, w, out, 768, 768, 512), dict(lda=768, ldb=768, ldc=768, accumulate=True))])      # accumulate needs fp32 C


@pytest.mark.parametrize("layout,M,N,K", [("NT", 64, 100, 3840), ("NN", 64, 100, 3840), ("TN", 100, 52, 2050), ("NT", 7, 19, 1024), ("NT", 128, 256, 1500)])
def test_gemm_skinny_fp32_k_slices(layout, M, N, K, monkeypatch):
    """The generic fp32 path cuts the contraction of skinny products (the 100-wide head over 3840 pooled features,
    nets/ShfitScaleFormer.py:958-975: 28 output tiles) into K slices over gridDim.z and sums them in slice order in a second launch:
    exact on integer data, every epilogue, ragged K, and the same bits as the unsliced kernel on such data."""
    ops = _ops()
    from deepmerge_amd._lib import DM_EPI_GELU, DM_EPI_NONE, DM_NN, DM_NT, DM_TN
    lay = {"NT": DM_NT, "NN": DM_NN, "TN": DM_TN}[layout]
    g = torch.Generator(device=DEV); g.manual_seed(M * N + K)
    A = torch.randint(-2, 3, (K, M) if layout == "TN" else (M, K), device=DEV, generator=g).float()
    B = torch.randint(-2, 3, (N, K) if layout == "NT" else (K, N), device=DEV, generator=g).float()
    bias = torch.randint(-3, 4, (N,), device=DEV, generator=g).float()
    res = torch.randint(-3, 4, (M, N), device=DEV, generator=g).float()
    ref = (A.t() if layout == "TN" else A) @ (B.t() if layout == "NT" else B) + bias + res
    outs = []
    for on in ("1", "0"):
        monkeypatch.setenv("DM_GEMM_SKINNY", on)      # (read once per process: the first value wins; both runs are checked against the exact result)
        C_ = torch.empty(M, N, device=DEV)
        ops.gemm(lay, A, B, C_, M, N, K, bias=bias, residual=res)
        assert torch.equal(C_, ref)
        Cg = torch.empty(M, N, device=DEV)
        aux = torch.empty(M, N, device=DEV)
        ops.gemm(lay, A * 0.25, B * 0.25, Cg, M, N, K, bias=bias, epilogue=DM_EPI_GELU, aux=aux, ldaux=N)
        pre = ref - res - bias
        assert torch.equal(aux, pre / 16 + bias)
        assert (Cg - torch.nn.functional.gelu(aux)).abs().max().item() < 1e-5
        acc = torch.ones(M, N, device=DEV)
        ops.gemm(lay, A, B, acc, M, N, K, accumulate=True)
        assert torch.equal(acc, ref - bias - res + 1)


def test_gemm_user_split_k_is_bounds_checked():
    """A caller-chosen split_k above the automatic one used to write past the split-K slab (GPU fault, round 2).  The library now
    refuses a workspace that is too small, and ops.gemm sizes the workspace for the requested slice count."""
    ops = _ops()
    from deepmerge_amd import _lib
    from deepmerge_amd._lib import DM_BF16, DM_F32, DM_TN, DmGemmArgs
    import ctypes as C
    rng = np.random.default_rng(2)
    T, M, N = 4096, 768, 768
    dy, x = _ints(rng, (T, M), -1, 2).to(DEV).to(torch.bfloat16), _ints(rng, (T, N), -1, 2).to(DEV).to(torch.bfloat16)
    want = dy.double().T @ x.double()
    for split in (2, 3, 8):
        G = torch.zeros((M, N), device=DEV)
        ops.gemm(DM_TN, dy, x, G, M, N, T, lda=M, ldb=N, ldc=N, split_k=split)
        assert torch.equal(G.double(), want), split
    a = DmGemmArgs()
    a.layout, a.ab_dtype, a.c_dtype, a.aux_dtype = DM_TN, DM_BF16, DM_F32, DM_F32
    a.M, a.N, a.K, a.split_k = M, N, T, 8
    G = torch.zeros((M, N), device=DEV)
    small = torch.empty(2 * M * N * 4, dtype=torch.uint8, device=DEV)
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = dy.data_ptr(), M, x.data_ptr(), N, G.data_ptr(), N
    a.workspace, a.workspace_bytes = small.data_ptr(), small.numel()
    rc = _lib.lib().dm_gemm(C.byref(a), torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"split_k" in _lib.lib().dm_last_error()


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("B,N,H,D,with_bias", [(2, 257, 3, 80, False), (1, 300, 2, 32, True), (2, 70, 2, 128, False), (1, 64, 2, 48, True)])
def test_attention_generic_shapes(mode, B, N, H, D, with_bias):
    """Head dims other than 64 / more than 256 tokens (ViT-H/14's 80 x 257 among them) run the generic fp32 kernel family
    (dm_attention_generic.hip): forward, lse and the three gradients against a float64 reference."""
    ops = _ops()
    dt = DT[mode]
    g = torch.Generator().manual_seed(N + D)
    qkv = (torch.randn(B, N, 3, H, D, generator=g) * 0.7).to(dt)
    bias = torch.randn(H, N, N, generator=g) * 0.5 if with_bias else None
    dout = torch.randn(B, N, H * D, generator=g).to(dt)
    scale = D ** -0.5
    q, k, v = [qkv.double()[:, :, i].permute(0, 2, 1, 3).clone().requires_grad_(True) for i in range(3)]      # [B,H,N,D]
    s = q @ k.transpose(-1, -2) * scale
    if bias is not None:
        s = s + bias.double()[None]
    p = torch.softmax(s, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B, N, H * D)
    (o * dout.double()).sum().backward()
    out, lse = ops.attention_fwd(qkv.to(DEV), None if bias is None else bias.to(DEV), B, N, H, D, scale)
    tol = 2e-5 if mode == "fp32" else 2e-2
    np.testing.assert_allclose(out.float().cpu().double().numpy(), o.detach().numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(lse.cpu().double().numpy(), torch.logsumexp(s, -1).detach().numpy(), rtol=1e-5, atol=1e-5)
    # backward from the reference's own (rounded) forward output so both sides differentiate the same function
    dqkv, slab, _ = ops.attention_bwd(qkv.to(DEV), None if bias is None else bias.to(DEV), out, dout.to(DEV), lse, B, N, H, D, scale)
    assert slab is None
    want = torch.stack([t.grad.permute(0, 2, 1, 3) for t in (q, k, v)], 2)                                      # [B,N,3,H,D]
    got = dqkv.float().cpu().double()
    err = float((got - want).norm() / want.norm())
    assert err < (1e-4 if mode == "fp32" else 2e-2), err
    if with_bias:
        with pytest.raises(ValueError):
            ops.attention_bwd(qkv.to(DEV), bias.to(DEV), out, dout.to(DEV), lse, B, N, H, D, scale, torch.zeros(N, N, dtype=torch.int32, device=DEV), 10)


# ---- split-bf16 ("bf16x3") products ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols,ld", [(37, 24, 32), (37, 20, 32), (300, 768, 772), (9, 264, 264), (70000, 64, 64)])
@pytest.mark.parametrize("stack,pattern", [(0, 0b100), (0, 0b010), (1, 0b100), (1, 0b010)])
def test_split_bf16_images(stack, pattern, rows, cols, ld):
    """dm_split_bf16: hi = bf16(x), lo = bf16(x - hi), three pieces side by side (stack 0) or one under the other (stack 1).  Shapes
    with cols % 8 == 0 take the 8-columns-per-thread kernel (rows walked by the grid's second dimension), the others the generic one."""
    from deepmerge_amd import _lib
    torch.manual_seed(3)
    src = (torch.randn(rows, ld, device=DEV) * torch.logspace(-3, 3, ld, device=DEV)).contiguous()
    dst = torch.full((3 * rows * cols,), float("nan"), dtype=torch.bfloat16, device=DEV)
    rc = _lib.lib().dm_split_bf16(src.data_ptr(), ld, rows, cols, dst.data_ptr(), stack, pattern, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    x = src[:, :cols]
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    pieces = [lo if (pattern >> j) & 1 else hi for j in range(3)]
    want = torch.cat(pieces, dim=0 if stack else 1)
    got = dst.view(3 * rows, cols) if stack else dst.view(rows, 3 * cols)
    assert torch.equal(got, want)
    # the pair carries x to ~2^-17 relative
    assert ((hi.float() + lo.float() - x).abs() <= x.abs() * 2.0 ** -16).all()
    assert _lib.lib().dm_split_bf16(src.data_ptr(), ld, rows, 22, dst.data_ptr(), stack, pattern, None) != 0     # cols % 4


@pytest.mark.parametrize("rows,cols,ld,stack", [(300, 768, 772, 1), (16384, 3072, 3072, 1), (37, 24, 32, 0), (70000, 64, 64, 1)])
def test_split_bf16_with_column_sums(rows, cols, ld, stack):
    """dm_split_bf16_colsum: the split image of dm_split_bf16, bit for bit, plus partial column sums of the fp32 source whose
    row-ordered reduction (dm_partial_reduce_batch) equals the float64 column sums to fp32 accuracy -- run-to-run identical."""
    import ctypes as C
    from deepmerge_amd import _lib
    lib = _lib.lib()
    torch.manual_seed(5)
    src = torch.randn(rows, ld, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    want = torch.empty(3 * rows * cols, dtype=torch.bfloat16, device=DEV)
    assert lib.dm_split_bf16(src.data_ptr(), ld, rows, cols, want.data_ptr(), stack, 0b100, st) == 0
    outs = []
    for _ in range(2):
        dst = torch.empty_like(want)
        part = torch.empty(lib.dm_split_colsum_partial_floats(rows, cols), device=DEV)
        n = C.c_int32(0)
        assert lib.dm_split_bf16_colsum(src.data_ptr(), ld, rows, cols, dst.data_ptr(), stack, 0b100, part.data_ptr(), C.byref(n), st) == 0
        assert 0 < n.value * cols <= part.numel() and torch.equal(dst, want)
        out = torch.full((cols,), 3.0, device=DEV)
        item = (_lib.DmReduceItem * 1)()
        item[0].partial, item[0].out0, item[0].out1 = part.data_ptr(), out.data_ptr(), out.data_ptr()
        item[0].nrows, item[0].width, item[0].split, item[0].accumulate = n.value, cols, cols, 1
        assert lib.dm_partial_reduce_batch(item, 1, st) == 0
        outs.append(out)
    ref = src[:, :cols].double().sum(0) + 3.0
    assert (outs[0].double() - ref).abs().max().item() < 1e-4 * max(1.0, rows ** 0.5)
    assert torch.equal(outs[0], outs[1])
    assert lib.dm_split_bf16_colsum(src.data_ptr(), ld, rows, 20, want.data_ptr(), stack, 0b100, outs[0].data_ptr(), C.byref(n), st) != 0     # cols % 8


def test_split_bf16_wgrad_with_an_offset_operand():
    """the bf16x3 weight-gradient path fuses the bias gradient's column sums into the operand split only where the
    vector kernel's preconditions hold; a dy view that starts 4 bytes into its storage (not 16-byte aligned) takes the two-pass
    route and gives the same product and column sums (reference use: Linear backward, nets/ShfitScaleFormer.py:53-56)."""
    from deepmerge_amd import ops
    from deepmerge_amd._lib import DM_TN
    torch.manual_seed(17)
    K, M, N = 1024, 384, 512
    store = torch.randn(K * M + 4, device=DEV)
    x = torch.randn(K, N, device=DEV)
    outs = []
    for off in (0, 1):                     # aligned (fused pass) and offset by one float (two passes)
        dy = store[off:off + K * M].view(K, M)
        assert (dy.data_ptr() % 16 == 0) == (off == 0)
        dw = torch.empty(M, N, device=DEV)
        db = torch.zeros(M, device=DEV)
        with ops.fp32_products("bf16x3"):
            ops.gemm(DM_TN, dy, x, dw, M, N, K, colsum_out=db)
        ref = dy.double().t() @ x.double()
        assert ((dw.double() - ref).abs().max() / ref.abs().max()).item() < 3e-5
        assert (db.double() - dy.double().sum(0)).abs().max().item() < 1e-3
        outs.append((dw, db))


@pytest.mark.parametrize("layout,M,N,K", [("NT", 2048, 768, 768), ("NT", 1024, 768, 3072), ("NT", 320, 192, 128), ("NN", 2048, 768, 2304),
                                          ("NN", 4096, 3072, 768), ("TN", 768, 768, 4096), ("TN", 2304, 768, 16384), ("TN", 192, 320, 1024),
                                          ("NN", 24576, 768, 2304)])      # (1.5 rounds of 256 x 192 tiles: round 5 routes such dgrads to the 4-wave kernel)
@pytest.mark.parametrize("family", ["tiles", "w4", "lds"])
def test_folded_split_product_equals_the_image_product(layout, M, N, K, family, monkeypatch):
    """DmGemmArgs.k_fold (ABI 4): the bf16x3 product on hi / lo PLANE PAIRS, three K segments that re-read the planes in place, is the
    same sum in the same order as the product over the three-piece images of dm_split_bf16 -- bit-identical on the same kernel
    (the 128 x 128 / 64 x 64 tiles, the 4-wave persistent kernel, or the LDS-DMA kernels; the other families are off for both runs), with
    the fused epilogues and the split-K / K-slice paths."""
    from deepmerge_amd import ops
    from deepmerge_amd._lib import DM_EPI_MUL, DM_EPI_NONE, DM_NN, DM_NT, DM_TN
    for k in ("DM_GEMM_W4", "DM_GEMM_256", "DM_GEMM_RING"):
        monkeypatch.setenv(k, "0")
    if family == "w4":                                 # the 4-wave persistent kernel wherever it is legal (256 x 192 tiles, K % 128 == 0)
        monkeypatch.setenv("DM_GEMM_W4", "2")
        monkeypatch.setenv("DM_GEMM_W4_TN", "2")
    if family == "lds":                                # the LDS-DMA kernels: two-workgroup ring (NT) and the 256 x 256 pipeline
        monkeypatch.setenv("DM_GEMM_RING", "2")
        monkeypatch.setenv("DM_GEMM_256", "2")
    torch.manual_seed(M + N + K)
    lay = {"NT": DM_NT, "NN": DM_NN, "TN": DM_TN}[layout]
    A = torch.randn((K, M) if layout == "TN" else (M, K), device=DEV)
    B = torch.randn((N, K) if layout == "NT" else (K, N), device=DEV)
    ref = (A.double().t() if layout == "TN" else A.double()) @ (B.double().t() if layout == "NT" else B.double())
    bias, res, aux = torch.randn(N, device=DEV), torch.randn(M, N, device=DEV), torch.randn(M, N, device=DEV)
    kw = dict(bias=bias, residual=res) if layout == "NT" else dict(epilogue=DM_EPI_MUL, aux=aux, ldaux=N) if layout == "NN" else {}
    if layout == "NT":
        ref = ref + bias.double() + res.double()
    elif layout == "NN":
        ref = ref * aux.double()
    outs = []
    for planes in (False, True):
        monkeypatch.setattr(ops, "_PLANES", planes)
        C_ = torch.empty(M, N, device=DEV)
        with ops.fp32_products("bf16x3"):
            ops.gemm(lay, A, B, C_, M, N, K, **kw)
        outs.append(C_)
    if family == "w4":      # the 4-wave kernel's folded form sums hi.hi + hi.lo + lo.hi per 32 contraction positions (one staging of both
        # pieces for three MFMA sets), not segment after segment: same terms, another fp32 summation order
        assert ((outs[0].double() - outs[1].double()).abs().max() / ref.abs().max()).item() < 1e-5
    else:
        assert torch.equal(outs[0], outs[1])
    assert ((outs[1].double() - ref).abs().max() / ref.abs().max()).item() < 3e-5
    # a plane pair made once serves both sides and every layout: A as the left operand here, then as the right operand of a weight gradient
    monkeypatch.setattr(ops, "_PLANES", True)
    if layout != "TN":
        Ap = ops.split_planes(A)
        C2 = torch.empty(M, N, device=DEV)
        ops.gemm(lay, Ap, B, C2, M, N, K, **kw)
        if M * N * K >= ops._SPLIT_MIN_WORK:           # (below that ops.gemm keeps fp32 operands on the fp32 MFMA kernel)
            assert torch.equal(C2, outs[1])
        assert ((C2.double() - ref).abs().max() / ref.abs().max()).item() < 3e-5
        # the result as a plane pair (DM_BF16_PAIR): what splitting the fp32 result would give, without the pass
        Cp = ops.Planes(torch.empty(2, M, N, dtype=torch.bfloat16, device=DEV))
        ops.gemm(lay, Ap, B, Cp, M, N, K, **kw)
        assert torch.equal(Cp.t, ops.split_planes(C2).t)
        dy = torch.randn(M, 256, device=DEV)
        db = torch.zeros(256, device=DEV)
        dyp = ops.split_planes(dy, colsum_out=db)
        dW = torch.empty(256, K, device=DEV)
        ops.gemm(DM_TN, dyp, Ap, dW, 256, K, M)
        wref = dy.double().t() @ A.double()
        assert ((dW.double() - wref).abs().max() / wref.abs().max()).item() < 3e-5
        assert (db.double() - dy.double().sum(0)).abs().max().item() < 1e-3 * max(1.0, M / 1024)
        # the column sums of a plane pair ride on the weight gradient itself: colsum(hi) + colsum(lo), the duplicate hi segment skipped
        db2 = torch.full((256,), 3.0, device=DEV)
        dW2 = torch.empty(256, K, device=DEV)
        ops.gemm(DM_TN, dyp, Ap, dW2, 256, K, M, colsum_out=db2, colsum_accumulate=True)
        assert torch.equal(dW2, dW)
        pair = dyp.t.float().sum(0)                                        # hi + lo, as fp32 values
        assert (db2.double() - 3.0 - pair.double().sum(0)).abs().max().item() < 2e-3 * max(1.0, M / 1024)


@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
def test_split_bf16_gemm_against_float64(layout):
    """ops.gemm under fp32_products("bf16x3"): every layout, with bias / GELU / residual epilogues and the fused column sums,
    against float64 -- error ~1e-5 of the result's scale where one bf16 pass gives ~3e-3."""
    from deepmerge_amd import ops
    from deepmerge_amd._lib import DM_EPI_GELU, DM_EPI_NONE, DM_NN, DM_NT, DM_TN
    torch.manual_seed(11)
    M, N, K = 512, 384, 1024
    if layout == "NT":
        A, B, lay = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV), DM_NT
        ref = A.double() @ B.double().t()
    elif layout == "NN":
        A, B, lay = torch.randn(M, K, device=DEV), torch.randn(K, N, device=DEV), DM_NN
        ref = A.double() @ B.double()
    else:
        A, B, lay = torch.randn(K, M, device=DEV), torch.randn(K, N, device=DEV), DM_TN
        ref = A.double().t() @ B.double()
    bias = torch.randn(N, device=DEV)
    res = torch.randn(M, N, device=DEV)
    scale = ref.abs().max().item()

    def run(kind, **kw):
        out = torch.empty(M, N, device=DEV)
        with ops.fp32_products(kind):
            ops.gemm(lay, A, B, out, M, N, K, **kw)
        return out

    exact = run("mfma_f32")
    split = run("bf16x3")
    one_pass = torch.empty(M, N, device=DEV)
    ops.gemm(lay, A.bfloat16(), B.bfloat16(), one_pass, M, N, K)
    e_exact = (exact.double() - ref).abs().max().item() / scale
    e_split = (split.double() - ref).abs().max().item() / scale
    e_bf16 = (one_pass.double() - ref).abs().max().item() / scale
    print(f"{layout}: max error / scale  fp32 MFMA {e_exact:.1e}   split-bf16 {e_split:.1e}   one bf16 pass {e_bf16:.1e}")
    assert e_split < 2e-5 and e_bf16 > 20 * e_split
    if layout != "TN":
        y = run("bf16x3", bias=bias, residual=res)
        assert ((y.double() - (ref + bias.double() + res.double())).abs().max().item() / scale) < 2e-5
        g = run("bf16x3", bias=bias, epilogue=DM_EPI_GELU)
        want = torch.nn.functional.gelu(ref + bias.double())
        assert ((g.double() - want).abs().max().item() / scale) < 2e-5
    else:
        cs = torch.zeros(M, device=DEV)
        with ops.fp32_products("bf16x3"):
            out = torch.empty(M, N, device=DEV)
            ops.gemm(lay, A, B, out, M, N, K, colsum_out=cs)
        assert torch.allclose(cs.double(), A.double().sum(0), rtol=0, atol=1e-4 * K ** 0.5)
        assert ((out.double() - ref).abs().max().item() / scale) < 2e-5
        acc = split.clone()
        with ops.fp32_products("bf16x3"):
            ops.gemm(lay, A, B, acc, M, N, K, accumulate=True)
        assert ((acc.double() - 2 * ref).abs().max().item() / scale) < 4e-5
    assert ops.get_fp32_products() == "mfma_f32"
