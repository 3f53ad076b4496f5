"""GPU: size-independent properties at BASELINE's full sizes (configs[1]: 32 pairs -> 64 samples x 256 tokens x 768; config 4:
a 4096x4096x4 tile, ~60 k sample points) -- the oracle cannot run these sizes in seconds, the properties can."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _ops():
    from deepmerge_amd import ops
    return ops


@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
def test_gemm_full_size_selection_and_linearity(layout):
    """Stage-0 shapes of the step.  (1) a 0/1 selection matrix as one operand must copy columns of the other exactly;
    (2) gemm(a1 + a2) == gemm(a1) + gemm(a2) exactly on integer data (fp32 accumulation of integers is associative)."""
    ops = _ops()
    from deepmerge_amd._lib import DM_NN, DM_NT, DM_TN
    g = torch.Generator(device=DEV); g.manual_seed(3)
    M, N, K = 16384, 3072, 768
    ri = lambda shape, lo, hi: torch.randint(lo, hi, shape, device=DEV, generator=g).to(BF)
    if layout == "NT":                                     # y[M,N] = x[M,K] W[N,K]^T
        x1, x2 = ri((M, K), -3, 4), ri((M, K), -3, 4)
        sel = torch.zeros((N, K), device=DEV, dtype=BF)
        cols = torch.randint(0, K, (N,), device=DEV, generator=g)
        sel[torch.arange(N, device=DEV), cols] = 1
        run = lambda a: ops.gemm(DM_NT, a, sel, torch.empty((M, N), device=DEV, dtype=BF), M, N, K, lda=K, ldb=K, ldc=N)
        assert torch.equal(run(x1), x1[:, cols])
        w = ri((N, K), -2, 3)
        f = lambda a: ops.gemm(DM_NT, a, w, torch.empty((M, N), device=DEV), M, N, K, lda=K, ldb=K, ldc=N)
    elif layout == "NN":                                   # dx[M,K'] = dy[M,N'] W[N',K']  (here M x 768 = M x 3072 @ 3072 x 768)
        Kn, Nn = 3072, 768
        x1, x2 = ri((M, Kn), -3, 4), ri((M, Kn), -3, 4)
        sel = torch.zeros((Kn, Nn), device=DEV, dtype=BF)
        rows = torch.randint(0, Kn, (Nn,), device=DEV, generator=g)
        sel[rows, torch.arange(Nn, device=DEV)] = 1
        run = lambda a: ops.gemm(DM_NN, a, sel, torch.empty((M, Nn), device=DEV, dtype=BF), M, Nn, Kn, lda=Kn, ldb=Nn, ldc=Nn)
        assert torch.equal(run(x1), x1[:, rows])
        w = ri((Kn, Nn), -2, 3)
        f = lambda a: ops.gemm(DM_NN, a, w, torch.empty((M, Nn), device=DEV), M, Nn, Kn, lda=Kn, ldb=Nn, ldc=Nn)
    else:                                                  # dW[N,K] = dy[M,N]^T x[M,K]: one-hot rows of x pick rows of dy... summed
        Mo, No = 768, 3072                                 # dW [768, 3072] = dy[M,768]^T x[M,3072]
        x1, x2 = ri((M, Mo), -1, 2), ri((M, Mo), -1, 2)
        xb = ri((M, No), -1, 2)
        f = lambda a: ops.gemm(DM_TN, a, xb, torch.empty((Mo, No), device=DEV), Mo, No, M, lda=Mo, ldb=No, ldc=No)
        ones = torch.ones((M, Mo), device=DEV, dtype=BF)   # dy = 1: every row of dW is the column sum of x
        assert torch.equal(f(ones), xb.float().sum(0)[None].expand(Mo, No))
    s = (x1.float() + x2.float()).to(BF)                   # exact in bf16 (|values| <= 6)
    assert torch.equal(f(s), f(x1) + f(x2))


def test_attention_full_size_softmax_properties():
    """B = 64, N = 256, 12 heads (the pipelined kernels): V = 1 -> out = 1; zero scores + zero bias -> out = mean of V;
    backward: rows of dS sum to zero (so does the bias-gradient slab), zero dO -> zero gradients."""
    ops = _ops()
    B, N, H, D = 64, 256, 12, 64
    g = torch.Generator(device=DEV); g.manual_seed(5)
    qkv = torch.randn((B, N, 3, H, D), device=DEV, generator=g).to(BF)
    bias = torch.randn((H, N, N), device=DEV, generator=g)
    q1 = qkv.clone(); q1[:, :, 2] = 1
    out, lse = ops.attention_fwd(q1, bias, B, N, H, D, 0.125)
    assert float((out.float() - 1).abs().max()) <= 2.0 ** -7            # sum of bf16-rounded probabilities
    q0 = qkv.clone(); q0[:, :, 0] = 0
    v_int = torch.randint(-4, 5, (B, N, H, D), device=DEV, generator=g).to(BF)
    q0[:, :, 2] = v_int
    out0, lse0 = ops.attention_fwd(q0, torch.zeros_like(bias), B, N, H, D, 0.125)
    want = v_int.float().mean(dim=1)                                     # [B,H,D]
    assert float((out0.float().view(B, N, H, D) - want[:, None]).abs().max()) <= 0.05
    assert torch.allclose(lse0, torch.full_like(lse0, float(np.log(N))), atol=1e-5)
    # backward
    out, lse = ops.attention_fwd(qkv, bias, B, N, H, D, 0.125)
    dout = torch.randn((B, N, H * D), device=DEV, generator=g).to(BF)
    idx = torch.arange(N * N, device=DEV, dtype=torch.int32).reshape(N, N) % 1575
    dqkv, slab, info = ops.attention_bwd(qkv, bias, out, dout, lse, B, N, H, D, 0.125, idx, 1575)
    rowsum = slab.sum(dim=(0, 3))                                        # [H, N]: sum over chunks and keys of dS
    scale = float(slab.abs().sum(dim=(0, 3)).max())
    assert float(rowsum.abs().max()) <= 2e-2 * scale                     # bf16 P / dP rounding only
    dq0, slab0, _ = ops.attention_bwd(qkv, bias, out, torch.zeros_like(dout), lse, B, N, H, D, 0.125, idx, 1575)
    assert float(dq0.float().abs().max()) == 0.0 and float(slab0.abs().max()) == 0.0


def test_rows_adam_and_loss_properties_full_size():
    ops = _ops()
    g = torch.Generator(device=DEV); g.manual_seed(7)
    rows, C = 16384, 768
    x = torch.randn((rows, C), device=DEV, generator=g) * 3 + 1
    y, mean, rstd = ops.layernorm_fwd(x, torch.ones(C, device=DEV), torch.zeros(C, device=DEV), 1e-5, torch.float32)
    assert float(y.mean(1).abs().max()) < 1e-5 and float((y.var(1, unbiased=False) - 1).abs().max()) < 1e-3
    # Adam with a zero gradient: weights unchanged, moments decay by exactly beta
    n = 48_700_000
    p = torch.randn(n, device=DEV, generator=g); p0 = p.clone()
    m = torch.randn(n, device=DEV, generator=g); v = torch.rand(n, device=DEV, generator=g)
    m0, v0 = m.clone(), v.clone()
    z = torch.zeros(n, device=DEV)
    m.zero_()
    ops.adam_step(p, z, m, v, 7, lr=1e-4)
    assert torch.equal(p, p0) and torch.equal(m, torch.zeros_like(m)) and torch.equal(v, v0 * np.float32(0.999))
    # contrastive loss is symmetric in its two sides, bit for bit, and so are the gradients (with opposite sign)
    from deepmerge_amd.Losses import Loss
    a = torch.randn((4096, 100), device=DEV, generator=g).requires_grad_(True)
    b = torch.randn((4096, 100), device=DEV, generator=g).requires_grad_(True)
    flag = torch.randint(0, 2, (4096,), device=DEV, generator=g)
    l1 = Loss(1.0, 0.1, 0)(a, b, flag); l1.backward()
    ga, gb = a.grad.clone(), b.grad.clone()
    a.grad = b.grad = None
    l2 = Loss(1.0, 0.1, 0)(b, a, flag); l2.backward()
    assert float(l1.detach()) == float(l2.detach()) and torch.equal(ga, a.grad) and torch.equal(gb, b.grad) and torch.equal(ga, -gb)


def test_sweep_patch_and_rag_properties_config4_size():
    ops = _ops()
    from deepmerge_amd import rag
    from deepmerge_amd.patches import point_batch
    g = torch.Generator(device=DEV); g.manual_seed(9)
    S, k, D = 19881, 3, 100
    P = S * k
    F = torch.randn((P, D), device=DEV, generator=g)
    ptr = (torch.arange(S + 1, device=DEV) * k).to(torch.int32)
    idx = torch.arange(P, device=DEV, dtype=torch.int32)
    side = 141
    grid = torch.arange(S, device=DEV).reshape(side, side)
    edges = torch.cat([torch.stack([grid[:, :-1].reshape(-1), grid[:, 1:].reshape(-1)], 1),
                       torch.stack([grid[:-1].reshape(-1), grid[1:].reshape(-1)], 1)]).to(torch.int32)
    pooled = ops.segment_mean(F, ptr, idx)
    # a segment of 4 identical rows pools to that row exactly (x+x+x+x and /4 are exact in binary floating point)
    base = F[:S].contiguous()
    ptr4 = (torch.arange(S + 1, device=DEV) * 4).to(torch.int32)
    idx4 = torch.arange(4 * S, device=DEV, dtype=torch.int32)
    assert torch.equal(ops.segment_mean(base.repeat_interleave(4, 0).contiguous(), ptr4, idx4), base)
    # the per-edge distance is symmetric bit for bit; merge == (simi < margin); permuting the edge list permutes the result
    simi, merge = ops.edge_similarity(pooled, edges, 1.0)
    simi_r, merge_r = ops.edge_similarity(pooled, edges.flip(1).contiguous(), 1.0)
    assert torch.equal(simi, simi_r) and torch.equal(merge, merge_r)
    assert torch.equal(merge.bool(), simi < 1.0)
    perm = torch.randperm(edges.shape[0], device=DEV, generator=g)
    sp, _ = ops.edge_similarity(pooled, edges[perm].contiguous(), 1.0)
    assert torch.equal(sp, simi[perm])
    # patch pyramid: a constant tile gives constant patches (v / 255) wherever the window is inside the raster, zeros outside
    tile = torch.full((4, 4096, 4096), 200, dtype=torch.uint8, device=DEV)
    xy = torch.tensor([[2048, 2048], [100000, 100000]], dtype=torch.int32, device=DEV)
    inner, obj = torch.tensor([24, 24]), torch.tensor([50, 50])
    patches, _ = point_batch(tile, xy, inner, obj, torch.zeros((2, 15), device=DEV))
    for pt in patches:
        assert torch.equal(pt[0], torch.full_like(pt[0], np.float32(200) / np.float32(255.0)))
        assert float(pt[1].abs().max()) == 0.0
    # RAG on a 4096^2 raster: every pixel is counted once, every shared pixel edge is seen from both sides
    lab = (torch.arange(4096, device=DEV)[:, None] // 29 * 142 + torch.arange(4096, device=DEV)[None, :] // 29).to(torch.int32)
    Sr = 142 * 142
    st = rag.label_stats(lab.contiguous(), tile, Sr)
    assert int(st["count"].sum()) == 4096 * 4096
    e, w = rag.rag_edges(lab.contiguous(), Sr)
    assert int(st["peri"][:, 0].sum()) == 2 * int(w.sum())
    assert int(st["peri"][:, 1].sum()) == 4 * 4096                       # the raster's own border
    root = rag.merge_components(e, torch.ones(e.shape[0], dtype=torch.bool, device=DEV), Sr)
    assert int(root.max()) == 0                                          # merging along every edge leaves one region


def test_whole_model_side_swap_symmetry_full_size():
    """configs[1] model and batch (32 pairs, 4 scales x 4 channels, bf16): swapping the two sides of every pair swaps the
    two embeddings bit for bit (no kernel mixes samples, and a row's arithmetic does not depend on its position in the batch),
    leaves the loss unchanged, and the parameter gradients agree to rounding (their summation order over samples changes)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import synth_batch
    from deepmerge_amd.Losses import Loss
    from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
    scales, in_c = [32, 64, 128, 256], 4
    torch.manual_seed(0)
    net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=list(scales), depth=[3, 2, 1], in_c=in_c, numerics="bf16").to(DEV).train()
    left, ld, right, rd, flag = synth_batch(32, scales, in_c, DEV, 77)
    crit = Loss(1.0, 0.1, 0)
    fa, fb = net(left, ld, right, rd)
    l1 = crit(fa, fb, flag); l1.backward()
    g1 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=True)
    fb2, fa2 = net(right, rd, left, ld)
    l2 = crit(fb2, fa2, flag); l2.backward()
    assert torch.equal(fa, fa2) and torch.equal(fb, fb2)
    assert float(l1.detach()) == float(l2.detach())
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        a, b = g1[n].float(), p.grad.float()
        assert float((a - b).norm()) <= 2e-2 * float(a.norm()) + 1e-7, n


def test_config5_model_full_size_properties():
    """BASELINE configs[4]'s per-GPU step: v3 depth [6,4,2], 4 scales x 4 channels, 120 pairs (config.py:20), bf16, through
    PairTrainer.  Size-independent checks: side swap swaps the embeddings bit for bit; the captured (hipGraph) step and the
    segmented data-parallel schedule give bit-identical weights to the eager step; loss is finite and moves."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import synth_batch
    from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
    from deepmerge_amd.trainer import PairTrainer
    scales, in_c, depth, B = [32, 64, 128, 256], 4, [6, 4, 2], 120
    nets = []
    for _ in range(3):
        torch.manual_seed(0)
        nets.append(ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=list(scales), depth=list(depth), in_c=in_c, numerics="bf16").to(DEV).train())
    left, ld, right, rd, flag = synth_batch(B, scales, in_c, DEV, 5)
    with torch.no_grad():
        fa, fb = nets[0](left, ld, right, rd)
        fb2, fa2 = nets[0](right, rd, left, ld)
    assert torch.equal(fa, fa2) and torch.equal(fb, fb2) and fa.shape == (B, 100) and bool(torch.isfinite(fa).all())
    eager, graph, seg = PairTrainer(nets[0], lr=1e-4), PairTrainer(nets[1], lr=1e-4), PairTrainer(nets[2], lr=1e-4, segmented=True)
    graph.enable_graph(warmup=1)
    seg.enable_graph(warmup=1)
    losses = []
    for _ in range(3):
        le, lg, ls = eager.step(left, ld, right, rd, flag), graph.step(left, ld, right, rd, flag), seg.step(left, ld, right, rd, flag)
        assert float(le) == float(lg) == float(ls)
        losses.append(float(le))
    assert torch.equal(eager.fp.flat, graph.fp.flat) and torch.equal(eager.fp.flat, seg.fp.flat)
    assert len(seg.bucket_slices) == 8 and all(np.isfinite(losses)) and losses[2] != losses[0]
