"""Algorithmic work of the pair-encoder step (the figures bench.py and DESIGN.md price against), the synthetic pair batch of the
reference's tensor contract, and the one-GPU runners of BASELINE.json configs 3 / 4 / 5 that `bench.py` reports as `extras`
(`tools/bench_configs.py` is their command line)."""
import json
import time
from typing import Sequence

PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_TBPS = 8.0
PEAK_F32_TFLOPS = 157.3


def encoder_forward_flops(scales: Sequence[int], in_c: int, depth: Sequence[int], grid: int = 8, dim: int = 768,
                          n_designed: int = 19, out_dim: int = 100) -> float:
    """Forward FLOPs per encoder sample (SURVEY 8d / BASELINE.md section 3):
    sum_scales 2*64*(in_c*p^2)*C + sum_stages depth*(24*N*C^2 + 4*N^2*C) + 2*(19*C + 2*C^2) + 2*(S+1)*C*100."""
    S, C = len(scales), dim
    f = sum(2.0 * grid * grid * (in_c * int(s / grid) ** 2) * C for s in scales)
    for stage in range(3):
        n = S * (grid >> stage) ** 2
        f += depth[stage] * (24.0 * n * C * C + 4.0 * n * n * C)
    f += 2.0 * (n_designed * C + 2 * C * C)
    f += 2.0 * (S + 1) * C * out_dim
    return f


def pair_step_flops(scales, in_c, depth) -> float:
    """2 sides x 3 (forward + dgrad + wgrad) x forward FLOPs."""
    return 6.0 * encoder_forward_flops(scales, in_c, depth)


def synth_batch(B, scales, in_c, device, seed):
    """Synthetic pair batch of the reference's tensor contract (MyUtils1.py:41-77): per side a list of
    [B, in_c, s, s] float32 patches in [0,1] on a uint8 grid, designed features [B,1,19], flag [B]."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    left = [torch.randint(0, 256, (B, in_c, s, s), generator=g, dtype=torch.uint8).float().div_(255.0) for s in scales]
    right = [torch.randint(0, 256, (B, in_c, s, s), generator=g, dtype=torch.uint8).float().div_(255.0) for s in scales]
    flag = (torch.arange(B) % 2 == 0).to(torch.int64)
    for i in range(len(scales)):        # positives: jittered copy of the left crop (so some d < margin)
        right[i][flag == 1] = (left[i][flag == 1] * 0.9 + 0.1 * right[i][flag == 1])
    ld = torch.exp(torch.empty(B, 1, 19).uniform_(-4.6, 6.9, generator=g))
    rd = torch.where(flag.view(B, 1, 1) == 1, ld * 1.05, torch.exp(torch.empty(B, 1, 19).uniform_(-4.6, 6.9, generator=g)))
    mv = lambda t: t.to(device)
    return [mv(t) for t in left], mv(ld), [mv(t) for t in right], mv(rd), mv(flag)


# ---- BASELINE.json configs 3 / 4 / 5 on one GPU ------------------------------------------------------------------------------------
DEV = "cuda:0"


def timed(fn, steps, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def ev(fn, iters=10):
    import torch
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def config3(steps=10, numerics="bf16", graph=True):
    """BASELINE configs[2]: ViT-B/16 pair encoder on 224x224x3 patches, 128 pairs per step (fwd + loss + bwd + Adam)."""
    import torch
    from deepmerge_amd.trainer import PairTrainer
    from deepmerge_amd.vit_model import vit_base_patch16_224_in21k
    B = 128
    net = vit_base_patch16_224_in21k(num_classes=100, has_logits=False, numerics=numerics).to(DEV)

    class Pair(torch.nn.Module):      # adapt the 2-tensor pair signature to PairTrainer's 4-argument step
        def __init__(self, n):
            super().__init__(); self.n = n; self.numerics = numerics
        def forward(self, a, _1, b, _2):        # (the trainer's graph mode keeps its inputs as lists of tensors: one image tensor per side)
            return self.n(a[0], b[0])
    tr = PairTrainer(Pair(net), margin=1.0, lr=1e-4)
    if graph:                                  # the eager step is host-bound (~330 launches of 10-100 us): replay it as one hipGraph like the headline
        tr.enable_graph(warmup=1)
    g = torch.Generator().manual_seed(0)
    x1 = torch.rand(B, 3, 224, 224, generator=g).to(DEV); x2 = torch.rand(B, 3, 224, 224, generator=g).to(DEV)
    flag = (torch.arange(B) % 2).to(DEV)
    dt = timed(lambda: tr.step([x1], None, [x2], None, flag), steps, warm=3 if graph else 2)
    gf = 210.6
    tf = B / dt * gf / 1e3
    return {"config": f"3: ViT-B/16 pair encoder 224x224x3, 128 pairs/step, {numerics}, fwd+loss+bwd+Adam, {steps} timed steps" + (", hipGraph replay" if graph and tr.graph_error is None else ""), "pairs_per_s": round(B / dt, 1),
            "ms_per_step": round(dt * 1e3, 2), "model_TFLOPs": round(tf, 1), "roofline_frac": round(tf / PEAK_BF16_TFLOPS, 4)}


def config5(steps=10, graph=False, three_scale=False, numerics="bf16"):
    """SURVEY 8d config 5 on one GPU: the 4-scale / 4-channel 256x256 variant (headline) or, three_scale=True, the reference's
    default 3-scale / 3-channel geometry (config.py: scales [32, 64, 128])."""
    from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
    from deepmerge_amd.trainer import PairTrainer
    scales, in_c, depth, B = ([32, 64, 128], 3, [6, 4, 2], 120) if three_scale else ([32, 64, 128, 256], 4, [6, 4, 2], 120)
    net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=list(scales), depth=list(depth), in_c=in_c, numerics=numerics).to(DEV)
    tr = PairTrainer(net, margin=1.0, lr=1e-4)
    if graph:
        tr.enable_graph(warmup=1)
    batch = synth_batch(B, scales, in_c, DEV, 7)
    dt = timed(lambda: tr.step(*batch), steps, warm=3 if graph else 2)
    gf = pair_step_flops(scales, in_c, depth) / 1e9
    tf = B / dt * gf / 1e3
    return {"config": f"5 (1 GPU): v3 [6,4,2], {len(scales)} scales x {in_c} ch, 120 pairs/step, {numerics}, fwd+loss+bwd+Adam, {steps} timed steps" + (", hipGraph replay" if graph else ""),
            "pairs_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 2), "gflop_per_pair": round(gf, 1), "model_TFLOPs": round(tf, 1),
            "roofline_frac": round(tf / PEAK_BF16_TFLOPS, 4)}


def config4(passes=2):
    """BASELINE configs[3]: ExtractFeatures on a 4096x4096x4 tile (patch gather -> v3 [6,4,2] eval forward -> superpixel pooling ->
    pairwise similarity sweep over the RAG edges).  `roofline_frac`: the encode against the bf16 MFMA peak (it is > 99.9 % of the
    pipeline's time), the pooling / sweep kernels against 8 TB/s of HBM."""
    import torch
    from deepmerge_amd import ops
    from deepmerge_amd.ExtractFeatures import FeatureIO, rag_similarity_sweep
    from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
    from deepmerge_amd.patches import point_batch
    torch.manual_seed(0)
    # SURVEY 8d geometry: a jittered-Voronoi superpixel raster (cell 29 px -> 142 x 142 = 20 164 superpixels on the 4096^2 tile), 3 sample
    # points per superpixel around its seed, and the region-adjacency edges FROM the raster (deepmerge_amd.rag.rag_edges: ~59 k unique
    # 4-neighbour label pairs) -- not a regular grid with its 2 S edges
    from deepmerge_amd import rag
    bands, H, W, k = 4, 4096, 4096, 3
    lab, cy, cx, S = voronoi_raster(H, W, 29)
    tile = torch.randint(0, 256, (bands, H, W), dtype=torch.uint8, device=DEV)
    P = S * k
    xy = torch.stack([(cx.reshape(-1, 1).cpu() + torch.randint(-6, 7, (S, k))).reshape(-1), (cy.reshape(-1, 1).cpu() + torch.randint(-6, 7, (S, k))).reshape(-1)], 1)
    xy = xy.clamp(0, H - 1).to(torch.int32).to(DEV)
    inner = torch.randint(20, 29, (P,)); obj = inner + torch.randint(20, 29, (P,))
    feats = torch.rand(P, 15, device=DEV)
    ptr = (torch.arange(S + 1) * k).to(torch.int32).to(DEV); idx = torch.arange(P, dtype=torch.int32, device=DEV)
    edges, _w = rag.rag_edges(lab, S)
    edges = edges.to(torch.int32).contiguous()
    del lab
    net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=[32, 64, 128], depth=[6, 4, 2], in_c=bands, numerics="bf16")
    fio = FeatureIO(net, None, DEV)
    bs = 2000
    out = {}
    t_g = ev(lambda: point_batch(tile, xy[:bs], inner[:bs], obj[:bs], feats[:bs]), 5)
    win_bytes = float(((inner[:bs] ** 2 + obj[:bs] ** 2 + (2 * obj[:bs] - inner[:bs]) ** 2) * bands).sum())
    out_bytes = bs * bands * (32 * 32 + 64 * 64 + 128 * 128) * 4
    out["patch_gather"] = {"points_per_s": round(bs / t_g), "GBps_algorithmic(read window bytes + write fp32 patches)": round((win_bytes + out_bytes) / t_g / 1e9, 1)}

    from deepmerge_amd.patches import point_batch_cols
    t_gf = ev(lambda: point_batch_cols(tile, xy[:bs], inner[:bs], obj[:bs], feats[:bs]), 5)
    out["patch_gather_fused(bf16 patch-embed rows, no fp32 patches / im2col pass)"] = {
        "points_per_s": round(bs / t_gf), "GBps_algorithmic(read window bytes + write bf16 rows)": round((win_bytes + out_bytes / 2) / t_gf / 1e9, 1)}

    def encode_all(fused):
        F = torch.empty((P, 100), device=DEV)
        gather = point_batch_cols if fused else point_batch
        with torch.no_grad():
            for s in range(0, P, bs):
                e = min(P, s + bs)
                patches, designed = gather(tile, xy[s:e], inner[s:e], obj[s:e], feats[s:e])
                F[s:e] = net(patches, designed)
        return F
    for fused in (False, True):
        for _ in range(passes):
            t0 = time.perf_counter(); F = encode_all(fused); torch.cuda.synchronize(); t_enc = time.perf_counter() - t0
        key = "encode(gather + v3[6,4,2] eval, batch 2000)" if fused else "encode_unfused(fp32 patches -> im2col -> embed)"
        enc_tf = P / t_enc * encoder_forward_flops([32, 64, 128], bands, [6, 4, 2]) / 1e12
        out[key] = {"points_per_s": round(P / t_enc), "seconds_for_tile": round(t_enc, 2), "model_TFLOPs": round(enc_tf, 1),
                    "roofline_frac": round(enc_tf / PEAK_BF16_TFLOPS, 4)}
    t_p = ev(lambda: ops.segment_mean(F, ptr, idx, validate=False), 20)      # (the CSR was validated by the call below)
    out["segment_mean"] = {"us": round(t_p * 1e6, 1), "GBps_algorithmic": round((P * 404 + S * 400) / t_p / 1e9, 1),
                           "roofline_frac": round((P * 404 + S * 400) / t_p / 1e12 / PEAK_HBM_TBPS, 4)}
    pooled = ops.segment_mean(F, ptr, idx)
    t_e = ev(lambda: ops.edge_similarity(pooled, edges, 1.0, validate=False), 20)
    E = edges.shape[0]
    out["edge_similarity"] = {"us": round(t_e * 1e6, 1), "edges_per_s": round(E / t_e), "GBps_algorithmic": round(E * 812 / t_e / 1e9, 1),
                              "roofline_frac": round(E * 812 / t_e / 1e12 / PEAK_HBM_TBPS, 4)}
    _, simi, merge = rag_similarity_sweep(F, ptr, idx, edges, 1.0)
    from deepmerge_amd.ExtractFeatures import near_margin_count
    out["summary"] = {"points": P, "superpixels": S, "edges": E, "merge_fraction": round(float(merge.float().mean()), 3),
                      "edges_within_1e-4_of_margin": near_margin_count(simi, 1.0),
                      "sweep_total_us(pool+edges)": round((t_p + t_e) * 1e6, 1)}
    return {"config": "4: ExtractFeatures pipeline, 4096x4096x4 tile", **out}


def voronoi_raster(H, W, cell):
    """Jittered-Voronoi label raster: one seed per cell x cell square (jittered inside its middle 60 %), every pixel takes the nearest of the
    nine seeds around its square.  Returns (labels int32 [H, W], seed rows [gy, gx], seed columns [gy, gx], number of superpixels)."""
    import torch
    gy, gx = (H + cell - 1) // cell, (W + cell - 1) // cell
    cy = (torch.arange(gy, device=DEV)[:, None] + torch.rand(gy, gx, device=DEV) * 0.6 + 0.2) * cell
    cx = (torch.arange(gx, device=DEV)[None, :] + torch.rand(gy, gx, device=DEV) * 0.6 + 0.2) * cell
    yy, xx = torch.meshgrid(torch.arange(H, device=DEV), torch.arange(W, device=DEV), indexing="ij")
    best = torch.full((H, W), float("inf"), device=DEV); lab = torch.zeros((H, W), dtype=torch.int32, device=DEV)
    by, bx = yy // cell, xx // cell
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            ny, nx = (by + dy).clamp(0, gy - 1), (bx + dx).clamp(0, gx - 1)
            d = (yy - cy[ny, nx]) ** 2 + (xx - cx[ny, nx]) ** 2
            upd = d < best
            best = torch.where(upd, d, best); lab = torch.where(upd, (ny * gx + nx).to(torch.int32), lab)
    return lab, cy.clamp(0, H - 1), cx.clamp(0, W - 1), gy * gx


def config4r():
    """RAG + designed attributes from a 4096x4096 label raster: HBM-bound integer passes."""
    import torch
    from deepmerge_amd import rag
    torch.manual_seed(0)
    bands, H, W, cell = 4, 4096, 4096, 29
    lab, _cy, _cx, S = voronoi_raster(H, W, cell)
    tile = torch.randint(0, 256, (bands, H, W), dtype=torch.uint8, device=DEV)
    t_s = ev(lambda: rag.label_stats(lab, tile, S), 10)
    st = rag.label_stats(lab, tile, S)
    t_f = ev(lambda: rag.designed_features(st), 10)
    t_e = ev(lambda: rag.rag_edges(lab, S), 10)
    edges, w = rag.rag_edges(lab, S)
    px = H * W
    print(json.dumps({"config": "4r: RAG + designed attributes from a 4096x4096 label raster", "superpixels": S, "edges": int(edges.shape[0]),
                      "label_stats": {"us": round(t_s * 1e6, 1), "GBps_algorithmic(labels + 3 bands)": round(px * 7 / t_s / 1e9, 1)},
                      "designed_features": {"us": round(t_f * 1e6, 1)},
                      "rag_edges(incl. canonical sort + host check)": {"us": round(t_e * 1e6, 1), "GBps_algorithmic(labels)": round(px * 4 / t_e / 1e9, 1)}}), flush=True)
