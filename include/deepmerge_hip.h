/*
 * deepmerge_hip.h -- C-ABI of libdeepmerge_hip.so, the MI355X (gfx950) kernels behind the
 * DeepMerge pair-encoder hot path.
 *
 * The reference (lvxianwei/DeepMerge) has NO native/FFI boundary: the path sits behind plain
 * Python torch.nn.Module classes that call stock torch ops.  Each entry point below therefore
 * cites the reference *Python* statement(s) whose arithmetic it replaces (file:line relative to
 * the reference tree); INTEGRATION.md shows the ctypes binding and the module-level drop-in.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (torch storage); the library
 *     never allocates, frees or retains memory and never synchronises the device;
 *   - every call enqueues on the caller's `stream` (hipStream_t passed as void*; NULL = default);
 *   - return value: 0 on success, negative DmStatus otherwise; dm_last_error() gives a
 *     thread-local message; no C++ exception crosses the boundary;
 *   - `dtype` arguments take DmDtype; "T" below means the activation type of the numerics mode:
 *     DM_BF16 (throughput mode: bf16 operands, fp32 accumulate) or DM_F32 (parity mode: fp32
 *     operands on the f32-input MFMA, bit-equivalent to an fmaf chain);
 *   - all matrices are row-major with an explicit leading dimension in ELEMENTS.
 */
#ifndef DEEPMERGE_HIP_H
#define DEEPMERGE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { DM_F32 = 0, DM_BF16 = 1,
               DM_BF16_PAIR = 2      /* ABI 4, DmGemmArgs.c_dtype only: C is written as a hi / lo plane pair (see c_plane) */
} DmDtype;

typedef enum {
  DM_OK = 0,
  DM_ERR_BAD_SHAPE = -1,
  DM_ERR_BAD_DTYPE = -2,
  DM_ERR_BAD_ALIGN = -3,
  DM_ERR_WORKSPACE = -4,
  DM_ERR_HIP = -5,
  DM_ERR_UNSUPPORTED = -6
} DmStatus;

/* ---- library ------------------------------------------------------------------------- */
int dm_abi_version(void);   /* 2: dm_patch_pyramid / dm_patch_pyramid_cols take a resize rule; 3: table-reading and split-bf16 attention entry points, dm_split_bf16_colsum (round 3);
                             * 4: DmGemmArgs.k_fold / a_fold / b_fold, dm_split_bf16_planes (round 4); 5: dm_pair_batch_gather (round 5); 6: dm_gemm_grouped (round 5) */
const char *dm_last_error(void);
/* Name of the code object architecture the library was built for ("gfx950"). */
const char *dm_arch(void);

/* ---- GEMM family (MFMA, LDS-staged 128x128 tiles) ---------------------------------------
 * Replaces every nn.Linear / k=stride Conv2d / k=1 Conv1d on the path and their autograd:
 *   qkv, proj        nets/ShfitScaleFormer.py:119, :134        (vit_model.py:119, :133)
 *   fc1, fc2         nets/ShfitScaleFormer.py:53, :56          (vit_model.py:152-156)
 *   PatchEmbed.proj  nets/ShfitScaleFormer.py:35               (after dm_patchify)
 *   FeatureEmbed     nets/ShfitScaleFormer.py:76-79, final Linear :948 / :967
 * layout: DM_NT  C[m,n] = sum_k A[m,k] * B[n,k]   (forward  y = x W^T,  W = [out,in])
 *         DM_NN  C[m,n] = sum_k A[m,k] * B[k,n]   (dgrad    dx = dy W)
 *         DM_TN  C[m,n] = sum_k A[k,m] * B[k,n]   (wgrad    dW = dy^T x)
 * epilogue, applied in this order on the fp32 accumulator v of element (m,n):
 *   v += bias[n]                     if bias != NULL
 *   aux[m,n] = v (as aux_dtype)      if epilogue == DM_EPI_GELU     (pre-activation saved for backward; aux may be NULL:
 *                                       inference, nothing saved)
 *   v  = gelu_erf(v)                 if epilogue == DM_EPI_GELU     (nn.GELU, erf form)
 *   v *= gelu_erf'(aux[m,n])         if epilogue == DM_EPI_DGELU    (backward through GELU)
 *   aux[m,n] = gelu_erf'(v); v = gelu_erf(v)   if epilogue == DM_EPI_GELU_GRAD  (the derivative is saved instead of the
 *                                       pre-activation: same bytes, and the backward epilogue becomes one multiply)
 *   v *= aux[m,n]                    if epilogue == DM_EPI_MUL      (backward through GELU with the saved derivative)
 *   v += residual[m,n]               if residual != NULL (fp32)
 *   v += C_old[m,n]                  if accumulate (C must be fp32; not with DM_EPI_DGELU / DM_EPI_MUL: DM_ERR_UNSUPPORTED)
 *   C[m,n] = v (as c_dtype)
 * split_k > 1 (DM_TN only): the contraction is cut into split_k slices whose fp32 partial tiles
 * go to `workspace` ([split_k, M, N] floats) and are summed deterministically by a second
 * kernel that applies the epilogue.  split_k == 0 lets the library choose.
 */
typedef enum { DM_NT = 0, DM_NN = 1, DM_TN = 2 } DmGemmLayout;
typedef enum { DM_EPI_NONE = 0, DM_EPI_GELU = 1, DM_EPI_DGELU = 2, DM_EPI_GELU_GRAD = 3, DM_EPI_MUL = 4 } DmEpilogue;

typedef struct {
  int32_t layout;        /* DmGemmLayout */
  int32_t ab_dtype;      /* DmDtype of A and B */
  int32_t c_dtype;       /* DmDtype of C */
  int32_t aux_dtype;     /* DmDtype of aux */
  int32_t M, N, K;
  int32_t epilogue;      /* DmEpilogue */
  int32_t accumulate;    /* 0/1 */
  int32_t split_k;       /* 0 = auto, 1 = none */
  const void *A; int64_t lda;
  const void *B; int64_t ldb;
  void *C; int64_t ldc;
  const float *bias;                     /* [N] or NULL */
  const float *residual; int64_t ldr;    /* [M,N] fp32 or NULL */
  void *aux; int64_t ldaux;              /* see epilogue */
  /* Optional two-level row addressing of C / residual / aux: row m lives at
   * (m / rows_per_group) * group_stride + (m % rows_per_group) * ld.  rows_per_group == 0: plain.
   * Used to write each scale's 64 patch tokens into its slice of the token cube (torch.cat at
   * nets/ShfitScaleFormer.py:877). */
  int32_t rows_per_group; int64_t group_stride;
  void *workspace; int64_t workspace_bytes;
  /* DM_TN only, optional: colsum_a[m] (+)= sum_k A[k,m], fp32 [M] -- the bias gradient that goes with a weight gradient
   * dW = dy^T x (autograd of nn.Linear's bias: column sums of dy).  On the 256x256 wgrad pipeline it costs a few extra
   * MFMAs against a ones fragment instead of another pass over dy; otherwise dm_gemm runs the column-sum kernels itself.
   * Needs the workspace (dm_gemm_workspace_bytes covers it). */
  float *colsum_a; int32_t colsum_accumulate;
  /* ABI 4, optional: folded contraction for the "bf16x3" products on hi / lo PLANE PAIRS (dm_split_bf16_planes).  k_fold > 0:
   * K == 3 * k_fold, and the K segment s = 0, 1, 2 of A is the plain operand (same layout, same lda, contraction length k_fold)
   * that starts a_fold[s] ELEMENTS behind A -- {0, 0, plane} for the left operand (hi, hi, lo), {0, plane, 0} for the right one
   * (hi, lo, hi), plane = rows * ld of the plane pair.  Every tensor is then split ONCE, whatever side and layout its consumers
   * read it in, and the hi plane is fetched twice from the same addresses instead of being stored twice.
   * bf16 operands only; k_fold % 64 == 0; 0 <= offsets < 2^30.  With colsum_a (DM_TN) the sums are colsum(hi plane) +
   * colsum(lo plane) of A, fp32 -- the bias gradient to the accuracy of the folded product itself, NOT the exact fp32 column sum of
   * the unsplit tensor (the segment that re-reads the hi plane is skipped).  Shapes the folded kernels do not take return
   * DM_ERR_UNSUPPORTED; nothing falls back inside the library (a caller may split into dm_split_bf16 images and call again). */
  int32_t k_fold;
  int64_t a_fold[3], b_fold[3];
  /* ABI 4: c_dtype == DM_BF16_PAIR writes the result as the hi / lo plane pair a later folded product reads (the split pass of
   * that tensor disappears): hi = bf16(v) at C[m * ldc + n], lo = bf16(v - hi) at C[c_plane + m * ldc + n], C a bf16 pointer, ldc and
   * c_plane in bf16 elements (multiples of 8).  Not with accumulate; NT / NN products on the MFMA path (DM_ERR_UNSUPPORTED otherwise). */
  int64_t c_plane;
} DmGemmArgs;

int dm_gemm(const DmGemmArgs *args, void *stream);
/* Bytes of workspace dm_gemm may use for these dimensions (upper bound over split_k choices). */
int64_t dm_gemm_workspace_bytes(int32_t layout, int32_t M, int32_t N, int32_t K);
/* ABI 6: n INDEPENDENT products (no output overlaps another product's operands or output), results as n dm_gemm calls in order --
 * weight gradients up to the order of the fp32 additions over K.  The four weight gradients of a transformer block (dW = dy^T x for
 * qkv / proj / fc1 / fc2: nets/ShfitScaleFormer.py:35, :119, :134 and vit_model.py:112-135, :160-176 under autograd) feed nothing else
 * in the block's backward pass.  Alone each has 12 .. 48 output tiles of 256 x 192: it either leaves most CUs idle or pays up to 16 K
 * slices, a slab round trip of as many partial gradients and a reduction launch.  Fast path: 1 .. 8 bf16 DM_TN products (plain operands
 * or hi / lo plane pairs through k_fold; fp32 C, no epilogue operands, split_k == 0, M % 256 == N % 192 == 0, K % 128 == 0) in ONE launch:
 *   contraction <= 12288, 2+ products, all tiles within one round of the CUs: one K slice per tile, the gradient stored / accumulated
 *   in place (the form the training step uses: -3.0 % on the headline step);
 *   contraction > 12288, the same for all products, tiles x slices >= 0.85 of the CUs: the SAME K slices for every product, (product,
 *   tile, slice) per workgroup, partial tiles to each product's own workspace slab and its own reduction behind (the 12-tile proj gradient
 *   next to the qkv gradient: 5 slices on 240 workgroups instead of 16 + 7 on two launches; -1.1 % on the headline step);
 *   DM_GEMM_GROUPED=3, with `workspace` (dm_gemm_grouped_workspace_bytes): "stream-K" -- every workgroup takes the same number of
 *   consecutive K steps of its product's tile-major step space, a tile is left as 2-3 partial pieces and ONE fix-up launch sums them in
 *   workgroup order (deterministic).  Exact, tested, and slower than the separate launches (no operand panel is shared between
 *   workgroups that sit at different K offsets: csrc/dm_gemm_w4.hip); never chosen by the default rule.
 * colsum_a is produced by the same launches.  Every other group: the calls one after the other (same errors as dm_gemm).  Each args[i]
 * carries its own workspace, as for dm_gemm; `workspace` (16-byte aligned, may be NULL) belongs to the group. */
int64_t dm_gemm_grouped_workspace_bytes(const DmGemmArgs *args, int32_t n);
int dm_gemm_grouped(const DmGemmArgs *args, int32_t n, void *workspace, int64_t workspace_bytes, void *stream);

/* ---- fused attention with 3-D relative-position bias -------------------------------------
 * Replaces nets/ShfitScaleFormer.py:119-133 (reshape/permute, q*scale, q@k^T, bias add, softmax,
 * attn@v, transpose/reshape) and vit_model.py:119-133 (bias == NULL; the scale 64^-0.5 = 2^-3 is
 * exact, so applying it before or after q@k^T is bit-identical).
 *   qkv   [B, N, 3, H, D]  T   (output of the qkv Linear).  D = 64 and N <= 256 run the MFMA kernels; any other head dim <= 128 /
 *                            N <= 4096 (ViT-H/14: D = 80, N = 257) a plain fp32 kernel family (no bias-table gradient there)
 *   bias  [H, N, N] fp32 or NULL  (dense, from dm_relpos_bias_gather)
 *   out   [B, N, H*D]      T
 *   lse   [B, H, N] fp32   row log-sum-exp of the biased scores (saved for backward)
 */
int dm_attention_fwd(const void *qkv, const float *bias, void *out, float *lse,
                     int32_t B, int32_t N, int32_t H, int32_t D, float scale, int32_t dtype, void *stream);
/* The same forward with the bias taken inside the kernel from relative_position_bias_table [n_bins, H] fp32 of a
 * (cube_s, cube_h, cube_w) token cube (tokens scale-major then row-major, index rule of nets/ShfitScaleFormer.py:139-156):
 * the dense [H,N,N] rows of dm_relpos_bias_gather are never formed.  dm_attention_relpos_inkernel returns 1 for the
 * shapes this takes (bf16, D = 64, cube (3|4, 8, 8), N = 64 cube_s, B*H >= 96) and 0 otherwise: for those, gather and
 * call dm_attention_fwd (dm_attention_fwd_relpos returns DM_ERR_UNSUPPORTED without launching anything).  Results equal dm_attention_fwd's
 * on the gathered bias up to fp32 rounding of bias / scale. */
int32_t dm_attention_relpos_inkernel(int32_t B, int32_t N, int32_t H, int32_t D, int32_t cube_s, int32_t cube_h,
                                     int32_t cube_w, int32_t dtype);
int dm_attention_fwd_relpos(const void *qkv, const float *table, int32_t cube_s, int32_t cube_h, int32_t cube_w,
                            void *out, float *lse, int32_t B, int32_t N, int32_t H, int32_t D, float scale,
                            int32_t dtype, void *stream);
/* Backward: dqkv [B,N,3,H,D] T (fully written).  If dbias_slab != NULL the gradient of the dense bias is written
 * too: dbias_slab[c][h][i][j] = sum over the samples of batch chunk c of dS[b,h,i,j], fp32, fully written,
 * dm_attention_bwd_batch_chunks(B,N,H) * H * N * N floats; fold it into the table's gradient with
 * dm_relpos_bias_reduce.  No atomics are used anywhere, so every output is run-to-run deterministic.
 * delta is a [B,H,N] fp32 scratch.  bias_t (optional) is the per-head transpose of bias, bias_t[h][key][q], which
 * lets the key-major kernel read the bias with coalesced vector loads; NULL falls back to strided reads of `bias`. */
int dm_attention_bwd(const void *qkv, const float *bias, const float *bias_t, const void *out, const void *dout,
                     const float *lse, void *dqkv, float *delta, float *dbias_slab, int32_t B, int32_t N, int32_t H,
                     int32_t D, float scale, int32_t dtype, void *stream);
/* dm_attention_bwd for the shapes dm_attention_relpos_inkernel takes: both passes form the bias from the table inside
 * the kernel, so bias / bias_t may be NULL (they are only read when the table kernels are switched off for A/B runs).
 * dbias_slab as for dm_attention_bwd (same chunk count, same layout, deterministic); its entries are sums of dS values
 * rounded to bf16 (the operand the dK product consumes), accumulated in fp32. */
int dm_attention_bwd_relpos(const void *qkv, const float *table, int32_t cube_s, int32_t cube_h, int32_t cube_w,
                            const float *bias, const float *bias_t, const void *out, const void *dout, const float *lse,
                            void *dqkv, float *delta, float *dbias_slab, int32_t B, int32_t N, int32_t H, int32_t D,
                            float scale, int32_t dtype, void *stream);
/* Attention of the "bf16x3" numerics mode: fp32 tensors, every product as a split-bf16 triple on the matrix pipe (hi.hi +
 * lo.hi + hi.lo into fp32: ~2^-17 relative per product, where the plain fp32 entry points above use fp32 FMA / fp32 MFMA
 * arithmetic at a fraction of the rate).  Replaces the same reference statements as dm_attention_fwd.  qkv_hi / qkv_lo:
 * caller-allocated bf16 tensors of qkv's shape, WRITTEN here (hi = bf16(x), lo = bf16(x - hi)) and kept by the caller for
 * the backward pass -- or, with qkv == NULL (ABI 4), READ here: the caller filled them (the qkv product with a DM_BF16_PAIR result).
 * table: NULL (no bias) or the relative-position table of a (cube_s, 8, 8) token cube as for
 * dm_attention_fwd_relpos.  dm_attention_split_ok returns 1 for the shapes taken (D = 64, 128 < N <= 256, cube (3|4, 8, 8)
 * when a table is given). */
int32_t dm_attention_split_ok(int32_t B, int32_t N, int32_t H, int32_t D, int32_t has_table, int32_t cube_s, int32_t cube_h,
                              int32_t cube_w);
int dm_attention_split_fwd(const float *qkv, void *qkv_hi, void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h,
                           int32_t cube_w, float *out, float *lse, int32_t B, int32_t N, int32_t H, int32_t D, float scale,
                           void *stream);
/* The same with the hi / lo PLANE PAIR of `out` written on the side (out_pair bf16 [2][B*N, H*64]; ABI 4): the output projection of the
 * "bf16x3" mode reads it as its folded left operand (DmGemmArgs.k_fold). */
int dm_attention_split_fwd_pair(const float *qkv, void *qkv_hi, void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h,
                                int32_t cube_w, float *out, void *out_pair, float *lse, int32_t B, int32_t N, int32_t H, int32_t D,
                                float scale, void *stream);
/* Backward of dm_attention_split_fwd: dqkv [B,N,3,H,64] fp32 fully written; qkv_hi / qkv_lo as the forward left them;
 * dout_hi / dout_lo: caller-allocated bf16 scratch of dout's shape (written here); delta [B,H,N] scratch; dbias_slab
 * (needs the table): dm_attention_split_bwd_chunks(B,N,H) x H x N x N floats, layout and reduction as for dm_attention_bwd. */
int32_t dm_attention_split_bwd_chunks(int32_t B, int32_t N, int32_t H);
int dm_attention_split_bwd(const void *qkv_hi, const void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h,
                           int32_t cube_w, const float *out, const float *dout, void *dout_hi, void *dout_lo,
                           const float *lse, float *dqkv, float *delta, float *dbias_slab, int32_t B, int32_t N,
                           int32_t H, int32_t D, float scale, void *stream);
/* The same with dqkv written as a hi / lo PLANE PAIR (dqkv_pair bf16 [2][B,N,3,H,64]; ABI 4) instead of fp32: the operand format of the
 * folded qkv weight / data gradient products (DmGemmArgs.k_fold), which then need no split pass. */
int dm_attention_split_bwd_pair(const void *qkv_hi, const void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h,
                                int32_t cube_w, const float *out, const float *dout, void *dout_hi, void *dout_lo,
                                const float *lse, void *dqkv_pair, float *delta, float *dbias_slab, int32_t B, int32_t N,
                                int32_t H, int32_t D, float scale, void *stream);
/* Number of batch chunks dm_attention_bwd uses for this problem size and dtype (first dimension of dbias_slab). */
int32_t dm_attention_bwd_batch_chunks(int32_t B, int32_t N, int32_t H, int32_t dtype);

/* relative_position_bias_table[index.view(-1)].view(N,N,H).permute(2,0,1)
 * (nets/ShfitScaleFormer.py:123-128): table [n_bins,H] fp32, index int32 [N,N] -> bias [H,N,N] and,
 * if bias_t != NULL, its per-head transpose bias_t[h][j][i] = bias[h][i][j]. */
int dm_relpos_bias_gather(const float *table, const int32_t *index, float *bias, float *bias_t,
                          int32_t N, int32_t H, int32_t n_bins, void *stream);
/* Autograd of that gather (the index_put the reference's autograd performs for table[index], :123-128):
 *   dtable[bin,h] (+)= sum_c sum_{(i,j): index[i,j] == bin} dbias_slab[c][h][i][j]
 * `positions` (int32, flat i*N+j, ascending within a bin) and `offsets` (int32 [n_bins+1]) are the CSR inverse of the
 * index: positions[offsets[bin] .. offsets[bin+1]) are the entries that read table row `bin`.  Fixed summation
 * order: deterministic.  The slab is scratch: its first chunk is overwritten with the sum over chunks. */
int dm_relpos_bias_reduce(float *dbias_slab, const int32_t *positions, const int32_t *offsets, float *dtable,
                          int32_t chunks, int32_t H, int32_t N, int32_t n_bins, int32_t accumulate, void *stream);

/* ---- row kernels (HBM-bound) ------------------------------------------------------------ */
/* nn.LayerNorm over the last dim (nets/ShfitScaleFormer.py:182-183, :902, :915, :926, :941;
 * vit_model.py:183-184, :498): x [rows, cols] fp32 -> y (y_dtype), saving mean / rstd [rows].
 * y_dtype == DM_BF16_PAIR (ABI 4, cols <= 1024): y is the hi / lo plane pair [2, rows, cols] of the result. */
int dm_layernorm_fwd(const float *x, const float *gamma, const float *beta, void *y, int32_t y_dtype,
                     float *mean, float *rstd, int32_t rows, int32_t cols, float eps, void *stream);
/* dx = (dres ? dres : 0) + LN'(dy); dx_lp (optional) receives the same values as bf16 (the operand
 * copy the previous layer's backward GEMMs consume); dgamma/dbeta (+)= column sums.  partial: fp32 scratch of
 * dm_layernorm_bwd_partial_floats(cols) floats. */
int dm_layernorm_bwd(const void *dy, int32_t dy_dtype, const float *x, const float *gamma,
                     const float *mean, const float *rstd, const float *dres, float *dx, void *dx_lp,
                     float *dgamma, float *dbeta, int32_t accumulate_params, float *partial,
                     int32_t rows, int32_t cols, void *stream);
int64_t dm_layernorm_bwd_partial_floats(int32_t cols);
/* The same backward WITHOUT the final reduction of the per-workgroup [dgamma | dbeta] partial rows: `*n_partial` rows of 2 * cols floats
 * are left in `partial` (keep it until they are reduced).  The reductions of a whole backward pass -- one per LayerNorm application,
 * nets/ShfitScaleFormer.py:170-183 runs two per block -- are then done by ONE dm_partial_reduce_batch launch instead of one small
 * launch each (out0 = dgamma, out1 = dbeta, width = 2 * cols, split = cols).  Results are bit-identical to dm_layernorm_bwd. */
int dm_layernorm_bwd_partials(const void *dy, int32_t dy_dtype, const float *x, const float *gamma,
                              const float *mean, const float *rstd, const float *dres, float *dx, void *dx_lp,
                              float *partial, int32_t rows, int32_t cols, int32_t *n_partial, void *stream);
/* The same with the hi / lo PLANE PAIR of dx on the side (dx_pair bf16 [2, rows, cols]; ABI 4): the gradient that leaves a LayerNorm
 * is the left operand of the next weight / data gradient products of the "bf16x3" mode, which then need no split pass.  cols <= 1024. */
int dm_layernorm_bwd_partials_pair(const void *dy, int32_t dy_dtype, const float *x, const float *gamma,
                                   const float *mean, const float *rstd, const float *dres, float *dx, void *dx_pair, float *partial,
                                   int32_t rows, int32_t cols, int32_t *n_partial, void *stream);
/* One job of dm_partial_reduce_batch: out0[j] (+)= sum_r partial[r][j] for j < split, out1[j - split] (+)= ... for split <= j < width;
 * rows are summed in a fixed order (deterministic).  Jobs of one batch must not share output elements. */
typedef struct DmReduceItem {
  const float *partial;
  float *out0, *out1;
  int32_t nrows, width, split, accumulate;
} DmReduceItem;
/* items: HOST array of n jobs (passed to the kernel by value, 32 per launch). */
int dm_partial_reduce_batch(const DmReduceItem *items, int32_t n, void *stream);

/* Per-scale 2x2 average pooling of the token grid (AvgPool2d(2,2) on [B,C,side,side] views,
 * nets/ShfitScaleFormer.py:892-901, :905-914): x [B, S*side*side, C] -> y [B, S*(side/2)^2, C], fp32. */
int dm_token_pool_fwd(const float *x, float *y, int32_t B, int32_t S, int32_t side, int32_t C, void *stream);
int dm_token_pool_bwd(const float *dy, float *dx, int32_t B, int32_t S, int32_t side, int32_t C, void *stream);
/* Mean over groups of `g` consecutive rows (AdaptiveAvgPool1d(1) per scale, :930-938):
 * x [rows*g, C] -> y [rows, C]; backward broadcasts dy/g. */
int dm_group_mean_fwd(const float *x, float *y, int32_t rows, int32_t g, int32_t C, void *stream);
int dm_group_mean_bwd(const float *dy, float *dx, int32_t rows, int32_t g, int32_t C, void *stream);

/* out[n] (+)= sum_m X[m,n]  (bias gradients).  partial: dm_colsum_partial_floats(N) floats. */
int dm_colsum(const void *X, int32_t dtype, int64_t ldx, float *out, int32_t M, int32_t N,
              int32_t accumulate, float *partial, void *stream);
int64_t dm_colsum_partial_floats(int32_t N);

/* fp32 -> T element-wise copy (weights / activations), n elements. */
int dm_cast(const float *src, void *dst, int32_t dst_dtype, int64_t n, void *stream);

/* Split-bf16 operand image for the "bf16x3" numerics mode: x = hi + lo with hi = bf16(x), lo = bf16(x - hi), so that an fp32
 * product A.B is recovered to ~2^-17 relative by ONE bf16 GEMM over a 3x longer contraction, [Ah | Ah | Al] . [Bh | Bl | Bh]
 * (the lo.lo term is dropped).  This replaces the reference's plain fp32 `nn.Linear` arithmetic (nets/ShfitScaleFormer.py:58-66,
 * :115-132) where bf16 alone misses the 1e-3 tolerance.  src fp32 [rows, cols] with leading dimension ld; dst bf16:
 *   stack = 0: [rows, 3*cols], piece j at column offset j*cols (the contraction runs along a row: NT's A and B, NN's A);
 *   stack = 1: [3*rows, cols], piece j at row offset j*rows    (the contraction runs down the rows: NN's B, TN's A and B).
 * pattern bit j set = piece j is the lo part: 0b100 for the left operand (hi, hi, lo), 0b010 for the right (hi, lo, hi).
 * cols % 4 == 0. */
int dm_split_bf16(const float *src, int64_t ld, int64_t rows, int64_t cols, void *dst, int32_t stack, int32_t pattern, void *stream);
/* The same with the column sums of src on the side (the bias gradient db = colsum(dy) that goes with dW = dy^T x,
 * nets/ShfitScaleFormer.py:58-66 under autograd: dy is read once for its split image and its sums): partial receives
 * *n_partial rows of `cols` floats (dm_split_colsum_partial_floats(rows, cols) floats at most), to be summed in row order --
 * e.g. by dm_partial_reduce_batch.  cols % 8 == 0, ld % 4 == 0, 16-byte aligned tensors (DM_ERR_UNSUPPORTED otherwise). */
int64_t dm_split_colsum_partial_floats(int64_t rows, int64_t cols);
/* The hi / lo PLANE PAIR of src: dst bf16 [2, rows, cols] (plane 0 = hi, plane 1 = lo, each with leading dimension cols), the operand
 * format of DmGemmArgs.k_fold.  partial / n_partial as for dm_split_bf16_colsum, or both NULL (no column sums).
 * cols % 8 == 0, ld % 4 == 0, 16-byte aligned tensors (DM_ERR_UNSUPPORTED otherwise). */
int dm_split_bf16_planes(const float *src, int64_t ld, int64_t rows, int64_t cols, void *dst, float *partial, int32_t *n_partial, void *stream);
int dm_split_bf16_colsum(const float *src, int64_t ld, int64_t rows, int64_t cols, void *dst, int32_t stack, int32_t pattern,
                         float *partial, int32_t *n_partial, void *stream);

/* Patch extraction for the k=stride Conv2d of PatchEmbed (nets/ShfitScaleFormer.py:25, :35):
 * x [B, C, side, side] fp32 -> cols [B*(side/p)^2, C*p*p] T, column order (c, dy, dx) = the
 * conv weight's [out, C, p, p] flattening, row order (b, py, px) = flatten(2).transpose(1,2). */
int dm_patchify(const float *x, void *cols, int32_t dtype, int32_t B, int32_t C, int32_t side, int32_t p, void *stream);

/* ---- loss / optimiser -------------------------------------------------------------------- */
/* Losses.py:34-38: d = sum((a-b)^2, 1); l = flag*d + (1-flag)*relu(margin-d); loss = mean(l).
 * Writes loss[0], and (if da/db != NULL) da = upstream*dloss/da, db likewise.  flag: fp32 [B]. */
int dm_contrastive_loss(const float *a, const float *b, const float *flag, float margin, float upstream,
                        float *loss, float *da, float *db, int32_t B, int32_t D, void *stream);

/* nn.CrossEntropyLoss, mean reduction (Losses.py:52-53 `MultiLoss.Loss_Class`, :83-84 `ClassLoss.Loss_Class`):
 * logits fp32 [B,K]; targets either class indices (int64 [B], target_prob NULL) or class probabilities
 * (fp32 [B,K], target_index NULL).  Writes loss[0] and, if dlogits != NULL, upstream * dloss/dlogits.
 * Out-of-range class indices are the caller's responsibility (torch raises; this reads out of bounds). */
int dm_cross_entropy(const float *logits, const int64_t *target_index, const float *target_prob, float upstream,
                     float *loss, float *dlogits, int32_t B, int32_t K, void *stream);

/* torch.optim.Adam single step over a flat fp32 buffer (Train_SMT.py:192-193, :300): in-place on
 * param/m/v; `step` is 1-based; if param_lp != NULL also writes the bf16 copy of the new weights.
 * grad_scale multiplies the gradient first (1/world_size after an RCCL sum all-reduce).
 * Hyper-parameters are doubles: torch derives 1-beta and the bias corrections in double precision
 * before rounding to fp32, and 1-0.999f differs from (float)(1-0.999) by 1.3e-5 relative. */
int dm_adam_step(float *param, const float *grad, float *m, float *v, void *param_lp, int64_t n,
                 int32_t step, double lr, double beta1, double beta2, double eps, double grad_scale, void *stream);

/* The same step for use inside a captured hipGraph: the two step-dependent scalars come from device memory,
 * hyper_dev = {lr / (1 - beta1^step), sqrt(1 - beta2^step)} (fp32[2]), which the host rewrites before each replay;
 * dm_adam_hyper computes that pair on the host exactly as dm_adam_step does. */
int dm_adam_hyper(int32_t step, double lr, double beta1, double beta2, float *hyper_host);
int dm_adam_step_dev(float *param, const float *grad, float *m, float *v, void *param_lp, int64_t n, const float *hyper_dev,
                     double beta1, double beta2, double eps, double grad_scale, void *stream);
/* ABI 6: the same, and the updated weights also as the hi / lo bf16 plane pair of the "bf16x3" products (param_hi[i] = bf16(p),
 * param_lo[i] = bf16(p - param_hi[i]): dm_split_bf16_planes' split): the next step's folded GEMMs read the pair the optimizer left instead of
 * splitting every weight matrix again (24 launches per step of the headline model). */
int dm_adam_step_dev_pair(float *param, const float *grad, float *m, float *v, void *param_hi, void *param_lo, int64_t n,
                          const float *hyper_dev, double beta1, double beta2, double eps, double grad_scale, void *stream);

/* ---- ExtractFeatures sweep ---------------------------------------------------------------- */
/* Per-superpixel mean pooling (ExtractFeatures.py:190-212): F [P,D] fp32, CSR ptr[S+1] / idx[*]
 * (int32) -> pooled [S,D]; rows are added in idx order then divided by the count (np.mean axis 0). */
int dm_segment_mean(const float *F, const int32_t *ptr, const int32_t *idx, float *pooled,
                    int32_t S, int32_t D, void *stream);
/* Per-edge distance (ExtractFeatures.py:139-147, :215-216): simi[e] = sqrt(max(0, |a|^2+|b|^2-2a.b))
 * for a = pooled[L], b = pooled[R]; edges int32 [E,2]; an edge with L == -1 or R == -1 yields NaN and
 * merge 0 (MyUtils2.py:184-186 skips them).  merge[e] = simi[e] < margin (uint8), may be NULL.
 * Summation order is fixed and documented in oracle/sweep_strict.c (bit-exact contract). */
int dm_edge_similarity(const float *pooled, const int32_t *edges, float *simi, uint8_t *merge,
                       int32_t E, int32_t D, float margin, void *stream);

/* ---- patch pyramid gather --------------------------------------------------------------------
 * Replaces, for one scale, the per-point loader work of MyUtils1.py:116-223 / MyUtils2.py:286-437:
 * calculate_left_top_point_and_size (top-left = int(mid - L/2), truncation toward zero), cut_image (window
 * clipped to the raster, zero padded) and resize_data (per band resize to target x target on uint8, /255).
 * resize_rule: DM_RESIZE_OPENCV = cv::resize(..., INTER_AREA) on uint8 restated branch by branch (integer-ratio fast path incl. the
 * half-up 2x case, float area tables, 11-bit fixed-point bilinear with area coordinates when the window is smaller than the target);
 * DM_RESIZE_EXACT_AREA = the exact rational area average of rounds 1-2.  Both are specified in oracle/patches.py (no cv2 fixture
 * exists: the OpenCV rule is faithful to the published algorithm, unpinned against a build).
 *   tile [bands,H,W] uint8; xy int32 [P,2] = (XPixel, YLine); windows int32 [P] = window side L for this scale,
 *   every L <= max_window <= 384; out float32 [P, bands, target, target]. */
#define DM_RESIZE_OPENCV 0
#define DM_RESIZE_EXACT_AREA 1
int dm_patch_pyramid(const uint8_t *tile, int32_t bands, int32_t H, int32_t W, const int32_t *xy, const int32_t *windows,
                     int32_t max_window, int32_t P, int32_t target, int32_t resize_rule, float *out, void *stream);
/* The same gather emitting the patch-embed GEMM's operand rows directly ("patch pyramid gather fused into
 * patch-embed"): cols [P * grid * grid, bands * ps * ps] with ps = target / grid, row (p * grid + py) * grid + px, column
 * (c * ps + dy) * ps + dx -- the im2col order of Conv2d(k = ps, stride = ps) (nets/ShfitScaleFormer.py:28-37) -- in bf16 or
 * fp32.  Equal, bit for bit, to dm_patch_pyramid followed by dm_patchify. */
int dm_patch_pyramid_cols(const uint8_t *tile, int32_t bands, int32_t H, int32_t W, const int32_t *xy, const int32_t *windows,
                          int32_t max_window, int32_t P, int32_t target, int32_t grid, int32_t resize_rule, void *cols, int32_t dtype,
                          void *stream);

/* ---- training feed: one scale of a pair batch from resident tiles and a DEVICE sample table (ABI 5) -----------------------
 * Replaces, for the training loop, the host-side batch assembly of Train_SMT.py:212-262 (DataLoader items of MyUtils1.py:41-77:
 * get_scales :130-156, the crop / pad / resize chain :116-223, the designed-feature row :60-77) without any host round trip:
 * no window arithmetic on the host, no per-tile launches, nothing read back.  Sample p is cut from tile tile_id[p] (tiles uint8
 * [n_tiles, bands, H, W], tile_id may be NULL: one tile) around pixel xy[p] with the window side the reference derives from
 * inner[p] / obj[p]: (inner, obj, obj + (obj - inner), obj + 2 (obj - inner))[scale_index].  Same crop / pad / resize arithmetic
 * and results as dm_patch_pyramid / dm_patch_pyramid_cols, bit for bit.
 *   grid == 0: out float32 [P, bands, target, target];  grid > 0: out = patch-embed rows [P * grid * grid, bands * ps * ps],
 *   dtype DM_BF16 / DM_F32 (the layout of dm_patch_pyramid_cols).
 *   region_features [P, 15] + designed [P, 19] (both or neither): designed[p] = region_features[p] || window sides / (32, 64, 128, 1)
 *   (patches.CONFIG_SCALES, config.py:32) -- the `designed features` tensor of MyUtils1.py:74-77.
 *   max_window bounds the LDS staging (<= 384).  A sample whose tile id or window side is out of range produces zeros and sets
 *   error_flag[0] |= 1 (int32 on the device, may be NULL): the caller reads it when it reads the loss, not per step. */
int dm_pair_batch_gather(const uint8_t *tiles, int32_t n_tiles, int32_t bands, int32_t H, int32_t W, const int32_t *tile_id,
                         const int32_t *xy, const int32_t *inner, const int32_t *obj, int32_t scale_index, int32_t max_window,
                         int32_t P, int32_t target, int32_t grid, int32_t resize_rule, void *out, int32_t dtype,
                         const float *region_features, float *designed, int32_t *error_flag, void *stream);

/* ---- region-adjacency graph + superpixel statistics from a label raster ---------------------------
 * Replaces, on the device, the inputs the reference reads from files written by external GIS software: the RAG edge
 * list (`LEFT_FID` / `RIGHT_FID` of lines.shp, MyUtils2.py:155-193, consumed at ExtractFeatures.py:188-219) and the 15
 * designed attributes (MyUtils1.py:79-114).  The definitions are this build's (oracle/rag.py); all results are exact
 * integers or fixed double-precision formulas of exact integers.
 *   labels int32 [H,W] (superpixel id, ids outside [0,S) are ignored); tile uint8 [bands,H,W].
 * dm_label_stats: count[S], sum / sumsq [S, min(bands,3)] (int64), bbox int32 [S,4] = xmin,ymin,xmax,ymax
 *   (INT_MAX,INT_MAX,-1,-1 for an id that never occurs), peri int64 [S,2] = pixel edges shared with another label /
 *   lying on the raster border.  All outputs are (re)initialised by the call.
 * dm_label_features: float32 [S,15] = area, peri, len, width, smooth, std0..2, mean0..2, shapeness, compact, bright,
 *   border (the attribute order of MyUtils1.py:79-114).
 * dm_rag_edges: unique unordered 4-neighbour label pairs a < b as keys a*S+b with the number of shared pixel edges.
 *   table_keys / table_counts: scratch of 2^capacity_log2 entries (int64 / int32); edge_keys / edge_counts: up to
 *   max_edges results in ARBITRARY order (sort by key for a canonical list); n_edges[0] = number found (may exceed
 *   max_edges: then the output is truncated); overflow[0] = 1 if the table was too small. */
int dm_label_stats(const int32_t *labels, const uint8_t *tile, int32_t bands, int32_t H, int32_t W, int32_t S,
                   int64_t *count, int64_t *sum, int64_t *sumsq, int32_t *bbox, int64_t *peri, void *stream);
int dm_label_features(const int64_t *count, const int64_t *sum, const int64_t *sumsq, const int32_t *bbox, const int64_t *peri,
                      int32_t S, int32_t bands, float *features, void *stream);
int dm_rag_edges(const int32_t *labels, int32_t H, int32_t W, int32_t S, int64_t *table_keys, int32_t *table_counts,
                 int32_t capacity_log2, int64_t *edge_keys, int32_t *edge_counts, int32_t max_edges, int32_t *n_edges,
                 int32_t *overflow, void *stream);

/* One round of the merge step that follows the sweep (optional; the reference leaves merging to external
 * GIS tooling): union-find over the edges with merge[e] != 0.  parent int32 [S] (init != 0: reset to the identity first);
 * after the call parent[s] is a root candidate and changed[0] tells whether any root was hooked -- repeat with init = 0
 * until changed[0] == 0; then parent[s] = the smallest superpixel id of s's component (order-independent). */
int dm_merge_round(const int32_t *edges, const uint8_t *merge, int32_t E, int32_t S, int32_t *parent, int32_t *changed,
                   int32_t init, void *stream);

/* BatchNorm2d (+ ReLU, + Dropout2d mask) of the auxiliary heads (reference nets/ShfitScaleFormer.py:329-368: Conv2d ->
 * BatchNorm2d -> ReLU -> Dropout2d(0.3)) on the channels-last matrix the convolution GEMM produces: x, y fp32 [M, C] with
 * M = samples * rows_per_sample.  training != 0: batch statistics (biased variance, eps inside the sqrt), running_mean /
 * running_var updated in place with `momentum` (unbiased variance), as torch.nn.BatchNorm2d; training == 0: the running
 * statistics.  mask: NULL or fp32 [samples, C] multipliers (0 or 1/(1-p): Dropout2d drops whole channels of a sample);
 * relu != 0 applies max(0, .) between the normalisation and the mask.  save_mean / save_rstd [C] are kept for backward.
 * workspace: dm_batchnorm_workspace_bytes(M, C), 8-byte aligned. */
int64_t dm_batchnorm_workspace_bytes(int32_t M, int32_t C);
int dm_batchnorm_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                     const float *mask, int32_t rows_per_sample, float *y, float *save_mean, float *save_rstd, int32_t M,
                     int32_t C, float eps, float momentum, int32_t training, int32_t relu, void *workspace, void *stream);
/* Gradient of the above (relu != 0: y > 0 decides the ReLU branch; pass the forward call's relu / training / mask): dx [M, C];
 * dgamma / dbeta [C] written, or added to when accumulate != 0. */
int dm_batchnorm_bwd(const float *dy, const float *x, const float *y, const float *gamma, const float *mask, int32_t rows_per_sample,
                     const float *save_mean, const float *save_rstd, float *dx, float *dgamma, float *dbeta, int32_t accumulate,
                     int32_t M, int32_t C, int32_t training, int32_t relu, void *workspace, void *stream);

/* ---- GRU cell (reference Nets.py:60-66: `nn.GRU(28, 80, num_layers=4, bidirectional=True)` of the MNIST sandbox net `RNN`) ----
 * PyTorch gate order r, z, n.  gi [B, 3H] with row stride gi_stride floats (a time slice of x W_ih^T + b_ih for all steps),
 * gh [B, 3H] = h W_hh^T + b_hh, h [B, H]:  r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n),
 * h_new = (1 - z) * n + z * h.  saved [B, 4H] keeps r, z, n, gh_n for the backward call, which returns dgi, dgh [B, 3H] and
 * dh [B, H] (the direct path to the previous state; the caller adds the path through gh). */
int dm_gru_cell_fwd(const float *gi, int64_t gi_stride, const float *gh, const float *h, float *h_new, float *saved,
                    int32_t B, int32_t H, void *stream);
int dm_gru_cell_bwd(const float *dh_new, const float *saved, const float *h, float *dgi, float *dgh, float *dh, int32_t B,
                    int32_t H, void *stream);

/* ---- optional in-library kernel timing ------------------------------------------------------
 * While enabled, the GEMM and attention entry points bracket their main kernel with hipEvents on
 * the caller's stream.  dm_prof_collect waits for the recorded events, aggregates them per kernel
 * name (launch count, total milliseconds, total algorithmic FLOPs and bytes as computed from the
 * call's dimensions) and clears the log.  Used by bench.py for the roofline figure. */
typedef struct {
  char name[64];
  int64_t launches;
  double total_ms;
  double total_flops;
  double total_bytes;
} DmProfRow;
int dm_prof_enable(int32_t on);
int32_t dm_prof_collect(DmProfRow *rows, int32_t max_rows);

#ifdef __cplusplus
}
#endif
#endif /* DEEPMERGE_HIP_H */
