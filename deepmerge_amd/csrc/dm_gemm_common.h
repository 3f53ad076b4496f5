// Pieces shared by the GEMM kernels (dm_gemm.hip: 128x128 / 64x64 register-staged tiles; dm_gemm256.hip: the
// 256x256 LDS-DMA pipeline): launch parameters and the fused epilogue.
#pragma once
#include "dm_common.h"

struct GemmParams {
  const void *A, *B;
  void *C;
  const float *bias;
  const float *residual;
  void *aux;
  long long lda, ldb, ldc, ldr, ldaux, group_stride;
  int M, N, K;
  int epilogue, accumulate, c_dtype, aux_dtype, rows_per_group;
  int tiles_m, tiles_n, split_k, k_per_split;
  int group_m;   // > 0: tiles are walked in bands of group_m row tiles, rows fastest inside a band (L2 reuse)
  float *workspace;
  int debug;     // ring kernel ablations (-DDM_RING_ABLATE builds, DM_RING_DEBUG: 1 no DMA in the loop, 2 no MFMA, 4 no fragment reads, 8 no stores); 0 in production
  float *colsum_slab;   // TN pipeline: partial column sums of A, [split_k * 4][M] (NULL: not wanted)
  // Folded contraction ("bf16x3" products on hi / lo plane pairs, DmGemmArgs.k_fold): K = 3 * k_fold; K segment s of A starts a_fold[s]
  // elements behind A (b_fold: B) and is addressed inside the segment as a plain operand of contraction length k_fold.  0: plain.
  int k_fold;
  long long a_fold[3], b_fold[3];
  long long c_plane;    // c_dtype == DM_BF16_PAIR: element offset of the lo plane behind C (the hi plane)
};

// dm_gemm_grouped -> dm_gemm_w4_grouped: where the column sums of A go (stream-K form) and the group's workspace
struct DmGroupedExtra {
  float *cs_out[8]; int cs_acc[8]; void *ws; long long ws_bytes;
  float *slab[8]; long long slab_bytes[8]; float *cs_region[8];      // the sliced form: every product's own split-K slab / column-sum rows (its dm_gemm workspace)
};

// The result strip as a hi / lo plane pair (c_dtype == DM_BF16_PAIR): hi = bf16(v), lo = bf16(v - hi) -- the split of dm_split_bf16.
__device__ __forceinline__ void dm_store_pair4(const GemmParams &p, long long off, const f32x4 &v) {
  bf16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (bf16_t)v[e];
    l[e] = (bf16_t)(v[e] - (float)h[e]);
  }
  bf16_t *c = reinterpret_cast<bf16_t *>(p.C) + off;
  *reinterpret_cast<bf16x4 *>(c) = h;
  *reinterpret_cast<bf16x4 *>(c + p.c_plane) = l;
}

// Folded contraction + column sums of A (the bias gradient on a weight gradient over plane pairs): a K position counts when its
// segment is the FIRST one that reads its piece of A -- (hi, hi, lo): segments 0 and 2, so that the sums are colsum(hi) + colsum(lo).
__device__ __forceinline__ bool dm_fold_counts(const GemmParams &p, int k) {
  if (p.k_fold <= 0 || k < p.k_fold) return true;
  if (k < 2 * p.k_fold) return p.a_fold[1] != p.a_fold[0];
  return p.a_fold[2] != p.a_fold[0] && p.a_fold[2] != p.a_fold[1];
}

struct DmGemmRow { long long c, r, x; };

// Element offsets of output row m in C / residual / aux (grouped-row addressing writes patch tokens straight
// into the token cube: rows_per_group consecutive rows share a base that advances by group_stride).
__device__ __forceinline__ DmGemmRow dm_gemm_row(const GemmParams &p, int m) {
  DmGemmRow rb;
  if (p.rows_per_group > 0) {
    const long long grp = m / p.rows_per_group, rr = m % p.rows_per_group;
    rb.c = grp * p.group_stride + rr * p.ldc;
    rb.r = grp * p.group_stride + rr * p.ldr;
    rb.x = grp * p.group_stride + rr * p.ldaux;
  } else {
    rb.c = (long long)m * p.ldc;
    rb.r = (long long)m * p.ldr;
    rb.x = (long long)m * p.ldaux;
  }
  return rb;
}

// One 1x4 strip (row rb, columns n..n+3) of the accumulator through the fused epilogue: bias, GELU (+ saved
// pre-activation) or GELU', fp32 residual add, accumulate, bf16 / fp32 store.  FAST selects the bf16-mode GELU.
template <bool FAST>
__device__ __forceinline__ void dm_gemm_emit(const GemmParams &p, f32x4 v, const DmGemmRow &rb, int n) {
  if (p.bias) v += dm_load4(p.bias + n);
  if (p.epilogue == DM_EPI_GELU) {
    if (p.aux) {      // aux == NULL: inference, nothing is saved for a backward pass
      if (p.aux_dtype == DM_F32) dm_store4(reinterpret_cast<float *>(p.aux) + rb.x + n, v);
      else dm_store4(reinterpret_cast<bf16_t *>(p.aux) + rb.x + n, v);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = FAST ? dm_gelu_fast(v[e]) : dm_gelu(v[e]);
  } else if (p.epilogue == DM_EPI_GELU_GRAD) {
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (FAST) {
        float cdf, pdf;
        dm_gelu_parts_fast(v[e], cdf, pdf);          // one exp for both
        d[e] = fmaf(v[e], pdf, cdf);
        v[e] = v[e] * cdf;
      } else {
        d[e] = dm_dgelu(v[e]);
        v[e] = dm_gelu(v[e]);
      }
    }
    if (p.aux_dtype == DM_F32) dm_store4(reinterpret_cast<float *>(p.aux) + rb.x + n, d);
    else dm_store4(reinterpret_cast<bf16_t *>(p.aux) + rb.x + n, d);
  } else if (p.epilogue == DM_EPI_DGELU || p.epilogue == DM_EPI_MUL) {
    f32x4 u = (p.aux_dtype == DM_F32) ? dm_load4(reinterpret_cast<const float *>(p.aux) + rb.x + n)
                                      : dm_load4(reinterpret_cast<const bf16_t *>(p.aux) + rb.x + n);
    if (p.epilogue == DM_EPI_MUL) {
      v *= u;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= FAST ? dm_dgelu_fast(u[e]) : dm_dgelu(u[e]);
    }
  }
  if (p.residual) v += dm_load4(p.residual + rb.r + n);
  if (p.c_dtype == DM_F32) {
    float *c = reinterpret_cast<float *>(p.C) + rb.c + n;
    if (p.accumulate) v += dm_load4(c);
    dm_store4(c, v);
  } else if (p.c_dtype == DM_BF16) {
    dm_store4(reinterpret_cast<bf16_t *>(p.C) + rb.c + n, v);
  } else {
    dm_store_pair4(p, rb.c + n, v);
  }
}

// The same strip in two steps, for kernels that keep the MFMA layout in their epilogue (64 x 64 tiles): dm_gemm_emit loads its read
// operands (bias, residual, aux, old C) right before it stores, so in a sequence of strips every load sits behind the previous
// strip's store and waits for that store's acknowledgement (vmcnt retires in order): a memory round trip per strip.  The caller
// issues the loads of ALL its strips first (dm_gemm_strip_load), then computes and stores (dm_gemm_strip_store).
struct DmStripPre { f32x4 bias, res, y; };
__device__ __forceinline__ void dm_gemm_strip_load(const GemmParams &p, const DmGemmRow &rb, int n, DmStripPre &pre) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  pre.bias = p.bias ? dm_load4(p.bias + n) : zero;
  pre.res = p.residual ? dm_load4(p.residual + rb.r + n) : zero;
  pre.y = zero;
  if (p.epilogue == DM_EPI_DGELU || p.epilogue == DM_EPI_MUL)
    pre.y = (p.aux_dtype == DM_F32) ? dm_load4(reinterpret_cast<const float *>(p.aux) + rb.x + n) : dm_load4(reinterpret_cast<const bf16_t *>(p.aux) + rb.x + n);
  else if (p.c_dtype == DM_F32 && p.accumulate)
    pre.y = dm_load4(reinterpret_cast<const float *>(p.C) + rb.c + n);
}
template <bool FAST>
__device__ __forceinline__ void dm_gemm_strip_store(const GemmParams &p, f32x4 v, const DmGemmRow &rb, int n, const DmStripPre &pre) {
  v += pre.bias;
  if (p.epilogue == DM_EPI_GELU) {
    if (p.aux) {
      if (p.aux_dtype == DM_F32) dm_store4(reinterpret_cast<float *>(p.aux) + rb.x + n, v);
      else dm_store4(reinterpret_cast<bf16_t *>(p.aux) + rb.x + n, v);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = FAST ? dm_gelu_fast(v[e]) : dm_gelu(v[e]);
  } else if (p.epilogue == DM_EPI_GELU_GRAD) {
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (FAST) {
        float cdf, pdf;
        dm_gelu_parts_fast(v[e], cdf, pdf);
        d[e] = fmaf(v[e], pdf, cdf);
        v[e] = v[e] * cdf;
      } else {
        d[e] = dm_dgelu(v[e]);
        v[e] = dm_gelu(v[e]);
      }
    }
    if (p.aux_dtype == DM_F32) dm_store4(reinterpret_cast<float *>(p.aux) + rb.x + n, d);
    else dm_store4(reinterpret_cast<bf16_t *>(p.aux) + rb.x + n, d);
  } else if (p.epilogue == DM_EPI_MUL) {
    v *= pre.y;
  } else if (p.epilogue == DM_EPI_DGELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= FAST ? dm_dgelu_fast(pre.y[e]) : dm_dgelu(pre.y[e]);
  }
  v += pre.res;
  if (p.c_dtype == DM_F32) {
    if (p.accumulate) v += pre.y;      // (accumulate never goes with an aux-reading epilogue: dm_gemm refuses the combination)
    dm_store4(reinterpret_cast<float *>(p.C) + rb.c + n, v);
  } else if (p.c_dtype == DM_BF16) {
    dm_store4(reinterpret_cast<bf16_t *>(p.C) + rb.c + n, v);
  } else {
    dm_store_pair4(p, rb.c + n, v);
  }
}

// ---- whole-line epilogue shared by the LDS-DMA kernels (dm_gemm_ring.hip, dm_gemm256.hip) ----------------------------------
// A wave owns a (WM * 16) x 64 block of outputs as acc[WM][4] (row i * 16 + (lane & 15), columns j * 16 + 4 * (lane >> 4) ..+3).
// Stored from that layout an instruction touches 16 rows x 32..64 B.  Instead the wave transposes ROWS rows at a time through a
// private LDS region (rows padded to DM_EPI_PITCH bytes: ds_write_b128 of 8 consecutive lanes then hits 8 distinct 16-byte
// slots) and walks them row by row, 8 consecutive columns per lane: bias / residual / aux reads and the C stores are whole
// 128-byte lines.
constexpr int DM_EPI_PITCH = 272;

// 8 consecutive outputs of row (rb) starting at column n through the fused epilogue (same semantics as dm_gemm_emit).
__device__ __forceinline__ void dm_gemm_emit8(const GemmParams &p, f32x4 lo, f32x4 hi, const DmGemmRow &rb, int n) {
  if (p.bias) { lo += dm_load4(p.bias + n); hi += dm_load4(p.bias + n + 4); }
  if (p.epilogue == DM_EPI_GELU) {
    if (p.aux) {
      if (p.aux_dtype == DM_F32) { dm_store4(reinterpret_cast<float *>(p.aux) + rb.x + n, lo); dm_store4(reinterpret_cast<float *>(p.aux) + rb.x + n + 4, hi); }
      else {
        bf16x8 o = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
        *reinterpret_cast<bf16x8 *>(reinterpret_cast<bf16_t *>(p.aux) + rb.x + n) = o;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { lo[e] = dm_gelu_fast(lo[e]); hi[e] = dm_gelu_fast(hi[e]); }
  } else if (p.epilogue == DM_EPI_GELU_GRAD) {
    f32x4 dl, dh;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float cdf, pdf;
      dm_gelu_parts_fast(lo[e], cdf, pdf);
      dl[e] = fmaf(lo[e], pdf, cdf);
      lo[e] = lo[e] * cdf;
      dm_gelu_parts_fast(hi[e], cdf, pdf);
      dh[e] = fmaf(hi[e], pdf, cdf);
      hi[e] = hi[e] * cdf;
    }
    if (p.aux_dtype == DM_F32) { dm_store4(reinterpret_cast<float *>(p.aux) + rb.x + n, dl); dm_store4(reinterpret_cast<float *>(p.aux) + rb.x + n + 4, dh); }
    else {
      bf16x8 o = {(bf16_t)dl[0], (bf16_t)dl[1], (bf16_t)dl[2], (bf16_t)dl[3], (bf16_t)dh[0], (bf16_t)dh[1], (bf16_t)dh[2], (bf16_t)dh[3]};
      *reinterpret_cast<bf16x8 *>(reinterpret_cast<bf16_t *>(p.aux) + rb.x + n) = o;
    }
  } else if (p.epilogue == DM_EPI_DGELU || p.epilogue == DM_EPI_MUL) {
    f32x4 ul, uh;
    if (p.aux_dtype == DM_F32) { ul = dm_load4(reinterpret_cast<const float *>(p.aux) + rb.x + n); uh = dm_load4(reinterpret_cast<const float *>(p.aux) + rb.x + n + 4); }
    else {
      const bf16x8 u = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const bf16_t *>(p.aux) + rb.x + n);
      ul = (f32x4){(float)u[0], (float)u[1], (float)u[2], (float)u[3]};
      uh = (f32x4){(float)u[4], (float)u[5], (float)u[6], (float)u[7]};
    }
    if (p.epilogue == DM_EPI_MUL) { lo *= ul; hi *= uh; }
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo[e] *= dm_dgelu_fast(ul[e]); hi[e] *= dm_dgelu_fast(uh[e]); }
    }
  }
  if (p.residual) { lo += dm_load4(p.residual + rb.r + n); hi += dm_load4(p.residual + rb.r + n + 4); }
  if (p.c_dtype == DM_F32) {
    float *c = reinterpret_cast<float *>(p.C) + rb.c + n;
    if (p.accumulate) { lo += dm_load4(c); hi += dm_load4(c + 4); }
    dm_store4(c, lo);
    dm_store4(c + 4, hi);
  } else if (p.c_dtype == DM_BF16) {
    bf16x8 o = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
    *reinterpret_cast<bf16x8 *>(reinterpret_cast<bf16_t *>(p.C) + rb.c + n) = o;
  } else {      // hi / lo plane pair: two 16-byte stores, as an fp32 row piece would take
    bf16x8 h = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
    bf16x8 l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      l[e] = (bf16_t)(lo[e] - (float)h[e]);
      l[4 + e] = (bf16_t)(hi[e] - (float)h[4 + e]);
    }
    bf16_t *c = reinterpret_cast<bf16_t *>(p.C) + rb.c + n;
    *reinterpret_cast<bf16x8 *>(c) = h;
    *reinterpret_cast<bf16x8 *>(c + p.c_plane) = l;
  }
}


// Staging layout of one wave: rows of PITCH bytes holding 64 fp32 columns.  SWZ = false: padded rows (PITCH = DM_EPI_PITCH = 272),
// chunk c of a row at c * 16; SWZ = true: exact rows (PITCH = 256, for kernels whose LDS is full), the 16-byte chunk c of row r at
// slot c ^ (r & 15) -- both patterns (ds_write_b128 of an accumulator tile: 16 rows x 4 chunks; ds_read_b128 of a row's chunk pair by 8
// lanes x 8 rows) are conflict-free in the bank model of the guide.
template <int PITCH, bool SWZ> __device__ __forceinline__ int dm_epi_slot(int row, int chunk) {
  return row * PITCH + ((SWZ ? (chunk ^ (row & 15)) : chunk) << 4);
}

template <int WM, int ROWS, bool SKIP_STORES = false, int PITCH = DM_EPI_PITCH, bool SWZ = false>
__device__ __forceinline__ void dm_epilogue_rows_generic(const GemmParams &p, f32x4 (&acc)[WM][4], char *mine, int m_wave, int n_wave, int lane) {
  const int g = lane >> 4, li = lane & 15;
  constexpr int PASS_TILES = ROWS / 16;
#pragma unroll
  for (int ps = 0; ps < WM / PASS_TILES; ++ps) {
#pragma unroll
    for (int ii = 0; ii < PASS_TILES; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4 *>(mine + dm_epi_slot<PITCH, SWZ>(ii * 16 + li, j * 4 + g)) = acc[ps * PASS_TILES + ii][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int n = n_wave + (lane & 7) * 8;
#pragma unroll
    for (int r = 0; r < ROWS / 8; ++r) {
      const int row = r * 8 + (lane >> 3);
      const int m = m_wave + ps * ROWS + row;
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(mine + dm_epi_slot<PITCH, SWZ>(row, (lane & 7) * 2));
      const f32x4 hi = *reinterpret_cast<const f32x4 *>(mine + dm_epi_slot<PITCH, SWZ>(row, (lane & 7) * 2 + 1));
      if constexpr (SKIP_STORES) { if (lo[0] == 12345.678f && m < p.M) dm_gemm_emit8(p, lo, hi, dm_gemm_row(p, m), n); }
      else if (m < p.M && n < p.N) dm_gemm_emit8(p, lo, hi, dm_gemm_row(p, m), n);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next pass overwrites the region
    __builtin_amdgcn_sched_barrier(0);
  }
}


// ---- lean whole-line epilogue (round 4) ------------------------------------------------------------------------------------------
// Same arithmetic as dm_gemm_emit8 row by row, restructured after reading the ISA of the form above (tools/isa_hazards.py listings):
//   * the generic form pays ~25 quarter-rate integer multiplies per 8-column item (64-bit row offsets, the grouped-row division)
//     and, worse, every item's bias / residual / aux LOADS sit behind the previous item's STORES in program order: vmcnt retires in
//     order, so each item waits for a store acknowledgement (a memory round trip per 8 rows: 8-24 of them per tile);
//   * here a wave builds three buffer descriptors per tile (C, residual, aux: base = the wave's first row / column, extent = the
//     valid rows below it, so rows past M are dropped / read zero in hardware and the code has no branches), per-lane byte offsets
//     are computed ONCE (row lane >> 3, columns 8 (lane & 7)), an item only advances a SCALAR offset (8 rows), the bias is loaded
//     once per tile, and the loads of item q + 1 are issued BEFORE the stores of item q (software pipeline over the tile's items):
//     no load ever waits for a store.
// Preconditions (the caller falls back to the generic form otherwise): plain rows (rows_per_group == 0) and 32-bit offsets inside a
// wave's block (16 WM rows x leading dimension x 4 B < 2^31).
// A 16-byte buffer store followed closely by a VALU write of its data registers: observed on gfx950 (exact-integer GELU + aux test,
// ~1 wave-instruction in 3000) that the LAST FOUR LANES of each 16-lane group stored the new value of a data register that a VALU
// instruction three slots behind the store overwrote -- beyond the one wait state the ISA manual asks for and the compiler
// provides.  The data registers are therefore kept alive (and idle) for 16 more cycles behind every store (8 made the test pass; the margin is cheap: < 1 % of an item).
#ifndef DM_EPI_STORE_NOP
#define DM_EPI_STORE_NOP "s_nop 7\n\ts_nop 7"
#endif
// DM_EPI_STORE_AUX: cache policy of the epilogue's stores (experiment builds: 2 = nt, 16 = sc1, 17 = sc0 sc1; default 0)
#ifndef DM_EPI_STORE_AUX
#define DM_EPI_STORE_AUX 0
#endif
// The operand an epilogue SAVES for the backward pass (fc1's GELU' / pre-activation: 100 MB per stage-0 block) is not read again before
// the backward pass: non-temporal stores (nt) keep it from displacing what the next kernels read (-DDM_EPI_AUX_POLICY=0 for A/B builds).
#ifndef DM_EPI_AUX_POLICY
#define DM_EPI_AUX_POLICY 2
#endif
// cache policy of the epilogue's read-once operands (fp32 residual, the saved GELU'): nt like the saved operand's stores.  Step A/B on one box,
// 100 steps x 3 alternating rounds: default 5.807 / 5.884 / 5.796 ms, nt aux stores 5.803 / 5.797 / 5.867, + nt loads 5.792 / 5.785 / 5.790 (-D...=0 for A/B builds)
#ifndef DM_EPI_LOAD_POLICY
#define DM_EPI_LOAD_POLICY 2
#endif
// Experiment builds (tools/epi_store_hazard.sh, profiles/r05_epi_store_hazard.txt): -DDM_EPI_SOFF_V=1 adds the scalar row step to the
// VECTOR offset and leaves soffset 0 -- the form for which LLVM's hazard recogniser pads a > 64-bit MUBUF store followed by a VALU
// write of its data registers (it deliberately does not when soffset is a register) -- and -DDM_EPI_STORE_PAD=n picks the pad
// behind the store (0 none, 1 / 2 / 4 / 8 cycles ... via s_nop).
#ifndef DM_EPI_SOFF_V
#define DM_EPI_SOFF_V 0
#endif
#ifdef DM_EPI_STORE_PAD
#undef DM_EPI_STORE_NOP
#if DM_EPI_STORE_PAD == 0
#define DM_EPI_STORE_NOP ""
#elif DM_EPI_STORE_PAD == 1
#define DM_EPI_STORE_NOP "s_nop 0"
#elif DM_EPI_STORE_PAD == 2
#define DM_EPI_STORE_NOP "s_nop 1"
#elif DM_EPI_STORE_PAD == 4
#define DM_EPI_STORE_NOP "s_nop 3"
#elif DM_EPI_STORE_PAD == 8
#define DM_EPI_STORE_NOP "s_nop 7"
#else
#define DM_EPI_STORE_NOP "s_nop 7\n\ts_nop 7"
#endif
#endif
#define DM_EPI_BSTORE(data, rsrc, vo, so, aux)                                  \
  do {                                                                          \
    const u32x4 dm_bs_ = (data);                                                \
    if (DM_EPI_SOFF_V) __builtin_amdgcn_raw_buffer_store_b128(dm_bs_, rsrc, (vo) + (unsigned)(so), 0, (aux) | DM_EPI_STORE_AUX); \
    else __builtin_amdgcn_raw_buffer_store_b128(dm_bs_, rsrc, vo, so, (aux) | DM_EPI_STORE_AUX); \
    asm volatile(DM_EPI_STORE_NOP ::"v"(dm_bs_) : "memory");                    \
  } while (0)
struct DmEpiPre { f32x4 r0, r1; u32x4 y0, y1; };      // residual; the old C (accumulate) OR the aux operand (never both: see dm_epilogue_rows)

__device__ __forceinline__ int dm_epi_records(long long bytes) { return (int)(bytes < 0 ? 0 : (bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes)); }

// RT = true: which operands exist is read from `p` at run time (any combination; the wave-uniform branches around the memory
// instructions make the compiler's waitcnt bookkeeping conservative: vmcnt(0) at the merge points).  RT = false: the STRUCTURE of an
// item -- RES: fp32 residual read; YL: 0 none / 1 old C (accumulate) / 2 aux read, bf16 / 3 aux read, fp32; C32: fp32 C; XS: 0 no aux
// store / 1 bf16 / 2 fp32 -- is a template argument, so every memory instruction is straight-line code with counted waits; only the
// arithmetic kind (GELU / GELU' / multiply) stays a run-time branch.  dm_epilogue_rows dispatches the combinations the encoder uses.
template <int WM, int ROWS, bool SKIP_STORES = false, bool RT = true, bool RES = false, int YL = 0, bool C32 = false, int XS = 0,
          int PITCH = DM_EPI_PITCH, bool SWZ = false>
__device__ __forceinline__ void dm_epilogue_rows_lean(const GemmParams &p, f32x4 (&acc)[WM][4], char *mine, int m_wave_in, int n_wave_in, int lane) {
  // wave-uniform by construction, but derived from threadIdx in some callers: without the readfirstlane the descriptors below live
  // in VGPRs and every buffer access becomes a waterfall loop
  const int m_wave = __builtin_amdgcn_readfirstlane(m_wave_in), n_wave = __builtin_amdgcn_readfirstlane(n_wave_in);
  constexpr int PASS_TILES = ROWS / 16, NPASS = WM / PASS_TILES, R = ROWS / 8, Q = NPASS * R;
  const int g = lane >> 4, li = lane & 15, c8 = lane & 7, rl = lane >> 3;
  const int n = n_wave + c8 * 8;
  const unsigned kill = (n < p.N) ? 0u : 0x80000000u;              // columns past N (N % 8 == 0: whole groups)
  const bool c32 = RT ? (p.c_dtype == DM_F32) : C32;
  const bool x32 = RT ? (p.aux_dtype == DM_F32) : (YL == 3 || XS == 2);
  const int csz = c32 ? 4 : 2, xsz = x32 ? 4 : 2;
  const bool live = m_wave < p.M;
  const long long rows_below = (long long)(p.M - 1 - m_wave), cols_right = (long long)(p.N - n_wave);
  const bool has_res = RT ? (p.residual != nullptr) : RES;
  const bool has_acc = RT ? (c32 && p.accumulate) : (YL == 1);
  const bool aux_load = RT ? (p.aux && (p.epilogue == DM_EPI_DGELU || p.epilogue == DM_EPI_MUL)) : (YL >= 2);
  const bool aux_store = RT ? (p.aux && (p.epilogue == DM_EPI_GELU || p.epilogue == DM_EPI_GELU_GRAD)) : (XS != 0);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char *>(p.C) + ((long long)m_wave * p.ldc + n_wave) * csz, 0, live ? dm_epi_records((rows_below * p.ldc + cols_right) * csz) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.residual) + (has_res ? (long long)m_wave * p.ldr + n_wave : 0), 0,
      (live && has_res) ? dm_epi_records((rows_below * p.ldr + cols_right) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char *>(p.aux) + (p.aux ? ((long long)m_wave * p.ldaux + n_wave) * xsz : 0), 0,
      (live && p.aux) ? dm_epi_records((rows_below * p.ldaux + cols_right) * xsz) : 0, 0x00020000);
  // run-time form only: the result as a hi / lo plane pair (DM_BF16_PAIR) -- a second descriptor, lo plane p.c_plane elements behind
  const bool pair = RT && p.c_dtype == DM_BF16_PAIR;
  const __amdgpu_buffer_rsrc_t rsC2 = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char *>(p.C) + ((pair ? p.c_plane : 0) + (long long)m_wave * p.ldc + n_wave) * csz, 0,
      (live && pair) ? dm_epi_records((rows_below * p.ldc + cols_right) * csz) : 0, 0x00020000);
  const unsigned voC = (unsigned)((rl * (int)p.ldc + c8 * 8) * csz) | kill;
  const unsigned voR = (unsigned)((rl * (int)p.ldr + c8 * 8) * 4) | kill;
  const unsigned voX = (unsigned)((rl * (int)p.ldaux + c8 * 8) * xsz) | kill;
  const int stepC = 8 * (int)p.ldc * csz, stepR = 8 * (int)p.ldr * 4, stepX = 8 * (int)p.ldaux * xsz;      // 8 rows
  // bias of the lane's 8 columns, once per tile; through a descriptor too (a null one reads zeros: no branch in front of the items)
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.bias), 0, p.bias ? dm_epi_records((long long)p.N * 4) : 0, 0x00020000);
  const f32x4 b_lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(n * 4) | kill, 0, 0));
  const f32x4 b_hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, ((unsigned)(n * 4) | kill) + 16, 0, 0));

  auto prefetch = [&](DmEpiPre &pre, int q) __attribute__((always_inline)) {
    if (has_res) {
      pre.r0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, voR, q * stepR, DM_EPI_LOAD_POLICY));
      pre.r1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, voR + 16, q * stepR, DM_EPI_LOAD_POLICY));
    }
    if (has_acc) {
      pre.y0 = __builtin_amdgcn_raw_buffer_load_b128(rsC, voC, q * stepC, 0);
      pre.y1 = __builtin_amdgcn_raw_buffer_load_b128(rsC, voC + 16, q * stepC, 0);
    }
    if (aux_load) {
      pre.y0 = __builtin_amdgcn_raw_buffer_load_b128(rsX, voX, q * stepX, DM_EPI_LOAD_POLICY);
      if (x32) pre.y1 = __builtin_amdgcn_raw_buffer_load_b128(rsX, voX + 16, q * stepX, DM_EPI_LOAD_POLICY);
    }
  };
  auto pack8 = [](const f32x4 &a, const f32x4 &b) __attribute__((always_inline)) {
    const bf16x8 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    return __builtin_bit_cast(u32x4, o);
  };
  auto emit = [&](f32x4 lo, f32x4 hi, const DmEpiPre &pre, int q) __attribute__((always_inline)) {
    lo += b_lo; hi += b_hi;
    if (p.epilogue == DM_EPI_GELU) {
      if (aux_store) {
        if (x32) {
          DM_EPI_BSTORE(__builtin_bit_cast(u32x4, lo), rsX, voX, q * stepX, DM_EPI_AUX_POLICY);
          DM_EPI_BSTORE(__builtin_bit_cast(u32x4, hi), rsX, voX + 16, q * stepX, DM_EPI_AUX_POLICY);
        } else {
          DM_EPI_BSTORE(pack8(lo, hi), rsX, voX, q * stepX, DM_EPI_AUX_POLICY);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo[e] = dm_gelu_fast(lo[e]); hi[e] = dm_gelu_fast(hi[e]); }
    } else if (p.epilogue == DM_EPI_GELU_GRAD) {
      f32x4 dl, dh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float cdf, pdf;
        dm_gelu_parts_fast(lo[e], cdf, pdf);
        dl[e] = fmaf(lo[e], pdf, cdf);
        lo[e] = lo[e] * cdf;
        dm_gelu_parts_fast(hi[e], cdf, pdf);
        dh[e] = fmaf(hi[e], pdf, cdf);
        hi[e] = hi[e] * cdf;
      }
      if (aux_store) {
        if (x32) {
          DM_EPI_BSTORE(__builtin_bit_cast(u32x4, dl), rsX, voX, q * stepX, DM_EPI_AUX_POLICY);
          DM_EPI_BSTORE(__builtin_bit_cast(u32x4, dh), rsX, voX + 16, q * stepX, DM_EPI_AUX_POLICY);
        } else {
          DM_EPI_BSTORE(pack8(dl, dh), rsX, voX, q * stepX, DM_EPI_AUX_POLICY);
        }
      }
    } else if (aux_load) {
      f32x4 ul, uh;
      if (x32) { ul = __builtin_bit_cast(f32x4, pre.y0); uh = __builtin_bit_cast(f32x4, pre.y1); }
      else {
        const bf16x8 u = __builtin_bit_cast(bf16x8, pre.y0);
        ul = (f32x4){(float)u[0], (float)u[1], (float)u[2], (float)u[3]};
        uh = (f32x4){(float)u[4], (float)u[5], (float)u[6], (float)u[7]};
      }
      if (p.epilogue == DM_EPI_MUL) { lo *= ul; hi *= uh; }
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] *= dm_dgelu_fast(ul[e]); hi[e] *= dm_dgelu_fast(uh[e]); }
      }
    }
    if (has_res) { lo += pre.r0; hi += pre.r1; }
    if (c32) {
      if (has_acc) { lo += __builtin_bit_cast(f32x4, pre.y0); hi += __builtin_bit_cast(f32x4, pre.y1); }
      DM_EPI_BSTORE(__builtin_bit_cast(u32x4, lo), rsC, voC, q * stepC, 0);
      DM_EPI_BSTORE(__builtin_bit_cast(u32x4, hi), rsC, voC + 16, q * stepC, 0);
    } else if (pair) {
      const u32x4 h8 = pack8(lo, hi);
      const bf16x8 hb = __builtin_bit_cast(bf16x8, h8);
      f32x4 rl0, rl1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { rl0[e] = lo[e] - (float)hb[e]; rl1[e] = hi[e] - (float)hb[4 + e]; }
      DM_EPI_BSTORE(h8, rsC, voC, q * stepC, 0);
      DM_EPI_BSTORE(pack8(rl0, rl1), rsC2, voC, q * stepC, 0);
    } else {
      DM_EPI_BSTORE(pack8(lo, hi), rsC, voC, q * stepC, 0);
    }
  };

  DmEpiPre pre[2];
  prefetch(pre[0], 0);
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
    for (int ii = 0; ii < PASS_TILES; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4 *>(mine + dm_epi_slot<PITCH, SWZ>(ii * 16 + li, j * 4 + g)) = acc[ps * PASS_TILES + ii][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int q = ps * R + r;
      if (q + 1 < Q) prefetch(pre[(q + 1) & 1], q + 1);          // (issued before this item's stores: see the header)
      __builtin_amdgcn_sched_barrier(0);
      const int row = r * 8 + rl;
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(mine + dm_epi_slot<PITCH, SWZ>(row, c8 * 2));
      const f32x4 hi = *reinterpret_cast<const f32x4 *>(mine + dm_epi_slot<PITCH, SWZ>(row, c8 * 2 + 1));
      if constexpr (SKIP_STORES) { if (lo[0] == 12345.678f) emit(lo, hi, pre[q & 1], q); }
      else emit(lo, hi, pre[q & 1], q);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next pass overwrites the region
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Structure key of an epilogue item (RES | YL << 1 | C32 << 3 | XS << 4) if the lean form's preconditions hold for a wave block of
// `rows` rows, -1 otherwise (grouped rows, 32-bit offsets, DM_GEMM_EPI_LEAN=0 = p.debug bit 0x400).  Host and device.
__host__ __device__ inline int dm_epi_lean_key(const GemmParams &p, int rows) {
  const long long lim = (1LL << 31) / ((long long)rows * 4);
  const bool aux_read = p.aux && (p.epilogue == DM_EPI_DGELU || p.epilogue == DM_EPI_MUL);
  const bool aux_write = p.aux && (p.epilogue == DM_EPI_GELU || p.epilogue == DM_EPI_GELU_GRAD);
  const bool lean = p.rows_per_group == 0 && !(p.debug & 0x400) && p.ldc < lim && p.ldr < lim && p.ldaux < lim &&
                    !(aux_read && p.accumulate);          // (one prefetch slot serves the old C or the aux operand)
  if (!lean) return -1;
  if (p.c_dtype == DM_BF16_PAIR) return 1 << 8;           // plane-pair results: the run-time lean form (no straight-line instance)
  const bool c32 = p.c_dtype == DM_F32, x32 = p.aux_dtype == DM_F32;
  const int yl = (c32 && p.accumulate) ? 1 : aux_read ? (x32 ? 3 : 2) : 0;
  const int xs = aux_write ? (x32 ? 2 : 1) : 0;
  return (p.residual ? 1 : 0) | (yl << 1) | ((c32 ? 1 : 0) << 3) | (xs << 4);
}
// the keys with a straight-line instance (every memory instruction unconditional, counted waits)
__host__ __device__ inline bool dm_epi_key_specialised(int key) {
  return key == 0 || key == (1 | (1 << 3)) || key == (1 << 3) || key == ((1 << 1) | (1 << 3)) || key == (1 << 4) || key == (2 << 1);
}

// LEAN_ONLY: the caller (a kernel whose K loop must not contain the generic form: the persistent LDS-DMA pipeline, where any load the
// compiler cannot count drains the DMA queue at the loop head) guarantees on the host that dm_epi_key_specialised holds.
template <int WM, int ROWS, bool SKIP_STORES = false, int PITCH = DM_EPI_PITCH, bool SWZ = false, bool LEAN_ONLY = false>
__device__ __forceinline__ void dm_epilogue_rows(const GemmParams &p, f32x4 (&acc)[WM][4], char *mine, int m_wave, int n_wave, int lane) {
  const int key = dm_epi_lean_key(p, 16 * WM);
  if constexpr (!LEAN_ONLY) {
    if (key < 0) { dm_epilogue_rows_generic<WM, ROWS, SKIP_STORES, PITCH, SWZ>(p, acc, mine, m_wave, n_wave, lane); return; }
  }
#define DM_EPI_CASE(RES, YL, C32, XS) \
  case ((RES) | ((YL) << 1) | ((C32) << 3) | ((XS) << 4)): \
    dm_epilogue_rows_lean<WM, ROWS, SKIP_STORES, false, (RES) != 0, (YL), (C32) != 0, (XS), PITCH, SWZ>(p, acc, mine, m_wave, n_wave, lane); break;
  switch (key) {
    DM_EPI_CASE(0, 0, 0, 0)      // bf16 C (+ bias / GELU without a saved derivative): qkv forward, the dgrads, inference fc1
    DM_EPI_CASE(1, 0, 1, 0)      // fp32 C + fp32 residual: proj / fc2 forward
    DM_EPI_CASE(0, 0, 1, 0)      // fp32 C: split-K slabs, unsplit weight gradients, fp32 activations
    DM_EPI_CASE(0, 1, 1, 0)      // fp32 C accumulated in place
    DM_EPI_CASE(0, 0, 0, 1)      // bf16 C + bf16 aux written: fc1 forward (GELU + saved GELU')
    DM_EPI_CASE(0, 2, 0, 0)      // bf16 C, bf16 aux read: dgrad of fc2 (multiply by the saved GELU')
    default:
      if constexpr (!LEAN_ONLY) dm_epilogue_rows_lean<WM, ROWS, SKIP_STORES, true, false, 0, false, 0, PITCH, SWZ>(p, acc, mine, m_wave, n_wave, lane);
      break;
  }
#undef DM_EPI_CASE
}
