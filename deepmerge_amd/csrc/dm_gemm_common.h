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
};

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
  } else {
    dm_store4(reinterpret_cast<bf16_t *>(p.C) + rb.c + n, v);
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
  } else {
    bf16x8 o = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
    *reinterpret_cast<bf16x8 *>(reinterpret_cast<bf16_t *>(p.C) + rb.c + n) = o;
  }
}


template <int WM, int ROWS, bool SKIP_STORES = false>
__device__ __forceinline__ void dm_epilogue_rows(const GemmParams &p, f32x4 (&acc)[WM][4], char *mine, int m_wave, int n_wave, int lane) {
  const int g = lane >> 4, li = lane & 15;
  constexpr int PASS_TILES = ROWS / 16;
#pragma unroll
  for (int ps = 0; ps < WM / PASS_TILES; ++ps) {
#pragma unroll
    for (int ii = 0; ii < PASS_TILES; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4 *>(mine + (ii * 16 + li) * DM_EPI_PITCH + (j * 16 + 4 * g) * 4) = acc[ps * PASS_TILES + ii][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int n = n_wave + (lane & 7) * 8;
#pragma unroll
    for (int r = 0; r < ROWS / 8; ++r) {
      const int row = r * 8 + (lane >> 3);
      const int m = m_wave + ps * ROWS + row;
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(mine + row * DM_EPI_PITCH + (lane & 7) * 32);
      const f32x4 hi = *reinterpret_cast<const f32x4 *>(mine + row * DM_EPI_PITCH + (lane & 7) * 32 + 16);
      if constexpr (SKIP_STORES) { if (lo[0] == 12345.678f && m < p.M) dm_gemm_emit8(p, lo, hi, dm_gemm_row(p, m), n); }
      else if (m < p.M && n < p.N) dm_gemm_emit8(p, lo, hi, dm_gemm_row(p, m), n);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next pass overwrites the region
    __builtin_amdgcn_sched_barrier(0);
  }
}
