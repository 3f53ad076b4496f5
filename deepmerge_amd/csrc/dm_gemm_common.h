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
