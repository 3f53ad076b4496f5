// ExtractFeatures region-adjacency sweep on the GPU (gfx950): per-superpixel mean pooling of the
// point embeddings and the per-edge Euclidean "simi" distance + merge decision.  HBM-bound
// gather / streaming kernels; the arithmetic ORDER is part of the contract (bit-exact against
// oracle/sweep_strict.c), so products and sums are written with explicit round-to-nearest
// intrinsics and can never be contracted into FMAs.
#include "dm_common.h"

// The summation order is a bit-exact contract: forbid FMA contraction of the a*b + c chains below
// (hipcc defaults to -ffp-contract=fast, and __fmul_rn/__fadd_rn are plain * and + in the HIP headers).
#pragma clang fp contract(off)

namespace {

// One wave per superpixel; lane owns columns lane, lane+64, ...  Rows are added in idx order and the
// sum is divided by the count (np.mean(axis=0) on a C-contiguous [k,D] float32 block:
// sequential row accumulation, then true_divide).
__global__ __launch_bounds__(256) void segment_mean_kernel(const float *__restrict__ F, const int *__restrict__ ptr,
                                                           const int *__restrict__ idx, float *__restrict__ pooled,
                                                           int S, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s = blockIdx.x * 4 + wave; s < S; s += gridDim.x * 4) {
    const int beg = ptr[s], end = ptr[s + 1];
    for (int c = lane; c < D; c += 64) {
      float acc = 0.f;
      if (end > beg) {
        acc = F[(long long)idx[beg] * D + c];
        for (int k = beg + 1; k < end; ++k) acc = __fadd_rn(acc, F[(long long)idx[k] * D + c]);
        acc = __fdiv_rn(acc, (float)(end - beg));
      }
      pooled[(long long)s * D + c] = acc;
    }
  }
}

// 8 lanes per edge (one per pairwise-sum accumulator), 8 edges per wave.
// Order (numpy pairwise sum for 8 <= D <= 128, see oracle/sweep_strict.c):
//   r[j] = t[j]; r[j] += t[8i + j] for 8i + j < D - D%8; res = ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7));
//   res += t[i] for the D%8 tail, in order.           t = x*x, y*y or x*y (rounded products).
__global__ __launch_bounds__(256) void edge_similarity_kernel(const float *__restrict__ pooled, const int *__restrict__ edges,
                                                              float *__restrict__ simi, unsigned char *__restrict__ merge,
                                                              int E, int D, float margin) {
  const int lane = threadIdx.x & 63;
  const int j = lane & 7;
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const bool live = e < E;
  int L = -1, R = -1;
  if (live) { L = edges[2 * e]; R = edges[2 * e + 1]; }
  const bool ok = live && L >= 0 && R >= 0;
  const float *x = pooled + (long long)(ok ? L : 0) * D;
  const float *y = pooled + (long long)(ok ? R : 0) * D;
  const int body = D - (D & 7);
  float xx = 0.f, yy = 0.f, xy = 0.f;
  if (D >= 8) {
    float a = x[j], b = y[j];
    xx = __fmul_rn(a, a); yy = __fmul_rn(b, b); xy = __fmul_rn(a, b);
    for (int i = 8; i < body; i += 8) {
      a = x[i + j]; b = y[i + j];
      xx = __fadd_rn(xx, __fmul_rn(a, a));
      yy = __fadd_rn(yy, __fmul_rn(b, b));
      xy = __fadd_rn(xy, __fmul_rn(a, b));
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {   // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)): commutative butterflies
      xx = __fadd_rn(xx, __shfl_xor(xx, o, 64));
      yy = __fadd_rn(yy, __shfl_xor(yy, o, 64));
      xy = __fadd_rn(xy, __shfl_xor(xy, o, 64));
    }
    for (int i = body; i < D; ++i) {
      const float a = x[i], b = y[i];
      xx = __fadd_rn(xx, __fmul_rn(a, a));
      yy = __fadd_rn(yy, __fmul_rn(b, b));
      xy = __fadd_rn(xy, __fmul_rn(a, b));
    }
  } else {
    for (int i = 0; i < D; ++i) {
      const float a = x[i], b = y[i];
      xx = __fadd_rn(xx, __fmul_rn(a, a));
      yy = __fadd_rn(yy, __fmul_rn(b, b));
      xy = __fadd_rn(xy, __fmul_rn(a, b));
    }
  }
  if (live && j == 0) {
    float d = __fsub_rn(__fadd_rn(xx, yy), __fmul_rn(2.0f, xy));
    if (d < 0.f) d = 0.f;                 // D[D < 0] = 0 (NaN stays NaN)
    // correctly rounded float sqrt: a double sqrt rounded once more to float is exact-rounded because
    // 53 >= 2*24 + 2 (the single-precision intrinsic forms are not guaranteed IEEE on this target)
    float sm = (float)sqrt((double)d);
    if (!ok) sm = __builtin_nanf("");
    simi[e] = sm;
    if (merge) merge[e] = (sm < margin) ? 1 : 0;
  }
}

}  // namespace

extern "C" int dm_segment_mean(const float *F, const int32_t *ptr, const int32_t *idx, float *pooled, int32_t S, int32_t D, void *stream) {
  DM_REQUIRE(F && ptr && idx && pooled && S > 0 && D > 0, DM_ERR_BAD_SHAPE, "dm_segment_mean: bad arguments");
  int grid = (S + 3) / 4;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(segment_mean_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), F, ptr, idx, pooled, S, D);
  DM_LAUNCH_CHECK("dm_segment_mean");
  return DM_OK;
}

extern "C" int dm_edge_similarity(const float *pooled, const int32_t *edges, float *simi, uint8_t *merge, int32_t E, int32_t D,
                                  float margin, void *stream) {
  DM_REQUIRE(pooled && edges && simi && E > 0, DM_ERR_BAD_SHAPE, "dm_edge_similarity: bad arguments");
  DM_REQUIRE(D > 0 && D <= 128, DM_ERR_UNSUPPORTED, "dm_edge_similarity: feature dim %d outside the pinned summation order (1..128)", D);
  const int grid = (int)(((long long)E * 8 + 255) / 256);
  hipLaunchKernelGGL(edge_similarity_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pooled, edges, simi, merge, E, D, margin);
  DM_LAUNCH_CHECK("dm_edge_similarity");
  return DM_OK;
}
