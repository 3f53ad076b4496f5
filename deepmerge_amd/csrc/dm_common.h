// Shared device/host helpers for libdeepmerge_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "deepmerge_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define DM_WAVE 64
#define DM_LDS __attribute__((address_space(3)))

// ---- host-side error plumbing (thread-local message, int status) ---------------------------
void dm_set_error(const char *fmt, ...);

#define DM_REQUIRE(cond, code, ...)      \
  do {                                   \
    if (!(cond)) {                       \
      dm_set_error(__VA_ARGS__);         \
      return (code);                     \
    }                                    \
  } while (0)

#define DM_LAUNCH_CHECK(what)                                                        \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      dm_set_error("%s: launch failed: %s", (what), hipGetErrorString(e__));         \
      return DM_ERR_HIP;                                                             \
    }                                                                                \
  } while (0)

static inline bool dm_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers ------------------------------------------------------------------------
template <typename T> struct DmTypeInfo;
template <> struct DmTypeInfo<float> { static constexpr int kDtype = DM_F32; static constexpr int kPerChunk = 4; };
template <> struct DmTypeInfo<bf16_t> { static constexpr int kDtype = DM_BF16; static constexpr int kPerChunk = 8; };

__device__ __forceinline__ float dm_to_float(float v) { return v; }
__device__ __forceinline__ float dm_to_float(bf16_t v) { return (float)v; }

template <typename T> __device__ __forceinline__ T dm_from_float(float v);
template <> __device__ __forceinline__ float dm_from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t dm_from_float<bf16_t>(float v) { return (bf16_t)v; }

// 4 consecutive elements, vectorised.
__device__ __forceinline__ f32x4 dm_load4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 dm_load4(const bf16_t *p) {
  bf16x4 v = *reinterpret_cast<const bf16x4 *>(p);
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
__device__ __forceinline__ void dm_store4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
__device__ __forceinline__ void dm_store4(bf16_t *p, f32x4 v) {
  bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  *reinterpret_cast<bf16x4 *>(p) = r;
}

// nn.GELU() (erf form) and its derivative.
__device__ __forceinline__ float dm_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dm_dgelu(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Throughput-mode variants (results are rounded to bf16 afterwards): erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7) on v_rcp_f32 / v_exp_f32 instead of the ~40-instruction erff; GELU and its
// derivative share the single exp(-x^2/2).
__device__ __forceinline__ void dm_gelu_parts_fast(float x, float &cdf, float &pdf) {
#ifdef DM_GELU_ABLATE      // (tools/gelu_cost.sh: what the epilogue's GELU arithmetic costs -- a three-instruction stand-in, wrong results)
  cdf = fmaf(0.1f, x, 0.5f);
  pdf = 0.3f;
  return;
#endif
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float e = __expf(-z * z);                                    // exp(-x^2/2)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float erf_abs = fmaf(-poly, e, 1.0f);
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ float dm_gelu_fast(float x) {
  float cdf, pdf;
  dm_gelu_parts_fast(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ float dm_dgelu_fast(float x) {
  float cdf, pdf;
  dm_gelu_parts_fast(x, cdf, pdf);
  return fmaf(x, pdf, cdf);
}

__device__ __forceinline__ float dm_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float dm_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware remap of a 1-D block id: blocks that share an XCD (same id % 8 under the
// observed round-robin dispatch) get a contiguous chunk of the logical tile order, so neighbouring
// tiles reuse operand panels in that XCD's L2.  Speed only; any placement is correct.
__device__ __forceinline__ int dm_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}
