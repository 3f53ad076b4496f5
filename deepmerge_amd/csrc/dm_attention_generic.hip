// Attention for shapes outside the tiled kernels' range (head dim != 64 or more than 256 tokens): ViT-H/14 of the reference's
// factories (vit_model.py:649-662) has head dim 80 and 257 tokens.  Same contract as dm_attention_fwd / _bwd (packed qkv
// [B, N, 3, H, D], optional dense bias [H, N, N], out [B, N, H*D], lse [B, H, N]); fp32 arithmetic throughout, no MFMA, no
// atomics.  A correctness path: one wave per query row (forward, dQ) or per key row (dK/dV), O(N * D) work per row with the row's
// scores staged in LDS.  D <= 256, N <= 4096.
#include "dm_common.h"

namespace {

constexpr int GEN_MAX_N = 4096;
constexpr int GEN_MAX_D = 256;     // (Nets.RNN's attention_net: one head of 160)
constexpr int ROWS_PER_WG = 4;

struct GenParams {
  const void *qkv, *out, *dout;
  const float *bias, *lse;
  void *o, *dqkv;
  float *lse_out, *delta;
  int B, N, H, D;
  float scale;
};

template <typename T> __device__ __forceinline__ float ldf(const T *p) { return dm_to_float(*p); }

// s_j = scale * q . k_j + bias_ij for the wave's row; lanes stride over keys.  qv: the row's q (or dO) in LDS.
template <typename T>
__device__ __forceinline__ float dot_row(const float *qv, const T *krow, int D) {
  float acc = 0.f;
  for (int d = 0; d < D; ++d) acc = fmaf(qv[d], dm_to_float(krow[d]), acc);
  return acc;
}

// which = 0: forward (writes o, lse); which = 1: dQ (writes dq, delta)
template <typename T, int WHICH>
__global__ __launch_bounds__(64 * ROWS_PER_WG) void attn_generic_row_kernel(const GenParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * ROWS_PER_WG + w, h = blockIdx.y, b = blockIdx.z;
  const int N = p.N, D = p.D, H = p.H;
  float *qv = sm + w * (2 * GEN_MAX_D + 2 * N);        // [D] q, [D] dO, [N] p, [N] dS
  float *dov = qv + GEN_MAX_D, *pv = dov + GEN_MAX_D, *dsv = pv + N;
  if (i >= N) return;                                   // (no barrier below: waves are independent)
  const long long tok = 3LL * H * D;
  const T *qkv = reinterpret_cast<const T *>(p.qkv);
  const T *qrow = qkv + ((long long)b * N + i) * tok + (long long)h * D;
  const T *kbase = qkv + (long long)b * N * tok + (long long)(H + h) * D;
  const T *vbase = qkv + (long long)b * N * tok + (long long)(2 * H + h) * D;
  for (int d = lane; d < D; d += 64) qv[d] = ldf(qrow + d);
  const long long orow = ((long long)b * N + i) * H * D + (long long)h * D;
  float delta = 0.f;
  if (WHICH == 1) {
    const T *dop = reinterpret_cast<const T *>(p.dout) + orow, *op = reinterpret_cast<const T *>(p.out) + orow;
    for (int d = lane; d < D; d += 64) {
      const float g = ldf(dop + d);
      dov[d] = g;
      delta += g * ldf(op + d);
    }
    delta = dm_wave_sum(delta);
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float *brow = p.bias ? p.bias + ((long long)h * N + i) * N : nullptr;
  // scores
  float mx = -INFINITY;
  for (int j = lane; j < N; j += 64) {
    float s = p.scale * dot_row(qv, kbase + (long long)j * tok, D);
    if (brow) s += brow[j];
    pv[j] = s;
    mx = fmaxf(mx, s);
  }
  float lse;
  if (WHICH == 0) {
    mx = dm_wave_max(mx);
    float l = 0.f;
    for (int j = lane; j < N; j += 64) {
      const float e = __expf(pv[j] - mx);
      pv[j] = e;
      l += e;
    }
    l = dm_wave_sum(l);
    lse = mx + __logf(l);
    const float inv = 1.f / l;
    for (int j = lane; j < N; j += 64) pv[j] *= inv;
    if (lane == 0) p.lse_out[((long long)b * H + h) * N + i] = lse;
  } else {
    lse = p.lse[((long long)b * H + h) * N + i];
    for (int j = lane; j < N; j += 64) {
      const float pr = __expf(pv[j] - lse);
      const float dP = dot_row(dov, vbase + (long long)j * tok, D);
      pv[j] = pr;
      dsv[j] = pr * (dP - delta);
    }
    if (lane == 0) p.delta[((long long)b * H + h) * N + i] = delta;
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // forward: o_d = sum_j p_j v_jd ; dQ: dq_d = scale * sum_j dS_j k_jd      (lanes over d, coalesced row reads)
  const float *wv = (WHICH == 0) ? pv : dsv;
  const T *mat = (WHICH == 0) ? vbase : kbase;
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int j = 0; j < N; ++j) acc = fmaf(wv[j], ldf(mat + (long long)j * tok + d), acc);
    if (WHICH == 0) reinterpret_cast<T *>(p.o)[orow + d] = dm_from_float<T>(acc);
    else reinterpret_cast<T *>(p.dqkv)[((long long)b * N + i) * tok + (long long)h * D + d] = dm_from_float<T>(acc * p.scale);
  }
}

// dK_j = scale * sum_i dS_ij q_i ; dV_j = sum_i p_ij dO_i      (one wave per key row j; lanes stride over queries i)
template <typename T>
__global__ __launch_bounds__(64 * ROWS_PER_WG) void attn_generic_key_kernel(const GenParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * ROWS_PER_WG + w, h = blockIdx.y, b = blockIdx.z;
  const int N = p.N, D = p.D, H = p.H;
  float *kv = sm + w * (2 * GEN_MAX_D + 2 * N);        // [D] k_j, [D] v_j, [N] p_ij, [N] dS_ij
  float *vv = kv + GEN_MAX_D, *pv = vv + GEN_MAX_D, *dsv = pv + N;
  if (j >= N) return;
  const long long tok = 3LL * H * D;
  const T *qkv = reinterpret_cast<const T *>(p.qkv);
  const T *qbase = qkv + (long long)b * N * tok + (long long)h * D;
  const T *krow = qkv + ((long long)b * N + j) * tok + (long long)(H + h) * D;
  const T *vrow = qkv + ((long long)b * N + j) * tok + (long long)(2 * H + h) * D;
  const T *dobase = reinterpret_cast<const T *>(p.dout) + (long long)b * N * H * D + (long long)h * D;
  for (int d = lane; d < D; d += 64) { kv[d] = ldf(krow + d); vv[d] = ldf(vrow + d); }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float *lse = p.lse + ((long long)b * H + h) * N, *delta = p.delta + ((long long)b * H + h) * N;
  for (int i = lane; i < N; i += 64) {
    float s = p.scale * dot_row(kv, qbase + (long long)i * tok, D);
    if (p.bias) s += p.bias[((long long)h * N + i) * N + j];
    const float pr = __expf(s - lse[i]);
    const float dP = dot_row(vv, dobase + (long long)i * H * D, D);
    pv[i] = pr;
    dsv[i] = pr * (dP - delta[i]);
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  T *dk = reinterpret_cast<T *>(p.dqkv) + ((long long)b * N + j) * tok + (long long)(H + h) * D;
  T *dv = reinterpret_cast<T *>(p.dqkv) + ((long long)b * N + j) * tok + (long long)(2 * H + h) * D;
  for (int d = lane; d < D; d += 64) {
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < N; ++i) {
      ak = fmaf(dsv[i], ldf(qbase + (long long)i * tok + d), ak);
      av = fmaf(pv[i], ldf(dobase + (long long)i * H * D + d), av);
    }
    dk[d] = dm_from_float<T>(ak * p.scale);
    dv[d] = dm_from_float<T>(av);
  }
}

size_t gen_lds(int N) { return (size_t)ROWS_PER_WG * (2 * GEN_MAX_D + 2 * N) * sizeof(float); }

template <typename K> bool raise_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

}  // namespace

// true if this shape belongs to the generic path (and is inside its limits)
bool dm_attn_generic_shape(int N, int D) { return (D != 64 || N > 256) && D >= 1 && D <= GEN_MAX_D && N >= 1 && N <= GEN_MAX_N; }

int dm_attn_generic_fwd(const void *qkv, const float *bias, void *out, float *lse, int B, int N, int H, int D, float scale, int dtype,
                        hipStream_t s) {
  GenParams p{};
  p.qkv = qkv; p.bias = bias; p.o = out; p.lse_out = lse; p.B = B; p.N = N; p.H = H; p.D = D; p.scale = scale;
  const dim3 grid((N + ROWS_PER_WG - 1) / ROWS_PER_WG, H, B), block(64 * ROWS_PER_WG);
  const size_t lds = gen_lds(N);
  if (dtype == DM_BF16) {
    if (!raise_lds(attn_generic_row_kernel<bf16_t, 0>, lds)) return DM_ERR_HIP;
    hipLaunchKernelGGL((attn_generic_row_kernel<bf16_t, 0>), grid, block, lds, s, p);
  } else {
    if (!raise_lds(attn_generic_row_kernel<float, 0>, lds)) return DM_ERR_HIP;
    hipLaunchKernelGGL((attn_generic_row_kernel<float, 0>), grid, block, lds, s, p);
  }
  return DM_OK;
}

int dm_attn_generic_bwd(const void *qkv, const float *bias, const void *out, const void *dout, const float *lse, void *dqkv, float *delta,
                        int B, int N, int H, int D, float scale, int dtype, hipStream_t s) {
  GenParams p{};
  p.qkv = qkv; p.bias = bias; p.out = out; p.dout = dout; p.lse = lse; p.dqkv = dqkv; p.delta = delta;
  p.B = B; p.N = N; p.H = H; p.D = D; p.scale = scale;
  const dim3 grid((N + ROWS_PER_WG - 1) / ROWS_PER_WG, H, B), block(64 * ROWS_PER_WG);
  const size_t lds = gen_lds(N);
  if (dtype == DM_BF16) {
    if (!raise_lds(attn_generic_row_kernel<bf16_t, 1>, lds) || !raise_lds(attn_generic_key_kernel<bf16_t>, lds)) return DM_ERR_HIP;
    hipLaunchKernelGGL((attn_generic_row_kernel<bf16_t, 1>), grid, block, lds, s, p);
    hipLaunchKernelGGL((attn_generic_key_kernel<bf16_t>), grid, block, lds, s, p);
  } else {
    if (!raise_lds(attn_generic_row_kernel<float, 1>, lds) || !raise_lds(attn_generic_key_kernel<float>, lds)) return DM_ERR_HIP;
    hipLaunchKernelGGL((attn_generic_row_kernel<float, 1>), grid, block, lds, s, p);
    hipLaunchKernelGGL((attn_generic_key_kernel<float>), grid, block, lds, s, p);
  }
  return DM_OK;
}
