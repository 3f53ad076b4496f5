// BatchNorm2d (+ ReLU, + per-(sample, channel) dropout mask) of the auxiliary heads (AuxBolck, nets/ShfitScaleFormer.py:329-368
// in the reference: Conv2d -> BatchNorm2d -> ReLU -> Dropout2d), on the channels-LAST matrix the convolution GEMM produces:
// x [M = samples * positions, C] fp32.  BatchNorm2d over (sample, y, x) per channel == column statistics of that matrix.
// Bound: HBM (three passes over M x C forward, three backward); the statistics are summed in double so one pass gives both
// moments without cancellation.  No atomics: partial sums per row slice, reduced in slice order (run-to-run deterministic).
#include "dm_common.h"

namespace {

constexpr int BN_COLS = 64;      // columns per workgroup (256 B of every row: coalesced)
constexpr int BN_GROUPS = 4;     // row groups per workgroup (256 threads)

// partial[slice][which][c]: which = 0 sum a, 1 sum b.  MODE 0: a = x, b = x^2.  MODE 1 (backward): g = dy * (y > 0) * mask,
// a = g, b = g * xhat.
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ y,
                                                         const float *__restrict__ mask, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                         double *__restrict__ partial, int M, int C, int rows_per_sample, int rows_per_slice, int relu) {
  __shared__ double sh[2][BN_GROUPS][BN_COLS];
  const int c = blockIdx.x * BN_COLS + (threadIdx.x & (BN_COLS - 1));
  const int grp = threadIdx.x / BN_COLS;
  const int r0 = blockIdx.y * rows_per_slice, r1 = min(M, r0 + rows_per_slice);
  double a = 0.0, b = 0.0;
  if (c < C) {
    float mu = 0.f, rs = 0.f;
    if (MODE == 1) { mu = mean[c]; rs = rstd[c]; }
    for (int r = r0 + grp; r < r1; r += BN_GROUPS) {
      const long long i = (long long)r * C + c;
      if (MODE == 0) {
        const float v = x[i];
        a += v;
        b += (double)v * v;
      } else {
        float g = (!relu || y[i] > 0.f) ? dy[i] : 0.f;
        if (mask) g *= mask[(long long)(r / rows_per_sample) * C + c];
        a += g;
        b += (double)g * ((x[i] - mu) * rs);
      }
    }
  }
  sh[0][grp][threadIdx.x & (BN_COLS - 1)] = a;
  sh[1][grp][threadIdx.x & (BN_COLS - 1)] = b;
  __syncthreads();
  if (grp == 0 && c < C) {
    const int t = threadIdx.x;
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int q = 0; q < BN_GROUPS; ++q) { sa += sh[0][q][t]; sb += sh[1][q][t]; }
    partial[((long long)blockIdx.y * 2 + 0) * C + c] = sa;
    partial[((long long)blockIdx.y * 2 + 1) * C + c] = sb;
  }
}

// forward statistics: mean, 1/sqrt(biased var + eps); running statistics as torch.nn.BatchNorm2d updates them
__global__ void bn_finalize_fwd_kernel(const double *__restrict__ partial, int slices, int M, int C, float eps, float momentum,
                                       float *__restrict__ mean, float *__restrict__ rstd, float *__restrict__ running_mean,
                                       float *__restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < slices; ++k) { s += partial[((long long)k * 2) * C + c]; q += partial[((long long)k * 2 + 1) * C + c]; }
  const double mu = s / M;
  double var = q / M - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = (M > 1) ? var * ((double)M / (M - 1)) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

// y = relu((x - mean) * rstd * gamma + beta) * mask[sample][c]
__global__ void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ mean, const float *__restrict__ rstd,
                                const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ mask,
                                float *__restrict__ y, long long n4, int C, int rows_per_sample, int relu) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int c = (int)(e % C);
    const long long r = e / C;
    const f32x4 v = dm_load4(x + e), mu = dm_load4(mean + c), rs = dm_load4(rstd + c), ga = dm_load4(gamma + c), be = dm_load4(beta + c);
    f32x4 o = (v - mu) * rs * ga + be;
    if (relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], 0.f);
    }
    if (mask) o *= dm_load4(mask + (r / rows_per_sample) * C + c);
    dm_store4(y + e, o);
  }
}

// dgamma, dbeta (+= when accumulate) and the two per-channel means the dx kernel needs
__global__ void bn_finalize_bwd_kernel(const double *__restrict__ partial, int slices, int M, int C, float *__restrict__ dgamma,
                                       float *__restrict__ dbeta, int accumulate, float *__restrict__ mean_g, float *__restrict__ mean_gx) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < slices; ++k) { s += partial[((long long)k * 2) * C + c]; q += partial[((long long)k * 2 + 1) * C + c]; }
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)q;
  mean_g[c] = (float)(s / M);
  mean_gx[c] = (float)(q / M);
}

// dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat));  eval mode (batch_stats = 0): dx = gamma * rstd * g
__global__ void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ y,
                                    const float *__restrict__ mask, const float *__restrict__ mean, const float *__restrict__ rstd,
                                    const float *__restrict__ gamma, const float *__restrict__ mean_g, const float *__restrict__ mean_gx,
                                    float *__restrict__ dx, long long n4, int C, int rows_per_sample, int batch_stats, int relu) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int c = (int)(e % C);
    const long long r = e / C;
    const f32x4 yy = dm_load4(y + e);
    f32x4 g = dm_load4(dy + e);
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = (!relu || yy[k] > 0.f) ? g[k] : 0.f;
    if (mask) g *= dm_load4(mask + (r / rows_per_sample) * C + c);
    const f32x4 rs = dm_load4(rstd + c), ga = dm_load4(gamma + c);
    f32x4 o = g;
    if (batch_stats) {
      const f32x4 xh = (dm_load4(x + e) - dm_load4(mean + c)) * rs;
      o = g - dm_load4(mean_g + c) - xh * dm_load4(mean_gx + c);
    }
    dm_store4(dx + e, o * ga * rs);
  }
}

// eval mode: the "batch" statistics are the running ones
__global__ void bn_eval_stats_kernel(const float *__restrict__ running_mean, const float *__restrict__ running_var, float *__restrict__ mean,
                                     float *__restrict__ rstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { mean[c] = running_mean[c]; rstd[c] = 1.0f / sqrtf(running_var[c] + eps); }
}

inline int bn_slices(int M) {
  int s = (M + 255) / 256;
  return s < 1 ? 1 : (s > 64 ? 64 : s);
}
inline int bn_grid(long long n4) {
  long long g = (n4 + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int64_t dm_batchnorm_workspace_bytes(int32_t M, int32_t C) { return (int64_t)bn_slices(M) * 2 * C * 8 + 2LL * C * 4; }

extern "C" int dm_batchnorm_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                                const float *mask, int32_t rows_per_sample, float *y, float *save_mean, float *save_rstd, int32_t M,
                                int32_t C, float eps, float momentum, int32_t training, int32_t relu, void *workspace, void *stream) {
  DM_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && rows_per_sample > 0 && M % rows_per_sample == 0, DM_ERR_BAD_SHAPE,
             "dm_batchnorm_fwd: M=%d C=%d rows_per_sample=%d (C %% 4 == 0, M a multiple of rows_per_sample)", M, C, rows_per_sample);
  DM_REQUIRE(x && gamma && beta && y && save_mean && save_rstd, DM_ERR_BAD_SHAPE, "dm_batchnorm_fwd: null pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (training) {
    DM_REQUIRE(workspace != nullptr, DM_ERR_BAD_SHAPE, "dm_batchnorm_fwd: training mode needs dm_batchnorm_workspace_bytes(M, C) of workspace");
    const int slices = bn_slices(M), rps = (M + slices - 1) / slices;
    double *partial = reinterpret_cast<double *>(workspace);
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3((C + BN_COLS - 1) / BN_COLS, slices), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr, nullptr,
                       partial, M, C, rows_per_sample, rps, 0);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, s, partial, slices, M, C, eps, momentum, save_mean, save_rstd,
                       running_mean, running_var);
  } else {
    DM_REQUIRE(running_mean && running_var, DM_ERR_BAD_SHAPE, "dm_batchnorm_fwd: eval mode needs the running statistics");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, s, running_mean, running_var, save_mean, save_rstd, C, eps);
  }
  hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_grid((long long)M * C / 4)), dim3(256), 0, s, x, save_mean, save_rstd, gamma, beta, mask, y,
                     (long long)M * C / 4, C, rows_per_sample, relu);
  DM_LAUNCH_CHECK("dm_batchnorm_fwd");
  return DM_OK;
}

extern "C" int dm_batchnorm_bwd(const float *dy, const float *x, const float *y, const float *gamma, const float *mask, int32_t rows_per_sample,
                                const float *save_mean, const float *save_rstd, float *dx, float *dgamma, float *dbeta, int32_t accumulate,
                                int32_t M, int32_t C, int32_t training, int32_t relu, void *workspace, void *stream) {
  DM_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && rows_per_sample > 0 && M % rows_per_sample == 0, DM_ERR_BAD_SHAPE,
             "dm_batchnorm_bwd: M=%d C=%d rows_per_sample=%d", M, C, rows_per_sample);
  DM_REQUIRE(dy && x && y && gamma && save_mean && save_rstd && dx && dgamma && dbeta && workspace, DM_ERR_BAD_SHAPE, "dm_batchnorm_bwd: null pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int slices = bn_slices(M), rps = (M + slices - 1) / slices;
  double *partial = reinterpret_cast<double *>(workspace);
  float *mean_g = reinterpret_cast<float *>(partial + (long long)slices * 2 * C), *mean_gx = mean_g + C;
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3((C + BN_COLS - 1) / BN_COLS, slices), dim3(256), 0, s, x, dy, y, mask, save_mean, save_rstd, partial,
                     M, C, rows_per_sample, rps, relu);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, s, partial, slices, M, C, dgamma, dbeta, accumulate, mean_g, mean_gx);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bn_grid((long long)M * C / 4)), dim3(256), 0, s, x, dy, y, mask, save_mean, save_rstd, gamma, mean_g,
                     mean_gx, dx, (long long)M * C / 4, C, rows_per_sample, training, relu);
  DM_LAUNCH_CHECK("dm_batchnorm_bwd");
  return DM_OK;
}
