// Row kernels for shapes outside the hot kernels' range (ViT-H/14 of the reference's factories, vit_model.py:649-662:
// embed dim 1280 > the 1024 columns the register-resident LayerNorm keeps per wave, patch side 14 not a multiple of 4).
// Same contracts as dm_layernorm_fwd / _bwd / dm_patchify; the rows are streamed from L1/L2 in passes instead of being
// held in registers, and the parameter gradients come from a column kernel that recomputes xhat.  fp32 statistics, no atomics.
#include "dm_common.h"

namespace {

// one wave per row, cols % 4 == 0, any width
template <typename TY>
__global__ __launch_bounds__(256) void layernorm_wide_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, TY *__restrict__ y,
                                                                 float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                                 int rows, int cols, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = cols >> 2;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float *xr = x + (long long)row * cols;
    float s = 0.f;
    for (int c = lane; c < nch; c += 64) {
      const f32x4 v = dm_load4(xr + 4 * c);
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = dm_wave_sum(s) / (float)cols;
    float q = 0.f;
    for (int c = lane; c < nch; c += 64) {
      const f32x4 v = dm_load4(xr + 4 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[e] - mean;
        q += d * d;
      }
    }
    const float rstd = rsqrtf(dm_wave_sum(q) / (float)cols + eps);
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
    TY *yr = y + (long long)row * cols;
    for (int c = lane; c < nch; c += 64) {
      const f32x4 v = dm_load4(xr + 4 * c), g = dm_load4(gamma + 4 * c), b = dm_load4(beta + 4 * c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * g[e] + b[e];
      dm_store4(yr + 4 * c, o);
    }
  }
}

// dx = dres + rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma      (one wave per row, two passes)
template <typename TDY>
__global__ __launch_bounds__(256) void layernorm_wide_dx_kernel(const TDY *__restrict__ dy, const float *__restrict__ x,
                                                                const float *__restrict__ gamma, const float *__restrict__ mean,
                                                                const float *__restrict__ rstd, const float *__restrict__ dres,
                                                                float *__restrict__ dx, bf16_t *__restrict__ dx_lp, int rows, int cols) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = cols >> 2;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const long long off = (long long)row * cols;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < nch; c += 64) {
      const f32x4 xv = dm_load4(x + off + 4 * c), d = dm_load4(dy + off + 4 * c), gm = dm_load4(gamma + 4 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g = d[e] * gm[e];
        s1 += g;
        s2 += g * ((xv[e] - mu) * rs);
      }
    }
    const float c1 = dm_wave_sum(s1) / (float)cols, c2 = dm_wave_sum(s2) / (float)cols;
    for (int c = lane; c < nch; c += 64) {
      const f32x4 xv = dm_load4(x + off + 4 * c), d = dm_load4(dy + off + 4 * c), gm = dm_load4(gamma + 4 * c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rs * (d[e] * gm[e] - c1 - ((xv[e] - mu) * rs) * c2);
      if (dres) o += dm_load4(dres + off + 4 * c);
      dm_store4(dx + off + 4 * c, o);
      if (dx_lp) dm_store4(dx_lp + off + 4 * c, o);
    }
  }
}

// partial[slice][0:cols] = sum_rows dy*xhat, partial[slice][cols:2cols] = sum_rows dy over the slice's rows.
// Workgroup = 64 columns (16 lanes x float4) x 16 row groups; grid (cols/64 rounded up, slices).
template <typename TDY>
__global__ __launch_bounds__(256) void layernorm_wide_param_kernel(const TDY *__restrict__ dy, const float *__restrict__ x,
                                                                   const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                   float *__restrict__ partial, int rows, int cols) {
  __shared__ f32x4 red[2][16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int col = (blockIdx.x * 16 + tx) * 4;
  const int slices = gridDim.y;
  const int per = (rows + slices - 1) / slices;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
  if (col < cols) {
    for (int r = r0 + ty; r < r1; r += 16) {
      const long long off = (long long)r * cols + col;
      const f32x4 xv = dm_load4(x + off), d = dm_load4(dy + off);
      const float mu = mean[r], rs = rstd[r];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dg[e] += d[e] * ((xv[e] - mu) * rs);
        db[e] += d[e];
      }
    }
  }
  red[0][ty][tx] = dg;
  red[1][ty][tx] = db;
  __syncthreads();
  if (ty < 2 && col < cols) {          // ty = 0: dgamma, ty = 1: dbeta ; fixed order over the 16 row groups
    f32x4 a = red[ty][0][tx];
    for (int k = 1; k < 16; ++k) a += red[ty][k][tx];
    dm_store4(partial + (long long)blockIdx.y * 2 * cols + (long long)ty * cols + col, a);
  }
}

// im2col of non-overlapping p x p patches, any p dividing side (element per thread)
template <typename T>
__global__ void patchify_any_kernel(const float *__restrict__ x, T *__restrict__ cols, int B, int C, int side, int p) {
  const int grid = side / p;
  const long long K = (long long)C * p * p;
  const long long total = (long long)B * grid * grid * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int dx = (int)(r % p); r /= p;
    const int dy = (int)(r % p); r /= p;
    const int c = (int)(r % C); r /= C;     // r = row = (b*grid + py)*grid + px
    const int px = (int)(r % grid);
    const long long bp = r / grid;
    const int py = (int)(bp % grid);
    const long long b = bp / grid;
    cols[i] = dm_from_float<T>(x[((b * C + c) * side + (py * p + dy)) * (long long)side + px * p + dx]);
  }
}

int wide_grid(long long items, int per_block, int cap) {
  long long g = (items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace

int dm_layernorm_wide_fwd(const float *x, const float *gamma, const float *beta, void *y, int y_dtype, float *mean, float *rstd, int rows,
                          int cols, float eps, hipStream_t s) {
  const int grid = wide_grid(rows, 4, 4096);
  if (y_dtype == DM_F32)
    hipLaunchKernelGGL(layernorm_wide_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, x, gamma, beta, (float *)y, mean, rstd, rows, cols, eps);
  else if (y_dtype == DM_BF16)
    hipLaunchKernelGGL(layernorm_wide_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, x, gamma, beta, (bf16_t *)y, mean, rstd, rows, cols, eps);
  else return DM_ERR_BAD_DTYPE;
  return DM_OK;
}

// Writes `*slices` partial rows of 2*cols floats (dgamma | dbeta) into `partial` (capacity max_slices rows); the caller reduces them.
int dm_layernorm_wide_bwd(const void *dy, int dy_dtype, const float *x, const float *gamma, const float *mean, const float *rstd,
                          const float *dres, float *dx, void *dx_lp, float *partial, int max_slices, int *slices, int rows, int cols,
                          hipStream_t s) {
  const int grid = wide_grid(rows, 4, 4096);
  int ns = rows / 64;
  if (ns < 1) ns = 1;
  if (ns > 64) ns = 64;
  if (ns > max_slices) ns = max_slices;
  if (ns < 1) return DM_ERR_BAD_SHAPE;
  *slices = ns;
  const dim3 pgrid((cols / 4 + 15) / 16, ns);
  if (dy_dtype == DM_F32) {
    hipLaunchKernelGGL(layernorm_wide_dx_kernel<float>, dim3(grid), dim3(256), 0, s, (const float *)dy, x, gamma, mean, rstd, dres, dx, (bf16_t *)dx_lp, rows, cols);
    hipLaunchKernelGGL(layernorm_wide_param_kernel<float>, pgrid, dim3(256), 0, s, (const float *)dy, x, mean, rstd, partial, rows, cols);
  } else if (dy_dtype == DM_BF16) {
    hipLaunchKernelGGL(layernorm_wide_dx_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)dy, x, gamma, mean, rstd, dres, dx, (bf16_t *)dx_lp, rows, cols);
    hipLaunchKernelGGL(layernorm_wide_param_kernel<bf16_t>, pgrid, dim3(256), 0, s, (const bf16_t *)dy, x, mean, rstd, partial, rows, cols);
  } else return DM_ERR_BAD_DTYPE;
  return DM_OK;
}

int dm_patchify_any(const float *x, void *cols, int dtype, int B, int C, int side, int p, hipStream_t s) {
  const long long total = (long long)B * (side / p) * (side / p) * ((long long)C * p * p);
  const int grid = wide_grid(total, 256, 65536);
  if (dtype == DM_F32) hipLaunchKernelGGL(patchify_any_kernel<float>, dim3(grid), dim3(256), 0, s, x, (float *)cols, B, C, side, p);
  else if (dtype == DM_BF16) hipLaunchKernelGGL(patchify_any_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, x, (bf16_t *)cols, B, C, side, p);
  else return DM_ERR_BAD_DTYPE;
  return DM_OK;
}
