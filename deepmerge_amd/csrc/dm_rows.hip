// HBM-bound row / element kernels of the DeepMerge hot path (gfx950): LayerNorm fwd/bwd,
// token pooling, column sums, casts, patch extraction, contrastive loss, Adam, and the
// relative-position-bias gather / gradient reduction.  All loads/stores are 16-byte vectors
// along the contiguous (channel) axis; reductions are wave64 butterflies.
#include <cstdlib>

#include "dm_common.h"

// dm_rows_wide.hip: streamed variants for rows wider than the register-resident kernels hold, and patch sides that are not
// multiples of 4 (ViT-H/14).
int dm_layernorm_wide_fwd(const float *x, const float *gamma, const float *beta, void *y, int y_dtype, float *mean, float *rstd, int rows,
                          int cols, float eps, hipStream_t s);
int dm_layernorm_wide_bwd(const void *dy, int dy_dtype, const float *x, const float *gamma, const float *mean, const float *rstd,
                          const float *dres, float *dx, void *dx_lp, float *partial, int max_slices, int *slices, int rows, int cols,
                          hipStream_t s);
int dm_patchify_any(const float *x, void *cols, int dtype, int B, int C, int side, int p, hipStream_t s);

namespace {

constexpr int MAXCH = 4;          // float4 chunks per lane per row -> cols <= 1024 in the register-resident kernels
constexpr int LN_WIDE_MAX = 8192; // widest row of the streamed kernels
static int ln_max_wg() {           // workgroups of the LayerNorm backward (each writes one partial row pair)
  static const int v = [] { const char *e = getenv("DM_LN_WG"); return e ? atoi(e) : 768; }();      // 3 workgroups (12 waves) per CU; 512 / 768 / 1024: 51 / 45 / 47 us at 16384 x 768 (tools/mb_ln.py)
  return v;
}
constexpr int CS_MAX_SLICES = 64;

// ---------------------------------------------------------------------------------------------
// LayerNorm forward: one wave per row.
// ---------------------------------------------------------------------------------------------
// PAIR (TY = bf16): y is the hi / lo plane pair [2, rows, cols] of the normalised rows (the operand format of the folded "bf16x3"
// products: no fp32 y, no split pass).
template <typename TY, bool PAIR = false>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, TY *__restrict__ y,
                                                            float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                            int rows, int cols, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = cols >> 2;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float *xr = x + (long long)row * cols;
    f32x4 v[MAXCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < nch) ? dm_load4(xr + 4 * c) : (f32x4){0.f, 0.f, 0.f, 0.f};
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = dm_wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[i][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(dm_wave_sum(q) / (float)cols + eps);
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
    TY *yr = y + (long long)row * cols;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const f32x4 g = dm_load4(gamma + 4 * c), b = dm_load4(beta + 4 * c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
        if constexpr (PAIR) {
          bf16x4 h, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            h[e] = (bf16_t)o[e];
            l[e] = (bf16_t)(o[e] - (float)h[e]);
          }
          *reinterpret_cast<bf16x4 *>(yr + 4 * c) = h;
          *reinterpret_cast<bf16x4 *>(yr + (long long)rows * cols + 4 * c) = l;
        } else {
          dm_store4(yr + 4 * c, o);
        }
      }
    }
  }
}

// LayerNorm backward: dx = dres + rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma.
// Each workgroup keeps per-column partial sums of dy*xhat (dgamma) and dy (dbeta) in registers
// over its grid-stride rows, combines its 4 waves through LDS and writes one partial row.
// CH = float4 chunks per lane per row: 3 for rows of up to 768 columns (the encoder's width: 112 registers, 4 workgroups per CU),
// 4 up to 1024.
template <typename TDY, int CH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TDY *__restrict__ dy, const float *__restrict__ x,
                                                            const float *__restrict__ gamma, const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, const float *__restrict__ dres,
                                                            float *__restrict__ dx, bf16_t *__restrict__ dx_lp,
                                                            float *__restrict__ partial, int rows, int cols, long long lp_plane) {
  // lp_plane > 0: dx_lp is the hi / lo plane pair of dx (lo plane lp_plane elements behind), not a rounded bf16 copy
  __shared__ float red[4][2][CH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = cols >> 2;
  f32x4 dg[CH], db[CH], gam[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    dg[i] = db[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int c = lane + 64 * i;
    gam[i] = (c < nch) ? dm_load4(gamma + 4 * c) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const long long off = (long long)row * cols;
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[CH], g[CH], dr[CH];
    float s1 = 0.f, s2 = 0.f;
    // every load of the row is issued before anything waits (the residual gradient used to be requested after the two wave
    // reductions: a second full memory latency per row with only 8 waves per CU to hide it -- 3.4 TB/s at 16384 x 768)
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = lane + 64 * i;
      dr[i] = (dres && c < nch) ? dm_load4(dres + off + 4 * c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const f32x4 xv = dm_load4(x + off + 4 * c);
        const f32x4 d = dm_load4(dy + off + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] = (xv[e] - mu) * rs;
          g[i][e] = d[e] * gam[i][e];
          s1 += g[i][e];
          s2 += g[i][e] * xh[i][e];
          dg[i][e] += d[e] * xh[i][e];
          db[i][e] += d[e];
        }
      } else {
        xh[i] = g[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    const float c1 = dm_wave_sum(s1) / (float)cols, c2 = dm_wave_sum(s2) / (float)cols;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (g[i][e] - c1 - xh[i][e] * c2);
        if (dres) o += dr[i];
        dm_store4(dx + off + 4 * c, o);
        if (dx_lp) {
          if (lp_plane > 0) {
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              h[e] = (bf16_t)o[e];
              l[e] = (bf16_t)(o[e] - (float)h[e]);
            }
            *reinterpret_cast<bf16x4 *>(dx_lp + off + 4 * c) = h;
            *reinterpret_cast<bf16x4 *>(dx_lp + lp_plane + off + 4 * c) = l;
          } else {
            dm_store4(dx_lp + off + 4 * c, o);
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < CH; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[wave][0][(lane + 64 * i) * 4 + e] = dg[i][e];
      red[wave][1][(lane + 64 * i) * 4 + e] = db[i][e];
    }
  }
  __syncthreads();
  float *prow = partial + (long long)blockIdx.x * 2 * cols;
  for (int c = threadIdx.x; c < 2 * cols; c += 256) {
    const int which = c / cols, col = c % cols;
    prow[c] = (red[0][which][col] + red[1][which][col]) + (red[2][which][col] + red[3][which][col]);
  }
}

// out[j] (+)= sum_r partial[r][j].  Block = 16 columns x 16 row-slices (many small workgroups: the input is a few
// MB spread over up to 1024 rows); each slice sums its rows in order, slices are combined in a fixed tree ->
// deterministic.
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float *__restrict__ partial, float *__restrict__ out0,
                                                             float *__restrict__ out1, int nrows, int width, int split,
                                                             int accumulate) {
  // `partial` rows are `width` wide; columns [0,split) go to out0, [split,width) to out1.
  __shared__ float red[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + tx;
  float s = 0.f;
  if (j < width) {
    // four independent accumulators: the loads of a slice are in flight together instead of one per round trip
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = ty;
    for (; r + 48 < nrows; r += 64) {
      s0 += partial[(long long)r * width + j];
      s1 += partial[(long long)(r + 16) * width + j];
      s2 += partial[(long long)(r + 32) * width + j];
      s3 += partial[(long long)(r + 48) * width + j];
    }
    for (; r < nrows; r += 16) s0 += partial[(long long)r * width + j];
    s = (s0 + s1) + (s2 + s3);
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < width) {
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = red[k][tx];
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
      for (int k = 0; k < w; ++k) t[k] = t[2 * k] + t[2 * k + 1];
    s = t[0];
    float *o = (j < split) ? out0 + j : out1 + (j - split);
    *o = accumulate ? *o + s : s;
  }
}

// The same reduction for up to 32 independent (partial rows -> out0 | out1) jobs in ONE launch (dm_partial_reduce_batch): the
// LayerNorm / bias-gradient reductions of a backward pass are 96-column-block launches of ~5 us each (29 per step of the headline
// model); queued and reduced together they fill the chip once.  Same arithmetic per job as partial_reduce_kernel (bit-identical results).
struct ReduceBatch {
  DmReduceItem it[32];
  int start[33];            // first workgroup of job i; start[n] = grid size
  int n;
};
__global__ __launch_bounds__(256) void partial_reduce_batch_kernel(const ReduceBatch rb) {
  __shared__ float red[16][17];
  int job = 0;
  while (job + 1 < rb.n && (int)blockIdx.x >= rb.start[job + 1]) ++job;
  const DmReduceItem &it = rb.it[job];
  const float *__restrict__ partial = it.partial;
  const int nrows = it.nrows, width = it.width;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int j = ((int)blockIdx.x - rb.start[job]) * 16 + tx;
  float s = 0.f;
  if (j < width) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = ty;
    for (; r + 48 < nrows; r += 64) {
      s0 += partial[(long long)r * width + j];
      s1 += partial[(long long)(r + 16) * width + j];
      s2 += partial[(long long)(r + 32) * width + j];
      s3 += partial[(long long)(r + 48) * width + j];
    }
    for (; r < nrows; r += 16) s0 += partial[(long long)r * width + j];
    s = (s0 + s1) + (s2 + s3);
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < width) {
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = red[k][tx];
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
      for (int k = 0; k < w; ++k) t[k] = t[2 * k] + t[2 * k + 1];
    s = t[0];
    float *o = (j < it.split) ? it.out0 + j : it.out1 + (j - it.split);
    *o = it.accumulate ? *o + s : s;
  }
}

// ---------------------------------------------------------------------------------------------
// token pooling
// ---------------------------------------------------------------------------------------------
__global__ void token_pool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int S, int side, int C) {
  const int half = side >> 1, c4 = C >> 2;
  const long long total = (long long)B * S * half * half * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4);
    long long r = i / c4;
    const int px = (int)(r % half); r /= half;
    const int py = (int)(r % half); r /= half;   // r = b*S + s
    const float *src = x + ((r * side + 2 * py) * side + 2 * px) * (long long)C + 4 * c;
    const f32x4 a = dm_load4(src), b = dm_load4(src + C), d = dm_load4(src + (long long)side * C), e = dm_load4(src + (long long)side * C + C);
    dm_store4(y + ((r * half + py) * half + px) * (long long)C + 4 * c, ((a + b) + (d + e)) * 0.25f);
  }
}
__global__ void token_pool_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int B, int S, int side, int C) {
  const int half = side >> 1, c4 = C >> 2;
  const long long total = (long long)B * S * side * side * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4);
    long long r = i / c4;
    const int xx = (int)(r % side); r /= side;
    const int yy = (int)(r % side); r /= side;
    const f32x4 g = dm_load4(dy + ((r * half + (yy >> 1)) * half + (xx >> 1)) * (long long)C + 4 * c);
    dm_store4(dx + ((r * side + yy) * side + xx) * (long long)C + 4 * c, g * 0.25f);
  }
}
__global__ void group_mean_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int rows, int g, int C) {
  const int c4 = C >> 2;
  const long long total = (long long)rows * c4;
  const float inv = 1.0f / (float)g;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4);
    const long long r = i / c4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < g; ++k) s += dm_load4(x + (r * g + k) * C + 4 * c);
    dm_store4(y + r * C + 4 * c, s * inv);
  }
}
__global__ void group_mean_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int rows, int g, int C) {
  const int c4 = C >> 2;
  const long long total = (long long)rows * g * c4;
  const float inv = 1.0f / (float)g;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4);
    const long long r = i / c4;
    dm_store4(dx + r * C + 4 * c, dm_load4(dy + (r / g) * C + 4 * c) * inv);
  }
}

// ---------------------------------------------------------------------------------------------
// column sums (bias gradients): slices of rows -> partial[slice][N]
// ---------------------------------------------------------------------------------------------
template <typename T>
// direct >= 0 (one slice only): the sums go straight to `partial` = the destination, added to what is there when direct == 1 -- the
// reduction launch of a one-slice column sum (the 64-row bias gradients of the head and the FeatureEmbed layers) does nothing else
__global__ __launch_bounds__(256) void colsum_kernel(const T *__restrict__ X, long long ldx, float *__restrict__ partial,
                                                     int M, int N, int rows_per_slice, int direct) {
  __shared__ f32x4 red[16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int n = blockIdx.x * 64 + tx * 4;
  const int r0 = blockIdx.y * rows_per_slice, r1 = min(M, r0 + rows_per_slice);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (n < N)
    for (int r = r0 + ty; r < r1; r += 16) s += dm_load4(X + (long long)r * ldx + n);
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) {
    f32x4 a = red[0][tx];
#pragma unroll
    for (int k = 1; k < 16; ++k) a += red[k][tx];
    if (direct == 1) a = dm_load4(partial + n) + a;
    dm_store4(partial + (long long)blockIdx.y * N + n, a);
  }
}

// any N / leading dimension (odd layer widths such as Nets.py's 250 and 10): one thread per column and row slice
template <typename T>
__global__ __launch_bounds__(256) void colsum_generic_kernel(const T *__restrict__ X, long long ldx, float *__restrict__ partial,
                                                             int M, int N, int rows_per_slice) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int r0 = blockIdx.y * rows_per_slice, r1 = min(M, r0 + rows_per_slice);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += (float)X[(long long)r * ldx + n];
  partial[(long long)blockIdx.y * N + n] = s;
}

// ---------------------------------------------------------------------------------------------
// cast / patchify
// ---------------------------------------------------------------------------------------------
__global__ void cast_bf16_kernel(const float *__restrict__ src, bf16_t *__restrict__ dst, long long n) {
  const long long n8 = n >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 a = dm_load4(src + 8 * i), b = dm_load4(src + 8 * i + 4);
    bf16x8 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    *reinterpret_cast<bf16x8 *>(dst + 8 * i) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(n8 << 3) + threadIdx.x] = (bf16_t)src[(n8 << 3) + threadIdx.x];
}
// hi / lo bf16 pieces of fp32 values for the split-bf16 ("bf16x3") products: see dm_split_bf16 in the header.
__global__ void split_bf16_kernel(const float *__restrict__ src, long long ld, long long rows, long long cols, bf16_t *__restrict__ dst,
                                  int stack, int pattern) {
  const long long c4 = cols >> 2, total = rows * c4;
  const long long piece = stack ? rows * cols : cols, ldd = stack ? cols : 3 * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c4, c = (i - r * c4) << 2;
    const f32x4 x = dm_load4(src + r * ld + c);
    bf16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hi[j] = (bf16_t)x[j];
      lo[j] = (bf16_t)(x[j] - (float)hi[j]);
    }
    bf16_t *d = dst + r * ldd + c;
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<bf16x4 *>(d + j * piece) = ((pattern >> j) & 1) ? lo : hi;
  }
}
// The same for cols % 8 == 0 (every operand of the model): 8 columns per thread (two 16-byte loads, three 16-byte stores), threads
// (32 x 8) = 256 columns x 8 rows per block, rows walked by blockIdx.y -- no 64-bit division per element, whole 128-byte lines.
// CS: also leave this block's column sums of the fp32 source in partial[blockIdx.y][cols] (the bias gradient that goes with a
// weight gradient: the operand is read once for both); rows in a fixed order per thread, the block's 8 row-threads in a fixed tree.
template <bool CS>
__global__ __launch_bounds__(256) void split_bf16_rows_kernel(const float *__restrict__ src, long long ld, int rows, int cols,
                                                              bf16_t *__restrict__ dst, int stack, int pattern, float *__restrict__ partial, int pieces) {
  __shared__ float red[CS ? 8 : 1][CS ? 264 : 1];
  const int c = (blockIdx.x * 32 + threadIdx.x) * 8;
  const bool live = c < cols;
  if (!CS && !live) return;
  const long long piece = stack ? (long long)rows * cols : cols, ldd = stack ? cols : 3LL * cols;      // (pieces == 2: the plane pair, stacked)
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = blockIdx.y * 8 + threadIdx.y; live && r < rows; r += gridDim.y * 8) {
    const float *sp = src + (long long)r * ld + c;
    const f32x4 x0 = dm_load4(sp), x1 = dm_load4(sp + 4);
    if constexpr (CS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { cs[j] += x0[j]; cs[4 + j] += x1[j]; }
    }
    u32x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = j < 2 ? x0[2 * j] : x1[2 * j - 4], b = j < 2 ? x0[2 * j + 1] : x1[2 * j - 3];
      const bf16x2 h2 = {(bf16_t)a, (bf16_t)b};
      const bf16x2 l2 = {(bf16_t)(a - (float)h2[0]), (bf16_t)(b - (float)h2[1])};
      hi[j] = __builtin_bit_cast(unsigned, h2);
      lo[j] = __builtin_bit_cast(unsigned, l2);
    }
    bf16_t *d = dst + (long long)r * ldd + c;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (j < pieces) *reinterpret_cast<u32x4 *>(d + j * piece) = ((pattern >> j) & 1) ? lo : hi;
  }
  if constexpr (CS) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.y][threadIdx.x * 8 + j] = cs[j];
    __syncthreads();
    if (threadIdx.y == 0 && live) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int x = threadIdx.x * 8 + j;
        partial[(long long)blockIdx.y * cols + c + j] = ((red[0][x] + red[1][x]) + (red[2][x] + red[3][x])) + ((red[4][x] + red[5][x]) + (red[6][x] + red[7][x]));
      }
    }
  }
}
__global__ void copy_f32_kernel(const float *__restrict__ src, float *__restrict__ dst, long long n) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    dm_store4(dst + 4 * i, dm_load4(src + 4 * i));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = src[(n4 << 2) + threadIdx.x];
}

template <typename T>
__global__ void patchify_kernel(const float *__restrict__ x, T *__restrict__ cols, int B, int C, int side, int p) {
  const int grid = side / p, p4 = p >> 2;
  const long long K = (long long)C * p * p;
  const long long total = (long long)B * grid * grid * (K >> 2);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int dx4 = (int)(r % p4); r /= p4;
    const int dy = (int)(r % p); r /= p;
    const int c = (int)(r % C); r /= C;     // r = row = (b*grid + py)*grid + px
    const int px = (int)(r % grid);
    const long long bp = r / grid;
    const int py = (int)(bp % grid);
    const long long b = bp / grid;
    const f32x4 v = dm_load4(x + ((b * C + c) * side + (py * p + dy)) * (long long)side + px * p + 4 * dx4);
    dm_store4(cols + r * K + ((long long)c * p + dy) * p + 4 * dx4, v);
  }
}

// ---------------------------------------------------------------------------------------------
// contrastive loss (single workgroup; B is a few hundred pairs at most)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void contrastive_loss_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                               const float *__restrict__ flag, float margin, float upstream,
                                                               float *__restrict__ loss, float *__restrict__ da,
                                                               float *__restrict__ db, int B, int D) {
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  const float gscale = upstream / (float)B;
  for (int r = wave; r < B; r += 4) {
    float d = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float t = a[(long long)r * D + c] - b[(long long)r * D + c];
      d += t * t;
    }
    d = dm_wave_sum(d);
    const float f = flag[r];
    const float hinge = margin - d;
    acc += f * d + (1.f - f) * fmaxf(hinge, 0.f);
    if (da) {
      // d(l)/d(d) = flag - (1-flag)*[margin - d > 0]   (relu'(0) = 0 as in torch)
      const float coef = (f - (1.f - f) * (hinge > 0.f ? 1.f : 0.f)) * 2.f * gscale;
      for (int c = lane; c < D; c += 64) {
        const float t = a[(long long)r * D + c] - b[(long long)r * D + c];
        da[(long long)r * D + c] = coef * t;
        db[(long long)r * D + c] = -coef * t;
      }
    }
  }
  if (lane == 0) wsum[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) / (float)B;
}

// ---------------------------------------------------------------------------------------------
// cross entropy (nn.CrossEntropyLoss, mean reduction; Losses.py:52-53, :83-84).  Single workgroup: a wave per
// row, rows reduced in a fixed order.  Targets are class indices (int64) or class probabilities (fp32 [B,K]).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float *__restrict__ logits, const long long *__restrict__ tidx,
                                                            const float *__restrict__ tprob, float upstream,
                                                            float *__restrict__ loss, float *__restrict__ dlogits, int B, int K) {
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float gscale = upstream / (float)B;
  float acc = 0.f;
  for (int r = wave; r < B; r += 4) {
    const float *x = logits + (long long)r * K;
    float mx = -INFINITY;
    for (int c = lane; c < K; c += 64) mx = fmaxf(mx, x[c]);
    mx = dm_wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < K; c += 64) se += expf(x[c] - mx);
    se = dm_wave_sum(se);
    const float lse = mx + logf(se);
    float l = 0.f, psum = 0.f;
    if (tidx) {
      const long long t = tidx[r];
      if (lane == 0) l = lse - x[t];
      psum = 1.f;
    } else {
      const float *q = tprob + (long long)r * K;
      for (int c = lane; c < K; c += 64) { l += q[c] * (lse - x[c]); psum += q[c]; }
      psum = dm_wave_sum(psum);
    }
    acc += dm_wave_sum(l);
    if (dlogits) {
      float *g = dlogits + (long long)r * K;
      for (int c = lane; c < K; c += 64) {
        const float sm = expf(x[c] - lse);
        const float tgt = tidx ? ((long long)c == tidx[r] ? 1.f : 0.f) : tprob[(long long)r * K + c];
        g[c] = (sm * psum - tgt) * gscale;
      }
    }
  }
  if (lane == 0) wsum[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) / (float)B;
}

// ---------------------------------------------------------------------------------------------
// Adam (flat buffers).  Operation order follows torch.optim.Adam's single-tensor path.
// ---------------------------------------------------------------------------------------------
__global__ void adam_kernel(float *__restrict__ param, const float *__restrict__ grad, float *__restrict__ m,
                            float *__restrict__ v, bf16_t *__restrict__ lp, long long n, float beta1, float beta2,
                            float omb1, float omb2, float eps, float step_size, float bc2_sqrt, float grad_scale,
                            const float *__restrict__ hyper, bf16_t *__restrict__ lo = nullptr) {
  if (hyper) {   // captured in a hipGraph: the step-dependent scalars live in device memory, rewritten before each replay
    step_size = hyper[0];
    bc2_sqrt = hyper[1];
  }
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    // every operand is touched once per step (1.3 GB for the headline model: nothing of it survives in the caches until the next step):
    // non-temporal loads / stores, except the bf16 weight copy that the next step's GEMMs read (86 MB: stays in the Infinity Cache)
    f32x4 p = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(param + 4 * i));
    f32x4 g = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(grad + 4 * i)) * grad_scale;
    f32x4 mm = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(m + 4 * i)), vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(v + 4 * i));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mm[e] = mm[e] * beta1 + g[e] * omb1;
      vv[e] = vv[e] * beta2 + (g[e] * g[e]) * omb2;
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      p[e] = p[e] - step_size * (mm[e] / denom);
    }
    __builtin_nontemporal_store(p, reinterpret_cast<f32x4 *>(param + 4 * i));
    __builtin_nontemporal_store(mm, reinterpret_cast<f32x4 *>(m + 4 * i));
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4 *>(v + 4 * i));
    if (lp) dm_store4(lp + 4 * i, p);
    if (lo) {      // the lo plane of the weight's hi / lo pair ("bf16x3" operands): bf16(p - bf16(p)), the split of dm_split_bf16_planes
      bf16x4 l;
#pragma unroll
      for (int e = 0; e < 4; ++e) l[e] = (bf16_t)(p[e] - (float)(bf16_t)p[e]);
      *reinterpret_cast<bf16x4 *>(lo + 4 * i) = l;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float g = grad[i] * grad_scale;
    const float mm = m[i] * beta1 + g * omb1;
    const float vv = v[i] * beta2 + (g * g) * omb2;
    const float p = param[i] - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    param[i] = p; m[i] = mm; v[i] = vv;
    if (lp) lp[i] = (bf16_t)p;
    if (lo) lo[i] = (bf16_t)(p - (float)(bf16_t)p);
  }
}

// ---------------------------------------------------------------------------------------------
// relative-position bias: gather and gradient reduction
// ---------------------------------------------------------------------------------------------
__global__ void relpos_gather_kernel(const float *__restrict__ table, const int *__restrict__ index,
                                     float *__restrict__ bias, float *__restrict__ bias_t, int N, int H, int n_bins) {
  const int NN = N * N;
  for (int ij = blockIdx.x * blockDim.x + threadIdx.x; ij < NN; ij += gridDim.x * blockDim.x) {
    int bin = index[ij];
    bin = min(max(bin, 0), n_bins - 1);
    for (int h = 0; h < H; ++h) bias[(long long)h * NN + ij] = table[bin * H + h];
    if (bias_t) {   // second pass with the roles of i and j swapped keeps both stores coalesced
      const int i = ij / N, j = ij % N;
      int bt = index[j * N + i];
      bt = min(max(bt, 0), n_bins - 1);
      for (int h = 0; h < H; ++h) bias_t[(long long)h * NN + ij] = table[bt * H + h];
    }
  }
}
// slab[0][e] = sum_c slab[c][e] in chunk order (coalesced 16-byte accesses); the gather below then works on one plane set
// that stays L2-resident instead of touching `chunks` scattered cache lines per position.
__global__ void chunk_sum_kernel(float *__restrict__ slab, long long plane4, int chunks) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < plane4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = dm_load4(slab + 4 * i);
    for (int c = 1; c < chunks; ++c) v += dm_load4(slab + ((long long)c * plane4 + i) * 4);
    dm_store4(slab + 4 * i, v);
  }
}

// d(table)[bin][h] = sum over chunks and over the positions (q,key) whose relative_position_index is `bin` of the
// dense d(bias) slab [chunks][H][N][N].  One wave per (bin, head); `pos` lists the flat positions q*N+key grouped by
// bin (CSR: off[bin]..off[bin+1]); lanes take items in a fixed interleave and the wave sum is a fixed butterfly.
__global__ __launch_bounds__(256) void relpos_reduce_kernel(const float *__restrict__ slab, const int *__restrict__ pos,
                                                            const int *__restrict__ off, float *__restrict__ dtable, int chunks,
                                                            int H, int N, int n_bins, int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bin = blockIdx.x, h = blockIdx.y * 4 + wave;
  if (h >= H) return;
  const int beg = off[bin], cnt = off[bin + 1] - beg;
  const long long plane = (long long)N * N;
  float acc = 0.f;
  for (int c = 0; c < chunks; ++c) {
    const float *pl = slab + ((long long)c * H + h) * plane;
#pragma unroll 4
    for (int j = lane; j < cnt; j += 64) acc += pl[pos[beg + j]];
  }
  acc = dm_wave_sum(acc);
  if (lane == 0) {
    float *o = dtable + (long long)bin * H + h;
    *o = accumulate ? *o + acc : acc;
  }
}

inline int grid_for(long long work_items, int block = 256, int cap = 4096) {
  long long g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

static int adam_grid(long long n) {
  static const int cap = [] { const char *e = getenv("DM_ADAM_GRID"); return e ? atoi(e) : 16384; }();   // 4096 -> 16384 workgroups: 293 -> 258 us for 48.7 M parameters (5.7 TB/s)
  return grid_for(n / 4 + 1, 256, cap);
}

}  // namespace

// =============================================================================================
extern "C" int dm_layernorm_fwd(const float *x, const float *gamma, const float *beta, void *y, int32_t y_dtype,
                                float *mean, float *rstd, int32_t rows, int32_t cols, float eps, void *stream) {
  DM_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= LN_WIDE_MAX, DM_ERR_BAD_SHAPE,
             "dm_layernorm_fwd: rows=%d cols=%d (cols must be a multiple of 4, <= %d)", rows, cols, LN_WIDE_MAX);
  DM_REQUIRE(x && gamma && beta && y && mean && rstd, DM_ERR_BAD_SHAPE, "dm_layernorm_fwd: null pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (cols > MAXCH * 256) {
    DM_REQUIRE(y_dtype != DM_BF16_PAIR, DM_ERR_UNSUPPORTED, "dm_layernorm_fwd: a plane-pair result needs cols <= %d", MAXCH * 256);
    const int rc = dm_layernorm_wide_fwd(x, gamma, beta, y, y_dtype, mean, rstd, rows, cols, eps, s);
    DM_REQUIRE(rc == DM_OK, rc, "dm_layernorm_fwd: bad y_dtype %d", y_dtype);
    DM_LAUNCH_CHECK("dm_layernorm_fwd(wide)");
    return DM_OK;
  }
  const int grid = grid_for((long long)rows, 4, 2048);
  if (y_dtype == DM_F32)
    hipLaunchKernelGGL(layernorm_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, x, gamma, beta, (float *)y, mean, rstd, rows, cols, eps);
  else if (y_dtype == DM_BF16)
    hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, x, gamma, beta, (bf16_t *)y, mean, rstd, rows, cols, eps);
  else if (y_dtype == DM_BF16_PAIR)
    hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, s, x, gamma, beta, (bf16_t *)y, mean, rstd, rows, cols, eps);
  else DM_REQUIRE(false, DM_ERR_BAD_DTYPE, "dm_layernorm_fwd: bad y_dtype %d", y_dtype);
  DM_LAUNCH_CHECK("dm_layernorm_fwd");
  return DM_OK;
}

extern "C" int64_t dm_layernorm_bwd_partial_floats(int32_t cols) { return (int64_t)ln_max_wg() * 2 * cols; }

// main kernel of the LayerNorm backward: dx (+ dres), optional bf16 copy, one [dgamma | dbeta] partial row per workgroup; *n_partial rows
static int ln_bwd_main(const void *dy, int32_t dy_dtype, const float *x, const float *gamma, const float *mean, const float *rstd,
                       const float *dres, float *dx, void *dx_lp, float *partial, int32_t rows, int32_t cols, int32_t *n_partial, hipStream_t s,
                       const char *who, bool lp_pair = false) {
  const long long lp_plane = lp_pair ? (long long)rows * cols : 0;
  DM_REQUIRE(!lp_pair || (dx_lp && cols <= MAXCH * 256), DM_ERR_UNSUPPORTED, "%s: the plane-pair copy needs dx_lp and cols <= %d", who, MAXCH * 256);
  DM_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= LN_WIDE_MAX, DM_ERR_BAD_SHAPE, "%s: rows=%d cols=%d", who, rows, cols);
  DM_REQUIRE(dy && x && gamma && mean && rstd && dx && partial && n_partial, DM_ERR_BAD_SHAPE, "%s: null pointer", who);
  if (cols > MAXCH * 256) {
    int slices = 0;
    const int rc = dm_layernorm_wide_bwd(dy, dy_dtype, x, gamma, mean, rstd, dres, dx, dx_lp, partial, ln_max_wg(), &slices, rows, cols, s);
    DM_REQUIRE(rc == DM_OK, rc, "%s: bad dy_dtype %d", who, dy_dtype);
    DM_LAUNCH_CHECK("dm_layernorm_bwd(wide)");
    *n_partial = slices;
    return DM_OK;
  }
  const int grid = grid_for((long long)rows, 4, ln_max_wg());
  DM_REQUIRE(dy_dtype == DM_F32 || dy_dtype == DM_BF16, DM_ERR_BAD_DTYPE, "%s: bad dy_dtype %d", who, dy_dtype);
  const bool narrow = cols <= 768;
  if (dy_dtype == DM_F32 && narrow)
    hipLaunchKernelGGL((layernorm_bwd_kernel<float, 3>), dim3(grid), dim3(256), 0, s, (const float *)dy, x, gamma, mean, rstd, dres, dx, (bf16_t *)dx_lp, partial, rows, cols, lp_plane);
  else if (dy_dtype == DM_F32)
    hipLaunchKernelGGL((layernorm_bwd_kernel<float, MAXCH>), dim3(grid), dim3(256), 0, s, (const float *)dy, x, gamma, mean, rstd, dres, dx, (bf16_t *)dx_lp, partial, rows, cols, lp_plane);
  else if (narrow)
    hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, 3>), dim3(grid), dim3(256), 0, s, (const bf16_t *)dy, x, gamma, mean, rstd, dres, dx, (bf16_t *)dx_lp, partial, rows, cols, lp_plane);
  else
    hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, MAXCH>), dim3(grid), dim3(256), 0, s, (const bf16_t *)dy, x, gamma, mean, rstd, dres, dx, (bf16_t *)dx_lp, partial, rows, cols, lp_plane);
  DM_LAUNCH_CHECK("dm_layernorm_bwd");
  *n_partial = grid;
  return DM_OK;
}

extern "C" int dm_layernorm_bwd(const void *dy, int32_t dy_dtype, const float *x, const float *gamma, const float *mean,
                                const float *rstd, const float *dres, float *dx, void *dx_lp, float *dgamma, float *dbeta,
                                int32_t accumulate_params, float *partial, int32_t rows, int32_t cols, void *stream) {
  DM_REQUIRE(dgamma && dbeta, DM_ERR_BAD_SHAPE, "dm_layernorm_bwd: null pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int32_t slices = 0;
  if (const int rc = ln_bwd_main(dy, dy_dtype, x, gamma, mean, rstd, dres, dx, dx_lp, partial, rows, cols, &slices, s, "dm_layernorm_bwd")) return rc;
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((2 * cols + 15) / 16), dim3(256), 0, s, partial, dgamma, dbeta, slices, 2 * cols, cols, accumulate_params);
  DM_LAUNCH_CHECK("dm_layernorm_bwd(reduce)");
  return DM_OK;
}

extern "C" int dm_layernorm_bwd_partials(const void *dy, int32_t dy_dtype, const float *x, const float *gamma, const float *mean,
                                         const float *rstd, const float *dres, float *dx, void *dx_lp, float *partial, int32_t rows,
                                         int32_t cols, int32_t *n_partial, void *stream) {
  return ln_bwd_main(dy, dy_dtype, x, gamma, mean, rstd, dres, dx, dx_lp, partial, rows, cols, n_partial, reinterpret_cast<hipStream_t>(stream),
                     "dm_layernorm_bwd_partials");
}

extern "C" int dm_layernorm_bwd_partials_pair(const void *dy, int32_t dy_dtype, const float *x, const float *gamma, const float *mean,
                                              const float *rstd, const float *dres, float *dx, void *dx_pair, float *partial, int32_t rows,
                                              int32_t cols, int32_t *n_partial, void *stream) {
  return ln_bwd_main(dy, dy_dtype, x, gamma, mean, rstd, dres, dx, dx_pair, partial, rows, cols, n_partial, reinterpret_cast<hipStream_t>(stream),
                     "dm_layernorm_bwd_partials_pair", true);
}

extern "C" int dm_partial_reduce_batch(const DmReduceItem *items, int32_t n, void *stream) {
  DM_REQUIRE(items && n > 0, DM_ERR_BAD_SHAPE, "dm_partial_reduce_batch: no items");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int32_t base = 0; base < n; base += 32) {
    ReduceBatch rb;
    rb.n = (n - base < 32) ? n - base : 32;
    int total = 0;
    for (int i = 0; i < rb.n; ++i) {
      const DmReduceItem &it = items[base + i];
      DM_REQUIRE(it.partial && it.out0 && it.out1 && it.nrows > 0 && it.width > 0 && it.split >= 0 && it.split <= it.width, DM_ERR_BAD_SHAPE,
                 "dm_partial_reduce_batch: item %d: nrows=%d width=%d split=%d", base + i, it.nrows, it.width, it.split);
      rb.it[i] = it;
      rb.start[i] = total;
      total += (it.width + 15) / 16;
    }
    for (int i = rb.n; i <= 32; ++i) rb.start[i] = total;
    hipLaunchKernelGGL(partial_reduce_batch_kernel, dim3(total), dim3(256), 0, s, rb);
    DM_LAUNCH_CHECK("dm_partial_reduce_batch");
  }
  return DM_OK;
}

extern "C" int dm_token_pool_fwd(const float *x, float *y, int32_t B, int32_t S, int32_t side, int32_t C, void *stream) {
  DM_REQUIRE(B > 0 && S > 0 && side >= 2 && side % 2 == 0 && C % 4 == 0, DM_ERR_BAD_SHAPE, "dm_token_pool_fwd: B=%d S=%d side=%d C=%d", B, S, side, C);
  const long long total = (long long)B * S * (side / 2) * (side / 2) * (C / 4);
  hipLaunchKernelGGL(token_pool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, B, S, side, C);
  DM_LAUNCH_CHECK("dm_token_pool_fwd");
  return DM_OK;
}
extern "C" int dm_token_pool_bwd(const float *dy, float *dx, int32_t B, int32_t S, int32_t side, int32_t C, void *stream) {
  DM_REQUIRE(B > 0 && S > 0 && side >= 2 && side % 2 == 0 && C % 4 == 0, DM_ERR_BAD_SHAPE, "dm_token_pool_bwd: B=%d S=%d side=%d C=%d", B, S, side, C);
  const long long total = (long long)B * S * side * side * (C / 4);
  hipLaunchKernelGGL(token_pool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, dx, B, S, side, C);
  DM_LAUNCH_CHECK("dm_token_pool_bwd");
  return DM_OK;
}
extern "C" int dm_group_mean_fwd(const float *x, float *y, int32_t rows, int32_t g, int32_t C, void *stream) {
  DM_REQUIRE(rows > 0 && g > 0 && C % 4 == 0, DM_ERR_BAD_SHAPE, "dm_group_mean_fwd: rows=%d g=%d C=%d", rows, g, C);
  hipLaunchKernelGGL(group_mean_fwd_kernel, dim3(grid_for((long long)rows * (C / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, rows, g, C);
  DM_LAUNCH_CHECK("dm_group_mean_fwd");
  return DM_OK;
}
extern "C" int dm_group_mean_bwd(const float *dy, float *dx, int32_t rows, int32_t g, int32_t C, void *stream) {
  DM_REQUIRE(rows > 0 && g > 0 && C % 4 == 0, DM_ERR_BAD_SHAPE, "dm_group_mean_bwd: rows=%d g=%d C=%d", rows, g, C);
  hipLaunchKernelGGL(group_mean_bwd_kernel, dim3(grid_for((long long)rows * g * (C / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, dx, rows, g, C);
  DM_LAUNCH_CHECK("dm_group_mean_bwd");
  return DM_OK;
}

extern "C" int64_t dm_colsum_partial_floats(int32_t N) { return (int64_t)CS_MAX_SLICES * N; }

extern "C" int dm_colsum(const void *X, int32_t dtype, int64_t ldx, float *out, int32_t M, int32_t N, int32_t accumulate,
                         float *partial, void *stream) {
  DM_REQUIRE(M > 0 && N > 0 && ldx >= N, DM_ERR_BAD_SHAPE, "dm_colsum: M=%d N=%d ldx=%lld", M, N, (long long)ldx);
  const bool vec = (N % 4 == 0) && (ldx % 4 == 0) && dm_aligned16(X);
  DM_REQUIRE(X && out && partial, DM_ERR_BAD_SHAPE, "dm_colsum: null pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int slices = (M + 255) / 256;
  if (slices > CS_MAX_SLICES) slices = CS_MAX_SLICES;
  const int rps = (M + slices - 1) / slices;
  slices = (M + rps - 1) / rps;
  dim3 grid((N + 63) / 64, slices), ggrid((N + 255) / 256, slices);
  // one slice, vector form: out[n] = (accumulate ? out[n] : 0) + sum -- what the reduction launch would compute from the single partial row
  const bool direct = slices == 1 && vec && dm_aligned16(out);
  float *dst = direct ? out : partial;
  const int dflag = direct ? (accumulate ? 1 : 0) : -1;
  if (dtype == DM_F32) {
    if (vec) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float *)X, (long long)ldx, dst, M, N, rps, dflag);
    else hipLaunchKernelGGL(colsum_generic_kernel<float>, ggrid, dim3(256), 0, s, (const float *)X, (long long)ldx, partial, M, N, rps);
  } else if (dtype == DM_BF16) {
    if (vec) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t *)X, (long long)ldx, dst, M, N, rps, dflag);
    else hipLaunchKernelGGL(colsum_generic_kernel<bf16_t>, ggrid, dim3(256), 0, s, (const bf16_t *)X, (long long)ldx, partial, M, N, rps);
  } else DM_REQUIRE(false, DM_ERR_BAD_DTYPE, "dm_colsum: bad dtype %d", dtype);
  DM_LAUNCH_CHECK("dm_colsum");
  if (direct) return DM_OK;
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((N + 15) / 16), dim3(256), 0, s, partial, out, out, slices, N, N, accumulate);
  DM_LAUNCH_CHECK("dm_colsum(reduce)");
  return DM_OK;
}

extern "C" int dm_cast(const float *src, void *dst, int32_t dst_dtype, int64_t n, void *stream) {
  DM_REQUIRE(src && dst && n > 0, DM_ERR_BAD_SHAPE, "dm_cast: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dst_dtype == DM_BF16) hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, s, src, (bf16_t *)dst, (long long)n);
  else if (dst_dtype == DM_F32) hipLaunchKernelGGL(copy_f32_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, src, (float *)dst, (long long)n);
  else DM_REQUIRE(false, DM_ERR_BAD_DTYPE, "dm_cast: bad dtype %d", dst_dtype);
  DM_LAUNCH_CHECK("dm_cast");
  return DM_OK;
}

static bool split_rows_ok(const float *src, int64_t ld, int64_t rows, int64_t cols, const void *dst, int32_t stack) {
  return cols % 8 == 0 && rows < (1LL << 31) && cols < (1LL << 31) && dm_aligned16(src) && dm_aligned16(dst) && ld % 4 == 0 &&
         (stack ? (rows * cols) % 8 == 0 : true);
}
static void split_rows_grid(int64_t rows, int64_t cols, int &gx, int &gy, bool colsum = false) {
  gx = (int)((cols / 8 + 31) / 32);
  long long y = (rows + 7) / 8;
  // ~32 blocks per CU in flight at most; rows beyond are walked.  With column sums every row of blocks leaves a partial row behind
  // (8 MB for a 16384 x 3072 operand at 8192 blocks, ~7 us to fold): a quarter of the blocks still fills the chip (8 per CU)
  const int blocks = colsum ? 2048 : 8192;
  const long long cap = blocks / gx > 0 ? blocks / gx : 1;
  gy = (int)(y > cap ? cap : y);
}

extern "C" int64_t dm_split_colsum_partial_floats(int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0 || cols % 8) return 0;
  int gx, gy;
  split_rows_grid(rows, cols, gx, gy, true);
  return (int64_t)gy * cols;
}

extern "C" int dm_split_bf16_colsum(const float *src, int64_t ld, int64_t rows, int64_t cols, void *dst, int32_t stack, int32_t pattern,
                                    float *partial, int32_t *n_partial, void *stream) {
  DM_REQUIRE(src && dst && partial && n_partial && rows > 0 && cols > 0 && ld >= cols && pattern >= 0 && pattern < 8, DM_ERR_BAD_SHAPE,
             "dm_split_bf16_colsum: bad arguments (rows=%lld cols=%lld ld=%lld)", (long long)rows, (long long)cols, (long long)ld);
  DM_REQUIRE(split_rows_ok(src, ld, rows, cols, dst, stack), DM_ERR_UNSUPPORTED,
             "dm_split_bf16_colsum: needs cols %% 8 == 0, ld %% 4 == 0 and 16-byte aligned tensors (cols=%lld ld=%lld)", (long long)cols, (long long)ld);
  int gx, gy;
  split_rows_grid(rows, cols, gx, gy, true);
  hipLaunchKernelGGL(split_bf16_rows_kernel<true>, dim3(gx, (unsigned)gy), dim3(32, 8), 0, reinterpret_cast<hipStream_t>(stream), src, (long long)ld,
                     (int)rows, (int)cols, (bf16_t *)dst, stack ? 1 : 0, pattern, partial, 3);
  DM_LAUNCH_CHECK("dm_split_bf16_colsum");
  *n_partial = gy;
  return DM_OK;
}

extern "C" int dm_split_bf16_planes(const float *src, int64_t ld, int64_t rows, int64_t cols, void *dst, float *partial, int32_t *n_partial,
                                    void *stream) {
  DM_REQUIRE(src && dst && rows > 0 && cols > 0 && ld >= cols && ((partial == nullptr) == (n_partial == nullptr)), DM_ERR_BAD_SHAPE,
             "dm_split_bf16_planes: bad arguments (rows=%lld cols=%lld ld=%lld)", (long long)rows, (long long)cols, (long long)ld);
  DM_REQUIRE(split_rows_ok(src, ld, rows, cols, dst, 1), DM_ERR_UNSUPPORTED,
             "dm_split_bf16_planes: needs cols %% 8 == 0, ld %% 4 == 0 and 16-byte aligned tensors (cols=%lld ld=%lld)", (long long)cols, (long long)ld);
  int gx, gy;
  split_rows_grid(rows, cols, gx, gy, partial != nullptr);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (partial)
    hipLaunchKernelGGL(split_bf16_rows_kernel<true>, dim3(gx, (unsigned)gy), dim3(32, 8), 0, s, src, (long long)ld, (int)rows, (int)cols, (bf16_t *)dst, 1, 0b10, partial, 2);
  else
    hipLaunchKernelGGL(split_bf16_rows_kernel<false>, dim3(gx, (unsigned)gy), dim3(32, 8), 0, s, src, (long long)ld, (int)rows, (int)cols, (bf16_t *)dst, 1, 0b10,
                       (float *)nullptr, 2);
  DM_LAUNCH_CHECK("dm_split_bf16_planes");
  if (n_partial) *n_partial = gy;
  return DM_OK;
}

extern "C" int dm_split_bf16(const float *src, int64_t ld, int64_t rows, int64_t cols, void *dst, int32_t stack, int32_t pattern,
                             void *stream) {
  DM_REQUIRE(src && dst && rows > 0 && cols > 0 && cols % 4 == 0 && ld >= cols && ld % 4 == 0, DM_ERR_BAD_SHAPE,
             "dm_split_bf16: rows=%lld cols=%lld ld=%lld (cols and ld must be multiples of 4)", (long long)rows, (long long)cols, (long long)ld);
  DM_REQUIRE(pattern >= 0 && pattern < 8, DM_ERR_BAD_SHAPE, "dm_split_bf16: pattern %d", pattern);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (split_rows_ok(src, ld, rows, cols, dst, stack)) {
    int gx, gy;
    split_rows_grid(rows, cols, gx, gy);
    hipLaunchKernelGGL(split_bf16_rows_kernel<false>, dim3(gx, (unsigned)gy), dim3(32, 8), 0, s, src, (long long)ld, (int)rows, (int)cols,
                       (bf16_t *)dst, stack ? 1 : 0, pattern, (float *)nullptr, 3);
  } else {
    hipLaunchKernelGGL(split_bf16_kernel, dim3(grid_for(rows * (cols / 4))), dim3(256), 0, s, src, (long long)ld, (long long)rows,
                       (long long)cols, (bf16_t *)dst, stack ? 1 : 0, pattern);
  }
  DM_LAUNCH_CHECK("dm_split_bf16");
  return DM_OK;
}

extern "C" int dm_patchify(const float *x, void *cols, int32_t dtype, int32_t B, int32_t C, int32_t side, int32_t p, void *stream) {
  DM_REQUIRE(B > 0 && C > 0 && p > 0 && side % p == 0, DM_ERR_BAD_SHAPE,
             "dm_patchify: B=%d C=%d side=%d patch=%d (patch must divide side)", B, C, side, p);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (p % 4 != 0) {
    const int rc = dm_patchify_any(x, cols, dtype, B, C, side, p, s);
    DM_REQUIRE(rc == DM_OK, rc, "dm_patchify: bad dtype %d", dtype);
    DM_LAUNCH_CHECK("dm_patchify(any)");
    return DM_OK;
  }
  const long long total = (long long)B * (side / p) * (side / p) * ((long long)C * p * p / 4);
  if (dtype == DM_F32) hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, x, (float *)cols, B, C, side, p);
  else if (dtype == DM_BF16) hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, x, (bf16_t *)cols, B, C, side, p);
  else DM_REQUIRE(false, DM_ERR_BAD_DTYPE, "dm_patchify: bad dtype %d", dtype);
  DM_LAUNCH_CHECK("dm_patchify");
  return DM_OK;
}

extern "C" int dm_cross_entropy(const float *logits, const int64_t *target_index, const float *target_prob, float upstream,
                                float *loss, float *dlogits, int32_t B, int32_t K, void *stream) {
  DM_REQUIRE(logits && loss && B > 0 && K > 0, DM_ERR_BAD_SHAPE, "dm_cross_entropy: bad arguments");
  DM_REQUIRE((target_index != nullptr) != (target_prob != nullptr), DM_ERR_BAD_SHAPE,
             "dm_cross_entropy: exactly one of target_index / target_prob must be given");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits,
                     reinterpret_cast<const long long *>(target_index), target_prob, upstream, loss, dlogits, B, K);
  DM_LAUNCH_CHECK("dm_cross_entropy");
  return DM_OK;
}

extern "C" int dm_contrastive_loss(const float *a, const float *b, const float *flag, float margin, float upstream,
                                   float *loss, float *da, float *db, int32_t B, int32_t D, void *stream) {
  DM_REQUIRE(a && b && flag && loss && B > 0 && D > 0, DM_ERR_BAD_SHAPE, "dm_contrastive_loss: bad arguments");
  DM_REQUIRE((da == nullptr) == (db == nullptr), DM_ERR_BAD_SHAPE, "dm_contrastive_loss: da and db must both be set or both NULL");
  hipLaunchKernelGGL(contrastive_loss_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b, flag, margin, upstream, loss, da, db, B, D);
  DM_LAUNCH_CHECK("dm_contrastive_loss");
  return DM_OK;
}

extern "C" int dm_adam_step(float *param, const float *grad, float *m, float *v, void *param_lp, int64_t n, int32_t step,
                            double lr, double beta1, double beta2, double eps, double grad_scale, void *stream) {
  DM_REQUIRE(param && grad && m && v && n > 0 && step >= 1, DM_ERR_BAD_SHAPE, "dm_adam_step: bad arguments");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(adam_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param, grad, m, v,
                     (bf16_t *)param_lp, (long long)n, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2),
                     (float)eps, step_size, bc2_sqrt, (float)grad_scale, (const float *)nullptr);
  DM_LAUNCH_CHECK("dm_adam_step");
  return DM_OK;
}

extern "C" int dm_adam_hyper(int32_t step, double lr, double beta1, double beta2, float *hyper_host) {
  DM_REQUIRE(step >= 1 && hyper_host, DM_ERR_BAD_SHAPE, "dm_adam_hyper: step is 1-based and hyper_host must be set");
  hyper_host[0] = (float)(lr / (1.0 - pow(beta1, (double)step)));
  hyper_host[1] = (float)sqrt(1.0 - pow(beta2, (double)step));
  return DM_OK;
}

extern "C" int dm_adam_step_dev(float *param, const float *grad, float *m, float *v, void *param_lp, int64_t n, const float *hyper_dev,
                                double beta1, double beta2, double eps, double grad_scale, void *stream) {
  DM_REQUIRE(param && grad && m && v && hyper_dev && n > 0, DM_ERR_BAD_SHAPE, "dm_adam_step_dev: bad arguments");
  DM_REQUIRE(dm_aligned16(param) && dm_aligned16(grad) && dm_aligned16(m) && dm_aligned16(v) && dm_aligned16(param_lp), DM_ERR_BAD_ALIGN,
             "dm_adam_step_dev: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel, dim3(adam_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param, grad, m, v,
                     (bf16_t *)param_lp, (long long)n, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2),
                     (float)eps, 0.f, 1.f, (float)grad_scale, hyper_dev);
  DM_LAUNCH_CHECK("dm_adam_step_dev");
  return DM_OK;
}

extern "C" int dm_adam_step_dev_pair(float *param, const float *grad, float *m, float *v, void *param_hi, void *param_lo, int64_t n,
                                     const float *hyper_dev, double beta1, double beta2, double eps, double grad_scale, void *stream) {
  DM_REQUIRE(param && grad && m && v && hyper_dev && param_hi && param_lo && n > 0, DM_ERR_BAD_SHAPE, "dm_adam_step_dev_pair: bad arguments");
  DM_REQUIRE(dm_aligned16(param) && dm_aligned16(grad) && dm_aligned16(m) && dm_aligned16(v) && dm_aligned16(param_hi) && dm_aligned16(param_lo),
             DM_ERR_BAD_ALIGN, "dm_adam_step_dev_pair: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel, dim3(adam_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param, grad, m, v,
                     (bf16_t *)param_hi, (long long)n, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2),
                     (float)eps, 0.f, 1.f, (float)grad_scale, hyper_dev, (bf16_t *)param_lo);
  DM_LAUNCH_CHECK("dm_adam_step_dev_pair");
  return DM_OK;
}

extern "C" int dm_relpos_bias_gather(const float *table, const int32_t *index, float *bias, float *bias_t, int32_t N, int32_t H,
                                     int32_t n_bins, void *stream) {
  DM_REQUIRE(table && index && bias && N > 0 && H > 0 && n_bins > 0, DM_ERR_BAD_SHAPE, "dm_relpos_bias_gather: bad arguments");
  hipLaunchKernelGGL(relpos_gather_kernel, dim3(grid_for((long long)N * N)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table, index, bias, bias_t, N, H, n_bins);
  DM_LAUNCH_CHECK("dm_relpos_bias_gather");
  return DM_OK;
}
extern "C" int dm_relpos_bias_reduce(float *dbias_slab, const int32_t *positions, const int32_t *offsets, float *dtable,
                                     int32_t chunks, int32_t H, int32_t N, int32_t n_bins, int32_t accumulate, void *stream) {
  DM_REQUIRE(dbias_slab && positions && offsets && dtable && chunks > 0 && H > 0 && N > 0 && n_bins > 0, DM_ERR_BAD_SHAPE,
             "dm_relpos_bias_reduce: bad arguments");
  DM_REQUIRE(n_bins <= 65535 * 32, DM_ERR_BAD_SHAPE, "dm_relpos_bias_reduce: too many bins (%d)", n_bins);
  const long long plane = (long long)H * N * N;
  if (chunks > 1 && plane % 4 == 0 && dm_aligned16(dbias_slab)) {
    hipLaunchKernelGGL(chunk_sum_kernel, dim3(grid_for(plane / 4, 256, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dbias_slab,
                       plane / 4, chunks);
    chunks = 1;
  }
  hipLaunchKernelGGL(relpos_reduce_kernel, dim3(n_bins, (H + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dbias_slab,
                     positions, offsets, dtable, chunks, H, N, n_bins, accumulate);
  DM_LAUNCH_CHECK("dm_relpos_bias_reduce");
  return DM_OK;
}
