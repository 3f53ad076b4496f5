// Fused attention with dense relative-position bias for the DeepMerge token cubes (gfx950).
//
// Problem shape: per (sample b, head h) one N x N attention with N <= 256 tokens and head dim
// D = 64 (N = 12..256 for ShfitScaleFormer stages, 197/198 for the ViT variants).  The whole
// score row fits in registers, so softmax is exact (no online rescaling) and nothing of size
// N x N ever goes to HBM.
//
// Forward  : workgroup = (64-query block, h, b), 4 waves x 16 query rows.
//            K tile -> LDS (k-contiguous image) -> S = scale*Q K^T + bias -> softmax in registers
//            (lane holds 4 consecutive keys of one query row; row reductions are 2 butterflies) ->
//            P -> LDS (wave-private, operand dtype) ; V tile -> LDS (transposed-read image) ->
//            O = P V.  K and V time-share one LDS buffer.
// Backward : two kernels so that no gradient is accumulated across workgroups:
//            dq kernel  (64-query block): recompute P, dP = dO V^T, dS = P*(dP - delta),
//                        dQ = scale * dS K; also delta = rowsum(dO*O) and the relative-position
//                        bias gradient binned into an LDS histogram (table bins) -> slab row.
//            dkv kernel (64-key block):   recompute P^T, dV = P^T dO, dK = scale * dS^T Q.
// MFMA operand conventions are those of dm_gemm.hip (16-byte fragments, swapped operands so a
// lane owns 4 consecutive columns of one row).
#include "dm_common.h"
#include "dm_mfma.h"
#include "dm_prof.h"

namespace {

constexpr int HD = 64;         // head dim
constexpr int QB = 64;         // rows (queries or keys) per workgroup
constexpr int MAX_BINS = 4096; // LDS histogram capacity (3-D table for a 4x8x8 cube has 1575 bins)

template <typename T> struct AttnLayout {
  static constexpr int RB = HD * (int)sizeof(T);       // bytes per 64-d row: 128 / 256
  static constexpr int CPRW = RB / 16;                 // 16-byte chunks per row: 8 / 16
  static constexpr int KBD = RB / 64;                  // 64-byte k-blocks across d: 2 / 4
  static constexpr int MROW = (sizeof(T) == 2) ? 128 : 272;   // row stride of the transposed-read image
  static constexpr int EPC = 16 / (int)sizeof(T);      // elements per chunk
};

__device__ __forceinline__ int fv_swz(int k) { return ((k >> 1) & 1) | (((k >> 3) & 1) << 1); }

// ---- global [rows][64] (row stride `ld` elements) -> LDS images ------------------------------
// k-contiguous image: addr(row, chunk) = row*RB + ((chunk ^ (row & (CPRW-1))) << 4)
template <typename T>
__device__ __forceinline__ void tile_to_lds_k(char *lds, const T *g, long long ld, int nvalid, int nrows, int t) {
  using L = AttnLayout<T>;
  const int c = t % L::CPRW;
  constexpr int RPI = 256 / L::CPRW;
  for (int row = t / L::CPRW; row < nrows; row += RPI) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < nvalid) v = *reinterpret_cast<const u32x4 *>(g + (long long)row * ld + c * L::EPC);
    *reinterpret_cast<u32x4 *>(lds + row * L::RB + ((c ^ (row & (L::CPRW - 1))) << 4)) = v;
  }
}
// transposed-read image (rows are the contraction index):
//   bf16: addr(k, d) = k*128 + (((d>>4) ^ fv(k)) << 5) + (d&15)*2     fp32: addr(k, d) = k*272 + d*4
template <typename T>
__device__ __forceinline__ void tile_to_lds_m(char *lds, const T *g, long long ld, int nvalid, int nrows, int t) {
  using L = AttnLayout<T>;
  const int c = t % L::CPRW;
  constexpr int RPI = 256 / L::CPRW;
  for (int row = t / L::CPRW; row < nrows; row += RPI) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < nvalid) v = *reinterpret_cast<const u32x4 *>(g + (long long)row * ld + c * L::EPC);
    if constexpr (sizeof(T) == 2)
      *reinterpret_cast<u32x4 *>(lds + row * 128 + ((((c >> 1) ^ fv_swz(row))) << 5) + ((c & 1) << 4)) = v;
    else
      *reinterpret_cast<u32x4 *>(lds + row * 272 + (c << 4)) = v;
  }
}
// fragment of a k-contiguous image: tile row `row`, 64-byte block kb
template <typename T> __device__ __forceinline__ u32x4 lds_frag_k(const char *lds, int row, int kb, int lane) {
  using L = AttnLayout<T>;
  const int chunk = kb * 4 + (lane >> 4);
  return *reinterpret_cast<const u32x4 *>(lds + row * L::RB + ((chunk ^ (row & (L::CPRW - 1))) << 4));
}
// fragment of a transposed-read image: columns d0..d0+15, contraction block kb
template <typename T> __device__ __forceinline__ u32x4 lds_frag_m(const char *lds, int d0, int kb, int lane);
template <> __device__ __forceinline__ u32x4 lds_frag_m<bf16_t>(const char *lds, int d0, int kb, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int d = d0 + 4 * p;
  u32x4 out;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int k = kb * 32 + 8 * g + 4 * half + q;
    const u32x2 w = dm_ds_read_tr16(lds + k * 128 + ((((d >> 4) ^ fv_swz(k))) << 5) + ((d & 15) << 1));
    out[2 * half] = w[0];
    out[2 * half + 1] = w[1];
  }
  return out;
}
template <> __device__ __forceinline__ u32x4 lds_frag_m<float>(const char *lds, int d0, int kb, int lane) {
  const int g = lane >> 4, i = lane & 15;
  u32x4 out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = kb * 16 + 4 * g + j;
    out[j] = *reinterpret_cast<const unsigned int *>(lds + k * 272 + ((d0 + i) << 2));
  }
  return out;
}
// fragment straight from global: row pointer `rowp` (64 contiguous d), block kb; zero if !valid
template <typename T> __device__ __forceinline__ u32x4 gl_frag(const T *rowp, bool valid, int kb, int lane) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (valid) v = *reinterpret_cast<const u32x4 *>(rowp + (kb * 4 + (lane >> 4)) * AttnLayout<T>::EPC);
  return v;
}

// reductions over the 4 lane groups that share a row (lanes i, i+16, i+32, i+48)
__device__ __forceinline__ float row_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float row_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

template <typename T> __device__ __forceinline__ void store_row4(char *p, f32x4 v);
template <> __device__ __forceinline__ void store_row4<float>(char *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
template <> __device__ __forceinline__ void store_row4<bf16_t>(char *p, f32x4 v) {
  bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  *reinterpret_cast<bf16x4 *>(p) = r;
}

struct AttnParams {
  const void *qkv;
  const float *bias;
  const void *out;     // forward: written; backward: read
  const void *dout;
  float *lse;
  float *delta;
  void *dqkv;
  const int *index;
  float *slab;
  int B, N, H, n_bins, nblk;
  float scale;
};

// Shared score computation: acc[kt] <- scale * (rows x tile^T) + bias, masked; `rows` fragments in
// `fa` (this wave's 16 rows, KBD blocks), tile in the k-contiguous LDS image.
// TRANSPOSED = false: rows are queries, columns keys  -> bias[h][row][col]
// TRANSPOSED = true : rows are keys, columns queries  -> bias[h][col][row]
template <typename T, int NKT, bool TRANSPOSED>
__device__ __forceinline__ void scores(f32x4 (&acc)[NKT], const u32x4 (&fa)[AttnLayout<T>::KBD], const char *lds,
                                       const float *bias_h, int N, int row, float scale, int lane) {
  using L = AttnLayout<T>;
  const int g = lane >> 4, li = lane & 15;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < L::KBD; ++kb) mma<T>(a, fa[kb], lds_frag_k<T>(lds, kt * 16 + li, kb, lane));
    const int col = kt * 16 + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = a[r] * scale;
      const bool ok = (row < N) && (col + r < N);
      if (bias_h && ok) s += TRANSPOSED ? bias_h[(long long)(col + r) * N + row] : bias_h[(long long)row * N + col + r];
      a[r] = ok ? s : -INFINITY;
    }
    acc[kt] = a;
  }
}

// =============================================================================================
// forward
// =============================================================================================
template <typename T, int NKT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
  using L = AttnLayout<T>;
  constexpr int NK = NKT * 16;
  constexpr int KV_BYTES = NK * (L::RB > L::MROW ? L::RB : L::MROW);
  constexpr int PS = NK * (int)sizeof(T) + 16;     // P row stride (bytes)
  constexpr int NKBP = NK * (int)sizeof(T) / 64;   // 64-byte blocks along keys
  __shared__ __attribute__((aligned(16))) char smem[KV_BYTES + 4 * 16 * PS];
  char *kv = smem;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, li = lane & 15;
  const int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int N = p.N, H = p.H;
  char *pw = smem + KV_BYTES + wave * 16 * PS;

  const long long tok_stride = 3LL * H * HD;
  const T *base = reinterpret_cast<const T *>(p.qkv) + (long long)b * N * tok_stride + (long long)h * HD;
  const T *Kg = base + (long long)H * HD, *Vg = base + 2LL * H * HD;
  const int q = qblk * QB + wave * 16 + li;

  tile_to_lds_k<T>(kv, Kg, tok_stride, N, NK, t);
  u32x4 fq[L::KBD];
#pragma unroll
  for (int kb = 0; kb < L::KBD; ++kb) fq[kb] = gl_frag<T>(base + (long long)q * tok_stride, q < N, kb, lane);
  __syncthreads();

  f32x4 s[NKT];
  scores<T, NKT, false>(s, fq, kv, p.bias ? p.bias + (long long)h * N * N : nullptr, N, q, p.scale, lane);

  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kt][r]);
  m = row_max(m);
  const float msafe = (m == -INFINITY) ? 0.f : m;
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = expf(s[kt][r] - msafe);
      s[kt][r] = e;
      l += e;
    }
  l = row_sum(l);
  const float inv = (l > 0.f) ? 1.f / l : 0.f;
  if (g == 0 && q < N) p.lse[((long long)b * H + h) * N + q] = msafe + logf(l);
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) store_row4<T>(pw + li * PS + (kt * 16 + 4 * g) * (int)sizeof(T), s[kt] * inv);

  __syncthreads();   // every wave is done reading the K image
  tile_to_lds_m<T>(kv, Vg, tok_stride, N, NK, t);
  __syncthreads();

  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < NKBP; ++kb) {
    const u32x4 fp = *reinterpret_cast<const u32x4 *>(pw + li * PS + (kb * 4 + g) * 16);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) mma<T>(o[dt], fp, lds_frag_m<T>(kv, dt * 16, kb, lane));
  }
  if (q < N) {
    T *orow = reinterpret_cast<T *>(const_cast<void *>(p.out)) + ((long long)b * N + q) * H * HD + (long long)h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dm_store4(orow + dt * 16 + 4 * g, o[dt]);
  }
}

// =============================================================================================
// backward, dQ + delta + bias-gradient histogram
// =============================================================================================
template <typename T, int NKT>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnParams p) {
  using L = AttnLayout<T>;
  constexpr int NK = NKT * 16;
  constexpr int KV_BYTES = NK * (L::RB > L::MROW ? L::RB : L::MROW);
  constexpr int PS = NK * (int)sizeof(T) + 16;
  constexpr int NKBP = NK * (int)sizeof(T) / 64;
  __shared__ __attribute__((aligned(16))) char smem[KV_BYTES + 4 * 16 * PS];
  __shared__ float bins[MAX_BINS];
  char *kv = smem;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, li = lane & 15;
  const int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int N = p.N, H = p.H;
  char *pw = smem + KV_BYTES + wave * 16 * PS;

  const long long tok_stride = 3LL * H * HD;
  const T *base = reinterpret_cast<const T *>(p.qkv) + (long long)b * N * tok_stride + (long long)h * HD;
  const T *Kg = base + (long long)H * HD, *Vg = base + 2LL * H * HD;
  const int q = qblk * QB + wave * 16 + li;
  const bool qok = q < N;
  const long long orow = ((long long)b * N + q) * H * HD + (long long)h * HD;
  const T *Og = reinterpret_cast<const T *>(p.out) + orow;
  const T *dOg = reinterpret_cast<const T *>(p.dout) + orow;

  if (p.index)
    for (int i = t; i < p.n_bins; i += 256) bins[i] = 0.f;
  tile_to_lds_k<T>(kv, Kg, tok_stride, N, NK, t);
  u32x4 fq[L::KBD], fdo[L::KBD];
#pragma unroll
  for (int kb = 0; kb < L::KBD; ++kb) {
    fq[kb] = gl_frag<T>(base + (long long)q * tok_stride, qok, kb, lane);
    fdo[kb] = gl_frag<T>(dOg, qok, kb, lane);
  }
  // delta[q] = sum_d dO[q][d] * O[q][d]  (lane group g covers d = 16g..16g+15)
  float dl = 0.f;
  if (qok) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 a = dm_load4(Og + 16 * g + 4 * c), d = dm_load4(dOg + 16 * g + 4 * c);
      dl += (a[0] * d[0] + a[1] * d[1]) + (a[2] * d[2] + a[3] * d[3]);
    }
  }
  dl = row_sum(dl);
  const long long rowid = ((long long)b * H + h) * N + q;
  if (g == 0 && qok) p.delta[rowid] = dl;
  const float lse = qok ? p.lse[rowid] : 0.f;
  __syncthreads();

  f32x4 s[NKT];
  scores<T, NKT, false>(s, fq, kv, p.bias ? p.bias + (long long)h * N * N : nullptr, N, q, p.scale, lane);
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[kt][r] = expf(s[kt][r] - lse);   // masked entries: exp(-inf) = 0

  __syncthreads();
  tile_to_lds_k<T>(kv, Vg, tok_stride, N, NK, t);
  __syncthreads();

  // dS = P * (dP - delta), dP = dO V^T
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < L::KBD; ++kb) mma<T>(a, fdo[kb], lds_frag_k<T>(kv, kt * 16 + li, kb, lane));
#pragma unroll
    for (int r = 0; r < 4; ++r) s[kt][r] = s[kt][r] * (a[r] - dl);
  }
  if (p.index && qok) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        if (key < N) {
          const int bin = p.index[(long long)q * N + key];
          if (bin >= 0 && bin < p.n_bins) atomicAdd(&bins[bin], s[kt][r]);
        }
      }
  }
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) store_row4<T>(pw + li * PS + (kt * 16 + 4 * g) * (int)sizeof(T), s[kt]);

  __syncthreads();
  tile_to_lds_m<T>(kv, Kg, tok_stride, N, NK, t);
  __syncthreads();

  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < NKBP; ++kb) {
    const u32x4 fp = *reinterpret_cast<const u32x4 *>(pw + li * PS + (kb * 4 + g) * 16);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) mma<T>(o[dt], fp, lds_frag_m<T>(kv, dt * 16, kb, lane));
  }
  if (qok) {
    T *dq = reinterpret_cast<T *>(p.dqkv) + ((long long)b * N + q) * tok_stride + (long long)h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dm_store4(dq + dt * 16 + 4 * g, o[dt] * p.scale);
  }
  if (p.index) {
    // bins are complete once every wave passed the barrier above (all atomics precede it)
    float *srow = p.slab + (((long long)b * H + h) * p.nblk + qblk) * p.n_bins;
    for (int i = t; i < p.n_bins; i += 256) srow[i] = bins[i];
  }
}

// =============================================================================================
// backward, dK + dV
// =============================================================================================
template <typename T, int NKT>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnParams p) {
  using L = AttnLayout<T>;
  constexpr int NK = NKT * 16;   // padded number of QUERIES here (columns)
  constexpr int KV_BYTES = NK * (L::RB > L::MROW ? L::RB : L::MROW);
  constexpr int PS = NK * (int)sizeof(T) + 16;
  constexpr int NKBP = NK * (int)sizeof(T) / 64;
  __shared__ __attribute__((aligned(16))) char smem[KV_BYTES + 4 * 16 * PS];
  char *qd = smem;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, li = lane & 15;
  const int kblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int N = p.N, H = p.H;
  char *pw = smem + KV_BYTES + wave * 16 * PS;

  const long long tok_stride = 3LL * H * HD;
  const T *base = reinterpret_cast<const T *>(p.qkv) + (long long)b * N * tok_stride + (long long)h * HD;
  const T *Qg = base;
  const long long o_stride = (long long)H * HD;
  const T *dOg = reinterpret_cast<const T *>(p.dout) + (long long)b * N * o_stride + (long long)h * HD;
  const int key = kblk * QB + wave * 16 + li;
  const bool kok = key < N;
  const T *krow = base + (long long)key * tok_stride + (long long)H * HD;
  const T *vrow = krow + (long long)H * HD;
  const float *lse = p.lse + ((long long)b * H + h) * N;
  const float *delta = p.delta + ((long long)b * H + h) * N;

  tile_to_lds_k<T>(qd, Qg, tok_stride, N, NK, t);
  u32x4 fk[L::KBD], fvv[L::KBD];
#pragma unroll
  for (int kb = 0; kb < L::KBD; ++kb) {
    fk[kb] = gl_frag<T>(krow, kok, kb, lane);
    fvv[kb] = gl_frag<T>(vrow, kok, kb, lane);
  }
  __syncthreads();

  // P^T[key][q] = exp(scale * K Q^T + bias[q][key] - lse[q])
  f32x4 s[NKT];
  scores<T, NKT, true>(s, fk, qd, p.bias ? p.bias + (long long)h * N * N : nullptr, N, key, p.scale, lane);
#pragma unroll
  for (int qt = 0; qt < NKT; ++qt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = qt * 16 + 4 * g + r;
      s[qt][r] = (qq < N) ? expf(s[qt][r] - lse[qq]) : 0.f;
    }
  // stash P^T (operand dtype) for dV = P^T dO
#pragma unroll
  for (int qt = 0; qt < NKT; ++qt) store_row4<T>(pw + li * PS + (qt * 16 + 4 * g) * (int)sizeof(T), s[qt]);

  __syncthreads();
  tile_to_lds_k<T>(qd, dOg, o_stride, N, NK, t);
  __syncthreads();
  // dS^T = P^T * (dP^T - delta[q]),  dP^T[key][q] = V dO^T
#pragma unroll
  for (int qt = 0; qt < NKT; ++qt) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < L::KBD; ++kb) mma<T>(a, fvv[kb], lds_frag_k<T>(qd, qt * 16 + li, kb, lane));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = qt * 16 + 4 * g + r;
      const float dlt = (qq < N) ? delta[qq] : 0.f;
      s[qt][r] = s[qt][r] * (a[r] - dlt);
    }
  }
  __syncthreads();
  tile_to_lds_m<T>(qd, dOg, o_stride, N, NK, t);
  __syncthreads();
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < NKBP; ++kb) {
    const u32x4 fp = *reinterpret_cast<const u32x4 *>(pw + li * PS + (kb * 4 + g) * 16);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) mma<T>(o[dt], fp, lds_frag_m<T>(qd, dt * 16, kb, lane));
  }
  T *dk = reinterpret_cast<T *>(p.dqkv) + ((long long)b * N + key) * tok_stride + (long long)H * HD + (long long)h * HD;
  T *dv = dk + (long long)H * HD;
  if (kok) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dm_store4(dv + dt * 16 + 4 * g, o[dt]);
  }
  // dK = scale * dS^T Q  (P region is wave-private: its reads above are complete for this wave)
#pragma unroll
  for (int qt = 0; qt < NKT; ++qt) store_row4<T>(pw + li * PS + (qt * 16 + 4 * g) * (int)sizeof(T), s[qt]);
  __syncthreads();
  tile_to_lds_m<T>(qd, Qg, tok_stride, N, NK, t);
  __syncthreads();
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < NKBP; ++kb) {
    const u32x4 fp = *reinterpret_cast<const u32x4 *>(pw + li * PS + (kb * 4 + g) * 16);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) mma<T>(o[dt], fp, lds_frag_m<T>(qd, dt * 16, kb, lane));
  }
  if (kok) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dm_store4(dk + dt * 16 + 4 * g, o[dt] * p.scale);
  }
}

// ---- dispatch -----------------------------------------------------------------------------------
template <typename T, int NKT> void launch3(int which, const AttnParams &p, dim3 grid, hipStream_t s) {
  if (which == 0) hipLaunchKernelGGL((attn_fwd_kernel<T, NKT>), grid, dim3(256), 0, s, p);
  else if (which == 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, NKT>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, NKT>), grid, dim3(256), 0, s, p);
}
template <typename T> int dispatch(int which, const AttnParams &p, hipStream_t s) {
  const dim3 grid((p.N + QB - 1) / QB, p.H, p.B);
  const int nkt = (p.N + 15) / 16;
  if (nkt <= 2) launch3<T, 2>(which, p, grid, s);
  else if (nkt <= 4) launch3<T, 4>(which, p, grid, s);
  else if (nkt <= 8) launch3<T, 8>(which, p, grid, s);
  else if (nkt <= 12) launch3<T, 12>(which, p, grid, s);
  else if (nkt <= 14) launch3<T, 14>(which, p, grid, s);
  else launch3<T, 16>(which, p, grid, s);
  return 0;
}

int check_common(const char *who, int B, int N, int H, int D, int dtype) {
  DM_REQUIRE(B > 0 && H > 0 && N > 0 && N <= 256, DM_ERR_BAD_SHAPE, "%s: need 0 < N <= 256 tokens (got N=%d, B=%d, H=%d)", who, N, B, H);
  DM_REQUIRE(D == HD, DM_ERR_BAD_SHAPE, "%s: head dim must be %d (got %d)", who, HD, D);
  DM_REQUIRE(dtype == DM_F32 || dtype == DM_BF16, DM_ERR_BAD_DTYPE, "%s: bad dtype %d", who, dtype);
  DM_REQUIRE(H <= 65535 && B <= 65535, DM_ERR_BAD_SHAPE, "%s: grid too large", who);
  return DM_OK;
}

}  // namespace

extern "C" int32_t dm_attention_bwd_slab_rows(int32_t N) { return (N + QB - 1) / QB; }

extern "C" int dm_attention_fwd(const void *qkv, const float *bias, void *out, float *lse, int32_t B, int32_t N, int32_t H,
                                int32_t D, float scale, int32_t dtype, void *stream) {
  if (int rc = check_common("dm_attention_fwd", B, N, H, D, dtype)) return rc;
  DM_REQUIRE(qkv && out && lse, DM_ERR_BAD_SHAPE, "dm_attention_fwd: null pointer");
  DM_REQUIRE(dm_aligned16(qkv) && dm_aligned16(out), DM_ERR_BAD_ALIGN, "dm_attention_fwd: qkv/out must be 16-byte aligned");
  AttnParams p{};
  p.qkv = qkv; p.bias = bias; p.out = out; p.lse = lse; p.B = B; p.N = N; p.H = H; p.scale = scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  {
    const double esz = (dtype == DM_BF16) ? 2.0 : 4.0;
    DmProfScope prof(dtype == DM_BF16 ? "attn_fwd_bf16" : "attn_fwd_f32", s, 4.0 * B * H * (double)N * N * HD,
                     esz * 4.0 * B * H * (double)N * HD);
    if (dtype == DM_BF16) dispatch<bf16_t>(0, p, s); else dispatch<float>(0, p, s);
  }
  DM_LAUNCH_CHECK("dm_attention_fwd");
  return DM_OK;
}

extern "C" int dm_attention_bwd(const void *qkv, const float *bias, const void *out, const void *dout, const float *lse,
                                void *dqkv, float *delta, const int32_t *index, int32_t n_bins, float *dtable_slab,
                                int32_t B, int32_t N, int32_t H, int32_t D, float scale, int32_t dtype, void *stream) {
  if (int rc = check_common("dm_attention_bwd", B, N, H, D, dtype)) return rc;
  DM_REQUIRE(qkv && out && dout && lse && dqkv && delta, DM_ERR_BAD_SHAPE, "dm_attention_bwd: null pointer");
  DM_REQUIRE(dm_aligned16(qkv) && dm_aligned16(out) && dm_aligned16(dout) && dm_aligned16(dqkv), DM_ERR_BAD_ALIGN,
             "dm_attention_bwd: tensors must be 16-byte aligned");
  DM_REQUIRE(index == nullptr || (dtable_slab != nullptr && n_bins > 0 && n_bins <= MAX_BINS), DM_ERR_BAD_SHAPE,
             "dm_attention_bwd: bias gradient needs a slab and 0 < n_bins <= %d (got %d)", MAX_BINS, n_bins);
  AttnParams p{};
  p.qkv = qkv; p.bias = bias; p.out = out; p.dout = dout; p.lse = const_cast<float *>(lse); p.delta = delta; p.dqkv = dqkv;
  p.index = index; p.slab = dtable_slab; p.n_bins = n_bins; p.nblk = (N + QB - 1) / QB;
  p.B = B; p.N = N; p.H = H; p.scale = scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  {
    const double esz = (dtype == DM_BF16) ? 2.0 : 4.0;
    DmProfScope prof(dtype == DM_BF16 ? "attn_bwd_bf16" : "attn_bwd_f32", s, 10.0 * B * H * (double)N * N * HD,
                     esz * 8.0 * B * H * (double)N * HD);
    if (dtype == DM_BF16) { dispatch<bf16_t>(1, p, s); dispatch<bf16_t>(2, p, s); }
    else { dispatch<float>(1, p, s); dispatch<float>(2, p, s); }
  }
  DM_LAUNCH_CHECK("dm_attention_bwd");
  return DM_OK;
}
