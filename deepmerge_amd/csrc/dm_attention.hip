// Fused attention with dense relative-position bias for the DeepMerge token cubes (gfx950).
//
// Problem shape: per (sample b, head h) one N x N attention with N <= 256 tokens and head dim
// D = 64 (N = 12..256 for ShfitScaleFormer stages, 197/198 for the ViT variants).  The whole
// score row fits in registers, so softmax is exact (no online rescaling) and nothing of size
// N x N ever goes to HBM or LDS.
//
// Workgroup = (64-row block, head, sample), 4 waves x 16 rows; one barrier per kernel.
//   forward : K and V tiles -> LDS; S = scale*Q K^T + bias (MFMA, K read by rows) -> softmax in
//             registers -> O = P V (MFMA, V read through the hardware transpose).
//   backward: two kernels so that no gradient is summed across workgroups:
//             dq  (rows = queries): P = exp(S - lse), dP = dO V^T, dS = P*(dP - delta),
//                  dQ = scale * dS K; also delta = rowsum(dO*O) and the bias-table gradient
//                  summed over the chunk's samples in registers -> one dense [N,N] slab per (chunk, head);
//             dkv (rows = keys):    P^T, dP^T, dS^T likewise, dV = P^T dO, dK = scale * dS^T Q.
//
// Two ideas carry the kernel:
//  (1) ONE LDS image per tile serves both the row reads (contraction over d: QK^T, dO V^T) and the
//      transposed reads (contraction over tokens: P V, dS K, P^T dO, dS^T Q).  Image: [token][64 d],
//      16-byte chunk index XOR s(token); bf16: s = ((t1^t2)<<2 | t0<<1 | t1), t = token>>1, which is
//      bank-conflict free for ds_read_b128 row fragments AND for ds_read_b64_tr_b16 in both row
//      patterns used here (checked by brute force against the gfx950 bank model); fp32: s = token&15
//      (conflict free for ds_read_b128 and the ds_read_b32 column reads).
//  (2) P / dS never leave registers: with the MFMA operands swapped a lane owns 4 consecutive
//      tokens of one row per 16x16 tile, and two adjacent tiles (bf16) or one tile (fp32) ARE a valid
//      operand fragment of the next MFMA for a permuted contraction order; the other operand is
//      fetched in the same permuted order (rows 4g+16*half+q of the transposed read).
#include "dm_common.h"
#include "dm_mfma.h"
#include "dm_attention_pipe.h"
#include "dm_attention_x3.h"
#include "dm_prof.h"

namespace {

constexpr int HD = 64;         // head dim
constexpr int QB = 64;         // rows (queries or keys) per workgroup

template <typename T> struct AL {
  static constexpr int RB = HD * (int)sizeof(T);       // bytes per 64-d row: 128 / 256
  static constexpr int CPRW = RB / 16;                 // 16-byte chunks per row: 8 / 16
  static constexpr int KBD = RB / 64;                  // 64-byte contraction blocks across d: 2 / 4
  static constexpr int EPC = 16 / (int)sizeof(T);      // elements per chunk: 8 / 4
  static constexpr int TPB = (sizeof(T) == 2) ? 2 : 1; // 16-token tiles per contraction block over tokens
};

template <typename T> __device__ __forceinline__ int swz(int row);
template <> __device__ __forceinline__ int swz<bf16_t>(int row) {
  const int t0 = (row >> 1) & 1, t1 = (row >> 2) & 1, t2 = (row >> 3) & 1;
  return ((t1 ^ t2) << 2) | (t0 << 1) | t1;
}
template <> __device__ __forceinline__ int swz<float>(int row) { return row & 15; }

template <typename T> __device__ __forceinline__ int img_off(int row, int chunk) {
  return row * AL<T>::RB + ((chunk ^ swz<T>(row)) << 4);
}

// global [rows][64] (row stride `ld` elements) -> LDS image; rows >= nvalid are zero-filled.
template <typename T>
__device__ __forceinline__ void tile_to_lds(char *lds, const T *g, long long ld, int nvalid, int nrows, int t) {
  using L = AL<T>;
  const int c = t % L::CPRW;
  constexpr int RPI = 256 / L::CPRW;
  for (int row = t / L::CPRW; row < nrows; row += RPI) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < nvalid) v = *reinterpret_cast<const u32x4 *>(g + (long long)row * ld + c * L::EPC);
    *reinterpret_cast<u32x4 *>(lds + img_off<T>(row, c)) = v;
  }
}

// row fragment: token `row`, 64-byte block kb across d.
template <typename T> __device__ __forceinline__ u32x4 frag_row(const char *lds, int row, int kb, int lane) {
  return *reinterpret_cast<const u32x4 *>(lds + img_off<T>(row, kb * 4 + (lane >> 4)));
}
// transposed fragment: columns d0..d0+15, contraction block m over tokens, in the order the
// register-resident P/dS fragments use: bf16 slot j of lane group g = token 32m + 4g + 16(j>>2) + (j&3);
// fp32 slot j = token 16m + 4g + j.
template <typename T> __device__ __forceinline__ u32x4 frag_tr(const char *lds, int d0, int m, int lane);
template <> __device__ __forceinline__ u32x4 frag_tr<bf16_t>(const char *lds, int d0, int m, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int d = d0 + 4 * p;
  u32x4 out;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int k = 32 * m + 4 * g + 16 * half + q;
    const u32x2 w = dm_ds_read_tr16(lds + img_off<bf16_t>(k, d >> 3) + ((d & 7) << 1));
    out[2 * half] = w[0];
    out[2 * half + 1] = w[1];
  }
  return out;
}
template <> __device__ __forceinline__ u32x4 frag_tr<float>(const char *lds, int d0, int m, int lane) {
  const int g = lane >> 4, d = d0 + (lane & 15);
  u32x4 out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 16 * m + 4 * g + j;
    out[j] = *reinterpret_cast<const unsigned int *>(lds + img_off<float>(k, d >> 2) + ((d & 3) << 2));
  }
  return out;
}
// fragment straight from global: row pointer `rowp` (64 contiguous d), block kb; zero if !valid
template <typename T> __device__ __forceinline__ u32x4 gl_frag(const T *rowp, bool valid, int kb, int lane) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (valid) v = *reinterpret_cast<const u32x4 *>(rowp + (kb * 4 + (lane >> 4)) * AL<T>::EPC);
  return v;
}

// accumulator tiles -> operand fragment of contraction block m
template <typename T, int NKT> __device__ __forceinline__ u32x4 acc_frag(const f32x4 (&s)[NKT], int m) {
  if constexpr (sizeof(T) == 2) {
    const f32x4 lo = s[2 * m], hi = s[2 * m + 1];
    bf16x8 r = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
    return __builtin_bit_cast(u32x4, r);
  } else {
    return __builtin_bit_cast(u32x4, s[m]);
  }
}

// out[dt] += sum over tokens of s[row][token] * img[token][d]   (P V, dS K, P^T dO, dS^T Q)
template <typename T, int NKT>
__device__ __forceinline__ void contract_tokens(f32x4 (&o)[4], const f32x4 (&s)[NKT], const char *img, int lane) {
#pragma unroll
  for (int m = 0; m < NKT / AL<T>::TPB; ++m) {
    const u32x4 fp = acc_frag<T, NKT>(s, m);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) mma<T>(o[dt], fp, frag_tr<T>(img, dt * 16, m, lane));
    if ((m & 1) == 1) __builtin_amdgcn_sched_barrier(0);
  }
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

template <typename T> __device__ __forceinline__ float fast_exp(float x);
template <> __device__ __forceinline__ float fast_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ float fast_exp<bf16_t>(float x) { return __expf(x); }

// 4 consecutive floats p[i..i+3] with i % 4 == 0; vector load when the row length allows it.
__device__ __forceinline__ f32x4 load4_guard(const float *p, int i, int n, bool vec) {
  if (vec) return (i < n) ? dm_load4(p + i) : (f32x4){0.f, 0.f, 0.f, 0.f};  // `vec` is a compile-time constant at every call site
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = (i + e < n) ? p[i + e] : 0.f;
  return r;
}

struct AttnParams {
  const void *qkv;
  const float *bias;    // [H,N,N]  bias[h][q][key]
  const float *bias_t;  // [H,N,N]  bias_t[h][key][q] or NULL
  const void *out;      // forward: written; backward: read
  const void *dout;
  float *lse;
  float *delta;
  void *dqkv;
  float *slab;
  int B, N, H, bchunk;
  float scale;
};

// Bias values of one 16x16 score tile for this lane: bias_rows[row][16kt + 4g .. +3]; when STRIDED the
// array is indexed [col*N + row] instead (transposed use without a transposed copy).  Zero when absent.
template <bool FAST, bool STRIDED>
__device__ __forceinline__ f32x4 bias_tile(const float *bias_rows, int N, int row, int lane, int kt) {
  const int col = kt * 16 + 4 * (lane >> 4);
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias_rows && row < N) {
    if constexpr (!STRIDED) {
      bv = load4_guard(bias_rows + (long long)row * N, col, N, FAST);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = (col + r < N) ? bias_rows[(long long)(col + r) * N + row] : 0.f;
    }
  }
  return bv;
}
// One 16x16 score tile: a[r] = scale * (this wave's row) . (tile row 16kt+4g+r) + bv[r];
// entries with row >= N or col >= N become -inf.
template <typename T>
__device__ __forceinline__ f32x4 score_tile(const u32x4 (&fa)[AL<T>::KBD], const char *img, const f32x4 bv,
                                            int N, int row, float scale, int lane, int kt) {
  using L = AL<T>;
  const int g = lane >> 4, li = lane & 15;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < L::KBD; ++kb) mma<T>(a, fa[kb], frag_row<T>(img, kt * 16 + li, kb, lane));
  const int col = kt * 16 + 4 * g;
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] = (row < N && col + r < N) ? a[r] * scale + bv[r] : -INFINITY;
  return a;
}

// pack TPB accumulator tiles (tokens 16*TPB*m ...) into the operand fragment of contraction block m
template <typename T> __device__ __forceinline__ u32x4 pack_tiles(const f32x4 (&tl)[AL<T>::TPB]) {
  if constexpr (sizeof(T) == 2) {
    bf16x8 r = {(bf16_t)tl[0][0], (bf16_t)tl[0][1], (bf16_t)tl[0][2], (bf16_t)tl[0][3],
                (bf16_t)tl[1][0], (bf16_t)tl[1][1], (bf16_t)tl[1][2], (bf16_t)tl[1][3]};
    return __builtin_bit_cast(u32x4, r);
  } else {
    return __builtin_bit_cast(u32x4, tl[0]);
  }
}

// out[dt] += sum over tokens of frag[row][token] * img[token][d], fragments already packed
template <typename T, int NB>
__device__ __forceinline__ void contract_frags(f32x4 (&o)[4], const u32x4 (&f)[NB], const char *img, int lane) {
#pragma unroll
  for (int m = 0; m < NB; ++m) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) mma<T>(o[dt], f[m], frag_tr<T>(img, dt * 16, m, lane));
    if ((m & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // bound how many transposed reads are in flight (VGPRs)
  }
}

// =============================================================================================
// forward
// =============================================================================================
template <typename T, int NKT, bool FAST>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnParams p) {
  using L = AL<T>;
  constexpr int NK = NKT * 16;
  constexpr int IMG = NK * L::RB;
  __shared__ __attribute__((aligned(16))) char smem[2 * IMG];
  char *kimg = smem, *vimg = smem + IMG;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, li = lane & 15;
  const int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int N = p.N, H = p.H;

  const long long tok_stride = 3LL * H * HD;
  const T *base = reinterpret_cast<const T *>(p.qkv) + (long long)b * N * tok_stride + (long long)h * HD;
  const int q = qblk * QB + wave * 16 + li;

  tile_to_lds<T>(kimg, base + (long long)H * HD, tok_stride, N, NK, t);
  tile_to_lds<T>(vimg, base + 2LL * H * HD, tok_stride, N, NK, t);
  u32x4 fq[L::KBD];
#pragma unroll
  for (int kb = 0; kb < L::KBD; ++kb) fq[kb] = gl_frag<T>(base + (long long)q * tok_stride, q < N, kb, lane);
  __syncthreads();

  f32x4 s[NKT];
  const float *brows = p.bias ? p.bias + (long long)h * N * N : nullptr;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) s[kt] = score_tile<T>(fq, kimg, bias_tile<FAST, false>(brows, N, q, lane, kt), N, q, p.scale, lane, kt);

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
      const float e = fast_exp<T>(s[kt][r] - msafe);
      s[kt][r] = e;
      l += e;
    }
  l = row_sum(l);
  const float inv = (l > 0.f) ? 1.f / l : 0.f;
  if (g == 0 && q < N) p.lse[((long long)b * H + h) * N + q] = msafe + logf(l);
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) s[kt] *= inv;

  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  contract_tokens<T, NKT>(o, s, vimg, lane);
  if (q < N) {
    T *orow = reinterpret_cast<T *>(const_cast<void *>(p.out)) + ((long long)b * N + q) * H * HD + (long long)h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dm_store4(orow + dt * 16 + 4 * g, o[dt]);
  }
}

// =============================================================================================
// backward, dQ + delta + the chunk's summed dS (dense bias gradient)
// =============================================================================================
template <typename T, int NKT, bool FAST>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnParams p) {
  // grid = (query block, head, batch chunk).  The workgroup walks the `bchunk` samples of its chunk and
  // sums dS over them in registers; the sum is written once per chunk as a dense [N][N] plane
  // (dm_relpos_bias_reduce folds it into the table's gradient: no atomics, deterministic).
  using L = AL<T>;
  constexpr int NK = NKT * 16;
  constexpr int IMG = NK * L::RB;
  constexpr int NB = NKT / L::TPB;
  __shared__ __attribute__((aligned(16))) char smem[2 * IMG];
  char *kimg = smem, *vimg = smem + IMG;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, li = lane & 15;
  const int qblk = blockIdx.x, h = blockIdx.y, chunk = blockIdx.z;
  const int N = p.N, H = p.H;
  const long long tok_stride = 3LL * H * HD;
  const int q = qblk * QB + wave * 16 + li;
  const bool qok = q < N;
  const float *brows = p.bias ? p.bias + (long long)h * N * N : nullptr;
  const bool want_bins = p.slab != nullptr;
  f32x4 hacc[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) hacc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int b_end = min(p.B, (chunk + 1) * p.bchunk);
  for (int b = chunk * p.bchunk; b < b_end; ++b) {
    const T *base = reinterpret_cast<const T *>(p.qkv) + (long long)b * N * tok_stride + (long long)h * HD;
    const long long orow = ((long long)b * N + q) * H * HD + (long long)h * HD;
    const T *Og = reinterpret_cast<const T *>(p.out) + orow;
    const T *dOg = reinterpret_cast<const T *>(p.dout) + orow;
    __syncthreads();                  // previous sample's reads of the images are complete
    tile_to_lds<T>(kimg, base + (long long)H * HD, tok_stride, N, NK, t);
    tile_to_lds<T>(vimg, base + 2LL * H * HD, tok_stride, N, NK, t);
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

    u32x4 fds[NB];
    // the bias tile of block m+1 is fetched while block m computes (the loop is pinned block by block to
    // bound registers, so the prefetch has to be explicit)
    f32x4 bnext[L::TPB];
#pragma unroll
    for (int u = 0; u < L::TPB; ++u) bnext[u] = bias_tile<FAST, false>(brows, N, q, lane, u);
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      f32x4 bcur[L::TPB];
#pragma unroll
      for (int u = 0; u < L::TPB; ++u) bcur[u] = bnext[u];
      if (m + 1 < NB) {
#pragma unroll
        for (int u = 0; u < L::TPB; ++u) bnext[u] = bias_tile<FAST, false>(brows, N, q, lane, (m + 1) * L::TPB + u);
      }
      f32x4 tl[L::TPB];
#pragma unroll
      for (int u = 0; u < L::TPB; ++u) {
        const int kt = m * L::TPB + u;
        const f32x4 sc = score_tile<T>(fq, kimg, bcur[u], N, q, p.scale, lane, kt);
        // dP = dO V^T for this tile, then dS = P * (dP - delta)
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < L::KBD; ++kb) mma<T>(a, fdo[kb], frag_row<T>(vimg, kt * 16 + li, kb, lane));
        f32x4 dsv;
#pragma unroll
        for (int r = 0; r < 4; ++r) dsv[r] = fast_exp<T>(sc[r] - lse) * (a[r] - dl);   // masked: exp(-inf) = 0
        if (want_bins) hacc[kt] += dsv;
        tl[u] = dsv;
      }
      fds[m] = pack_tiles<T>(tl);
      // Pin the packed fragment HERE: otherwise the optimiser sinks exp/pack of every block down to the
      // contraction below and keeps all raw score tiles + loaded bias alive (>256 VGPRs, spills).
      asm volatile("" : "+v"(fds[m]));
      __builtin_amdgcn_sched_barrier(0);
    }

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    contract_frags<T, NB>(o, fds, kimg, lane);
    if (qok) {
      T *dq = reinterpret_cast<T *>(p.dqkv) + ((long long)b * N + q) * tok_stride + (long long)h * HD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dm_store4(dq + dt * 16 + 4 * g, o[dt] * p.scale);
    }
  }

  if (want_bins && qok) {
    // dense d(bias)[chunk][h][q][key] = the chunk's summed dS; dm_relpos_bias_reduce folds it into the table's
    // gradient in a fixed order (no atomics anywhere: the bias-table gradient is run-to-run deterministic)
    float *srow = p.slab + (((long long)chunk * H + h) * N + q) * N;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int key = kt * 16 + 4 * g;
      if constexpr (FAST) {
        if (key < N) dm_store4(srow + key, hacc[kt]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (key + r < N) srow[key + r] = hacc[kt][r];
      }
    }
  }
}

// =============================================================================================
// backward, dK + dV  (rows of the workgroup are KEYS; columns are queries)
// =============================================================================================
template <typename T, int NKT, bool FAST>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnParams p) {
  using L = AL<T>;
  constexpr int NK = NKT * 16;
  constexpr int IMG = NK * L::RB;
  __shared__ __attribute__((aligned(16))) char smem[2 * IMG];
  char *qimg = smem, *doimg = smem + IMG;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, li = lane & 15;
  const int kblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int N = p.N, H = p.H;

  const long long tok_stride = 3LL * H * HD;
  const T *base = reinterpret_cast<const T *>(p.qkv) + (long long)b * N * tok_stride + (long long)h * HD;
  const long long o_stride = (long long)H * HD;
  const T *dOg = reinterpret_cast<const T *>(p.dout) + (long long)b * N * o_stride + (long long)h * HD;
  const int key = kblk * QB + wave * 16 + li;
  const bool kok = key < N;
  const T *krow = base + (long long)key * tok_stride + (long long)H * HD;
  const T *vrow = krow + (long long)H * HD;
  const float *lse = p.lse + ((long long)b * H + h) * N;
  const float *delta = p.delta + ((long long)b * H + h) * N;

  tile_to_lds<T>(qimg, base, tok_stride, N, NK, t);
  tile_to_lds<T>(doimg, dOg, o_stride, N, NK, t);
  u32x4 fk[L::KBD], fvv[L::KBD];
#pragma unroll
  for (int kb = 0; kb < L::KBD; ++kb) {
    fk[kb] = gl_frag<T>(krow, kok, kb, lane);
    fvv[kb] = gl_frag<T>(vrow, kok, kb, lane);
  }
  __syncthreads();

  // P^T[key][q] = exp(scale * K Q^T + bias[q][key] - lse[q]);   dS^T = P^T * (V dO^T - delta[q])
  constexpr int NB = NKT / L::TPB;
  u32x4 fpt[NB], fds[NB];
  // FAST: N % 4 == 0 and a transposed bias copy is available -> vector loads everywhere;
  // otherwise the generic form reads `bias` with a stride and lse/delta element-wise.
  const float *bsrc = FAST ? p.bias_t : p.bias;
  const float *brows = bsrc ? bsrc + (long long)h * N * N : nullptr;
  f32x4 bnext[L::TPB], lnext[L::TPB], dnext[L::TPB];
  auto fetch = [&](int m) {
#pragma unroll
    for (int u = 0; u < L::TPB; ++u) {
      const int qt = m * L::TPB + u;
      bnext[u] = bias_tile<FAST, !FAST>(brows, N, key, lane, qt);
      lnext[u] = load4_guard(lse, qt * 16 + 4 * g, N, FAST);
      dnext[u] = load4_guard(delta, qt * 16 + 4 * g, N, FAST);
    }
  };
  fetch(0);
#pragma unroll
  for (int m = 0; m < NB; ++m) {
    f32x4 bcur[L::TPB], lcur[L::TPB], dcur[L::TPB];
#pragma unroll
    for (int u = 0; u < L::TPB; ++u) { bcur[u] = bnext[u]; lcur[u] = lnext[u]; dcur[u] = dnext[u]; }
    if (m + 1 < NB) fetch(m + 1);
    f32x4 tp[L::TPB], td[L::TPB];
#pragma unroll
    for (int u = 0; u < L::TPB; ++u) {
      const int qt = m * L::TPB + u;
      const int q0 = qt * 16 + 4 * g;
      const f32x4 sc = score_tile<T>(fk, qimg, bcur[u], N, key, p.scale, lane, qt);
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < L::KBD; ++kb) mma<T>(a, fvv[kb], frag_row<T>(doimg, qt * 16 + li, kb, lane));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = (q0 + r < N) ? fast_exp<T>(sc[r] - lcur[u][r]) : 0.f;
        tp[u][r] = pv;
        td[u][r] = pv * (a[r] - dcur[u][r]);
      }
    }
    fpt[m] = pack_tiles<T>(tp);
    fds[m] = pack_tiles<T>(td);
    // Pin the packed fragments HERE: otherwise the optimiser sinks exp/pack of every block down to the
    // contractions below and keeps all raw score tiles + loaded bias/lse/delta alive (>256 VGPRs, spills).
    asm volatile("" : "+v"(fpt[m]), "+v"(fds[m]));
    __builtin_amdgcn_sched_barrier(0);
  }
  f32x4 o[4];
  T *dk = reinterpret_cast<T *>(p.dqkv) + ((long long)b * N + key) * tok_stride + (long long)H * HD + (long long)h * HD;
  T *dvp = dk + (long long)H * HD;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  contract_frags<T, NB>(o, fpt, doimg, lane);            // dV = P^T dO
  if (kok) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dm_store4(dvp + dt * 16 + 4 * g, o[dt]);
  }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  contract_frags<T, NB>(o, fds, qimg, lane);             // dK = scale * dS^T Q
  if (kok) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dm_store4(dk + dt * 16 + 4 * g, o[dt] * p.scale);
  }
}

// ---- dispatch -----------------------------------------------------------------------------------
template <typename T, int NKT, bool FAST> void launch3(int which, const AttnParams &p, dim3 grid, hipStream_t s) {
  if (which == 0) hipLaunchKernelGGL((attn_fwd_kernel<T, NKT, FAST>), grid, dim3(256), 0, s, p);
  else if (which == 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, NKT, FAST>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, NKT, FAST>), grid, dim3(256), 0, s, p);
}
template <typename T, bool FAST> void dispatch_nkt(int which, const AttnParams &p, hipStream_t s) {
  const dim3 grid((p.N + QB - 1) / QB, p.H, which == 1 ? (p.B + p.bchunk - 1) / p.bchunk : p.B);
  const int nkt = (p.N + 15) / 16;
  if (nkt <= 2) launch3<T, 2, FAST>(which, p, grid, s);
  else if (nkt <= 4) launch3<T, 4, FAST>(which, p, grid, s);
  else if (nkt <= 8) launch3<T, 8, FAST>(which, p, grid, s);
  else if (nkt <= 12) launch3<T, 12, FAST>(which, p, grid, s);
  else if (nkt <= 14) launch3<T, 14, FAST>(which, p, grid, s);
  else launch3<T, 16, FAST>(which, p, grid, s);
}
template <typename T> int dispatch(int which, const AttnParams &p, hipStream_t s) {
  // fast form: rows are a multiple of 4 tokens (vector loads of bias / lse / delta / index) and, for the
  // key-major kernel, either no bias or a transposed bias copy.
  const bool fast = (p.N % 4 == 0) && (which != 2 || p.bias == nullptr || p.bias_t != nullptr);
  if (fast) dispatch_nkt<T, true>(which, p, s);
  else dispatch_nkt<T, false>(which, p, s);
  return 0;
}

int check_common(const char *who, int B, int N, int H, int D, int dtype) {
  DM_REQUIRE(B > 0 && H > 0 && N > 0 && N <= 256, DM_ERR_BAD_SHAPE, "%s: need 0 < N <= 4096 tokens and head dim <= 256 (got N=%d, D=%d, B=%d, H=%d)", who, N, D, B, H);
  DM_REQUIRE(D == HD, DM_ERR_BAD_SHAPE, "%s: head dim must be in 1..256 (got %d)", who, D);
  DM_REQUIRE(dtype == DM_F32 || dtype == DM_BF16, DM_ERR_BAD_DTYPE, "%s: bad dtype %d", who, dtype);
  DM_REQUIRE(H <= 65535 && B <= 65535, DM_ERR_BAD_SHAPE, "%s: grid too large", who);
  return DM_OK;
}

}  // namespace

// samples one dq workgroup walks: as many as keeps >= ~1.2 workgroups per CU in flight
static int batch_chunk(int B, int N, int H) {
  const int per_sample = ((N + QB - 1) / QB) * H;
  for (int c = 8; c > 1; c >>= 1)
    if ((long long)per_sample * ((B + c - 1) / c) >= 300) return c;
  return 1;
}

extern "C" int32_t dm_attention_bwd_batch_chunks(int32_t B, int32_t N, int32_t H, int32_t dtype) {
  if (const int pc = dm_attn_bwd_pipe_chunks(B, N, H, dtype == DM_BF16)) return pc;     // pipelined kernels (dm_attention_pipe.hip)
  const int c = batch_chunk(B, N, H);
  return (B + c - 1) / c;
}

// dm_attention_generic.hip: any head dim <= 256 / up to 4096 tokens, fp32 arithmetic (ViT-H/14: D = 80, N = 257)
bool dm_attn_generic_shape(int N, int D);
int dm_attn_generic_fwd(const void *qkv, const float *bias, void *out, float *lse, int B, int N, int H, int D, float scale, int dtype, hipStream_t s);
int dm_attn_generic_bwd(const void *qkv, const float *bias, const void *out, const void *dout, const float *lse, void *dqkv, float *delta,
                        int B, int N, int H, int D, float scale, int dtype, hipStream_t s);

extern "C" int dm_attention_fwd(const void *qkv, const float *bias, void *out, float *lse, int32_t B, int32_t N, int32_t H,
                                int32_t D, float scale, int32_t dtype, void *stream) {
  if (dm_attn_generic_shape(N, D)) {
    DM_REQUIRE(B > 0 && H > 0 && H <= 65535 && B <= 65535 && (dtype == DM_F32 || dtype == DM_BF16) && qkv && out && lse, DM_ERR_BAD_SHAPE,
               "dm_attention_fwd: bad arguments (B=%d H=%d dtype=%d)", B, H, dtype);
    const int rc = dm_attn_generic_fwd(qkv, bias, out, lse, B, N, H, D, scale, dtype, reinterpret_cast<hipStream_t>(stream));
    DM_REQUIRE(rc == DM_OK, rc, "dm_attention_fwd: generic kernel could not be configured");
    DM_LAUNCH_CHECK("dm_attention_fwd(generic)");
    return DM_OK;
  }
  if (int rc = check_common("dm_attention_fwd", B, N, H, D, dtype)) return rc;
  DM_REQUIRE(qkv && out && lse, DM_ERR_BAD_SHAPE, "dm_attention_fwd: null pointer");
  DM_REQUIRE(dm_aligned16(qkv) && dm_aligned16(out) && dm_aligned16(bias), DM_ERR_BAD_ALIGN, "dm_attention_fwd: qkv/out/bias must be 16-byte aligned");
  AttnParams p{};
  p.qkv = qkv; p.bias = bias; p.out = out; p.lse = lse; p.B = B; p.N = N; p.H = H; p.scale = scale; p.bchunk = 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  {
    const double esz = (dtype == DM_BF16) ? 2.0 : 4.0;
    DmProfScope prof(dtype == DM_BF16 ? "attn_fwd_bf16" : "attn_fwd_f32", s, 4.0 * B * H * (double)N * N * HD,
                     esz * 4.0 * B * H * (double)N * HD);
    bool piped = false;
    if (dtype == DM_BF16) {      // persistent LDS-DMA pipeline for the big stage (dm_attention_pipe.hip)
      AttnPipeParams pp{qkv, bias, out, lse, B, N, H, scale};
      piped = dm_attn_fwd_q32(pp, s) || dm_attn_fwd_pipe(pp, s);     // 32 rows per wave (dm_attention_q32.hip), else 16 rows per wave
    }
    if (!piped) {
      if (dtype == DM_BF16) dispatch<bf16_t>(0, p, s); else dispatch<float>(0, p, s);
    }
  }
  DM_LAUNCH_CHECK("dm_attention_fwd");
  return DM_OK;
}

// Forward with the relative-position bias formed inside the kernel from the table (no dense [H, N, N] rows): the 32-rows-per-wave
// kernel with the head's table in LDS (dm_attention_q32.hip).  Token cube (cube_s, 8, 8), scale-major then row-major.
static bool relpos_inkernel(int32_t B, int32_t N, int32_t H, int32_t D, int32_t cube_s, int32_t cube_h, int32_t cube_w, int32_t dtype) {
  if (dtype != DM_BF16 || D != HD || cube_h != 8 || cube_w != 8 || B <= 0 || H <= 0) return false;
  AttnPipeParams pp{nullptr, nullptr, nullptr, nullptr, B, N, H, 1.f};
  pp.table = reinterpret_cast<const float *>(16);      // (shape decision only)
  pp.cube_s = cube_s;
  if (!dm_attn_fwd_q32_takes(pp)) return false;
  // the backward pass must take the table too: a caller told "in kernel" holds no dense rows, and the 16-row / generic
  // backward kernels would then run WITHOUT the bias (the forward and backward switches and B * H rules differ)
  AttnPipeBwdParams bp{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, N, H, 1.f};
  bp.table = pp.table;
  bp.cube_s = cube_s;
  return dm_attn_bwd_tab_takes(bp);
}

extern "C" int32_t dm_attention_relpos_inkernel(int32_t B, int32_t N, int32_t H, int32_t D, int32_t cube_s, int32_t cube_h, int32_t cube_w,
                                                int32_t dtype) {
  return relpos_inkernel(B, N, H, D, cube_s, cube_h, cube_w, dtype) ? 1 : 0;
}

extern "C" int dm_attention_fwd_relpos(const void *qkv, const float *table, int32_t cube_s, int32_t cube_h, int32_t cube_w, void *out,
                                       float *lse, int32_t B, int32_t N, int32_t H, int32_t D, float scale, int32_t dtype, void *stream) {
  DM_REQUIRE(relpos_inkernel(B, N, H, D, cube_s, cube_h, cube_w, dtype), DM_ERR_UNSUPPORTED,
             "dm_attention_fwd_relpos: shape not taken (B=%d N=%d H=%d D=%d cube=%dx%dx%d dtype=%d); gather the bias and call dm_attention_fwd",
             B, N, H, D, cube_s, cube_h, cube_w, dtype);
  DM_REQUIRE(qkv && table && out && lse, DM_ERR_BAD_SHAPE, "dm_attention_fwd_relpos: null pointer");
  DM_REQUIRE(dm_aligned16(qkv) && dm_aligned16(out), DM_ERR_BAD_ALIGN, "dm_attention_fwd_relpos: qkv/out must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  {
    DmProfScope prof("attn_fwd_bf16", s, 4.0 * B * H * (double)N * N * HD, 2.0 * 4.0 * B * H * (double)N * HD);
    AttnPipeParams pp{qkv, nullptr, out, lse, B, N, H, scale};
    pp.table = table;
    pp.cube_s = cube_s;
    DM_REQUIRE(dm_attn_fwd_q32(pp, s), DM_ERR_UNSUPPORTED, "dm_attention_fwd_relpos: kernel could not be configured");
  }
  DM_LAUNCH_CHECK("dm_attention_fwd_relpos");
  return DM_OK;
}

static int attention_bwd(const void *qkv, const float *bias, const float *bias_t, const float *table, int32_t cube_s, const void *out,
                         const void *dout, const float *lse, void *dqkv, float *delta, float *dbias_slab, int32_t B, int32_t N, int32_t H,
                         int32_t D, float scale, int32_t dtype, void *stream) {
  if (dm_attn_generic_shape(N, D)) {
    DM_REQUIRE(B > 0 && H > 0 && H <= 65535 && B <= 65535 && (dtype == DM_F32 || dtype == DM_BF16) && qkv && out && dout && lse && dqkv && delta,
               DM_ERR_BAD_SHAPE, "dm_attention_bwd: bad arguments (B=%d H=%d dtype=%d)", B, H, dtype);
    DM_REQUIRE(dbias_slab == nullptr, DM_ERR_UNSUPPORTED, "dm_attention_bwd: the bias-table gradient needs head dim 64 and N <= 256 (got D=%d N=%d)", D, N);
    const int rc = dm_attn_generic_bwd(qkv, bias, out, dout, lse, dqkv, delta, B, N, H, D, scale, dtype, reinterpret_cast<hipStream_t>(stream));
    DM_REQUIRE(rc == DM_OK, rc, "dm_attention_bwd: generic kernel could not be configured");
    DM_LAUNCH_CHECK("dm_attention_bwd(generic)");
    return DM_OK;
  }
  if (int rc = check_common("dm_attention_bwd", B, N, H, D, dtype)) return rc;
  DM_REQUIRE(qkv && out && dout && lse && dqkv && delta, DM_ERR_BAD_SHAPE, "dm_attention_bwd: null pointer");
  DM_REQUIRE(dm_aligned16(qkv) && dm_aligned16(out) && dm_aligned16(dout) && dm_aligned16(dqkv) && dm_aligned16(bias) &&
             dm_aligned16(bias_t) && dm_aligned16(lse) && dm_aligned16(delta) && dm_aligned16(dbias_slab), DM_ERR_BAD_ALIGN,
             "dm_attention_bwd: tensors must be 16-byte aligned");
  AttnParams p{};
  p.qkv = qkv; p.bias = bias; p.bias_t = bias ? bias_t : nullptr; p.out = out; p.dout = dout; p.lse = const_cast<float *>(lse);
  p.delta = delta; p.dqkv = dqkv; p.slab = dbias_slab;
  p.B = B; p.N = N; p.H = H; p.scale = scale; p.bchunk = batch_chunk(B, N, H);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  {
    const double esz = (dtype == DM_BF16) ? 2.0 : 4.0;
    DmProfScope prof(dtype == DM_BF16 ? "attn_bwd_bf16" : "attn_bwd_f32", s, 10.0 * B * H * (double)N * N * HD,
                     esz * 8.0 * B * H * (double)N * HD);
    bool piped = false;
    DM_REQUIRE(!(table && !bias) || dtype == DM_BF16, DM_ERR_UNSUPPORTED, "dm_attention_bwd_relpos: the table-reading kernels are bf16 only");
    if (dtype == DM_BF16) {
      AttnPipeBwdParams pp{qkv, bias, out, dout, lse, delta, dqkv, dbias_slab, B, N, H, scale};
      pp.table = table;
      pp.cube_s = cube_s;
      // a table without dense rows: only the two table-reading kernels may run -- every other kernel would drop the bias
      const bool table_only = table && !bias;
      DM_REQUIRE(!table_only || dm_attn_bwd_tab_takes(pp), DM_ERR_UNSUPPORTED,
                 "dm_attention_bwd_relpos: the table-reading backward kernels do not take this call (B=%d N=%d H=%d, switches "
                 "DM_ATTN_PIPE / DM_ATTN_Q32_BWD / DM_ATTN_Q32_TABKV) and no dense bias rows were given", B, N, H);
      if (dm_attn_bwd_pipe_ok(pp)) {      // 32 rows per wave where those kernels take the pass (dm_attention_q32_bwd.hip), else 16
        const bool dq = dm_attn_bwd_dq_q32(pp, s);
        const bool dkv = dq && dm_attn_bwd_dkv_q32(pp, s);
        DM_REQUIRE(!table_only || dkv, DM_ERR_HIP, "dm_attention_bwd_relpos: a table-reading backward kernel could not be configured");
        piped = dkv || dm_attn_bwd_pipe(pp, s, dq);
      }
    }
    if (!piped) {
      if (dtype == DM_BF16) { dispatch<bf16_t>(1, p, s); dispatch<bf16_t>(2, p, s); }
      else { dispatch<float>(1, p, s); dispatch<float>(2, p, s); }
    }
  }
  DM_LAUNCH_CHECK("dm_attention_bwd");
  return DM_OK;
}

extern "C" int dm_attention_bwd(const void *qkv, const float *bias, const float *bias_t, const void *out, const void *dout,
                                const float *lse, void *dqkv, float *delta, float *dbias_slab, int32_t B, int32_t N, int32_t H, int32_t D, float scale, int32_t dtype,
                                void *stream) {
  return attention_bwd(qkv, bias, bias_t, nullptr, 0, out, dout, lse, dqkv, delta, dbias_slab, B, N, H, D, scale, dtype, stream);
}

// The backward of dm_attention_fwd_relpos: both passes form the bias from the table inside the kernel (dQ: 8 waves, dK / dV + the
// table-gradient slab: 4 waves; dm_attention_q32_bwd.hip); bias / bias_t may be NULL.
extern "C" int dm_attention_bwd_relpos(const void *qkv, const float *table, int32_t cube_s, int32_t cube_h, int32_t cube_w, const float *bias,
                                       const float *bias_t, const void *out, const void *dout, const float *lse, void *dqkv, float *delta,
                                       float *dbias_slab, int32_t B, int32_t N, int32_t H, int32_t D, float scale, int32_t dtype, void *stream) {
  DM_REQUIRE(table, DM_ERR_BAD_SHAPE, "dm_attention_bwd_relpos: null table");
  // (with dense rows given, the A/B switches may route either pass to the kernels that read them; without, attention_bwd refuses)
  DM_REQUIRE(bias || relpos_inkernel(B, N, H, D, cube_s, cube_h, cube_w, dtype), DM_ERR_UNSUPPORTED,
             "dm_attention_bwd_relpos: shape not taken (B=%d N=%d H=%d D=%d cube=%dx%dx%d dtype=%d); call dm_attention_bwd", B, N, H, D, cube_s,
             cube_h, cube_w, dtype);
  return attention_bwd(qkv, bias, bias_t, table, cube_s, out, dout, lse, dqkv, delta, dbias_slab, B, N, H, D, scale, dtype, stream);
}

// ---- "bf16x3" numerics mode: fp32 tensors, split-bf16 products (dm_attention_x3.hip) ------------------------------------------------------------
static bool split_shape(int32_t B, int32_t N, int32_t H, int32_t D, int32_t has_table, int32_t cube_s, int32_t cube_h, int32_t cube_w) {
  if (D != HD) return false;
  if (has_table && (cube_h != 8 || cube_w != 8)) return false;
  return dm_attn_x3_shape(B, N, H, has_table != 0, cube_s);
}

extern "C" int32_t dm_attention_split_ok(int32_t B, int32_t N, int32_t H, int32_t D, int32_t has_table, int32_t cube_s, int32_t cube_h,
                                         int32_t cube_w) {
  return split_shape(B, N, H, D, has_table, cube_s, cube_h, cube_w) ? 1 : 0;
}

static int split_fwd_impl(const float *qkv, void *qkv_hi, void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h, int32_t cube_w, float *out,
                          void *out_pair, float *lse, int32_t B, int32_t N, int32_t H, int32_t D, float scale, void *stream);
extern "C" int dm_attention_split_fwd(const float *qkv, void *qkv_hi, void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h,
                                      int32_t cube_w, float *out, float *lse, int32_t B, int32_t N, int32_t H, int32_t D, float scale,
                                      void *stream) {
  return split_fwd_impl(qkv, qkv_hi, qkv_lo, table, cube_s, cube_h, cube_w, out, nullptr, lse, B, N, H, D, scale, stream);
}
extern "C" int dm_attention_split_fwd_pair(const float *qkv, void *qkv_hi, void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h,
                                           int32_t cube_w, float *out, void *out_pair, float *lse, int32_t B, int32_t N, int32_t H, int32_t D,
                                           float scale, void *stream) {
  DM_REQUIRE(out_pair && dm_aligned16(out_pair), DM_ERR_BAD_SHAPE, "dm_attention_split_fwd_pair: out_pair must be a 16-byte aligned pointer");
  return split_fwd_impl(qkv, qkv_hi, qkv_lo, table, cube_s, cube_h, cube_w, out, out_pair, lse, B, N, H, D, scale, stream);
}
static int split_fwd_impl(const float *qkv, void *qkv_hi, void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h, int32_t cube_w, float *out,
                          void *out_pair, float *lse, int32_t B, int32_t N, int32_t H, int32_t D, float scale, void *stream) {
  DM_REQUIRE(split_shape(B, N, H, D, table != nullptr, cube_s, cube_h, cube_w), DM_ERR_UNSUPPORTED,
             "dm_attention_split_fwd: shape not taken (B=%d N=%d H=%d D=%d cube=%dx%dx%d); use dm_attention_fwd", B, N, H, D, cube_s, cube_h, cube_w);
  DM_REQUIRE(qkv_hi && qkv_lo && out && lse, DM_ERR_BAD_SHAPE, "dm_attention_split_fwd: null pointer");
  DM_REQUIRE(dm_aligned16(qkv) && dm_aligned16(qkv_hi) && dm_aligned16(qkv_lo) && dm_aligned16(out), DM_ERR_BAD_ALIGN,
             "dm_attention_split_fwd: tensors must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  {
    DmProfScope prof("attn_fwd_x3", s, 3.0 * 4.0 * B * H * (double)N * N * HD, 4.0 * 4.0 * B * H * (double)N * HD);
    if (qkv) dm_attn_x3_split(qkv, qkv_hi, qkv_lo, (long long)B * N * 3 * H * HD, s);      // (NULL: the caller filled the two images -- a DM_BF16_PAIR product)
    AttnX3Params p{reinterpret_cast<const bf16_t *>(qkv_hi), reinterpret_cast<const bf16_t *>(qkv_lo), table, cube_s, out, lse, B, N, H, scale,
                   reinterpret_cast<bf16_t *>(out_pair)};
    DM_REQUIRE(dm_attn_fwd_x3(p, s), DM_ERR_UNSUPPORTED, "dm_attention_split_fwd: kernel could not be configured");
  }
  DM_LAUNCH_CHECK("dm_attention_split_fwd");
  return DM_OK;
}

extern "C" int32_t dm_attention_split_bwd_chunks(int32_t B, int32_t N, int32_t H) { return dm_attn_x3_chunks(B, N, H); }

static int split_bwd_impl(const void *qkv_hi, const void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h, int32_t cube_w,
                          const float *out, const float *dout, void *dout_hi, void *dout_lo, const float *lse, float *dqkv, void *dqkv_pair,
                          float *delta, float *dbias_slab, int32_t B, int32_t N, int32_t H, int32_t D, float scale, void *stream);
extern "C" int dm_attention_split_bwd(const void *qkv_hi, const void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h, int32_t cube_w,
                                      const float *out, const float *dout, void *dout_hi, void *dout_lo, const float *lse, float *dqkv,
                                      float *delta, float *dbias_slab, int32_t B, int32_t N, int32_t H, int32_t D, float scale, void *stream) {
  DM_REQUIRE(dqkv, DM_ERR_BAD_SHAPE, "dm_attention_split_bwd: null pointer");
  return split_bwd_impl(qkv_hi, qkv_lo, table, cube_s, cube_h, cube_w, out, dout, dout_hi, dout_lo, lse, dqkv, nullptr, delta, dbias_slab, B, N, H, D, scale, stream);
}
extern "C" int dm_attention_split_bwd_pair(const void *qkv_hi, const void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h, int32_t cube_w,
                                           const float *out, const float *dout, void *dout_hi, void *dout_lo, const float *lse, void *dqkv_pair,
                                           float *delta, float *dbias_slab, int32_t B, int32_t N, int32_t H, int32_t D, float scale, void *stream) {
  DM_REQUIRE(dqkv_pair, DM_ERR_BAD_SHAPE, "dm_attention_split_bwd_pair: null pointer");
  return split_bwd_impl(qkv_hi, qkv_lo, table, cube_s, cube_h, cube_w, out, dout, dout_hi, dout_lo, lse, nullptr, dqkv_pair, delta, dbias_slab, B, N, H, D, scale, stream);
}
static int split_bwd_impl(const void *qkv_hi, const void *qkv_lo, const float *table, int32_t cube_s, int32_t cube_h, int32_t cube_w,
                          const float *out, const float *dout, void *dout_hi, void *dout_lo, const float *lse, float *dqkv, void *dqkv_pair,
                          float *delta, float *dbias_slab, int32_t B, int32_t N, int32_t H, int32_t D, float scale, void *stream) {
  DM_REQUIRE(split_shape(B, N, H, D, table != nullptr, cube_s, cube_h, cube_w), DM_ERR_UNSUPPORTED,
             "dm_attention_split_bwd: shape not taken (B=%d N=%d H=%d D=%d cube=%dx%dx%d); use dm_attention_bwd", B, N, H, D, cube_s, cube_h, cube_w);
  DM_REQUIRE(qkv_hi && qkv_lo && out && dout && dout_hi && dout_lo && lse && (dqkv || dqkv_pair) && delta, DM_ERR_BAD_SHAPE, "dm_attention_split_bwd: null pointer");
  DM_REQUIRE(table || !dbias_slab, DM_ERR_BAD_SHAPE, "dm_attention_split_bwd: a bias-gradient slab needs the table");
  DM_REQUIRE(dm_aligned16(qkv_hi) && dm_aligned16(qkv_lo) && dm_aligned16(out) && dm_aligned16(dout) && dm_aligned16(dout_hi) &&
             dm_aligned16(dout_lo) && dm_aligned16(dqkv) && dm_aligned16(dqkv_pair) && dm_aligned16(dbias_slab), DM_ERR_BAD_ALIGN,
             "dm_attention_split_bwd: tensors must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  {
    DmProfScope prof("attn_bwd_x3", s, 3.0 * 10.0 * B * H * (double)N * N * HD, 4.0 * 8.0 * B * H * (double)N * HD);
    dm_attn_x3_split(dout, dout_hi, dout_lo, (long long)B * N * H * HD, s);
    AttnX3BwdParams p{reinterpret_cast<const bf16_t *>(qkv_hi), reinterpret_cast<const bf16_t *>(qkv_lo), reinterpret_cast<const bf16_t *>(dout_hi),
                      reinterpret_cast<const bf16_t *>(dout_lo), out, dout, lse, delta, dqkv, reinterpret_cast<bf16_t *>(dqkv_pair),
                      (long long)B * N * 3 * H * HD, dbias_slab, table, cube_s, B, N, H, scale};
    DM_REQUIRE(dm_attn_bwd_dq_x3(p, s) && dm_attn_bwd_dkv_x3(p, s), DM_ERR_UNSUPPORTED, "dm_attention_split_bwd: kernels could not be configured");
  }
  DM_LAUNCH_CHECK("dm_attention_split_bwd");
  return DM_OK;
}
