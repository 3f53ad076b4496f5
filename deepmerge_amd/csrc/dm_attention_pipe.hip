// Attention forward as a persistent, software-pipelined kernel (bf16, head dim 64, N a multiple of 16 up to 256), gfx950.
//
// The per-(sample, head, 64-row block) kernel of dm_attention.hip is latency-bound: a workgroup lives ~13 us of which < 1 us is
// MFMA time -- it waits for its K / V tiles, then for the bias rows, then computes, with only two workgroups per CU to overlap.
// Here a workgroup owns (head, block of 128 query rows) and walks a chunk of samples:
//   * the bias rows of its queries (fp32 [16 rows x N] per wave) are loaded ONCE into registers and reused for every sample;
//   * K and V of sample i+1 arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`) into the second of two LDS buffers while sample i
//     is computed; the Q fragments of sample i+1 are prefetched into registers the same way; one barrier per sample;
//   * 8 waves x 16 query rows, whole score row in registers (exact softmax), P stays in registers as the MFMA operand of P.V
//     (two adjacent 16x16 score tiles are a valid fragment for the key order 4g+r, 16+4g+r; V is fetched in that order with
//     the hardware-transposed LDS read).
// LDS images: K [N keys][128 B], 16-byte chunk index XOR (key & 7) (conflict-free ds_read_b128 fragments);
//             V [N keys][128 B], 32-byte slot index XOR ((key >> 1) & 3) (each half-wave of a transposed read covers all banks).
#include <cstdlib>

#include "dm_attention_pipe.h"
#include "dm_common.h"
#include "dm_mfma.h"

namespace dmpipe {

constexpr int HD = 64;          // head dim
constexpr int ROWS = 128;       // query rows per workgroup (8 waves x 16)
constexpr int WB_PITCH = 144;   // write-back staging: 16 rows x (128 B + 16 B pad) per wave
constexpr int WB_BYTES = 8 * 16 * WB_PITCH;

// A wave holds a 16-row x 64-column result as v[dt] = columns dt * 16 + 4 g .. + 3 of row li (the MFMA layout): stored from there,
// an instruction writes 8 bytes to each of 16 rows x 4 places.  Through a wave-private LDS block (no barrier: one wave, in-order LDS)
// lane L ends up with the 16-byte chunk L & 7 of rows L >> 3 and 8 + (L >> 3): two stores, each covering eight whole 128-byte rows.
// row0 = pointer to the wave's first row, stride = elements between rows, nvalid = rows of the 16 that exist.
__device__ __forceinline__ void wb_rows16(char *stage, const f32x4 (&v)[4], bf16_t *row0, long long stride, int lane, int nvalid) {
  const int g = lane >> 4, li = lane & 15;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const bf16x4 r = {(bf16_t)v[dt][0], (bf16_t)v[dt][1], (bf16_t)v[dt][2], (bf16_t)v[dt][3]};
    *reinterpret_cast<bf16x4 *>(stage + li * WB_PITCH + (dt * 16 + 4 * g) * 2) = r;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int r = lane >> 3, c = lane & 7;
  const u32x4 a = *reinterpret_cast<const u32x4 *>(stage + r * WB_PITCH + c * 16);
  const u32x4 b = *reinterpret_cast<const u32x4 *>(stage + (r + 8) * WB_PITCH + c * 16);
  if (r < nvalid) *reinterpret_cast<u32x4 *>(row0 + (long long)r * stride + c * 8) = a;
  if (r + 8 < nvalid) *reinterpret_cast<u32x4 *>(row0 + (long long)(r + 8) * stride + c * 8) = b;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the block is reused by the next call)
}

__device__ __forceinline__ u32x4 pack2(const f32x4 &a, const f32x4 &b);

#define DM_LDS_DMA(rsrc, dst, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst), 16, (int)(voff), (int)(soff), 0, 0)

__device__ __forceinline__ float row_max4(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float row_sum4(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// Workgroup -> (head, row block, sample chunk) from a 1-D grid.  The row blocks of one (head, chunk) stage the SAME K / V (Q / dO)
// rows; the hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with its own L2, so with the plain
// (head, block, chunk) order those workgroups landed on different XCDs and every image was fetched from the Infinity Cache / HBM once
// per row block (timing ablation, tools/mb_attn_abl.sh: without the K / V DMA the forward kernel takes 112 instead of 157 us --
// the largest single component).  Here ids L and L + 8 .. L + 8 (nblk - 1) -- same XCD, dispatched together -- are the row blocks
// of one group, so the second reader of a row hits that XCD's L2.
__device__ __forceinline__ bool pipe_coords(int nblk, int H, int chunks, int &h, int &rb, int &chunk) {
  if (chunks < 0) {                              // DM_ATTN_XCD=0: the plain order, for A/B runs
    const int L = blockIdx.x;
    h = L % H;
    rb = (L / H) % nblk;
    chunk = L / (H * nblk);
    return chunk < -chunks;
  }
  const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
  rb = j % nblk;
  const int group = xcd + 8 * (j / nblk);
  if (group >= H * chunks) return false;
  h = group % H;
  chunk = group / H;
  return true;
}
inline bool pipe_xcd_map() {
  static const bool on = [] { const char *e = getenv("DM_ATTN_XCD"); return !(e && atoi(e) == 0); }();
  return on;
}
inline int pipe_grid_size(int nblk, int H, int chunks) { return pipe_xcd_map() ? (H * chunks + 7) / 8 * 8 * nblk : H * chunks * nblk; }

// RAGGED: N is not NKT * 16 (ViT's 197 / 198, v5's 193): tokens >= N are zero-filled by the DMA descriptor and masked.
template <int NKT, bool RAGGED, bool PF>
__global__ __launch_bounds__(512) void attn_fwd_pipe_kernel(const AttnPipeParams p, int bchunk, int nblk, int chunks) {
  constexpr int NP = NKT * 16;                    // padded token count (LDS images, tile loops)
  const int N = RAGGED ? p.N : NP;
  constexpr int IMG = NP * 128;                   // one K or V image
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K image | V image]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, li = lane & 15;
  int h, rb, chunk;
  if (!pipe_coords(nblk, p.H, chunks, h, rb, chunk)) return;
  const int H = p.H;
  const int b0 = chunk * bchunk, b1 = min(p.B, b0 + bchunk);
  if (b0 >= b1) return;
  const int q = rb * ROWS + wave * 16 + li;       // this lane's query row
  const bool wave_live = rb * ROWS + wave * 16 < N;          // a wave with no valid row only takes part in the DMA / barriers
  const bool row_ok = q < N;                                // (ragged N: the last live wave has some invalid rows)
  const long long tok_stride = 3LL * H * HD;     // elements between consecutive tokens of qkv
  const bf16_t *qkv = reinterpret_cast<const bf16_t *>(p.qkv);

  // ---- bias rows of this wave's queries: registers for the whole chunk -------------------------------------------
  f32x4 bias[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) bias[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  if (p.bias && wave_live && row_ok) {
    const float *brow = p.bias + ((long long)h * N + q) * N + 4 * g;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {               // scores are kept in log2 units: exp2 is one instruction
      if constexpr (!RAGGED) {
        bias[kt] = dm_load4(brow + 16 * kt) * LOG2E;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * kt + 4 * g + r < N) bias[kt][r] = brow[16 * kt + r] * LOG2E;
      }
    }
  }
  const float scale2 = p.scale * LOG2E;

  // ---- DMA addressing: one wave-instruction = 8 keys x 128 B; every wave stages NKT/4 instructions of K and of V --------
  // lane -> key row (lane >> 3) of the instruction, 16-byte position lane & 7; the swizzles are applied on the source chunk
  const int dkey = lane >> 3;
  const unsigned srcK = (unsigned)(((lane & 7) ^ dkey) * 16);                         // chunk ^ (key & 7)
  const unsigned srcV = (unsigned)(((lane & 7) ^ (((dkey >> 1) & 3) << 1)) * 16);     // slot ^ ((key >> 1) & 3)
  auto stage = [&](int b, int buf) {
    // the sample's K / V rows: key stride tok_stride elements; descriptor covers exactly this sample's N tokens
    const bf16_t *base = qkv + (long long)b * N * tok_stride + (long long)h * HD;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(base), 0, (int)(N * tok_stride * 2), 0x00020000);
    char *kimg = smem + buf * (2 * IMG), *vimg = kimg + IMG;
#pragma unroll
    for (int j = 0; j < (NKT + 3) / 4; ++j) {
      const int inst = wave + 8 * j;                                // instruction index: keys 8*inst .. 8*inst+7
      if (inst < NP / 8) {
        const unsigned rowoff = (unsigned)((8 * inst + dkey) * tok_stride * 2);
        DM_LDS_DMA(rs, kimg + inst * 1024, rowoff + (unsigned)(1 * H * HD * 2) + srcK, 0);
        DM_LDS_DMA(rs, vimg + inst * 1024, rowoff + (unsigned)(2 * H * HD * 2) + srcV, 0);
      }
    }
  };
  auto load_q = [&](int b, u32x4 (&fq)[2]) {
    const bf16_t *qrow = qkv + ((long long)b * N + q) * tok_stride + (long long)h * HD;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      fq[ks] = (wave_live && row_ok) ? *reinterpret_cast<const u32x4 *>(qrow + (4 * ks + g) * 8) : (u32x4){0u, 0u, 0u, 0u};
  };

  // fragment offsets inside the images
  const int kswz0 = ((g) ^ (li & 7)) << 4, kswz1 = ((4 + g) ^ (li & 7)) << 4;   // K: row li of a 16-key tile, chunk 4ks+g
  // V transposed read: lane 4q'+p' of a 16-lane group addresses key row q', d columns 4p'..4p'+3 of the 16-wide d tile;
  // rows of a 32-key block are taken in the order 4g+q' (first read) and 16+4g+q' (second): the order P's registers hold.
  const int vq = li >> 2, vp = li & 3;
  const int vrow = 4 * g + vq;                                      // + 32*m (+16)
  const int vf = ((vrow >> 1) & 3);                                 // slot XOR of these rows ((row >> 1) & 3; +16 / +32m keep it)

  u32x4 fq[2], fq_next[2];
  // results of the previous sample, written one iteration late: the vmcnt(0) at the top of an iteration then only ever waits
  // for memory operations issued a whole sample earlier (never for stores it has just issued)
  f32x4 o_prev[4];
  float lse_prev = 0.f;
  char *wb_stage = smem + 4 * IMG + wave * (16 * WB_PITCH);
  const int q_wave = rb * ROWS + wave * 16;
  auto write_back = [&](int b) {
    if (!wave_live) return;
    bf16_t *orow0 = reinterpret_cast<bf16_t *>(p.out) + ((long long)b * N + q_wave) * H * HD + (long long)h * HD;
    wb_rows16(wb_stage, o_prev, orow0, (long long)H * HD, lane, N - q_wave);
    if (g == 0 && row_ok) p.lse[((long long)b * H + h) * N + q] = lse_prev;
  };
  stage(b0, 0);
  load_q(b0, fq);
  for (int b = b0; b < b1; ++b) {
    const int buf = (b - b0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this sample's K / V / Q have landed (issued one sample ago)
    __builtin_amdgcn_s_barrier();                                  // ... for every wave; and everyone is done with the other buffer
    if (b + 1 < b1) {
      stage(b + 1, buf ^ 1);
      load_q(b + 1, fq_next);
    }
    if (b > b0) write_back(b - 1);
    const char *kimg = smem + buf * (2 * IMG), *vimg = kimg + IMG;

    if constexpr (PF) {
      if (wave_live) {
        // Hand-pipelined form (timing ablation of the plain form, tools/mb_attn_abl.sh: fragment reads 11 us, softmax 9.5 us, DMA 5 us
        // on top of a 20 us floor, and the parts ADD -- hipcc issues every batch of fragment reads right in front of its MFMAs).
        // Here the fragments of batch i + 1 are requested before the MFMAs of batch i, the scale / bias of batch i - 1 and the
        // exponentials of block mb + 1 sit in the same scheduling group as the MFMAs of batch i / block mb.
        constexpr int NB = NKT / 2;                                  // batches of two 16-key tiles (NKT is even for every instance)
        f32x4 s[NKT];
        u32x4 kf[2][4];
        auto read_k = [&](int i, u32x4 (&f)[4]) {
          const char *r0 = kimg + (32 * i + li) * 128;
          f[0] = *reinterpret_cast<const u32x4 *>(r0 + kswz0);
          f[1] = *reinterpret_cast<const u32x4 *>(r0 + kswz1);
          f[2] = *reinterpret_cast<const u32x4 *>(r0 + 2048 + kswz0);
          f[3] = *reinterpret_cast<const u32x4 *>(r0 + 2048 + kswz1);
        };
        auto finish = [&](int kt, const f32x4 &a) {
          s[kt] = a * scale2 + bias[kt];
          if constexpr (RAGGED) {
            if (16 * kt + 16 > N) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (16 * kt + 4 * g + r >= N) s[kt][r] = -INFINITY;
            }
          }
        };
        f32x4 acc[2][2];
        read_k(0, kf[0]);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          if (i + 1 < NB) read_k(i + 1, kf[(i + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
          mma<bf16_t>(a0, fq[0], kf[i & 1][0]);
          mma<bf16_t>(a1, fq[0], kf[i & 1][2]);
          mma<bf16_t>(a0, fq[1], kf[i & 1][1]);
          mma<bf16_t>(a1, fq[1], kf[i & 1][3]);
          acc[i & 1][0] = a0;
          acc[i & 1][1] = a1;
          if (i > 0) {
            finish(2 * i - 2, acc[(i - 1) & 1][0]);
            finish(2 * i - 1, acc[(i - 1) & 1][1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        finish(NKT - 2, acc[(NB - 1) & 1][0]);
        finish(NKT - 1, acc[(NB - 1) & 1][1]);
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) m = fmaxf(fmaxf(m, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
        m = row_max4(m);
        float l0 = 0.f, l1 = 0.f;
        auto exp_block = [&](int mb) {                               // tiles 2 mb, 2 mb + 1 -> unnormalised probabilities
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e0 = __builtin_amdgcn_exp2f(s[2 * mb][r] - m), e1 = __builtin_amdgcn_exp2f(s[2 * mb + 1][r] - m);
            s[2 * mb][r] = e0;
            s[2 * mb + 1][r] = e1;
            l0 += e0;
            l1 += e1;
          }
        };
        u32x2 vfr[2][8];
        auto read_v = [&](int mb, u32x2 (&f)[8]) {
          const char *vblk = vimg + (32 * mb + vrow) * 128 + 8 * vp;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const int slot = (dt ^ vf) << 5;
            f[2 * dt] = dm_ds_read_tr16(vblk + slot);
            f[2 * dt + 1] = dm_ds_read_tr16(vblk + 16 * 128 + slot);
          }
        };
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        read_v(0, vfr[0]);
        exp_block(0);
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
          if (mb + 1 < NB) read_v(mb + 1, vfr[(mb + 1) & 1]);
          const u32x4 pf = pack2(s[2 * mb], s[2 * mb + 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            mma<bf16_t>(o[dt], pf, (u32x4){vfr[mb & 1][2 * dt][0], vfr[mb & 1][2 * dt][1], vfr[mb & 1][2 * dt + 1][0], vfr[mb & 1][2 * dt + 1][1]});
          if (mb + 1 < NB) exp_block(mb + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        const float l = row_sum4(l0 + l1);
        const float inv = 1.f / l;
        lse_prev = (m + __builtin_amdgcn_logf(l)) * LN2;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o_prev[dt] = o[dt] * inv;
      }
    } else
    if (wave_live) {
      // ---- S = scale * Q K^T + bias ---------------------------------------------------------------------------------
      f32x4 s[NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const char *krow = kimg + (16 * kt + li) * 128;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        mma<bf16_t>(a, fq[0], *reinterpret_cast<const u32x4 *>(krow + kswz0));
        mma<bf16_t>(a, fq[1], *reinterpret_cast<const u32x4 *>(krow + kswz1));
        s[kt] = a * scale2 + bias[kt];
        if constexpr (RAGGED) {                      // keys past N (zero rows of the image) take no probability mass
          if (16 * kt + 16 > N) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (16 * kt + 4 * g + r >= N) s[kt][r] = -INFINITY;
          }
        }
      }
      // ---- exact softmax over the row (64 values in this lane, 4 lanes per row) -----------------------------------------
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) m = fmaxf(fmaxf(m, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
      m = row_max4(m);
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[kt][r] - m);
          s[kt][r] = e;
          l += e;
        }
      l = row_sum4(l);
      const float inv = 1.f / l;
      lse_prev = (m + __builtin_amdgcn_logf(l)) * LN2;          // natural-log units for the backward kernels
      // ---- O = P V ---------------------------------------------------------------------------------------------------
      f32x4 o[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mb = 0; mb < NKT / 2; ++mb) {
        const f32x4 pa = s[2 * mb], pb = s[2 * mb + 1];           // unnormalised (<= 1): O is scaled by 1 / l at the end
        const bf16x8 pk = {(bf16_t)pa[0], (bf16_t)pa[1], (bf16_t)pa[2], (bf16_t)pa[3], (bf16_t)pb[0], (bf16_t)pb[1], (bf16_t)pb[2], (bf16_t)pb[3]};
        const u32x4 pf = __builtin_bit_cast(u32x4, pk);
        const char *vblk = vimg + (32 * mb + vrow) * 128 + 8 * vp;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int slot = (dt ^ vf) << 5;
          const u32x2 lo = dm_ds_read_tr16(vblk + slot);
          const u32x2 hi = dm_ds_read_tr16(vblk + 16 * 128 + slot);
          mma<bf16_t>(o[dt], pf, (u32x4){lo[0], lo[1], hi[0], hi[1]});
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o_prev[dt] = o[dt] * inv;
    }
    fq[0] = fq_next[0];
    fq[1] = fq_next[1];
  }
  write_back(b1 - 1);
}

// (Round 2: a two-workgroups-per-CU variant of the kernel above -- 4 waves x 16 rows, ONE K and ONE V image refilled in place -- was
// built to stop the two waves of a SIMD running in lockstep, measured no faster (42.0 vs 40.9 us) and served as the vehicle of the
// compile-time timing ablations in profiles/r02_attn_ablations.txt; removed from the product library in round 3, see git history.)

// (Round 2, tried and removed: a variant of the kernel above that software-pipelines ACROSS samples inside a wave -- raw scores of
// sample b + 1 on the matrix pipe while the VALU exponentiates sample b, scale / bias / max of b + 1 under P(b).V(b), K one
// sample ahead of V in the same four LDS images.  Correct (all attention tests), but 59 us against 47 us at N = 256 with a bias:
// two live score arrays + the bias rows need > 256 VGPRs (scratch spills), and hipcc places each K fragment read directly in
// front of its MFMA behind an `s_waitcnt lgkmcnt(0)`, so the LDS latency is exposed 16 times per sample either way.  At N = 197
// without spills it tied (182 us vs 186 us).  Counters of the kernels that ship: profiles/r02_attn_mfma_util.md.)

// =============================================================================================================================
// backward.  Same skeleton as the forward kernel; all LDS images here are "dual-use" (row fragments for the QK^T-like
// products, hardware-transposed fragments for the contractions over tokens): 16-byte chunk index XOR s(row),
// s = ((t1^t2)<<2 | t0<<1 | t1), t = row >> 1 -- the image layout of dm_attention.hip, filled by DMA here.
// =============================================================================================================================
__device__ __forceinline__ int dual_swz(int row) {
  const int t0 = (row >> 1) & 1, t1 = (row >> 2) & 1, t2 = (row >> 3) & 1;
  return ((t1 ^ t2) << 2) | (t0 << 1) | t1;
}

// Per-lane constants of the two fragment kinds (see frag_row / frag_tr in dm_attention.hip).
struct FragAddr {
  int row[2];     // row fragment of tile row `li`, 64-byte block ks: + tile * 2048
  int tr[4];      // transposed fragment of d-tile dt: + (32 * m + 16 * half) * 128
  __device__ __forceinline__ void init(int lane) {
    const int g = lane >> 4, li = lane & 15;
    const int sl = dual_swz(li);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) row[ks] = li * 128 + (((ks * 4 + g) ^ sl) << 4);
    const int qq = (lane >> 2) & 3, pp = lane & 3;
    const int k = 4 * g + qq;                      // + 32 m + 16 half: does not change the swizzle bits
    const int sk = dual_swz(k);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) tr[dt] = k * 128 + (((2 * dt + (pp >> 1)) ^ sk) << 4) + (pp & 1) * 8;
  }
};
__device__ __forceinline__ u32x4 read_row(const char *img, const FragAddr &fa, int tile, int ks) {
  return *reinterpret_cast<const u32x4 *>(img + tile * 2048 + fa.row[ks]);
}
__device__ __forceinline__ u32x4 read_tr(const char *img, const FragAddr &fa, int m, int dt) {
  const u32x2 lo = dm_ds_read_tr16(img + (32 * m) * 128 + fa.tr[dt]);
  const u32x2 hi = dm_ds_read_tr16(img + (32 * m + 16) * 128 + fa.tr[dt]);
  return (u32x4){lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ u32x4 pack2(const f32x4 &a, const f32x4 &b) {
  const bf16x8 r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
  return __builtin_bit_cast(u32x4, r);
}
// sum over the 8 bf16 pairs of two fragments
__device__ __forceinline__ float dot8(const u32x4 &x, const u32x4 &y) {
  const bf16x8 a = __builtin_bit_cast(bf16x8, x), b = __builtin_bit_cast(bf16x8, y);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += (float)a[i] * (float)b[i];
  return acc;
}

// ---- dQ (rows of the workgroup are queries) ---------------------------------------------------------------------------------
template <int NKT, bool RAGGED>
__global__ __launch_bounds__(512) void attn_bwd_dq_pipe_kernel(const AttnPipeBwdParams p, int bchunk, int nblk, int chunks) {
  constexpr int NP = NKT * 16;
  const int N = RAGGED ? p.N : NP;
  constexpr int IMG = NP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K image | V image]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, li = lane & 15;
  int h, rb, chunk;
  if (!pipe_coords(nblk, p.H, chunks, h, rb, chunk)) return;
  const int H = p.H;
  const int b0 = chunk * bchunk, b1 = min(p.B, b0 + bchunk);
  if (b0 >= b1) return;
  const int q = rb * ROWS + wave * 16 + li;
  const bool wave_live = rb * ROWS + wave * 16 < N;
  const bool row_ok = q < N;
  const long long tok_stride = 3LL * H * HD;
  const bf16_t *qkv = reinterpret_cast<const bf16_t *>(p.qkv);
  const bf16_t *outp = reinterpret_cast<const bf16_t *>(p.out);
  const bf16_t *dout = reinterpret_cast<const bf16_t *>(p.dout);
  FragAddr fa;
  fa.init(lane);

  f32x4 bias[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) bias[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr float LOG2E = 1.4426950408889634f;
  if (p.bias && wave_live && row_ok) {
    const float *brow = p.bias + ((long long)h * N + q) * N + 4 * g;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {               // log2 units, see the forward kernel
      if constexpr (!RAGGED) {
        bias[kt] = dm_load4(brow + 16 * kt) * LOG2E;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * kt + 4 * g + r < N) bias[kt][r] = brow[16 * kt + r] * LOG2E;
      }
    }
  }
  const float scale2 = p.scale * LOG2E;

  const int dkey = lane >> 3;
  auto stage = [&](int b, int buf) {
    const bf16_t *base = qkv + (long long)b * N * tok_stride + (long long)h * HD;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(base), 0, (int)(N * tok_stride * 2), 0x00020000);
    char *kimg = smem + buf * (2 * IMG), *vimg = kimg + IMG;
#pragma unroll
    for (int j = 0; j < (NKT + 3) / 4; ++j) {
      const int inst = wave + 8 * j;
      if (inst < NP / 8) {
        const int row = 8 * inst + dkey;
        const unsigned src = (unsigned)(row * tok_stride * 2) + (unsigned)((((lane & 7) ^ dual_swz(row))) * 16);
        DM_LDS_DMA(rs, kimg + inst * 1024, src + (unsigned)(1 * H * HD * 2), 0);
        DM_LDS_DMA(rs, vimg + inst * 1024, src + (unsigned)(2 * H * HD * 2), 0);
      }
    }
  };
  // per-sample register inputs of this lane's query row: Q, dO, O fragments (d = 8g..8g+7 and 32+8g..) and lse
  auto load_rows = [&](int b, u32x4 (&fq)[2], u32x4 (&fdo)[2], u32x4 (&fo)[2], float &lse) {
    const bf16_t *qrow = qkv + ((long long)b * N + q) * tok_stride + (long long)h * HD;
    const long long orow = ((long long)b * N + q) * H * HD + (long long)h * HD;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = (4 * ks + g) * 8;
      const bool ok = wave_live && row_ok;
      fq[ks] = ok ? *reinterpret_cast<const u32x4 *>(qrow + c) : (u32x4){0u, 0u, 0u, 0u};
      fdo[ks] = ok ? *reinterpret_cast<const u32x4 *>(dout + orow + c) : (u32x4){0u, 0u, 0u, 0u};
      fo[ks] = ok ? *reinterpret_cast<const u32x4 *>(outp + orow + c) : (u32x4){0u, 0u, 0u, 0u};
    }
    lse = (wave_live && row_ok) ? p.lse[((long long)b * H + h) * N + q] * LOG2E : 0.f;
  };

  u32x4 fq[2], fdo[2], fo[2], fq_n[2], fdo_n[2], fo_n[2];
  float lse = 0.f, lse_n = 0.f;
  f32x4 o_prev[4];
  float delta_prev = 0.f;
  char *wb_stage = smem + 4 * NP * 128 + wave * (16 * WB_PITCH);
  const int q_wave = rb * ROWS + wave * 16;
  auto write_back = [&](int b) {
    if (!wave_live) return;
    bf16_t *dq0 = reinterpret_cast<bf16_t *>(p.dqkv) + ((long long)b * N + q_wave) * tok_stride + (long long)h * HD;
    wb_rows16(wb_stage, o_prev, dq0, tok_stride, lane, N - q_wave);
    if (g == 0 && row_ok) p.delta[((long long)b * H + h) * N + q] = delta_prev;
  };

  stage(b0, 0);
  load_rows(b0, fq, fdo, fo, lse);
  for (int b = b0; b < b1; ++b) {
    const int buf = (b - b0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (b + 1 < b1) {
      stage(b + 1, buf ^ 1);
      load_rows(b + 1, fq_n, fdo_n, fo_n, lse_n);
    }
    if (b > b0) write_back(b - 1);
    const char *kimg = smem + buf * (2 * IMG), *vimg = kimg + IMG;
    if (wave_live) {
      const float dl = row_sum4(dot8(fdo[0], fo[0]) + dot8(fdo[1], fo[1]));      // delta[q] = sum_d dO O
      f32x4 o[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < NKT / 2; ++m) {
        f32x4 ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int kt = 2 * m + u;
          f32x4 sc = {0.f, 0.f, 0.f, 0.f}, a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            mma<bf16_t>(sc, fq[ks], read_row(kimg, fa, kt, ks));
            mma<bf16_t>(a, fdo[ks], read_row(vimg, fa, kt, ks));
          }
          sc = sc * scale2 + bias[kt];
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[u][r] = __builtin_amdgcn_exp2f(sc[r] - lse) * (a[r] - dl);
          if constexpr (RAGGED) {                    // keys past N: no probability, no gradient
            if (16 * kt + 16 > N) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (16 * kt + 4 * g + r >= N) ds[u][r] = 0.f;
            }
          }
        }
        const u32x4 fds = pack2(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) mma<bf16_t>(o[dt], fds, read_tr(kimg, fa, m, dt));
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o_prev[dt] = o[dt] * p.scale;
      delta_prev = dl;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { fq[ks] = fq_n[ks]; fdo[ks] = fdo_n[ks]; fo[ks] = fo_n[ks]; }
    lse = lse_n;
  }
  write_back(b1 - 1);
}

// ---- dK, dV (rows of the workgroup are keys; columns are queries), plus the chunk's summed dS for the bias gradient ----------
// BIAS = false (vit_model.py attention: no bias table): no transposed-bias rows and no dS accumulator in registers
template <int NKT, bool RAGGED, bool BIAS>
__global__ __launch_bounds__(512) void attn_bwd_dkv_pipe_kernel(const AttnPipeBwdParams p, int bchunk, int nblk, int chunks) {
  constexpr int NP = NKT * 16;
  const int N = RAGGED ? p.N : NP;
  constexpr int IMG = NP * 128;
  constexpr int BUF = 2 * IMG + 2 * 1024;                        // Q image | dO image | lse[256] | delta[256]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, li = lane & 15;
  int h, rb, chunk;
  if (!pipe_coords(nblk, p.H, chunks, h, rb, chunk)) return;
  const int H = p.H;
  const int b0 = chunk * bchunk, b1 = min(p.B, b0 + bchunk);
  if (b0 >= b1) return;
  const int key = rb * ROWS + wave * 16 + li;
  const bool wave_live = rb * ROWS + wave * 16 < N;
  const bool row_ok = key < N;
  const long long tok_stride = 3LL * H * HD;
  const bf16_t *qkv = reinterpret_cast<const bf16_t *>(p.qkv);
  const bf16_t *dout = reinterpret_cast<const bf16_t *>(p.dout);
  FragAddr fa;
  fa.init(lane);

  // bias^T rows of this wave's keys: biasT[qt][r] = bias[h][q = 16 qt + 4g + r][key]  (strided, once per workgroup).
  // (Natural-log units here: the log2-unit form of the other two kernels costs this one its last free registers.)
  constexpr int NBR = BIAS ? NKT : 1;
  f32x4 biasT[NBR], hacc[NBR];
#pragma unroll
  for (int qt = 0; qt < NBR; ++qt) { biasT[qt] = (f32x4){0.f, 0.f, 0.f, 0.f}; hacc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  if (BIAS && p.bias && wave_live && row_ok) {
    const float *bcol = p.bias + (long long)h * N * N + key;
#pragma unroll
    for (int qt = 0; qt < NKT; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!RAGGED || 16 * qt + 4 * g + r < N) biasT[qt][r] = bcol[(long long)(16 * qt + 4 * g + r) * N];
  }

  const int dkey = lane >> 3;
  auto stage = [&](int b, int buf) {
    const bf16_t *qbase = qkv + (long long)b * N * tok_stride + (long long)h * HD;
    const bf16_t *dbase = dout + (long long)b * N * H * HD + (long long)h * HD;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(qbase), 0, (int)(N * tok_stride * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(dbase), 0, (int)(N * H * HD * 2), 0x00020000);
    char *qimg = smem + buf * BUF, *dimg = qimg + IMG;
#pragma unroll
    for (int j = 0; j < (NKT + 3) / 4; ++j) {
      const int inst = wave + 8 * j;
      if (inst < NP / 8) {
        const int row = 8 * inst + dkey;
        const unsigned chunk16 = (unsigned)(((lane & 7) ^ dual_swz(row)) * 16);
        DM_LDS_DMA(rq, qimg + inst * 1024, (unsigned)(row * tok_stride * 2) + chunk16, 0);
        DM_LDS_DMA(rd, dimg + inst * 1024, (unsigned)(row * H * HD * 2) + chunk16, 0);
      }
    }
    if (wave < 2) {     // the sample's lse / delta rows (N floats each; lanes past N read zero)
      const float *src = (wave == 0 ? p.lse : p.delta) + ((long long)b * H + h) * N;
      const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, N * 4, 0x00020000);
      DM_LDS_DMA(rv, qimg + 2 * IMG + wave * 1024, (unsigned)(lane * 16), 0);
    }
  };
  auto load_rows = [&](int b, u32x4 (&fk)[2], u32x4 (&fv)[2]) {
    const bf16_t *krow = qkv + ((long long)b * N + key) * tok_stride + (long long)H * HD + (long long)h * HD;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = (4 * ks + g) * 8;
      fk[ks] = (wave_live && row_ok) ? *reinterpret_cast<const u32x4 *>(krow + c) : (u32x4){0u, 0u, 0u, 0u};
      fv[ks] = (wave_live && row_ok) ? *reinterpret_cast<const u32x4 *>(krow + (long long)H * HD + c) : (u32x4){0u, 0u, 0u, 0u};
    }
  };

  u32x4 fk[2], fv[2], fk_n[2], fv_n[2];
  f32x4 dk_prev[4], dv_prev[4];
  char *wb_stage = smem + 2 * (2 * NP * 128 + 2048) + wave * (16 * WB_PITCH);
  const int key_wave = rb * ROWS + wave * 16;
  auto write_back = [&](int b) {
    if (!wave_live) return;
    bf16_t *dk0 = reinterpret_cast<bf16_t *>(p.dqkv) + ((long long)b * N + key_wave) * tok_stride + (long long)H * HD + (long long)h * HD;
    // Measured (tools/mb_attn.py, one box): whole-row write-back makes the bias-free kernel faster (ViT, N = 197: backward
    // 418 -> 398 us) and the one that also carries the bias-gradient accumulators slower (N = 256: 139 -> 149 us): strips stay there.
    if constexpr (!BIAS) {
      wb_rows16(wb_stage, dk_prev, dk0, tok_stride, lane, N - key_wave);
      wb_rows16(wb_stage, dv_prev, dk0 + (long long)H * HD, tok_stride, lane, N - key_wave);
    } else if (row_ok) {
      bf16_t *dk = dk0 + (long long)li * tok_stride, *dv = dk + (long long)H * HD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dm_store4(dk + dt * 16 + 4 * g, dk_prev[dt]);
        dm_store4(dv + dt * 16 + 4 * g, dv_prev[dt]);
      }
    }
  };

  stage(b0, 0);
  load_rows(b0, fk, fv);
  for (int b = b0; b < b1; ++b) {
    const int buf = (b - b0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (b + 1 < b1) {
      stage(b + 1, buf ^ 1);
      load_rows(b + 1, fk_n, fv_n);
    }
    if (b > b0) write_back(b - 1);
    const char *qimg = smem + buf * BUF, *dimg = qimg + IMG;
    const float *lse_l = reinterpret_cast<const float *>(qimg + 2 * IMG), *delta_l = lse_l + 256;
    if (wave_live) {
      f32x4 odk[4], odv[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { odk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; odv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int m = 0; m < NKT / 2; ++m) {
        f32x4 pv[2], ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int qt = 2 * m + u;
          f32x4 sc = {0.f, 0.f, 0.f, 0.f}, a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            mma<bf16_t>(sc, fk[ks], read_row(qimg, fa, qt, ks));
            mma<bf16_t>(a, fv[ks], read_row(dimg, fa, qt, ks));
          }
          sc = sc * p.scale;
          if constexpr (BIAS) sc += biasT[qt];
          const f32x4 l4 = *reinterpret_cast<const f32x4 *>(lse_l + 16 * qt + 4 * g);
          const f32x4 d4 = *reinterpret_cast<const f32x4 *>(delta_l + 16 * qt + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pv[u][r] = __expf(sc[r] - l4[r]);
            ds[u][r] = pv[u][r] * (a[r] - d4[r]);
          }
          if constexpr (RAGGED) {                    // queries past N (zero rows of the images, lse = 0) contribute nothing
            if (16 * qt + 16 > N) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (16 * qt + 4 * g + r >= N) { pv[u][r] = 0.f; ds[u][r] = 0.f; }
            }
          }
          if constexpr (BIAS) {
            if (p.slab) hacc[qt] += ds[u];
          }
        }
        const u32x4 fpt = pack2(pv[0], pv[1]), fds = pack2(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          mma<bf16_t>(odv[dt], fpt, read_tr(dimg, fa, m, dt));
          mma<bf16_t>(odk[dt], fds, read_tr(qimg, fa, m, dt));
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { dk_prev[dt] = odk[dt] * p.scale; dv_prev[dt] = odv[dt]; }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { fk[ks] = fk_n[ks]; fv[ks] = fv_n[ks]; }
  }
  write_back(b1 - 1);
  if (BIAS && p.slab && wave_live && row_ok) {
    // dbias[chunk][h][q][key] = the chunk's summed dS (this lane holds 4 consecutive q of one key: scattered 4-byte stores,
    // once per chunk)
    float *plane = p.slab + ((long long)chunk * H + h) * N * N + key;
#pragma unroll
    for (int qt = 0; qt < NKT; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!RAGGED || 16 * qt + 4 * g + r < N) plane[(long long)(16 * qt + 4 * g + r) * N] = hacc[BIAS ? qt : 0][r];
  }
}

inline void pipe_grid(int B, int N, int H, int &nblk, int &chunks, int &bchunk) {
  nblk = (N + ROWS - 1) / ROWS;
  chunks = 256 / (H * nblk);
  if (chunks < 1) chunks = 1;
  if (chunks > B) chunks = B;
  bchunk = (B + chunks - 1) / chunks;
  chunks = (B + bchunk - 1) / bchunk;
}

template <int NKT, bool RAGGED> void launch_bwd(const AttnPipeBwdParams &p, hipStream_t s, bool dq_done) {
  constexpr int NP = NKT * 16;
  constexpr int LDS_DQ = 4 * NP * 128 + WB_BYTES, LDS_DKV = 2 * (2 * NP * 128 + 2048) + WB_BYTES;
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dq_pipe_kernel<NKT, RAGGED>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ) == hipSuccess &&
                         hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dkv_pipe_kernel<NKT, RAGGED, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV) == hipSuccess &&
                         hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dkv_pipe_kernel<NKT, RAGGED, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV) == hipSuccess;
  (void)ok;
  int nblk, chunks, bchunk;
  pipe_grid(p.B, p.N, p.H, nblk, chunks, bchunk);
  if (!dq_done)       // (dQ and delta may already have been written by the 32-row kernel of dm_attention_q32_bwd.hip)
    hipLaunchKernelGGL((attn_bwd_dq_pipe_kernel<NKT, RAGGED>), dim3(pipe_grid_size(nblk, p.H, chunks)), dim3(512), LDS_DQ, s, p, bchunk, nblk, pipe_xcd_map() ? chunks : -chunks);
  if (p.bias || p.slab)
    hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<NKT, RAGGED, true>), dim3(pipe_grid_size(nblk, p.H, chunks)), dim3(512), LDS_DKV, s, p, bchunk, nblk, pipe_xcd_map() ? chunks : -chunks);
  else
    hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<NKT, RAGGED, false>), dim3(pipe_grid_size(nblk, p.H, chunks)), dim3(512), LDS_DKV, s, p, bchunk, nblk, pipe_xcd_map() ? chunks : -chunks);
}


template <int NKT, bool RAGGED, bool PF> void launch_pf(const AttnPipeParams &p, hipStream_t s);
template <int NKT, bool RAGGED> void launch(const AttnPipeParams &p, hipStream_t s) {
  static const bool pf = [] { const char *e = getenv("DM_ATTN_PF"); return !(e && atoi(e) == 0); }();
  if (pf) launch_pf<NKT, RAGGED, true>(p, s);
  else launch_pf<NKT, RAGGED, false>(p, s);
}
template <int NKT, bool RAGGED, bool PF> void launch_pf(const AttnPipeParams &p, hipStream_t s) {
  constexpr int NP = NKT * 16;
  constexpr int LDS = 4 * NP * 128 + WB_BYTES;
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_pipe_kernel<NKT, RAGGED, PF>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)ok;
  // one workgroup per CU (two K/V buffers fill the LDS): as many chunks as fit one round of the 256 CUs
  int nblk, chunks, bchunk;
  pipe_grid(p.B, p.N, p.H, nblk, chunks, bchunk);
  hipLaunchKernelGGL((attn_fwd_pipe_kernel<NKT, RAGGED, PF>), dim3(pipe_grid_size(nblk, p.H, chunks)), dim3(512), LDS, s, p, bchunk, nblk, pipe_xcd_map() ? chunks : -chunks);
}

}  // namespace dmpipe

// Which instance serves N tokens: exact tilings 128 / 192 / 256 use the unmasked kernels, every other N in (128, 256]
// (ViT's 197 / 198, v5's 193, ...) the masked ones with the tile count rounded up to an even number.
static bool pipe_shape_ok(int B, int N, int H) {
  static const int mode = [] { const char *e = getenv("DM_ATTN_PIPE"); return e ? atoi(e) : 1; }();
  if (mode == 0) return false;
  if (N < 128 || N > 256) return false;
  if ((long long)N * 3 * H * 64 * 2 >= (1LL << 31)) return false;            // one sample's rows must fit a 32-bit DMA offset
  if (mode != 2 && B * H < 96) return false;                                  // too little work for persistent workgroups
  return true;
}
static int pipe_tiles(int N, bool &ragged) {
  ragged = !(N == 128 || N == 192 || N == 256);
  const int nkt = (N + 15) / 16;
  return ragged ? (nkt + 1) / 2 * 2 : nkt;
}

bool dm_attn_fwd_pipe(const AttnPipeParams &p, hipStream_t s) {
  if (!pipe_shape_ok(p.B, p.N, p.H)) return false;
  bool ragged;
  switch (pipe_tiles(p.N, ragged)) {
    case 8: dmpipe::launch<8, false>(p, s); return true;
    case 10: dmpipe::launch<10, true>(p, s); return true;
    case 12: if (ragged) dmpipe::launch<12, true>(p, s); else dmpipe::launch<12, false>(p, s); return true;
    case 14: dmpipe::launch<14, true>(p, s); return true;
    case 16: if (ragged) dmpipe::launch<16, true>(p, s); else dmpipe::launch<16, false>(p, s); return true;
    default: return false;
  }
}

int dm_attn_bwd_pipe_chunks(int B, int N, int H, int dtype_is_bf16) {
  if (!dtype_is_bf16 || !pipe_shape_ok(B, N, H)) return 0;
  bool ragged;
  pipe_tiles(N, ragged);
  if (ragged) return 0;          // a masked backward only runs without a bias (below), where no slab exists
  int nblk, chunks, bchunk;
  dmpipe::pipe_grid(B, N, H, nblk, chunks, bchunk);
  return chunks;
}

bool dm_attn_bwd_pipe_ok(const AttnPipeBwdParams &p) {
  if (!pipe_shape_ok(p.B, p.N, p.H)) return false;
  bool ragged;
  const int nkt = pipe_tiles(p.N, ragged);
  // masked + bias (v5's N = 193): the dK/dV kernel would need > 256 registers (it spills); the generic kernels take it
  if (ragged && (p.bias || p.slab)) return false;
  return nkt >= 8 && nkt <= 16 && nkt % 2 == 0;
}

bool dm_attn_bwd_pipe(const AttnPipeBwdParams &p, hipStream_t s, bool dq_done) {
  if (!dm_attn_bwd_pipe_ok(p)) return false;
  bool ragged;
  const int nkt = pipe_tiles(p.N, ragged);
  switch (nkt) {
    case 8: dmpipe::launch_bwd<8, false>(p, s, dq_done); return true;
    case 10: dmpipe::launch_bwd<10, true>(p, s, dq_done); return true;
    case 12: if (ragged) dmpipe::launch_bwd<12, true>(p, s, dq_done); else dmpipe::launch_bwd<12, false>(p, s, dq_done); return true;
    case 14: dmpipe::launch_bwd<14, true>(p, s, dq_done); return true;
    case 16: if (ragged) dmpipe::launch_bwd<16, true>(p, s, dq_done); else dmpipe::launch_bwd<16, false>(p, s, dq_done); return true;
    default: return false;
  }
}
