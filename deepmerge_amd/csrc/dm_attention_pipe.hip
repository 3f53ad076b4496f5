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

template <int NKT>
__global__ __launch_bounds__(512) void attn_fwd_pipe_kernel(const AttnPipeParams p, int bchunk) {
  constexpr int N = NKT * 16;
  constexpr int IMG = N * 128;                    // one K or V image
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K image | V image]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int h = blockIdx.x, rb = blockIdx.y, chunk = blockIdx.z;
  const int H = p.H;
  const int b0 = chunk * bchunk, b1 = min(p.B, b0 + bchunk);
  if (b0 >= b1) return;
  const int q = rb * ROWS + wave * 16 + li;       // this lane's query row
  const bool wave_live = rb * ROWS + wave * 16 < N;          // N % 16 == 0: a wave is entirely live or entirely idle
  const long long tok_stride = 3LL * H * HD;     // elements between consecutive tokens of qkv
  const bf16_t *qkv = reinterpret_cast<const bf16_t *>(p.qkv);

  // ---- bias rows of this wave's queries: registers for the whole chunk -------------------------------------------
  f32x4 bias[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) bias[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (p.bias && wave_live) {
    const float *brow = p.bias + ((long long)h * N + q) * N + 4 * g;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) bias[kt] = dm_load4(brow + 16 * kt);
  }

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
    for (int j = 0; j < NKT / 4; ++j) {
      const int inst = wave + 8 * j;                                // instruction index: keys 8*inst .. 8*inst+7
      if (inst < N / 8) {
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
      fq[ks] = wave_live ? *reinterpret_cast<const u32x4 *>(qrow + (4 * ks + g) * 8) : (u32x4){0u, 0u, 0u, 0u};
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
  auto write_back = [&](int b) {
    if (!wave_live) return;
    bf16_t *orow = reinterpret_cast<bf16_t *>(p.out) + ((long long)b * N + q) * H * HD + (long long)h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dm_store4(orow + dt * 16 + 4 * g, o_prev[dt]);
    if (g == 0) p.lse[((long long)b * H + h) * N + q] = lse_prev;
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

    if (wave_live) {
      // ---- S = scale * Q K^T + bias ---------------------------------------------------------------------------------
      f32x4 s[NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const char *krow = kimg + (16 * kt + li) * 128;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        mma<bf16_t>(a, fq[0], *reinterpret_cast<const u32x4 *>(krow + kswz0));
        mma<bf16_t>(a, fq[1], *reinterpret_cast<const u32x4 *>(krow + kswz1));
        s[kt] = a * p.scale + bias[kt];
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
          const float e = __expf(s[kt][r] - m);
          s[kt][r] = e;
          l += e;
        }
      l = row_sum4(l);
      const float inv = 1.f / l;
      lse_prev = m + logf(l);
      // ---- O = P V ---------------------------------------------------------------------------------------------------
      f32x4 o[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mb = 0; mb < NKT / 2; ++mb) {
        const f32x4 pa = s[2 * mb] * inv, pb = s[2 * mb + 1] * inv;
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
      for (int dt = 0; dt < 4; ++dt) o_prev[dt] = o[dt];
    }
    fq[0] = fq_next[0];
    fq[1] = fq_next[1];
  }
  write_back(b1 - 1);
}

template <int NKT> void launch(const AttnPipeParams &p, hipStream_t s) {
  constexpr int N = NKT * 16;
  constexpr int LDS = 4 * N * 128;
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_pipe_kernel<NKT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)ok;
  const int nblk = (N + ROWS - 1) / ROWS;
  // one workgroup per CU (two K/V buffers fill the LDS): as many chunks as fit one round of the 256 CUs
  int chunks = 256 / (p.H * nblk);
  if (chunks < 1) chunks = 1;
  if (chunks > p.B) chunks = p.B;
  const int bchunk = (p.B + chunks - 1) / chunks;
  chunks = (p.B + bchunk - 1) / bchunk;
  hipLaunchKernelGGL(attn_fwd_pipe_kernel<NKT>, dim3(p.H, nblk, chunks), dim3(512), LDS, s, p, bchunk);
}

}  // namespace dmpipe

bool dm_attn_fwd_pipe(const AttnPipeParams &p, hipStream_t s) {
  static const int mode = [] { const char *e = getenv("DM_ATTN_PIPE"); return e ? atoi(e) : 1; }();
  if (mode == 0) return false;
  if (p.N % 16 != 0 || p.N < 128 || p.N > 256) return false;
  if ((long long)p.B * p.N * 3 * p.H * 64 * 2 >= (1LL << 40)) return false;
  if ((long long)p.N * 3 * p.H * 64 * 2 >= (1LL << 31)) return false;      // one sample's rows must fit a 32-bit DMA offset
  if (mode != 2 && p.B * p.H < 96) return false;                             // too little work for persistent workgroups
  switch (p.N / 16) {
    case 8: dmpipe::launch<8>(p, s); return true;
    case 12: dmpipe::launch<12>(p, s); return true;
    case 16: dmpipe::launch<16>(p, s); return true;
    default: return false;
  }
}
