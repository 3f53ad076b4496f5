// 256x256x64 bf16 GEMM pipeline for the large products of the encoder, gfx950: NT (forward), NN (dgrad) and TN (wgrad,
// split-K) from one template -- an operand is either k-contiguous (image rows = tile rows, read with ds_read_b128) or
// m-contiguous (image = 4 bands of [64 k][64 columns], read with the hardware transpose ds_read_b64_tr_b16).
//
// Why a second kernel: the 128x128 register-staged tile of dm_gemm.hip tops out near 0.87 PFLOP/s (one barrier
// per K stage, every wave waits for the whole stage).  This kernel keeps the matrix pipe fed with the structure
// that needs one workgroup per CU:
//   * 8 waves = 2 (M) x 4 (N), 128x64 outputs per wave (32 accumulator tiles), one workgroup per CU;
//   * operands arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR round trip, hardware bounds check zero-fills
//     ragged M / N) into two 64 KiB K-tile buffers; the XOR swizzle of the LDS image is applied on the SOURCE
//     address (LDS destinations of one wave-instruction are lane-linear);
//   * a K tile is consumed in four phases (one 64x32 quadrant of the wave's outputs x K = 64, 16 MFMAs each);
//     each phase = [LDS fragment reads + DMA issue] barrier [MFMAs] barrier, and the two wave rows run ONE barrier
//     apart, so on every SIMD one wave issues MFMAs while its partner reads LDS / issues DMA;
//   * DMA completion is waited for with a COUNTED vmcnt once per K tile (6 pieces stay in flight across it),
//     never vmcnt(0) inside the loop.
//
// Hazard bookkeeping (phases p0..p3 of K tile t, buffer t&1; "piece" = 64 image rows = one DMA instruction per wave):
//   reads : p0 A sub0 + B sub0, p1 B sub1, p2 A sub1, p3 none       (sub = which half of the wave's rows / columns)
//   stage : p0 A-sub1 pieces of tile t+1; p2 A-sub0 + B-sub0 pieces of tile t+2; p3 B-sub1 pieces of tile t+2
//           -> every piece is re-staged >= 2 phases after its last read (needed because of the one-barrier stagger)
//   wait  : p3, vmcnt(6) before the phase's first barrier retires all of tile t+1 (the 6 newer pieces are tile t+2's);
//           tile t+1 is first read in the next phase.
#include <cstdlib>

#include "dm_common.h"
#include "dm_gemm_common.h"
#include "dm_mfma.h"

namespace dm256 {

constexpr int T256 = 256;                 // workgroup tile (rows and columns)
constexpr int BK256 = 64;                 // K per tile
constexpr int OPER_BYTES = T256 * 128;    // one operand's image of a K tile: 256 rows x 128 B
constexpr int BUF_BYTES = 2 * OPER_BYTES; // A image + B image
constexpr int EPI256 = 8 * 64 * DM_EPI_PITCH;   // whole-line epilogue staging: 64 rows x 272 B per wave
constexpr int LDS256 = (2 * BUF_BYTES > EPI256) ? 2 * BUF_BYTES : EPI256;     // two K-tile buffers = 128 KiB; staging 136 KiB

// image row of the B operand -> column of the tile.  Image rows are ordered [sub][wave column][32] so that the
// columns the waves read in the same phase are contiguous pieces (see the hazard table above).
__device__ __forceinline__ int b_image_to_col(int rho) { return ((rho >> 5) & 3) * 64 + (rho >> 7) * 32 + (rho & 31); }

#define DM_LDS_DMA(rsrc, dst, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst), 16, voff, soff, 0, 0)

// One operand of the pipeline.  MM = false: k-contiguous [rows][K] (A of NT/NN, B of NT); MM = true: m-contiguous [K][cols]
// (A of TN, B of NN/TN).  IS_B selects the image ordering of the B operand (sub-major, see b_image_to_col).
template <bool MM, bool IS_B> struct Operand {
  const bf16_t *pnl;     // start of this tile's panel and its byte extent (the buffer descriptor is built from these at
  int nbytes;            // the DMA sites: the descriptor type itself is device-only and cannot be a member here)
  unsigned vo[4];        // per-lane byte offset of DMA piece u (64 image rows, or one 64-column band)
  unsigned tile_step;    // soffset advance per K tile
  int frag[4];           // per-lane LDS byte offsets of the fragment reads (see load())

  // base: operand pointer; ld: leading dimension; o0 / extent: first row (column) of this tile and the operand's extent in
  // that dimension; kbeg / kend: K range of this workgroup.
  __device__ __forceinline__ void setup(const bf16_t *base, long long ld, int o0, int extent, int kbeg, int kend, int wave, int lane,
                                        int wrc /* wave row (A) or wave column (B) */) {
    const int g = lane >> 4, li = lane & 15;
    if constexpr (!MM) {
      pnl = base + (long long)o0 * ld + kbeg;
      const long long bytes = ((long long)(min(T256, extent - o0) - 1) * ld + (kend - kbeg)) * 2;
      nbytes = (int)min(bytes, 0x7fffffffLL);
      // a wave-instruction fills 8 image rows x 128 B; slot s of image row r holds the operand's 16-byte chunk s ^ (r & 7)
      const int srow = 8 * wave + (lane >> 3);
      const int chunk = (lane & 7) ^ (lane >> 3);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int img = 64 * u + srow;
        const int row = IS_B ? b_image_to_col(img) : img;
        vo[u] = (unsigned)(((long long)row * ld) * 2 + chunk * 16);
      }
      tile_step = BK256 * 2;
      const int sw0 = (g ^ (li & 7)) << 4, sw1 = ((4 + g) ^ (li & 7)) << 4;
      const int rowb = (IS_B ? wrc * 32 : wrc * 128) + li;
      frag[0] = rowb * 128 + sw0;
      frag[1] = rowb * 128 + sw1;
      frag[2] = frag[3] = 0;
    } else {
      pnl = base + (long long)kbeg * ld + o0;
      const long long bytes = ((long long)(kend - kbeg - 1) * ld + (extent - o0)) * 2;
      nbytes = (int)min(bytes, 0x7fffffffLL);
      // image = 4 bands of [64 k-rows][64 columns = 128 B]; a wave-instruction fills k-rows 8w..8w+7 of one band.
      // The 32-byte slot index of k-row r is XORed with f(r) = ((r >> 1) & 1) | (((r >> 3) & 1) << 1): the 8 k-rows a
      // half-wave of a transposed read touches (r = 8g + q, g in {0,1} or {2,3}, q = 0..3) then cover all 64 banks once.
      const int krow = 8 * wave + (lane >> 3);
      const int fk = ((krow >> 1) & 1) | (((krow >> 3) & 1) << 1);
      const int csrc = (lane & 7) ^ (fk << 1);                 // source 16-byte chunk of this lane's LDS position
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int col = IS_B ? ((u & 1) * 2 + (csrc >> 2)) * 64 + (u >> 1) * 32 + (csrc & 3) * 8 : u * 64 + csrc * 8;
        const unsigned kill = (o0 + col < extent) ? 0u : 0x80000000u;      // columns past the operand would alias the next row
        vo[u] = (unsigned)(((long long)krow * ld + col) * 2) | kill;
      }
      tile_step = (unsigned)(BK256 * ld * 2);
      // transposed read: lane 4q+p of a 16-lane group addresses k-row q, columns 4p..4p+3; group gq covers k = 8gq..8gq+7
      const int q = li >> 2, pq = li & 3;
      const int rbase = (8 * g + q) * 128 + 8 * pq;
      const int fr = ((q >> 1) & 1) | ((g & 1) << 1);           // f(r) of every k-row this lane addresses (r = 32ks + 8g + q [+4])
      if constexpr (!IS_B) {
#pragma unroll
        for (int i = 0; i < 4; ++i) frag[i] = rbase + ((i ^ fr) << 5);      // m-tile i of the band = 32-byte slot i
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) frag[j] = rbase + ((((wrc & 1) * 2 + j) ^ fr) << 5);
        frag[2] = frag[3] = 0;
      }
    }
  }
  // persistent kernel: panel start / extent of another tile (whole K; the per-lane offsets above do not depend on the tile as long as
  // no column of an m-contiguous panel lies past the operand: the plan requires N % 256 == 0 for that kernel)
  __device__ __forceinline__ void retile(const bf16_t *base, long long ld, int o0, int extent, int K) {
    if constexpr (!MM) {
      pnl = base + (long long)o0 * ld;
      nbytes = (int)min(((long long)(min(T256, extent - o0) - 1) * ld + K) * 2, 0x7fffffffLL);
    } else {
      pnl = base + o0;
      nbytes = (int)min(((long long)(K - 1) * ld + (extent - o0)) * 2, 0x7fffffffLL);
    }
  }
  // LDS byte offset (inside an operand image) of DMA piece u for this wave
  __device__ __forceinline__ static int piece_offset(int u, int wave) { return MM ? u * 8192 + wave * 1024 : (64 * u + 8 * wave) * 128; }

  // fragments of sub-tile `sub` (A: 64 rows = 4 tiles x 2 k-steps -> f[8]; B: 32 columns = 2 tiles x 2 k-steps -> f[4])
  template <int NT_> __device__ __forceinline__ void load(u32x4 (&f)[2 * NT_], const char *img, int sub, int wrc) const {
    if constexpr (!MM) {
      const int stride = IS_B ? 128 : 64;                       // image rows per sub
#pragma unroll
      for (int i = 0; i < NT_; ++i) {
        const char *r = img + (sub * stride + i * 16) * 128;
        f[2 * i] = *reinterpret_cast<const u32x4 *>(r + frag[0]);
        f[2 * i + 1] = *reinterpret_cast<const u32x4 *>(r + frag[1]);
      }
    } else {
      const int band = IS_B ? 2 * sub + (wrc >> 1) : 2 * wrc + sub;
      const char *b = img + band * 8192;
#pragma unroll
      for (int i = 0; i < NT_; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const u32x2 lo = dm_ds_read_tr16(b + frag[i] + (32 * ks) * 128);
          const u32x2 hi = dm_ds_read_tr16(b + frag[i] + (32 * ks + 4) * 128);
          f[2 * i + ks] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    }
  }
};

// FOLD: hi / lo plane pairs, three K segments of p.k_fold (GemmParams.k_fold): the panels start at segment offset 0 and a K tile's DMA
// offset is its segment's offset plus its position inside the segment.
template <int LAYOUT, bool FOLD = false>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmParams p) {
  constexpr bool AM = (LAYOUT == DM_TN), BMM = (LAYOUT != DM_NT);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, li = lane & 15;

  // ---- tile of this workgroup (XCD-contiguous ids, rows fastest inside bands of group_m row tiles) ----------
  int id = dm_xcd_remap(blockIdx.x, gridDim.x);
  const int per_z = p.tiles_m * p.tiles_n;
  const int z = id / per_z;
  id -= z * per_z;
  int tm, tn;
  if (p.group_m > 0) {
    const int band = id / (p.group_m * p.tiles_n);
    const int within = id - band * (p.group_m * p.tiles_n);
    const int gsz = min(p.group_m, p.tiles_m - band * p.group_m);
    tn = within / gsz;
    tm = band * p.group_m + (within - tn * gsz);
  } else {
    tn = id % p.tiles_n;
    tm = id / p.tiles_n;
  }
  const int m0 = tm * T256, n0 = tn * T256;
  const int kbeg = z * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
  const int ntile = (kend - kbeg + BK256 - 1) / BK256;

  Operand<AM, false> opA;
  Operand<BMM, true> opB;
  opA.setup(reinterpret_cast<const bf16_t *>(p.A), p.lda, m0, p.M, FOLD ? 0 : kbeg, FOLD ? p.k_fold : kend, wave, lane, wr);
  opB.setup(reinterpret_cast<const bf16_t *>(p.B), p.ldb, n0, p.N, FOLD ? 0 : kbeg, FOLD ? p.k_fold : kend, wave, lane, wc);
  if constexpr (FOLD) {      // the descriptors reach to the end of the farthest segment
    const long long a_far = max(p.a_fold[0], max(p.a_fold[1], p.a_fold[2])), b_far = max(p.b_fold[0], max(p.b_fold[1], p.b_fold[2]));
    opA.nbytes = (int)min((long long)opA.nbytes + a_far * 2, 0x7fffffffLL);
    opB.nbytes = (int)min((long long)opB.nbytes + b_far * 2, 0x7fffffffLL);
  }

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(opA.pnl), 0, opA.nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(opB.pnl), 0, opB.nbytes, 0x00020000);
  // piece u of the A (which = 0) / B (which = 1) image of K tile kt.  (The int casts matter: with unsigned arguments the
  // builtin call fails to instantiate in the HOST pass of this template -- silently -- and no launch stub is emitted.)
  const int seg_tiles = FOLD ? p.k_fold / BK256 : 1;
  auto stage = [&](int kt, int which, int u) {
    int sa = (int)(kt * opA.tile_step), sb = (int)(kt * opB.tile_step);
    if constexpr (FOLD) {
      const int tg = kbeg / BK256 + kt;
      const int seg = (tg >= 2 * seg_tiles) ? 2 : (tg >= seg_tiles) ? 1 : 0;
      const int kk = tg - seg * seg_tiles;
      sa = (int)((seg == 0 ? p.a_fold[0] : seg == 1 ? p.a_fold[1] : p.a_fold[2]) * 2) + (int)(kk * opA.tile_step);
      sb = (int)((seg == 0 ? p.b_fold[0] : seg == 1 ? p.b_fold[1] : p.b_fold[2]) * 2) + (int)(kk * opB.tile_step);
    }
    if (which == 0) DM_LDS_DMA(rsA, smem + (kt & 1) * BUF_BYTES + opA.piece_offset(u, wave), (int)opA.vo[u], sa);
    else DM_LDS_DMA(rsB, smem + (kt & 1) * BUF_BYTES + OPER_BYTES + opB.piece_offset(u, wave), (int)opB.vo[u], sb);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 fa[8], fb0[4], fb1[4];
  // Column sums of A (the bias gradient that goes with a weight gradient): the workgroups of column tile 0 multiply their A
  // fragments with a ones fragment; the four waves that hold the same A rows take every fourth K tile each (+6 % MFMAs).
  const bool colsum = AM && p.colsum_slab != nullptr && tn == 0;
  f32x4 accb[AM ? 8 : 1];
#pragma unroll
  for (int i = 0; i < (AM ? 8 : 1); ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};     // eight bf16 1.0
#define DM_COLSUM(MI)                                                                      \
  do {                                                                                     \
    if constexpr (AM) {                                                                    \
      if (colsum && (kt & 3) == wc && (!FOLD || dm_fold_counts(p, kbeg + kt * BK256))) {   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                      \
            mma<bf16_t>(accb[(MI) * 4 + i], fa[2 * i + ks], ones);                         \
      }                                                                                    \
    }                                                                                      \
  } while (0)

#define DM_QUAD(MI, NI, FB)                                                                \
  do {                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
        mma<bf16_t>(acc[(MI) * 4 + i][(NI) * 2 + j], fa[2 * i + ks], FB[2 * j + ks]);      \
    __builtin_amdgcn_s_setprio(0);                                                         \
  } while (0)
#define DM_PHASE_SYNC()                         \
  do {                                          \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);          \
  } while (0)
#define DM_PHASE_END()                 \
  do {                                 \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();      \
  } while (0)

  // ---- prologue: all of tile 0, and of tile 1 everything except the A-sub1 pieces (staged in phase 0) ---------
#pragma unroll
  for (int u = 0; u < 4; ++u) { stage(0, 0, u); stage(0, 1, u); }
  if (ntile > 1) {
    stage(1, 0, 0); stage(1, 0, 2);      // A sub0 pieces
    stage(1, 1, 0); stage(1, 1, 1);      // B sub0
    stage(1, 1, 2); stage(1, 1, 3);      // B sub1
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // the second wave row runs one barrier behind the first

  for (int kt = 0; kt < ntile; ++kt) {
    const char *imgA = smem + (kt & 1) * BUF_BYTES;
    const char *imgB = imgA + OPER_BYTES;
    // phase 0: quadrant (0,0)
    opB.template load<2>(fb0, imgB, 0, wc);
    __builtin_amdgcn_sched_barrier(0);
    opA.template load<4>(fa, imgA, 0, wr);
    if (kt + 1 < ntile) { stage(kt + 1, 0, 1); stage(kt + 1, 0, 3); }
    DM_PHASE_SYNC();
    DM_QUAD(0, 0, fb0);
    DM_COLSUM(0);
    DM_PHASE_END();
    // phase 1: quadrant (0,1)
    opB.template load<2>(fb1, imgB, 1, wc);
    DM_PHASE_SYNC();
    DM_QUAD(0, 1, fb1);
    DM_PHASE_END();
    // phase 2: quadrant (1,1)
    opA.template load<4>(fa, imgA, 1, wr);
    if (kt + 2 < ntile) { stage(kt + 2, 0, 0); stage(kt + 2, 0, 2); stage(kt + 2, 1, 0); stage(kt + 2, 1, 1); }
    DM_PHASE_SYNC();
    DM_QUAD(1, 1, fb1);
    DM_COLSUM(1);
    DM_PHASE_END();
    // phase 3: quadrant (1,0); retire tile kt+1's DMA
    if (kt + 2 < ntile) {
      stage(kt + 2, 1, 2); stage(kt + 2, 1, 3);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    DM_PHASE_SYNC();
    DM_QUAD(1, 0, fb0);
    DM_PHASE_END();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // balance the stagger

  // ---- epilogue ----------------------------------------------------------------------------------------------
  if constexpr (AM) {
    if (colsum && g == 0) {      // every column of an accb tile holds the same sum: lanes g == 0 write element 0
      float *row = p.colsum_slab + (long long)(z * 4 + wc) * p.M;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + li;
        if (m < p.M) row[m] = accb[i][0];
      }
    }
  }
  // whole-line epilogue (dm_gemm_common.h): every wave transposes its 128 x 64 block through a private LDS region.  N % 8 != 0
  // (only possible for fp32 outputs here) keeps the 4-column form.
  const bool rows_ok = (p.N % 8 == 0) && (p.ldc % 8 == 0) && (p.aux == nullptr || p.ldaux % 8 == 0) && (p.rows_per_group == 0 || p.group_stride % 8 == 0);
  if (rows_ok) {
    __builtin_amdgcn_s_barrier();               // every wave is done reading the K-tile buffers
    char *mine = smem + wave * (64 * DM_EPI_PITCH);
    if (p.split_k > 1) {                        // K slice: fp32 partial tile into the slab, summed in slice order by splitk_reduce_kernel
      GemmParams q = p;
      q.C = p.workspace + (long long)z * p.M * p.N;
      q.ldc = p.N; q.c_dtype = DM_F32; q.bias = nullptr; q.residual = nullptr; q.aux = nullptr; q.epilogue = DM_EPI_NONE;
      q.accumulate = 0; q.rows_per_group = 0;
      dm_epilogue_rows<8, 64>(q, acc, mine, m0 + wr * 128, n0 + wc * 64, lane);
    } else {
      dm_epilogue_rows<8, 64>(p, acc, mine, m0 + wr * 128, n0 + wc * 64, lane);
    }
    return;
  }
  if (p.split_k > 1) {          // K slice: fp32 partial tile into the slab, summed in slice order by splitk_reduce_kernel
    float *W = p.workspace + (long long)z * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wr * 128 + i * 16 + li;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + 4 * g;
        if (n < p.N) dm_store4(W + (long long)m * p.N + n, acc[i][j]);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + li;
    if (m >= p.M) continue;
    const DmGemmRow rb = dm_gemm_row(p, m);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + 4 * g;
      if (n >= p.N) continue;
      dm_gemm_emit<true>(p, acc[i][j], rb, n);
    }
  }
}

// ---- persistent form (round 4): forward (NT) and dgrad (NN) products with several tiles per CU ------------------------------------
// One workgroup per CU walks its tiles (L, L + G, ...) as ONE flattened sequence of K tiles: the LDS-DMA of the next tile's first two K
// tiles is issued during the current tile's last two (same phases, same hazard table as above -- the buffer is the flattened index & 1),
// so only the first tile of a launch pays a cold fill; the epilogue runs between two K tiles through a wave-PRIVATE 4 KiB staging block
// outside the K-tile buffers (2 x 64 KiB + 8 x 4 KiB = exactly 160 KiB; 16 rows x 256 B, XOR-swizzled: dm_epi_slot) in the lean form
// of dm_gemm_common.h, without barriers, while those DMA pieces land.  Measured per tile before (one workgroup per tile): ~20 us of
// fill + drain + epilogue around 12 x 1.2 us of K loop at K = 768.
constexpr int EPI_P = 16 * 256;                          // one wave's staging block
constexpr int LDS256P = 2 * BUF_BYTES + 8 * EPI_P;       // 163840

template <int LAYOUT>
__global__ __launch_bounds__(512) void gemm256p_kernel(const GemmParams p) {
  static_assert(LAYOUT != DM_TN, "weight gradients keep the one-tile form (split-K)");
  constexpr bool BMM = (LAYOUT != DM_NT);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int G = gridDim.x;
  const int L = dm_xcd_remap(blockIdx.x, G);
  const int tiles = p.tiles_m * p.tiles_n;
  const int ntile = p.K / BK256;
  const int n_my = (tiles - L + G - 1) / G;
  const int total = n_my * ntile;
  auto tile_mn = [&](int r, int &m0, int &n0) {
    const int id = L + r * G;
    int tm, tn;
    if (p.group_m > 0) {
      const int band = id / (p.group_m * p.tiles_n);
      const int within = id - band * (p.group_m * p.tiles_n);
      const int gsz = min(p.group_m, p.tiles_m - band * p.group_m);
      tn = within / gsz;
      tm = band * p.group_m + (within - tn * gsz);
    } else {
      tn = id % p.tiles_n;
      tm = id / p.tiles_n;
    }
    m0 = tm * T256;
    n0 = tn * T256;
  };
  int m_cur, n_cur;
  tile_mn(0, m_cur, n_cur);
  const bf16_t *Ab = reinterpret_cast<const bf16_t *>(p.A), *Bb = reinterpret_cast<const bf16_t *>(p.B);
  Operand<false, false> opA;
  Operand<BMM, true> opB;
  opA.setup(Ab, p.lda, m_cur, p.M, 0, p.K, wave, lane, wr);
  opB.setup(Bb, p.ldb, n_cur, p.N, 0, p.K, wave, lane, wc);
  // descriptors of the tile being computed (c) and of the next one (n): staging looks at most two K tiles ahead
  __amdgpu_buffer_rsrc_t rsAc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(opA.pnl), 0, opA.nbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsBc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(opB.pnl), 0, opB.nbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsAn = rsAc, rsBn = rsBc;
  auto make_next = [&](int r) {
    if (r < n_my) {
      int m0, n0;
      tile_mn(r, m0, n0);
      opA.retile(Ab, p.lda, m0, p.M, p.K);
      opB.retile(Bb, p.ldb, n0, p.N, p.K);
      rsAn = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(opA.pnl), 0, opA.nbytes, 0x00020000);
      rsBn = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(opB.pnl), 0, opB.nbytes, 0x00020000);
    }
  };
  make_next(1);
  int kt = 0, r = 0, flat = 0;
  // piece u of the A (which = 0) / B (which = 1) image of the K tile d steps ahead of the current one (d = 0 only in the prologue)
  auto stage = [&](int d, int which, int u) {
    int kk = kt + d;
    const bool nx = kk >= ntile;
    if (nx) kk -= ntile;
    char *buf = smem + ((flat + d) & 1) * BUF_BYTES;
    if (which == 0) {
      if (nx) DM_LDS_DMA(rsAn, buf + opA.piece_offset(u, wave), (int)opA.vo[u], (int)(kk * opA.tile_step));
      else DM_LDS_DMA(rsAc, buf + opA.piece_offset(u, wave), (int)opA.vo[u], (int)(kk * opA.tile_step));
    } else {
      if (nx) DM_LDS_DMA(rsBn, buf + OPER_BYTES + opB.piece_offset(u, wave), (int)opB.vo[u], (int)(kk * opB.tile_step));
      else DM_LDS_DMA(rsBc, buf + OPER_BYTES + opB.piece_offset(u, wave), (int)opB.vo[u], (int)(kk * opB.tile_step));
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 fa[8], fb0[4], fb1[4];
  constexpr bool AM = false;       // (the macros below were written for the one-tile kernel)
  const bool colsum = false;
  f32x4 accb[1];
  const u32x4 ones = {0, 0, 0, 0};
  (void)accb; (void)ones; (void)colsum;

  // ---- prologue: all of K tile 0, and of K tile 1 everything except the A-sub1 pieces (staged in phase 0) ------------------------
#pragma unroll
  for (int u = 0; u < 4; ++u) { stage(0, 0, u); stage(0, 1, u); }
  if (total > 1) {
    stage(1, 0, 0); stage(1, 0, 2);
    stage(1, 1, 0); stage(1, 1, 1);
    stage(1, 1, 2); stage(1, 1, 3);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // the second wave row runs one barrier behind the first

  // (two nested loops on purpose: with the epilogue inside ONE flattened loop the compiler's waitcnt bookkeeping puts a vmcnt(2) for the
  // epilogue's load destinations at the loop head -- every K tile then waits for the DMA pieces issued one phase earlier)
  for (; r < n_my;) {
  for (kt = 0; kt < ntile; ++kt, ++flat) {
    const char *imgA = smem + (flat & 1) * BUF_BYTES;
    const char *imgB = imgA + OPER_BYTES;
    // phase 0: quadrant (0,0)
    opB.template load<2>(fb0, imgB, 0, wc);
    __builtin_amdgcn_sched_barrier(0);
    opA.template load<4>(fa, imgA, 0, wr);
    if (flat + 1 < total) { stage(1, 0, 1); stage(1, 0, 3); }
    DM_PHASE_SYNC();
    DM_QUAD(0, 0, fb0);
    DM_PHASE_END();
    // phase 1: quadrant (0,1)
    opB.template load<2>(fb1, imgB, 1, wc);
    DM_PHASE_SYNC();
    DM_QUAD(0, 1, fb1);
    DM_PHASE_END();
    // phase 2: quadrant (1,1)
    opA.template load<4>(fa, imgA, 1, wr);
    if (flat + 2 < total) { stage(2, 0, 0); stage(2, 0, 2); stage(2, 1, 0); stage(2, 1, 1); }
    DM_PHASE_SYNC();
    DM_QUAD(1, 1, fb1);
    DM_PHASE_END();
    // phase 3: quadrant (1,0); retire the next K tile's DMA
    if (flat + 2 < total) {
      stage(2, 1, 2); stage(2, 1, 3);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    DM_PHASE_SYNC();
    DM_QUAD(1, 0, fb0);
    DM_PHASE_END();

  }
    {
      // tile finished: epilogue through this wave's private block (no barrier: the K-tile buffers are not touched and the next
      // tile's first two K tiles keep landing); the stores are waited for by the counted vmcnt of the next K tile's phase 3
      kt = ntile;         // (stage() is not called here; keeps the cursor meaningful for readers)
      dm_epilogue_rows<8, 16, false, 256, true, true>(p, acc, smem + 2 * BUF_BYTES + wave * EPI_P, m_cur + wr * 128, n_cur + wc * 64, lane);
      // A wait the COMPILER sees (the builtin, not asm): every specialised item structure issues at least two stores behind its last
      // load, so vmcnt(2) proves all VGPR-destination loads of the epilogue complete on every path.  Without it the waitcnt pass
      // assumes they may be pending at the K loop's head and puts a vmcnt(2) THERE -- executed every K tile, against the DMA queue.
      __builtin_amdgcn_s_waitcnt(0x0F72);      // vmcnt(2) expcnt(7) lgkmcnt(15)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ++r;
      if (r < n_my) tile_mn(r, m_cur, n_cur);
      rsAc = rsAn;
      rsBc = rsBn;
      make_next(r + 1);
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // balance the stagger
}

}  // namespace dm256

namespace {
template <int LAYOUT> bool set_lds_limit_p() {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(dm256::gemm256p_kernel<LAYOUT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             dm256::LDS256P) == hipSuccess;
}
int p256_cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    const char *e = getenv("DM_GEMM_CUS_RESERVED");
    const int rsv = e ? atoi(e) : 0;
    return (rsv > 0 && rsv < cus) ? cus - rsv : cus;
  }();
  return n;
}
template <int LAYOUT, bool FOLD = false> bool set_lds_limit() {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(dm256::gemm256_kernel<LAYOUT, FOLD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             dm256::LDS256) == hipSuccess;
}
}  // namespace
using namespace dm256;

// Decides whether the 256x256 pipeline runs this product and, if so, fills p.tiles_m / tiles_n / split_k / k_per_split.
// Called by dm_gemm after argument validation (alignment, N % 4, ...).
bool dm_gemm256_plan(GemmParams &p, int layout, int ab_dtype, bool can_split, long long workspace_bytes, int user_split) {
  const char *menv = getenv("DM_GEMM_256");              // 0 = off, 2 = always (A/B runs); read per call: tests flip it
  const int mode = menv ? atoi(menv) : 1;
  if (mode == 0 || ab_dtype != DM_BF16) return false;
  if (p.K < BK256) return false;
  if (p.k_fold > 0 && p.k_fold % BK256 != 0) return false;              // folded contraction: segments of whole K tiles
  const bool am = layout == DM_TN, bm = layout != DM_NT;
  if ((!am || !bm) && p.K % BK256 != 0) return false;                    // the K tail of k-contiguous rows is not masked
  if (am && p.M % 8 != 0) return false;                                  // m-contiguous rows are fetched in 16-byte chunks
  if (bm && p.N % 8 != 0) return false;
  // 32-bit DMA offsets: a 256-row panel (k-contiguous) or the whole K extent (m-contiguous) of an operand
  const long long spanA = am ? (long long)p.K * p.lda * 2 : 256LL * p.lda * 2;
  const long long spanB = bm ? (long long)p.K * p.ldb * 2 : 256LL * p.ldb * 2;
  if (spanA >= (1LL << 31) || spanB >= (1LL << 31)) return false;
  const int tiles_m = (p.M + T256 - 1) / T256, tiles_n = (p.N + T256 - 1) / T256;
  const long long tiles = (long long)tiles_m * tiles_n;
  int split = 1;
  if (layout == DM_TN && can_split && user_split != 1) {
    // one workgroup per CU: pick the slice count whose tiles * slices fills whole rounds of the 256 CUs best (fewest slices
    // among the near-best), with >= 8 K tiles per slice and the slab inside the caller's workspace
    double best = 0.0;
    for (int sp = 1; sp <= 32; ++sp) {
      if (sp > 1 && ((long long)p.K / sp < 8 * BK256 || (long long)sp * p.M * p.N * 4 > workspace_bytes)) break;
      const long long wg = tiles * sp;
      const double eff = (double)wg / (double)(((wg + 255) / 256) * 256);
      if (eff > best + 0.04) { best = eff; split = sp; }
    }
    if (user_split > 1) split = user_split;
  }
  const long long wgs = tiles * split;
  // Measured inside the encoder's step (activations are L2-cold there, unlike a warm microbenchmark): with one workgroup
  // per CU the ~1.5 K tiles of DMA in flight do not cover a first-touch miss, and the 256x256 epilogue cannot overlap
  // another workgroup's main loop.  The pipeline wins when every A panel is re-used by many column tiles (fc1: N = 3072,
  // +15 %) or when there are several rounds of tiles; narrow forward products (N = 768 / 2304 at M = 16384) stay on the
  // 128x128 kernel, whose 12 waves per CU hide the misses (-20 % otherwise).  mode 2 forces the pipeline (benchmarks).
  // Cold-operand microbenchmark (tools/mb_cold.py, 8 rotating operand sets): NT 98 us vs 113 us (128x128), TN 107 vs 117,
  // NN 107 vs 103 -- the m-contiguous images lose most of their warm-cache lead, so dgrad only takes the pipeline when
  // there are many rounds of tiles.
  bool take;
  static const int tn_min_tiles = [] { const char *e = getenv("DM_GEMM_256_TN_MINK"); return e ? atoi(e) : 8; }();   // K tiles per slice (8: the 4096-token stage's wgrads gain 12-30 % in the step; 16 left them on the 64x64 kernel)
  if (layout == DM_TN) take = wgs >= 200 && (long long)p.K / split >= (long long)tn_min_tiles * BK256;
  else if (layout == DM_NN) {
    // folded products (K = 3 x the layer's width: a tile's fill and epilogue weigh a third of what they do in bf16 mode): the dgrad of
    // fc2 (768 tiles, plane-pair result x saved GELU') 285 us here against 305 on 128 x 128 tiles (tools/mb_fold.py); DM_GEMM_FOLD_ROUTES=0: off
    static const bool fold_routes = [] { const char *e = getenv("DM_GEMM_FOLD_ROUTES"); return !(e && atoi(e) == 0); }();
    take = tiles >= 1024 || (fold_routes && p.k_fold > 0 && p.K >= 2048 && tiles >= 512);
  }
  else {
    // long-K forward products (fc2: K = 3072, 192 tiles) amortise the fill / epilogue: -0.03 ms/step measured
    static const int longk = [] { const char *e = getenv("DM_GEMM_256_NT_LONGK"); return e ? atoi(e) : 1; }();
    take = (tiles_n >= 10 && tiles >= 384) || tiles >= 1024 || (longk && p.K >= 2048 && tiles >= 180);
    // since the 128x128 kernel has the whole-line epilogue too it is ahead on the short-K wide products whose 128x128 tiles make whole
    // rounds of 3 workgroups per CU (fc1 forward in the step: 122 us here, 116 us there)
    if (p.K < 2048 && ((long long)((p.M + 127) / 128) * ((p.N + 127) / 128)) % 768 == 0) take = false;
  }
  if (mode == 2) take = true;
  if (!take) return false;
  static const bool attr_ok = set_lds_limit<DM_NT>() && set_lds_limit<DM_NN>() && set_lds_limit<DM_TN>() && set_lds_limit<DM_NT, true>() &&
                              set_lds_limit<DM_NN, true>() && set_lds_limit<DM_TN, true>();
  if (!attr_ok) return false;
  int kps = (int)((((long long)p.K + split - 1) / split + BK256 - 1) / BK256 * BK256);
  split = (p.K + kps - 1) / kps;
  p.tiles_m = tiles_m; p.tiles_n = tiles_n; p.split_k = split; p.k_per_split = kps;
  return true;
}

void dm_gemm256_launch(const GemmParams &p_in, int layout, hipStream_t s) {
  GemmParams p = p_in;
  static const int gm = [] { const char *e = getenv("DM_GEMM_256_GROUP_M"); return e ? atoi(e) : -1; }();
  if (gm >= 0) p.group_m = gm;
  {
    // persistent form: forward / dgrad, more tiles than CUs, the whole-line 8-column epilogue legal, no ragged N.  OFF by default
    // (DM_GEMM_256P=1: where tiles > CUs; 2: every legal launch; read per call so tests can flip it).  Measured (tools/mb_epi.py w4set,
    // cold operands, same box, persistent / one tile per workgroup): 16384 x 3072 x 768 plain 88-90 / 88-90 us, + GELU' 136-138 / 115-116,
    // dgrad x GELU' 132 / 117, qkv + bias 78-80 / 73-76.  Staging the next tile's K tiles under the current tile's tail does NOT
    // recover the ~15 us a tile spends outside its K loop: that time is the stores (128 KiB per tile at ~3.5 B/clk per CU), which only
    // overlap with matrix work when accumulators are double-buffered; and the 16-row staging block makes the epilogue itself slower.
    const char *penv = getenv("DM_GEMM_256P");
    const int pmode = penv ? atoi(penv) : 0;
    const int cus = p256_cu_count();
    const long long tiles = (long long)p.tiles_m * p.tiles_n;
    const bool rows_ok = (p.N % 8 == 0) && (p.ldc % 8 == 0) && (p.aux == nullptr || p.ldaux % 8 == 0) && (p.rows_per_group == 0 || p.group_stride % 8 == 0) &&
                         (p.residual == nullptr || p.ldr % 8 == 0);
    // (the persistent kernel has no folded-contraction / plane-pair forms: such a launch stays on gemm256_kernel<.., true>)
    if (pmode != 0 && layout != DM_TN && p.k_fold == 0 && p.c_dtype != DM_BF16_PAIR && p.split_k <= 1 && cus > 0 && (tiles > cus || pmode == 2) && p.N % 256 == 0 && p.K % BK256 == 0 && rows_ok &&
        dm_epi_key_specialised(dm_epi_lean_key(p, 128))) {
      static const bool attr_ok = set_lds_limit_p<DM_NT>() && set_lds_limit_p<DM_NN>();
      if (attr_ok) {
        const dim3 pg((unsigned)(tiles < cus ? tiles : cus));
        if (layout == DM_NT) hipLaunchKernelGGL(dm256::gemm256p_kernel<DM_NT>, pg, dim3(512), LDS256P, s, p);
        else hipLaunchKernelGGL(dm256::gemm256p_kernel<DM_NN>, pg, dim3(512), LDS256P, s, p);
        return;
      }
    }
  }
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n * p.split_k));
  if (p.k_fold > 0) {
    switch (layout) {
      case DM_NT: hipLaunchKernelGGL((dm256::gemm256_kernel<DM_NT, true>), grid, dim3(512), LDS256, s, p); break;
      case DM_NN: hipLaunchKernelGGL((dm256::gemm256_kernel<DM_NN, true>), grid, dim3(512), LDS256, s, p); break;
      default: hipLaunchKernelGGL((dm256::gemm256_kernel<DM_TN, true>), grid, dim3(512), LDS256, s, p); break;
    }
    return;
  }
  switch (layout) {
    case DM_NT: hipLaunchKernelGGL(dm256::gemm256_kernel<DM_NT>, grid, dim3(512), LDS256, s, p); break;
    case DM_NN: hipLaunchKernelGGL(dm256::gemm256_kernel<DM_NN>, grid, dim3(512), LDS256, s, p); break;
    default: hipLaunchKernelGGL(dm256::gemm256_kernel<DM_TN>, grid, dim3(512), LDS256, s, p); break;
  }
}
