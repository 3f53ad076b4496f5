// Persistent 256x192x64 bf16 GEMM pipeline (gfx950) for the forward (NT) and dgrad (NN) products of the 16384-token stage.
//
// Why a third LDS-DMA kernel.  The 256x256 pipeline (dm_gemm256.hip) reaches ~72 % MFMA issue in its K loop, but one launch of it
// costs ~20 us more than its K loops: every workgroup of the chip fills its pipeline at the same time (a 33 MB burst from
// HBM / the Infinity Cache), computes in lockstep, and stores its 256x256 block at the same time, so the memory system idles
// during the loops and the matrix pipes idle during the bursts -- once per round of tiles.  The vendor library (hipBLASLt,
// tools/mb_yardstick.py) runs these shapes 1.2-1.4x faster with persistent workgroups.  This kernel removes the per-tile
// fill / drain:
//   * one workgroup per CU, grid = min(tiles, CUs); a workgroup walks its tiles (L, L + grid, ...) as ONE flattened sequence
//     of K steps: the LDS-DMA for the next tile's first K steps is issued during the current tile's last ones, so only the
//     first tile of a launch pays a cold fill, and a tile's stores drain under the next tile's K loop;
//   * tile 256 x 192: N = 768 / 2304 / 3072 are multiples of 192, so M = 16384 gives 256 / 768 / 1024 tiles = whole rounds of
//     the 256 CUs (256x256 tiles give 192 / 576 / 768: 0.75 and 2.25 rounds);
//   * two K-step buffers of 56 KiB (A 256 x 128 B, B 192 x 128 B) + a 48 KiB epilogue staging region that is NOT shared with
//     them (the buffers hold the next tile's operands while a tile is stored) = exactly the 160 KiB of a CU;
//   * 8 waves = 4 (M) x 2 (N), 64 x 96 outputs per wave (24 accumulator tiles); a K step is three phases, one per 32-column
//     third of the wave's columns (16 MFMAs each); as in dm_gemm256.hip the two wave columns run one barrier apart, so on
//     every SIMD one wave issues MFMAs while its partner reads LDS / issues DMA, and DMA completion is a counted vmcnt.
//
// Hazard bookkeeping (step = one K step of the flattened sequence, buffer = step & 1; piece = 64 image rows = one DMA
// instruction per wave; A image = pieces A0..A3 (wave row wm reads piece wm), B image = pieces B0..B2 = the three thirds):
//   reads : p0 A (all) + B0, p1 B1, p2 B2
//   stage : p0 B1 of step+1, p1 B2 of step+1, p2 A0..A3 + B0 of step+2   (every piece >= 2 phases after its last read)
//   wait  : before each phase's first barrier, vmcnt(7): the 7 newest DMA instructions are younger than the piece the NEXT
//           phase reads (see the issue order above); fewer at the end of the sequence.
// Stores count in vmcnt on gfx9 and may retire out of order with loads; a counted wait is then conservative (loads retire
// in order among themselves), it can only wait longer.
#include <cstdlib>

#include "dm_common.h"
#include "dm_gemm_common.h"
#include "dm_mfma.h"

namespace dmp192 {

constexpr int TM = 256, TN = 192, BK = 64;
constexpr int A_BYTES = TM * 128;              // 32 KiB
constexpr int B_BYTES = TN * 128;              // 24 KiB
constexpr int BUF_BYTES = A_BYTES + B_BYTES;   // 56 KiB
constexpr int EPI_WAVE = 16 * 96 * 4;          // 16 rows x 96 fp32, XOR-swizzled (no padding)
constexpr int LDS_BYTES = 2 * BUF_BYTES + 8 * EPI_WAVE;   // 163840 = 160 KiB

#define DMP_LDS_DMA(rsrc, dst, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst), 16, voff, soff, 0, 0)

// DBG (ablation builds only, -DDM_P192_ABLATE): 4 no MFMA, 8 no DMA inside the loop, 16 no fragment reads
template <int LAYOUT, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_p192_kernel(const GemmParams p) {
  constexpr bool BMM = (LAYOUT == DM_NN);      // B m-contiguous [K][N] (dgrad) or k-contiguous [N][K] (forward)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 3, wn = wave >> 2;     // wave row (64 rows each), wave column (96 columns each) = stagger group
  const int g = lane >> 4, li = lane & 15;

  const int G = gridDim.x;
  const int L = dm_xcd_remap(blockIdx.x, G);
  const int tiles = p.tiles_m * p.tiles_n;
  const int ntile = p.K / BK;
  const int n_my = (tiles - L + G - 1) / G;    // tiles L, L + G, ...
  const int total = n_my * ntile;

  // ---- per-lane DMA source offsets -----------------------------------------------------------------------------------------
  // k-contiguous image: a wave-instruction fills 8 image rows x 128 B; slot s of image row r holds the operand's 16-byte chunk
  // s ^ (r & 7).  B image rows are ordered [third][wave column][32]: image row rho <-> tile column (rho>>5&1)*96 + (rho>>6)*32 + (rho&31).
  const int srow = 8 * wave + (lane >> 3);
  const int chunk = (lane & 7) ^ (lane >> 3);
  unsigned voA[4], voB[3];
#pragma unroll
  for (int u = 0; u < 4; ++u) voA[u] = (unsigned)(((long long)(64 * u + srow) * p.lda) * 2 + chunk * 16);
  if constexpr (!BMM) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int col = (srow >> 5) * 96 + u * 32 + (srow & 31);
      voB[u] = (unsigned)(((long long)col * p.ldb) * 2 + chunk * 16);
    }
  } else {
    // m-contiguous image: 3 bands of [64 k-rows][64 columns = 128 B]; a wave-instruction fills k-rows 8w..8w+7 of one band; the
    // 32-byte slot index of k-row r is XORed with f(r) = ((r >> 1) & 1) | (((r >> 3) & 1) << 1)   (as in dm_gemm256.hip)
    const int fk = ((srow >> 1) & 1) | (((srow >> 3) & 1) << 1);
    const int csrc = (lane & 7) ^ (fk << 1);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int col = (csrc >> 2) * 96 + u * 32 + (csrc & 3) * 8;
      voB[u] = (unsigned)(((long long)srow * p.ldb + col) * 2);
    }
  }
  const unsigned stepA = BK * 2;
  const unsigned stepB = BMM ? (unsigned)(BK * p.ldb * 2) : (unsigned)(BK * 2);
  const int pieceA = (8 * wave) * 128;                 // + 64 * u * 128
  const int pieceB = BMM ? wave * 1024 : (8 * wave) * 128;   // + u * 8192

  // ---- per-lane fragment offsets -----------------------------------------------------------------------------------------------
  const int sw0 = (g ^ (li & 7)) << 4, sw1 = ((4 + g) ^ (li & 7)) << 4;
  const int fragA0 = (wm * 64 + li) * 128 + sw0, fragA1 = (wm * 64 + li) * 128 + sw1;
  int fragB0, fragB1;
  if constexpr (!BMM) {
    fragB0 = (wn * 32 + li) * 128 + sw0;
    fragB1 = (wn * 32 + li) * 128 + sw1;
  } else {
    const int q = li >> 2, pq = li & 3;
    const int rbase = (8 * g + q) * 128 + 8 * pq;
    const int fr = ((q >> 1) & 1) | ((g & 1) << 1);
    fragB0 = rbase + (((wn * 2 + 0) ^ fr) << 5);
    fragB1 = rbase + (((wn * 2 + 1) ^ fr) << 5);
  }

  // ---- tile cursors (uniform) --------------------------------------------------------------------------------------------------
  const bf16_t *Ab = reinterpret_cast<const bf16_t *>(p.A), *Bb = reinterpret_cast<const bf16_t *>(p.B);
  auto tile_mn = [&](int r, int &m0, int &n0) {
    const int tid = L + r * G;
    const int tm = tid / p.tiles_n;
    m0 = tm * TM;
    n0 = (tid - tm * p.tiles_n) * TN;
  };
  auto make_a = [&](int m0) {
    const long long bytes = ((long long)(min(TM, p.M - m0) - 1) * p.lda + p.K) * 2;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(Ab + (long long)m0 * p.lda), 0, (int)min(bytes, 0x7fffffffLL), 0x00020000);
  };
  auto make_b = [&](int n0) {
    if constexpr (!BMM) {
      const long long bytes = ((long long)(min(TN, p.N - n0) - 1) * p.ldb + p.K) * 2;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(Bb + (long long)n0 * p.ldb), 0, (int)min(bytes, 0x7fffffffLL), 0x00020000);
    } else {
      const long long bytes = ((long long)(p.K - 1) * p.ldb + (p.N - n0)) * 2;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(Bb + n0), 0, (int)min(bytes, 0x7fffffffLL), 0x00020000);
    }
  };
  int m_cur, n_cur;
  tile_mn(0, m_cur, n_cur);
  // cursor 1 = step + 1 (stages B1, B2), cursor 2 = step + 2 (stages A, B0)
  int r1 = 0, k1 = 0, r2 = 0, k2 = 0;
  __amdgpu_buffer_rsrc_t rsB1 = make_b(n_cur), rsA2 = make_a(m_cur), rsB2 = rsB1;
  auto advance1 = [&]() {
    if (++k1 == ntile) {
      k1 = 0; ++r1;
      if (r1 < n_my) { int m0, n0; tile_mn(r1, m0, n0); rsB1 = make_b(n0); }
    }
  };
  auto advance2 = [&]() {
    if (++k2 == ntile) {
      k2 = 0; ++r2;
      if (r2 < n_my) { int m0, n0; tile_mn(r2, m0, n0); rsA2 = make_a(m0); rsB2 = make_b(n0); }
    }
  };
  // (the int casts matter: with unsigned arguments the builtin fails to instantiate in the host pass of a template, silently)
  auto stage_b1 = [&](int step, int u) {       // piece u (1 or 2) of cursor 1 into buffer step & 1
    DMP_LDS_DMA(rsB1, smem + (step & 1) * BUF_BYTES + A_BYTES + u * 8192 + pieceB, (int)voB[u], (int)(k1 * stepB));
  };
  auto stage_ab2 = [&](int step) {             // A0..A3 + B0 of cursor 2 into buffer step & 1
    char *buf = smem + (step & 1) * BUF_BYTES;
#pragma unroll
    for (int u = 0; u < 4; ++u) DMP_LDS_DMA(rsA2, buf + u * 8192 + pieceA, (int)voA[u], (int)(k2 * stepA));
    DMP_LDS_DMA(rsB2, buf + A_BYTES + pieceB, (int)voB[0], (int)(k2 * stepB));
  };

  f32x4 acc[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 fa[8], fb[4];
  if constexpr (DBG & 16) {
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = (u32x4){(unsigned)lane, 1u, 2u, 3u};
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = (u32x4){(unsigned)lane, 1u, 2u, 3u};
  }

  auto load_a = [&](const char *img) {
    if constexpr (DBG & 16) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[2 * i] = *reinterpret_cast<const u32x4 *>(img + i * 2048 + fragA0);
      fa[2 * i + 1] = *reinterpret_cast<const u32x4 *>(img + i * 2048 + fragA1);
    }
  };
  auto load_b = [&](const char *imgB, int third) {
    if constexpr (DBG & 16) return;
    const char *b = imgB + third * 8192;
    if constexpr (!BMM) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fb[2 * j] = *reinterpret_cast<const u32x4 *>(b + j * 2048 + fragB0);
        fb[2 * j + 1] = *reinterpret_cast<const u32x4 *>(b + j * 2048 + fragB1);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int f = j ? fragB1 : fragB0;
          const u32x2 lo = dm_ds_read_tr16(b + f + (32 * ks) * 128);
          const u32x2 hi = dm_ds_read_tr16(b + f + (32 * ks + 4) * 128);
          fb[2 * j + ks] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    }
  };

#define DMP_MMA(THIRD)                                                                      \
  do {                                                                                      \
    if constexpr (!(DBG & 64)) __builtin_amdgcn_s_setprio(1);                               \
    if constexpr (!(DBG & 4))                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
        mma<bf16_t>(acc[i][(THIRD) * 2 + j], fa[2 * i + ks], fb[2 * j + ks]);               \
    if constexpr (!(DBG & 64)) __builtin_amdgcn_s_setprio(0);                               \
  } while (0)
#define DMP_SYNC()                                                              \
  do {                                                                          \
    if constexpr (DBG & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    __builtin_amdgcn_s_barrier();                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                          \
  } while (0)
#define DMP_END()                      \
  do {                                 \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();      \
  } while (0)

  // ---- prologue: step 0 entirely, A + B0 of step 1 (issue order = the steady-state order) -------------------------------------
  stage_ab2(0);
  advance2();
  {  // B1, B2 of step 0 through cursor 1 (k1 = 0)
    stage_b1(0, 1);
    stage_b1(0, 2);
    advance1();
  }
  if (total > 1) {
    stage_ab2(1);
    advance2();
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");     // A + B0 of step 0 have landed
  } else {
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wn == 1) __builtin_amdgcn_s_barrier();             // the second wave column runs one barrier behind the first

  int kt = 0, r = 0;
  for (int step = 0; step < total; ++step) {
    const char *imgA = smem + (step & 1) * BUF_BYTES;
    const char *imgB = imgA + A_BYTES;
    const bool e1 = !(DBG & 8) && step + 1 < total, e2 = !(DBG & 8) && step + 2 < total;
    // phase 0: columns 0..31 of the wave; stage B1 of step + 1; B1 of this step must land before the next phase
    load_b(imgB, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_a(imgA);
    if (e1) {
      stage_b1(step + 1, 1);
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    }
    DMP_SYNC();
    DMP_MMA(0);
    DMP_END();
    // phase 1: columns 32..63; stage B2 of step + 1; B2 of this step must land
    load_b(imgB, 1);
    if (e1) {
      stage_b1(step + 1, 2);
      advance1();
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    DMP_SYNC();
    DMP_MMA(1);
    DMP_END();
    // phase 2: columns 64..95; stage A + B0 of step + 2; A + B0 of step + 1 must land
    load_b(imgB, 2);
    if (e2) {
      stage_ab2(step);
      advance2();
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    DMP_SYNC();
    DMP_MMA(2);
    DMP_END();

    if (++kt == ntile) {
      // ---- tile finished: fused epilogue through this wave's private staging rows (no barrier: the K-step buffers are not
      // touched, the DMA of the next tile keeps landing) ---------------------------------------------------------------------
      char *mine = smem + 2 * BUF_BYTES + wave * EPI_WAVE;
      const int m_wave = m_cur + wm * 64, n_wave = n_cur + wn * 96;
#ifdef DM_P192_ABLATE
      const bool no_epi = p.debug & 1, no_emit = p.debug & 2;
#else
      constexpr bool no_epi = false, no_emit = false;
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (no_epi) {
#pragma unroll
          for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          continue;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
          *reinterpret_cast<f32x4 *>(mine + li * 384 + (((j * 4 + g) ^ (li & 7)) << 4)) = acc[i][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3) {
          const int item = q3 * 64 + lane;           // 16 rows x 12 groups of 8 columns
          const int row = item / 12, cg = item - row * 12;
          const f32x4 lo = *reinterpret_cast<const f32x4 *>(mine + row * 384 + (((2 * cg) ^ (row & 7)) << 4));
          const f32x4 hi = *reinterpret_cast<const f32x4 *>(mine + row * 384 + (((2 * cg + 1) ^ (row & 7)) << 4));
          const int m = m_wave + i * 16 + row;
          if (m < p.M && !(no_emit && lo[0] != 12345.678f)) dm_gemm_emit8(p, lo, hi, dm_gemm_row(p, m), n_wave + cg * 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next row tile overwrites the region
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      kt = 0;
      ++r;
      if (r < n_my) tile_mn(r, m_cur, n_cur);
    }
  }
  if (wn == 0) __builtin_amdgcn_s_barrier();   // balance the stagger
}

}  // namespace dmp192

namespace {
template <int LAYOUT, int DBG = 0> bool p192_set_lds_limit() {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(dmp192::gemm_p192_kernel<LAYOUT, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             dmp192::LDS_BYTES) == hipSuccess;
}
int p192_cu_count() {
  // DM_GEMM_CUS_RESERVED = n plans the one-workgroup-per-CU grids for n CUs fewer than the device has: a collective running next to
  // the backward pass (RCCL kernels hold CUs for the length of an all-reduce) otherwise pushes the last workgroups of such a grid
  // into a second round.  Default 0 -- to be tuned on a multi-GPU node, none was available to this build.
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    const char *e = getenv("DM_GEMM_CUS_RESERVED");
    const int r = e ? atoi(e) : 0;
    return (r > 0 && r < cus) ? cus - r : cus;
  }();
  return n;
}
}  // namespace

// Decides whether the persistent 256x192 pipeline runs this product (bf16 NT / NN); fills p.tiles_m / tiles_n and returns the grid
// size (0 = not taken).  `aligned8`: the 8-column epilogue (dm_gemm_emit8) is legal for C / aux / grouped rows.
int dm_gemm_p192_plan(GemmParams &p, int layout, int ab_dtype, bool aligned8) {
  using namespace dmp192;
  const char *env = getenv("DM_GEMM_P192");       // 0 = off, 1 = routing rule, 2 = every legal product (read per call: tests flip it)
  const int mode = env ? atoi(env) : 0;      // off by default: dm_gemm_w4.hip supersedes it (kept for the ablation record)
  if (mode == 0 || ab_dtype != DM_BF16 || !aligned8) return 0;
  if (layout != DM_NT && layout != DM_NN) return 0;
  if (p.K < 2 * BK || p.K % BK != 0 || p.N % TN != 0) return 0;
  const long long spanA = 256LL * p.lda * 2;
  const long long spanB = (layout == DM_NN) ? (long long)p.K * p.ldb * 2 : 192LL * p.ldb * 2;
  if (spanA >= (1LL << 31) || spanB >= (1LL << 31)) return 0;
  const int tiles_m = (p.M + TM - 1) / TM, tiles_n = p.N / TN;
  const long long tiles = (long long)tiles_m * tiles_n;
  const int cus = p192_cu_count();
  if (cus <= 0) return 0;
  // one workgroup per CU: worth it when the tiles fill (nearly) whole rounds of the chip
  if (mode != 2) {
    if (tiles < cus) return 0;
    const long long rounds = (tiles + cus - 1) / cus;
    if ((double)tiles / (double)(rounds * cus) < 0.85) return 0;
  }
  static const bool attr_ok = p192_set_lds_limit<DM_NT>() && p192_set_lds_limit<DM_NN>();
  if (!attr_ok) return 0;
  p.tiles_m = tiles_m;
  p.tiles_n = tiles_n;
  p.split_k = 1;
  p.k_per_split = p.K;
  {
    const char *denv = getenv("DM_P192_DEBUG");        // ablations (-DDM_P192_ABLATE builds): 1 no epilogue, 2 epilogue without stores
    p.debug = denv ? atoi(denv) : 0;
  }
  return (int)(tiles < cus ? tiles : cus);
}

void dm_gemm_p192_launch(const GemmParams &p, int layout, int grid, hipStream_t s) {
#ifdef DM_P192_ABLATE
  if (layout == DM_NT && (p.debug & ~3)) {
    static const bool ok = p192_set_lds_limit<DM_NT, 4>() && p192_set_lds_limit<DM_NT, 8>() && p192_set_lds_limit<DM_NT, 16>() &&
                           p192_set_lds_limit<DM_NT, 32>() && p192_set_lds_limit<DM_NT, 64>() && p192_set_lds_limit<DM_NT, 96>() && p192_set_lds_limit<DM_NT, 40>() &&
                           p192_set_lds_limit<DM_NT, 12>() && p192_set_lds_limit<DM_NT, 20>() && p192_set_lds_limit<DM_NT, 24>() && p192_set_lds_limit<DM_NT, 28>();
    (void)ok;
    switch (p.debug & ~3) {
      case 4: hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 4>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p); return;
      case 8: hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 8>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p); return;
      case 16: hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 16>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p); return;
      case 12: hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 12>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p); return;
      case 20: hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 20>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p); return;
      case 24: hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 24>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p); return;
      case 32: hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 32>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p); return;
      case 64: hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 64>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p); return;
      case 96: hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 96>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p); return;
      case 40: hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 40>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p); return;
      case 28: hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 28>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p); return;
      default: break;
    }
  }
#endif
  if (layout == DM_NT) hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NT, 0>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p);
  else hipLaunchKernelGGL((dmp192::gemm_p192_kernel<DM_NN, 0>), dim3(grid), dim3(512), dmp192::LDS_BYTES, s, p);
}
