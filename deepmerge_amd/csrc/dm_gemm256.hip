// 256x256x64 bf16 GEMM for the large forward products (NT: y = x W^T, both operands k-contiguous), gfx950.
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

namespace {

constexpr int T256 = 256;                 // workgroup tile (rows and columns)
constexpr int BK256 = 64;                 // K per tile
constexpr int OPER_BYTES = T256 * 128;    // one operand's image of a K tile: 256 rows x 128 B
constexpr int BUF_BYTES = 2 * OPER_BYTES; // A image + B image
constexpr int LDS256 = 2 * BUF_BYTES;     // two K-tile buffers = 128 KiB

// image row of the B operand -> column of the tile.  Image rows are ordered [sub][wave column][32] so that the
// columns the waves read in the same phase are contiguous pieces (see the hazard table above).
__device__ __forceinline__ int b_image_to_col(int rho) { return ((rho >> 5) & 3) * 64 + (rho >> 7) * 32 + (rho & 31); }

#define DM_LDS_DMA(rsrc, dst, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst), 16, voff, soff, 0, 0)

template <bool FAST>
__global__ __launch_bounds__(512) void gemm256_nt_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, li = lane & 15;

  // ---- tile of this workgroup (XCD-contiguous ids, rows fastest inside bands of group_m row tiles) ----------
  int id = dm_xcd_remap(blockIdx.x, gridDim.x);
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

  // ---- descriptors of this tile's row panels (rows past the operand read zero) ------------------------------
  const bf16_t *A = reinterpret_cast<const bf16_t *>(p.A) + (long long)m0 * p.lda;
  const bf16_t *B = reinterpret_cast<const bf16_t *>(p.B) + (long long)n0 * p.ldb;
  const long long bytesA = ((long long)(min(T256, p.M - m0) - 1) * p.lda + p.K) * 2;
  const long long bytesB = ((long long)(min(T256, p.N - n0) - 1) * p.ldb + p.K) * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(A), 0, (int)min(bytesA, 0x7fffffffLL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(B), 0, (int)min(bytesB, 0x7fffffffLL), 0x00020000);

  // ---- DMA addressing: a wave-instruction fills 8 image rows x 128 B; lane -> (row lane>>3, slot lane&7);
  //      slot s of image row r holds the operand's 16-byte chunk s ^ (r & 7) -------------------------------------
  const int srow = 8 * wave + (lane >> 3);
  const int chunk = (lane & 7) ^ (lane >> 3);
  unsigned voA[4], voB[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    voA[u] = (unsigned)(((long long)(64 * u + srow) * p.lda) * 2 + chunk * 16);
    voB[u] = (unsigned)(((long long)b_image_to_col(64 * u + srow) * p.ldb) * 2 + chunk * 16);
  }
  const int ntile = p.K / BK256;
  // piece u of the A (which = 0) / B (which = 1) image of K tile kt
  auto stage = [&](int kt, int which, int u) {
    char *dst = smem + (kt & 1) * BUF_BYTES + which * OPER_BYTES + (64 * u + 8 * wave) * 128;
    if (which == 0) DM_LDS_DMA(rsA, dst, voA[u], kt * (BK256 * 2));
    else DM_LDS_DMA(rsB, dst, voB[u], kt * (BK256 * 2));
  };

  // ---- fragment addressing ----------------------------------------------------------------------------------
  const int sw0 = ((g) ^ (li & 7)) << 4, sw1 = ((4 + g) ^ (li & 7)) << 4;
  const int aoff = (wr * 128 + li) * 128;                 // + (sub*64 + i*16) * 128
  const int boff = OPER_BYTES + (wc * 32 + li) * 128;     // + (sub*128 + j*16) * 128

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 fa[8], fb0[4], fb1[4];

  auto load_a = [&](const char *buf, int sub) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char *r = buf + aoff + (sub * 64 + i * 16) * 128;
      fa[2 * i] = *reinterpret_cast<const u32x4 *>(r + sw0);
      fa[2 * i + 1] = *reinterpret_cast<const u32x4 *>(r + sw1);
    }
  };
  auto load_b = [&](const char *buf, int sub, u32x4(&fb)[4]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char *r = buf + boff + (sub * 128 + j * 16) * 128;
      fb[2 * j] = *reinterpret_cast<const u32x4 *>(r + sw0);
      fb[2 * j + 1] = *reinterpret_cast<const u32x4 *>(r + sw1);
    }
  };
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
    stage(1, 0, 0); stage(1, 0, 2);      // A sub0 pieces (rows 0-63 of each half)
    stage(1, 1, 0); stage(1, 1, 1);      // B sub0
    stage(1, 1, 2); stage(1, 1, 3);      // B sub1
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // the second wave row runs one barrier behind the first

  for (int kt = 0; kt < ntile; ++kt) {
    const char *buf = smem + (kt & 1) * BUF_BYTES;
    // phase 0: quadrant (0,0)
    load_b(buf, 0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    load_a(buf, 0);
    if (kt + 1 < ntile) { stage(kt + 1, 0, 1); stage(kt + 1, 0, 3); }
    DM_PHASE_SYNC();
    DM_QUAD(0, 0, fb0);
    DM_PHASE_END();
    // phase 1: quadrant (0,1)
    load_b(buf, 1, fb1);
    DM_PHASE_SYNC();
    DM_QUAD(0, 1, fb1);
    DM_PHASE_END();
    // phase 2: quadrant (1,1)
    load_a(buf, 1);
    if (kt + 2 < ntile) { stage(kt + 2, 0, 0); stage(kt + 2, 0, 2); stage(kt + 2, 1, 0); stage(kt + 2, 1, 1); }
    DM_PHASE_SYNC();
    DM_QUAD(1, 1, fb1);
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
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + li;
    if (m >= p.M) continue;
    const DmGemmRow rb = dm_gemm_row(p, m);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + 4 * g;
      if (n >= p.N) continue;
      dm_gemm_emit<FAST>(p, acc[i][j], rb, n);
    }
  }
}

}  // namespace

// Launches the 256x256 pipeline when the product suits it; returns false (nothing launched) otherwise.
// Called by dm_gemm after argument validation (alignment, N % 4, ...); dry_run only answers the question.
bool dm_gemm256_try(GemmParams &p, int layout, int ab_dtype, hipStream_t s, bool dry_run) {
  static const int mode = [] { const char *e = getenv("DM_GEMM_256"); return e ? atoi(e) : 1; }();   // 0 = off (A/B runs)
  if (mode == 0 || layout != DM_NT || ab_dtype != DM_BF16) return false;
  if (p.K % BK256 != 0 || p.K < BK256) return false;
  if (256LL * p.lda * 2 >= (1LL << 31) || 256LL * p.ldb * 2 >= (1LL << 31)) return false;
  const int tiles_m = (p.M + T256 - 1) / T256, tiles_n = (p.N + T256 - 1) / T256;
  const long long tiles = (long long)tiles_m * tiles_n;
  // Measured inside the encoder's step (activations are L2-cold there, unlike a warm microbenchmark): with one workgroup
  // per CU the ~1.5 K tiles of DMA in flight do not cover a first-touch miss, and the 256x256 epilogue cannot overlap
  // another workgroup's main loop.  The pipeline wins when every A panel is re-used by many column tiles (fc1: N = 3072,
  // +15 %) or when there are several rounds of tiles; narrow products (N = 768 / 2304 at M = 16384) stay on the
  // 128x128 kernel, whose 12 waves per CU hide the misses (-20 % otherwise).  mode 2 forces the pipeline (benchmarks).
  if (mode != 2 && !((tiles_n >= 10 && tiles >= 384) || tiles >= 1024)) return false;
  static const bool attr_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm256_nt_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS256) == hipSuccess;
  }();
  if (!attr_ok) return false;
  if (dry_run) return true;
  p.tiles_m = tiles_m; p.tiles_n = tiles_n; p.split_k = 1; p.k_per_split = p.K;
  hipLaunchKernelGGL(gemm256_nt_kernel<true>, dim3((unsigned)tiles), dim3(512), LDS256, s, p);
  return true;
}
