// bf16 NT GEMM for the encoder's forward and (through transposed weight mirrors) dgrad products, gfx950:
// C[M,N] = A[M,K] . B[N,K]^T, both operands k-contiguous, fused epilogue.
//
// Why a third kernel.  Measured on the step's own shapes with cache-cold operands (tools/mb_epi.py, round 2): the 256x256
// pipeline (dm_gemm256.hip, one 512-thread workgroup per CU) spends ~1.2 us per 64-deep K tile and ~20 us per TILE outside the
// K loop -- cold prologue fill, then an epilogue in which every CU of the chip stores at the same time while no MFMA runs.
// With K = 768 (qkv, proj, fc1 and the dgrads that mirror them) that fixed part is larger than the 12 K tiles of work, and the
// register-staged 128x128 kernel (3 workgroups per CU, slower main loop) ends up at the same time.  This kernel keeps the
// per-wave work of the pipeline (a 128x64 output block per wave = 32 accumulator tiles, LDS-DMA operand staging) but makes the
// workgroup HALF as large so that TWO independent workgroups share a CU: while one is in its prologue or epilogue (HBM-bound,
// VALU for GELU) the other one's main loop owns the matrix pipe.
//
//   * workgroup = 4 waves (2 x 2), one wave per SIMD, <= 256 VGPRs -> two workgroups per CU (80 KiB of LDS each);
//   * tile = 256 x 128 (WM = 8 m-tiles per wave) or 128 x 128 (WM = 4, for products with few tiles), K tile 64;
//   * operand rows are staged 128 B at a time: tools/mb_fill.py measures 29 TB/s of LDS-DMA fill from L2 with >= 128-byte row
//     pieces against 15 TB/s with 64-byte pieces (a first version of this kernel with K tiles of 32 was fill-bound at 9 TB/s);
//   * LDS = two A buffers (double-buffered) + ONE B buffer: a wave moves all of its B fragments of a K tile into registers
//     (8 x 16 B per lane) first, so the B buffer is re-filled for the next K tile while this one is computed.  Per K tile:
//         vmcnt(0), barrier          every wave's pieces of tile t have landed; every wave is done reading A[(t-1)&1]
//         B fragments -> registers, barrier      the B buffer is free
//         issue B(t+1) and A(t+1) -> A[(t+1)&1]; two k-steps of [8 A fragment reads, 32 MFMAs]
//   * image of a k-contiguous operand: rows of 128 B, 16-byte slot s of row r holds source chunk s ^ (r & 7) (the layout of
//     dm_gemm256.hip: conflict-free ds_read_b128 fragments).  The LDS destination of a DMA instruction is lane-linear, so the
//     permutation is applied to the per-lane SOURCE address;
//   * epilogue: the wave transposes its accumulators through a private LDS region (rows padded to 272 B) and walks them row
//     by row, 8 consecutive columns per lane, so that bias / residual / aux reads and the C stores are whole 128-byte lines
//     (the accumulator layout itself gives 8-byte pieces of 16 different rows per instruction).
#include <cstdlib>

#include "dm_common.h"
#include "dm_gemm_common.h"
#include "dm_mfma.h"

namespace dmring {

constexpr int BK = 64;
constexpr int BN = 128;

template <int WM> struct Cfg {
  static constexpr int BM = 2 * WM * 16;
  static constexpr int A_BYTES = BM * 128;
  static constexpr int B_BYTES = BN * 128;
  static constexpr int B_OFF = 2 * A_BYTES;
  static constexpr int RING = 2 * A_BYTES + B_BYTES;        // 80 KiB (WM = 8) / 48 KiB (WM = 4)
  static constexpr int EPI_ROWS = WM * 8;                    // rows staged per pass and wave
  static constexpr int EPI = 4 * EPI_ROWS * DM_EPI_PITCH;
  static constexpr int LDS = RING > EPI ? RING : EPI;
  static constexpr int NA = BM / 32;                         // A pieces (8 rows each) per wave and K tile
  static constexpr int NB = BN / 32;
};

#define DM_RING_DMA(rsrc, dst, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst), 16, voff, soff, 0, 0)

// FOLD: hi / lo plane pairs, three K segments of p.k_fold (GemmParams.k_fold), as in the other GEMM kernels.
template <int WM, int DBG = 0, bool FOLD = false>
__global__ __launch_bounds__(256, 2) void gemm_ring_kernel(const GemmParams p) {
  using C = Cfg<WM>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 1, wc = wave & 1;
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
  const int m0 = tm * C::BM, n0 = tn * BN;
  const int nk = p.K / BK;

  // ---- DMA addressing: piece q = 8 image rows x 128 B; this wave fills pieces wave + 4u ---------------------------
  const bf16_t *pa = reinterpret_cast<const bf16_t *>(p.A) + (long long)m0 * p.lda;
  const bf16_t *pb = reinterpret_cast<const bf16_t *>(p.B) + (long long)n0 * p.ldb;
  const long long a_far = FOLD ? max(p.a_fold[0], max(p.a_fold[1], p.a_fold[2])) : 0, b_far = FOLD ? max(p.b_fold[0], max(p.b_fold[1], p.b_fold[2])) : 0;
  const long long ba = ((long long)(min(C::BM, p.M - m0) - 1) * p.lda + (FOLD ? a_far + p.k_fold : (long long)p.K)) * 2;
  const long long bb = ((long long)(min(BN, p.N - n0) - 1) * p.ldb + (FOLD ? b_far + p.k_fold : (long long)p.K)) * 2;
  const int seg_tiles = FOLD ? p.k_fold / BK : 1;
  auto k_off = [&](int kt, bool is_b) -> int {       // byte offset of K tile kt inside a row of the operand
    if constexpr (!FOLD) return kt * (BK * 2);
    const int seg = (kt >= 2 * seg_tiles) ? 2 : (kt >= seg_tiles) ? 1 : 0;
    const long long o = is_b ? (seg == 0 ? p.b_fold[0] : seg == 1 ? p.b_fold[1] : p.b_fold[2]) : (seg == 0 ? p.a_fold[0] : seg == 1 ? p.a_fold[1] : p.a_fold[2]);
    return (int)(o * 2) + (kt - seg * seg_tiles) * (BK * 2);
  };
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(pa), 0, (int)min(ba, 0x7fffffffLL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(pb), 0, (int)min(bb, 0x7fffffffLL), 0x00020000);
  const int prow = lane >> 3;
  const int csrc = (lane & 7) ^ prow;
  int voA[C::NA], voB[C::NB];
#pragma unroll
  for (int u = 0; u < C::NA; ++u) voA[u] = (int)(((long long)((wave + 4 * u) * 8 + prow) * p.lda) * 2 + csrc * 16);
#pragma unroll
  for (int u = 0; u < C::NB; ++u) voB[u] = (int)(((long long)((wave + 4 * u) * 8 + prow) * p.ldb) * 2 + csrc * 16);

  // (destinations are written relative to the `smem` array and the offsets cast to int: otherwise the builtin fails to
  // instantiate in the HOST pass of this template -- silently -- and no launch stub is emitted)
  auto stage_a = [&](int kt, int u0, int u1) {
    const int sb = (kt & 1) * C::A_BYTES;
    const int ko = k_off(kt, false);
#pragma unroll
    for (int u = 0; u < C::NA; ++u)
      if (u >= u0 && u < u1) DM_RING_DMA(rsA, smem + sb + (wave + 4 * u) * 1024, (int)voA[u], (int)ko);
  };
  auto stage_b = [&](int kt) {
    const int ko = k_off(kt, true);
#pragma unroll
    for (int u = 0; u < C::NB; ++u) DM_RING_DMA(rsB, smem + C::B_OFF + (wave + 4 * u) * 1024, (int)voB[u], (int)ko);
  };

  // fragment reads: k-step ks of row r lives in slots (4 ks + g) ^ (r & 7)
  const int fr0 = li * 128 + ((g ^ (li & 7)) << 4), fr1 = li * 128 + (((4 + g) ^ (li & 7)) << 4);
  const int offA = wr * WM * 2048, offB = C::B_OFF + wc * 4 * 2048;

  f32x4 acc[WM][4];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 fb[2][4], fa[WM];

  // (Tried and dropped: a software L2 prefetch -- every wave touching one dword per 64 bytes of the operand lines two K tiles
  // ahead, left in flight by a counted vmcnt.  The extra 64-byte requests cost more than the latency they hid: 110 us against
  // 96 us on the fc1 product, tools/mb_ring.py.)
  if constexpr ((DBG & 1) == 0) { stage_b(0); stage_a(0, 0, C::NA); }
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fb[0][j] = *reinterpret_cast<const u32x4 *>(smem + ((DBG & 4) ? 0 : offB + j * 2048 + fr0));
      fb[1][j] = *reinterpret_cast<const u32x4 *>(smem + ((DBG & 4) ? 0 : offB + j * 2048 + fr1));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const bool more = (kt + 1 < nk) && !(DBG & 1);
    if (more) stage_b(kt + 1);
    const char *img = smem + (kt & 1) * C::A_BYTES + offA;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < WM; ++i) fa[i] = *reinterpret_cast<const u32x4 *>(img + ((DBG & 4) ? 0 : i * 2048 + (ks ? fr1 : fr0)));
      if (more) stage_a(kt + 1, ks * (C::NA / 2), (ks + 1) * (C::NA / 2));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr ((DBG & 2) != 0) acc[i][j][0] += __builtin_bit_cast(float, fa[i][0] ^ fb[ks][j][1]);
          else mma<bf16_t>(acc[i][j], fa[i], fb[ks][j]);
        }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: transpose through a wave-private LDS region, then whole-line row accesses ---------------------
  __builtin_amdgcn_s_barrier();                       // every wave is done with the ring
  char *mine = smem + wave * (C::EPI_ROWS * DM_EPI_PITCH);
  dm_epilogue_rows<WM, C::EPI_ROWS, (DBG & 8) != 0>(p, acc, mine, m0 + wr * (WM * 16), n0 + wc * 64, lane);
}

template <int WM, bool FOLD = false> bool set_lds_limit() {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_ring_kernel<WM, 0, FOLD>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<WM>::LDS) == hipSuccess;
}

}  // namespace dmring

// Decides whether the ring kernel runs this product; fills p.tiles_m / tiles_n / split_k and returns the wave height (8 or 4), 0 = no.
int dm_gemm_ring_plan(GemmParams &p, int layout, int ab_dtype, bool aligned8) {
  const char *env = getenv("DM_GEMM_RING");                      // 0 = off, 1 = routing rules, 2 = whenever legal (read per call: tests flip it)
  const int mode = env ? atoi(env) : 1;
  if (mode == 0 || layout != DM_NT || ab_dtype != DM_BF16 || !aligned8) return 0;
  if (p.K % dmring::BK != 0 || p.N % 8 != 0) return 0;
  if (p.k_fold > 0 && p.k_fold % dmring::BK != 0) return 0;      // folded contraction: segments of whole K tiles
  if (256LL * p.lda * 2 >= (1LL << 31) || 128LL * p.ldb * 2 >= (1LL << 31)) return 0;
  const char *fenv = getenv("DM_GEMM_RING_WM");
  const int force = fenv ? atoi(fenv) : 0;
  const long long t256 = (long long)((p.M + 255) / 256) * ((p.N + 127) / 128);
  // Measured with cache-cold operands (tools/mb_epi.py, round 2, us per launch: this kernel / best of the other two):
  //   M = 16384: qkv 72 / 83, fc1 + GELU' 119 / 133, proj + residual 53 / 50, fc2 + residual 107 / 102
  //   M =  4096: qkv 23 / 31, fc1 36 / 38, proj 19 / 16, fc2 45 / 42          M = 1024: never ahead
  // -> wide outputs (several rounds of tiles, so the two workgroups of a CU drift apart and overlap) take it
  if (mode == 1 && !(p.N >= 2048 && p.M >= 2048)) return 0;
  // (round 5, tools/routing_check.py: from ~100 k tokens on the 256 x 256 pipeline is ahead on the wide forward products -- the 2000-point evaluation
  // batch, 384 000 tokens: qkv 1364 against 1456 us, fc1 2393 against 2458 -- below that the two tie within 2 %)
  if (mode == 1 && p.k_fold == 0 && p.M >= 262144) return 0;
  // ... until the 128x128 kernel got the same whole-line epilogue (dm_gemm.hip): where its tiles make whole rounds of 3 workgroups per
  // CU it is now ahead inside the training step (tools/prof_shapes.py, per launch, same box: 16384 x 2304 90 -> 79 us,
  // 16384 x 3072 + GELU' 121 -> 116 us, 4096 x 3072 + GELU' 44 -> 36 us); 4096 x 2304 (576 tiles = 0.75 round) stays here (29 vs 31 us)
  // Round 4, both kernels on the lean epilogue (dm_gemm_common.h), same box, per step: 16384 x 3072 + GELU' 0.353 ms on 128 x 128 tiles,
  // 0.330 here (0.335 on the 256 x 256 pipeline); 16384 x 2304 0.235 / 0.235; 4096 x 3072 + GELU' 0.073 / 0.080 -> the widest product is back
  // folded products (tools/mb_fold.py, the qkv forward with a plane-pair result, 16384 x 2304 x 3*768): 192 us here, 199 on the 4-wave kernel,
  // 208 on the 256 x 256 pipeline, 236 on 128 x 128 tiles -- the whole-rounds exception below is a bf16-mode (K = 768) finding
  static const bool fold_routes = [] { const char *e = getenv("DM_GEMM_FOLD_ROUTES"); return !(e && atoi(e) == 0); }();
  const bool fold_wide = fold_routes && p.k_fold > 0 && p.N >= 2048 && p.M >= 8192;
  if (mode == 1 && !fold_wide && ((long long)((p.M + 127) / 128) * ((p.N + 127) / 128)) % 768 == 0 && !(p.N >= 3072 && p.M >= 8192)) return 0;
  int wm = t256 >= 384 ? 8 : 4;
  if (force == 8 || force == 4) wm = force;
  static const bool ok = dmring::set_lds_limit<8>() && dmring::set_lds_limit<4>() && dmring::set_lds_limit<8, true>() && dmring::set_lds_limit<4, true>();
  if (!ok) return 0;
  const int bm = wm * 32;
  p.tiles_m = (p.M + bm - 1) / bm;
  p.tiles_n = (p.N + 127) / 128;
  p.split_k = 1;
  p.k_per_split = p.K;
  const char *denv = getenv("DM_RING_DEBUG");
  p.debug = denv ? atoi(denv) : 0;
  return wm;
}

void dm_gemm_ring_launch(const GemmParams &p, int wm, hipStream_t s) {
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n));
#ifdef DM_RING_ABLATE
  if (wm == 8 && p.debug) {
    switch (p.debug) {
#define DM_ABL(D) case D: { static const bool k = hipFuncSetAttribute(reinterpret_cast<const void *>(dmring::gemm_ring_kernel<8, D>), hipFuncAttributeMaxDynamicSharedMemorySize, dmring::Cfg<8>::LDS) == hipSuccess; (void)k; hipLaunchKernelGGL((dmring::gemm_ring_kernel<8, D>), grid, dim3(256), dmring::Cfg<8>::LDS, s, p); return; }
      DM_ABL(1) DM_ABL(2) DM_ABL(4) DM_ABL(5) DM_ABL(6) DM_ABL(7) DM_ABL(8) DM_ABL(15)
#undef DM_ABL
      default: break;
    }
  }
#endif
  if (p.k_fold > 0) {
    if (wm == 8) hipLaunchKernelGGL((dmring::gemm_ring_kernel<8, 0, true>), grid, dim3(256), dmring::Cfg<8>::LDS, s, p);
    else hipLaunchKernelGGL((dmring::gemm_ring_kernel<4, 0, true>), grid, dim3(256), dmring::Cfg<4>::LDS, s, p);
    return;
  }
  if (wm == 8) hipLaunchKernelGGL(dmring::gemm_ring_kernel<8>, grid, dim3(256), dmring::Cfg<8>::LDS, s, p);
  else hipLaunchKernelGGL(dmring::gemm_ring_kernel<4>, grid, dim3(256), dmring::Cfg<4>::LDS, s, p);
}
