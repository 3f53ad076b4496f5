// Persistent 256x192x64 bf16 GEMM, four 512-register waves per CU (gfx950): forward (NT) and dgrad (NN) products of the
// 16384-token stage.
//
// Why this shape of kernel.  Ablations of the LDS-DMA pipelines (profiles/r02_p192_ablations.txt) show the matrix pipe alone
// runs a K step in 0.64 us and the operand fill alone in 0.6-0.9 us, but with two LDS buffers only ~1.3 K steps of fill are
// in flight, every step waits for its slowest piece (a first-touch L2 miss is ~1.5 us under load), and the 8-wave ping-pong
// exposes its fragment reads: ~1.15 us per step whatever the tile.  LDS (160 KiB) cannot hold a third buffer; the register
// file (512 KiB per CU) can.  So, like the vendor library's kernels for these shapes (tools/mb_yardstick.py):
//   * 4 waves (2 x 2), one per SIMD, 128 x 96 outputs each (48 accumulator tiles = 192 registers): 30 % fewer LDS fragment
//     bytes per MFMA than 8 waves of 64 x 96;
//   * operands go global -> registers -> LDS; two register sets hold the K steps s+2 and s+3 while step s+1 sits in the second
//     LDS buffer: 2-3 K steps of fill in flight;
//   * one barrier per K step, in the middle of it: [ks 0 MFMAs | fragment reads of ks 1, ds_write of step s+1, global loads of
//     step s+3] barrier [ks 1 MFMAs | fragment reads of step s+1, ks 0];
//   * persistent: grid = min(tiles, CUs), a workgroup walks its tiles (L, L + grid, ...) as one flattened sequence of K steps,
//     so the fill never drains at a tile boundary; 256 x 192 tiles make N = 768 / 2304 / 3072 at M = 16384 whole rounds of
//     the 256 CUs;
//   * fused epilogue through a per-wave LDS staging region that is not shared with the K-step buffers.
// Loads past the end of the sequence use a null buffer descriptor (zero-filled, no traffic): no tail special cases.
#include <cstdlib>
#include <type_traits>

#include "dm_common.h"
#include "dm_gemm_common.h"
#include "dm_mfma.h"

// -DDM_W4_STAMP (diagnostic build, tools/w4_stamps.py): wave 0 of the first 64 workgroups stamps s_memtime at six points of its K steps
// 8 .. 23 of the folded form's three-set step into a __device__ array
#ifdef DM_W4_STAMP
__device__ unsigned long long dmw4_stamps[64 * 16 * 8];
#define DMW4_T(i) do { if (wave == 0 && lane == 0 && blockIdx.x < 64 && sstep >= 8 && sstep < 24) dmw4_stamps[(blockIdx.x * 16 + (sstep - 8)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int dm_debug_w4_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dmw4_stamps), sizeof(dmw4_stamps)); }
// ... and every workgroup's wave 0 stamps the phases of its FIRST tile: 0 kernel entry, 1 addresses set up (first global load next), 2 prologue
// done (two K steps staged, first fragments read), 3 K loop done, 4 epilogue issued
__device__ unsigned long long dmw4_kstamps[512 * 8];
#define DMW4_K(i) do { if (wave == 0 && lane == 0 && blockIdx.x < 512) { dmw4_kstamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 0 || (i) == 4) dmw4_kstamps[blockIdx.x * 8 + ((i) == 0 ? 5 : 6)] = __builtin_amdgcn_s_memrealtime(); } } while (0)     /* (slots 5 / 6: the chip-wide 100 MHz clock at entry / end) */
extern "C" int dm_debug_w4_kstamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dmw4_kstamps), sizeof(dmw4_kstamps)); }
#else
#define DMW4_T(i) do { } while (0)
#define DMW4_K(i) do { } while (0)
#endif

namespace dmw4 {

constexpr int TM = 256, TN = 192, BK = 64;
constexpr int A_BYTES = TM * 128;              // 32 KiB
constexpr int B_BYTES = TN * 128;              // 24 KiB
constexpr int BUF_BYTES = A_BYTES + B_BYTES;   // 56 KiB
constexpr int EPI_WAVE = 16 * 96 * 4;          // 16 rows x 96 fp32, XOR-swizzled
constexpr int LDS_BYTES = 2 * BUF_BYTES + 4 * EPI_WAVE;   // 136 KiB

template <int V> using IC = std::integral_constant<int, V>;

// The accumulator tile stays in the SAME four AGPRs ("+a": destination tied to the addend).  With the builtin the register
// allocator, under this kernel's pressure, splits the 48 accumulator live ranges and shuffles them AGPR <-> VGPR around every MFMA.
// Operand order as in dm_mfma.h (swapped: a lane ends with 4 consecutive n of row m = lane & 15).
// acc = 0 in place (0 * 0 + 0 on the matrix pipe).  A plain assignment makes the compiler materialise all 48 zero tiles in AGPR-class
// registers next to the live accumulators (384 > 256 -> scratch spills inside the K loop).
__device__ __forceinline__ void zero_pinned(f32x4 &acc, const u32x4 &z) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "=a"(acc) : "v"(z));      // (the compiler cannot see the MFMA: its VALU-write -> MFMA-read wait states are inserted by hand)
}
__device__ __forceinline__ void mma_pinned(f32x4 &acc, const u32x4 &a, const u32x4 &b) {
  asm volatile("s_nop 0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
}

// DBG (ablation builds only, -DDM_W4_ABLATE): 1 no epilogue, 4 no global loads, 8 no LDS writes, 16 no fragment reads, 32 no MFMAs
// EK: 0 = the generic fused epilogue (any operand combination, grouped rows); k > 0 = the lean epilogue with item structure key
// k - 1 = RES | YL << 1 | C32 << 3 | XS << 4 (dm_gemm_common.h).  ONE epilogue per kernel instance: with two in one kernel the
// accumulators meet in phi nodes behind them and the register allocator spills all 48 tiles around every tile end (tried).
// FOLD ("bf16x3" products on hi / lo plane pairs, GemmParams.k_fold; the standard pattern A = (hi, hi, lo), B = (hi, lo, hi)): a K step
// covers 32 contraction positions and stages BOTH pieces of both operands -- the 128-byte row of an LDS image is [32 hi | 32 lo]
// (k-contiguous operands) resp. its k-rows 0..31 hi, 32..63 lo (m-contiguous ones), i.e. the images keep their format and only the
// global addresses of the staged pieces change -- and runs THREE MFMA sets on them: hi.hi, hi.lo, lo.hi.  Per 64 contraction
// positions that is 2 steps of staging for 6 MFMA sets where the generic folded form (three K segments, one MFMA set pair per step)
// needs 3 steps: the staging / fragment traffic that bounds this kernel's step drops by a third.
// The kernel body: workgroup L of G (XCD-remapped ids) of the launch described by p.  gemm_w4_kernel runs it for one product;
// gemm_w4_grouped_kernel for one of several independent weight gradients sharing a launch (dm_gemm_grouped).
// SK ("stream-K" weight gradients, gemm_w4_streamk_kernel): the workgroup owns `sk.count` consecutive K steps of the product's step space
// (tile-major: tile t is steps [t * sk.steps, (t + 1) * sk.steps)) starting at sk.g0, i.e. the tail of one tile, then whole tiles, then the
// head of another; every piece (segment) leaves a partial tile in slot sk.slot0 + r of the tile-local slab p.workspace [slot][256][192]
// (column sums: p.colsum_slab [slot][256]) and streamk_fixup_kernel sums a tile's pieces in workgroup order.
struct W4Sk { int g0, count, steps, slot0; };
template <int LAYOUT, int DBG, int EK, bool FOLD, bool CS, bool SK = false>
__device__ __forceinline__ void gemm_w4_body(const GemmParams &p, const int G, const int L, const W4Sk sk = W4Sk{0, 0, 1, 0}) {
  static_assert(!SK || (LAYOUT == DM_TN && EK != 0 && DBG == 0), "stream-K segments: weight gradients with the lean epilogue");
  constexpr int EPIU = 0;
  constexpr bool AMM = (LAYOUT == DM_TN);      // A m-contiguous [K][M] (wgrad) or k-contiguous [M][K]
  constexpr bool BMM = (LAYOUT != DM_NT);      // B m-contiguous [K][N] (dgrad, wgrad) or k-contiguous [N][K] (forward)
  constexpr int NB = 6;                        // global loads of B per thread and K step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int g = lane >> 4, li = lane & 15;

  DMW4_K(0);
  const int tiles = p.tiles_m * p.tiles_n;
  {
    // Experiment (DM_W4_STAGGER = n, off by default): the workgroups of every other XCD start n x 1024 cycles late, so that half of the
    // chip stores its tiles while the other half is in its K loop (all 256 workgroups otherwise reach every epilogue together and
    // the stores run at the memory system's burst rate with the matrix pipes idle).
    const int stag = (p.debug >> 16) & 0x7fff;
    if (stag && (blockIdx.x & 1)) {
      const long long t0 = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t0 < (long long)stag * 1024) __builtin_amdgcn_s_sleep(32);
    }
  }
  // wgrad (TN): one (tile, K slice) per workgroup, slices of k_per_split (the last one may be shorter); the partial tile goes to
  // slab z of the split-K workspace.  Forward / dgrad: whole K, tiles L, L + G, ...
  // (round 4: forward / dgrad products with few tiles and a long contraction are sliced the same way -- p.split_k > 1, partial
  // tiles to the slab, dm_gemm's splitk_epilogue_kernel applies the fused epilogue)
  const bool sliced = AMM || p.split_k > 1;
  const int zslice = sliced ? L / tiles : 0;
  constexpr int BKR = FOLD ? 32 : BK;                  // contraction positions per K step
  const int k_total = FOLD ? p.k_fold : p.K;
  const int kbeg = sliced ? zslice * p.k_per_split : 0;
  const int kend = sliced ? min(k_total, kbeg + p.k_per_split) : k_total;
  const int ntile = (kend - kbeg) / BKR;
  const long long a_lo = FOLD ? p.a_fold[2] : 0, b_lo = FOLD ? p.b_fold[1] : 0;      // element offset of the lo plane behind the hi plane
  const int sk_t0 = SK ? sk.g0 / sk.steps : 0;          // first tile this workgroup touches
  const int n_my = SK ? (sk.g0 + sk.count - 1) / sk.steps - sk_t0 + 1 : sliced ? 1 : (tiles - L + G - 1) / G;
  const int total = SK ? sk.count : n_my * ntile;
  // K steps / first contraction position / contraction length of work item r (SK: segment r; else every item is [kbeg, kend))
  auto nt = [&](int r) __attribute__((always_inline)) {
    if constexpr (!SK) return ntile;
    else return min((sk_t0 + r + 1) * sk.steps, sk.g0 + sk.count) - max((sk_t0 + r) * sk.steps, sk.g0);
  };
  auto kb = [&](int r) __attribute__((always_inline)) {
    if constexpr (!SK) return kbeg;
    else return (r == 0 ? sk.g0 - sk_t0 * sk.steps : 0) * BKR;
  };
  auto kl = [&](int r) __attribute__((always_inline)) {
    if constexpr (!SK) return kend - kbeg;
    else return nt(r) * BKR;
  };

  // ---- global -> register mapping -----------------------------------------------------------------------------------------------
  // k-contiguous operand: load u of a thread is the 16-byte chunk (t & 7) of tile row (t >> 3) + 32 u; LDS image = rows of 128 B,
  // slot s of row r holds chunk s ^ (r & 7).
  const int lrow = t >> 3, lchunk = t & 7;
  const int wA = lrow * 128 + ((lchunk ^ (lrow & 7)) << 4);           // + u * 4096
  int voA, strideA, stepA, wA0, wA1;
  // (FOLD: chunks 0..3 of a row come from the hi plane, 4..7 from the lo plane, 32 contraction positions each)
  if constexpr (!AMM) {
    voA = FOLD ? (int)(((long long)lrow * p.lda + (lchunk & 3) * 8 + (lchunk >= 4 ? a_lo : 0)) * 2) : (int)(((long long)lrow * p.lda + lchunk * 8) * 2);
    strideA = (int)(32 * p.lda * 2);
    stepA = BKR * 2;
    wA0 = wA1 = wA;
  } else {
    // m-contiguous A [K][M]: chunk c32 = t % 32 (8 rows of the tile) of k-row t / 32 + 8 u; image = 4 bands of [64 k-rows][64 m], as B below
    const int krow = t >> 5, c32 = t & 31, band = c32 >> 3, c8 = c32 & 7;
    voA = (int)(((long long)krow * p.lda + c32 * 8) * 2);
    strideA = (int)(8 * p.lda * 2);
    stepA = (int)(BKR * p.lda * 2);
    const int f0 = (krow >> 1) & 1;
    wA0 = band * 8192 + krow * 128 + (((c8 >> 1) ^ f0) << 5) + ((c8 & 1) << 4);          // even u: + u * 1024
    wA1 = band * 8192 + krow * 128 + (((c8 >> 1) ^ (f0 | 2)) << 5) + ((c8 & 1) << 4);    // odd u
  }
  int voB = 0, strideB = 0, stepB, wB0 = 0;
  int voB3[3] = {0, 0, 0}, wB3[3] = {0, 0, 0}, halfB = 0;
  if constexpr (!BMM) {
    voB = FOLD ? (int)(((long long)lrow * p.ldb + (lchunk & 3) * 8 + (lchunk >= 4 ? b_lo : 0)) * 2) : (int)(((long long)lrow * p.ldb + lchunk * 8) * 2);
    strideB = (int)(32 * p.ldb * 2);
    stepB = BKR * 2;
    wB0 = wA;                                                         // + u * 4096
  } else {
    // m-contiguous operand [K][N]: a half tile (32 k-rows x 192 columns) is 768 16-byte chunks = 3 pieces of all 256 threads -- piece i
    // of a thread is chunk c = t + 256 i: k-row c / 24, columns 8 (c % 24) .. + 7 -- and pieces 3..5 are the same chunks 32 k-rows
    // further down (plain) resp. of the lo plane (FOLD), i.e. a scalar offset on the global side and + 4096 in the image.  (Until round 4
    // threads 0..191 loaded 8 pieces of 8 k-rows each and wave 3 none: two load / write pairs more on the waves every barrier waits for.)
    // LDS image = 3 bands of [64 k-rows][64 columns = 128 B]; the 32-byte slot index of k-row r is XORed with
    // f(r) = ((r >> 1) & 1) | (((r >> 3) & 1) << 1).
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = t + 256 * i, krow = c / 24, c24 = c - krow * 24, band = c24 >> 3, c8 = c24 & 7;
      const int f = ((krow >> 1) & 1) | (((krow >> 3) & 1) << 1);
      voB3[i] = (int)(((long long)krow * p.ldb + c24 * 8) * 2);
      wB3[i] = band * 8192 + krow * 128 + (((c8 >> 1) ^ f) << 5) + ((c8 & 1) << 4);
    }
    halfB = FOLD ? (int)(b_lo * 2) : (int)(32 * p.ldb * 2);
    stepB = (int)(BKR * p.ldb * 2);
  }

  // ---- fragment offsets -----------------------------------------------------------------------------------------------------------
  // A fragments.  k-contiguous: two offsets (k-steps), + i * 2048 per row tile.  m-contiguous: row tile i of this wave is column block
  // wm * 8 + i of the image = band wm * 2 + (i >> 2), 32-byte slot i & 3: four offsets (slot), + (i >> 2) * 8192 + ks * 4096 (+ 512).
  int offA0, offA1, xsA[4];
  if constexpr (!AMM) {
    offA0 = (wm * 128 + li) * 128 + ((g ^ (li & 7)) << 4);
    offA1 = (wm * 128 + li) * 128 + (((4 + g) ^ (li & 7)) << 4);
    xsA[0] = xsA[1] = xsA[2] = xsA[3] = 0;
  } else {
    const int q = li >> 2, pq = li & 3;
    const int rbase = wm * 16384 + (8 * g + q) * 128 + 8 * pq;
    const int fr = ((q >> 1) & 1) | ((g & 1) << 1);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) xsA[sl] = rbase + ((sl ^ fr) << 5);
    offA0 = offA1 = 0;
  }
  int offB[6];      // NT: [0], [1] = the two k-steps (+ j * 2048); NN: one per column tile (+ ks * 4096, + 512 for the upper 4 k-rows)
  if constexpr (!BMM) {
    offB[0] = (wn * 96 + li) * 128 + ((g ^ (li & 7)) << 4);
    offB[1] = (wn * 96 + li) * 128 + (((4 + g) ^ (li & 7)) << 4);
    offB[2] = offB[3] = offB[4] = offB[5] = 0;
  } else {
    const int q = li >> 2, pq = li & 3;
    const int rbase = (8 * g + q) * 128 + 8 * pq;
    const int fr = ((q >> 1) & 1) | ((g & 1) << 1);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int cb = wn * 6 + j;
      offB[j] = (cb >> 2) * 8192 + rbase + (((cb & 3) ^ fr) << 5);
    }
  }

  // ---- tile cursors (uniform) -------------------------------------------------------------------------------------------------------
  const bf16_t *Ab = reinterpret_cast<const bf16_t *>(p.A), *Bb = reinterpret_cast<const bf16_t *>(p.B);
  auto tile_mn = [&](int r, int &m0, int &n0) __attribute__((always_inline)) {
    const int tid = SK ? sk_t0 + r : (DBG & 64) ? (L & 7) : sliced ? L - zslice * tiles : L + r * G;        // (ablation 64: every workgroup reads the same few L2-resident tiles)
    // (A BLOCKED order -- column blocks of 8 tiles walked row by row, an XCD keeping its own chunk over the rounds, so that its 32
    // workgroups share 4 panels of A and 8 of B instead of 2 and 16 -- was measured in round 4: 16384 x 3072 x 768 on 4 rounds 103 -> 97 us,
    // x 2304 on 3 rounds 80 -> 77 us, the sliced weight gradients unchanged; the multi-round products still lose to the kernels with 2-3
    // workgroups per CU inside the step (qkv forward 85 vs 77 us per launch), so the row-major order stayed.)
    const int tm = tid / p.tiles_n;
    m0 = tm * TM;
    n0 = (tid - tm * p.tiles_n) * TN;
  };
  // (FOLD: the descriptor also reaches over the lo plane's part of the panel -- rows past the operand's last one then still fall outside
  // it in the lo plane, i.e. never outside the plane pair; in the hi plane they read other rows of the pair: outputs nobody stores)
  auto make_a = [&](int m0, bool live, int r = 0) __attribute__((always_inline)) {
    const int kb0 = kb(live ? r : 0), kl0 = kl(live ? r : 0);
    if constexpr (!AMM) {
      const long long bytes = ((long long)(min(TM, p.M - m0) - 1) * p.lda + kl0 + a_lo) * 2;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(Ab + (long long)m0 * p.lda + kb0), 0, live ? (int)min(bytes, 0x7fffffffLL) : 0, 0x00020000);
    } else {
      const long long bytes = ((long long)(kl0 - 1) * p.lda + (p.M - m0) + a_lo) * 2;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(Ab + (long long)kb0 * p.lda + m0), 0, live ? (int)min(bytes, 0x7fffffffLL) : 0, 0x00020000);
    }
  };
  auto make_b = [&](int n0, bool live, int r = 0) __attribute__((always_inline)) {
    const int kb0 = kb(live ? r : 0), kl0 = kl(live ? r : 0);
    if constexpr (!BMM) {
      const long long bytes = ((long long)(min(TN, p.N - n0) - 1) * p.ldb + kl0 + b_lo) * 2;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(Bb + (long long)n0 * p.ldb + kb0), 0, live ? (int)min(bytes, 0x7fffffffLL) : 0, 0x00020000);
    } else {
      const long long bytes = ((long long)(kl0 - 1) * p.ldb + (p.N - n0) + b_lo) * 2;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(Bb + (long long)kb0 * p.ldb + n0), 0, live ? (int)min(bytes, 0x7fffffffLL) : 0, 0x00020000);
    }
  };
  int m_cur, n_cur;
  tile_mn(0, m_cur, n_cur);
  // Staging schedule.  The 8 + NB pieces (16-byte-per-thread loads) of a K step are split into two groups of PH = (8 + NB) / 2:
  // group X (pieces 0 .. PH-1) and group Y (the rest); piece q is A load q for q < 8, else B load q - 8.  K step j waits in register
  // set j & 1.  During K step s:
  //   k-step 0 (before the mid-step barrier): group Y of step s + 1 goes set -> LDS buffer (s + 1) & 1, the set entry is refilled
  //            with step s + 3;
  //   k-step 1 (after the barrier, when nobody reads buffer s & 1 any more): group X of step s + 2 goes set -> buffer s & 1, refilled
  //            with step s + 4.
  // So the ds_write_b128 (the VGPR -> LDS path moves ~79 B/clk per CU: 56 KiB of a step = ~730 cycles) are spread one per row tile
  // over the WHOLE step instead of two per row tile over half of it.  Two load cursors (one per group); past the end of the
  // sequence: null descriptors, the loads return zeros.
  constexpr int PH = (8 + NB) / 2;
  struct Cursor { int r, k; };
  Cursor cx{0, 0}, cy{0, 0};
  __amdgpu_buffer_rsrc_t rsAx = make_a(m_cur, true), rsBx = make_b(n_cur, true), rsAy = rsAx, rsBy = rsBx;
  auto advance_x = [&]() __attribute__((always_inline)) {
    if (++cx.k == nt(cx.r)) {
      cx.k = 0; ++cx.r;
      int m0 = 0, n0 = 0;
      const bool live = cx.r < n_my;
      if (live) tile_mn(cx.r, m0, n0);
      rsAx = make_a(m0, live, cx.r);
      rsBx = make_b(n0, live, cx.r);
    }
  };
  auto advance_y = [&]() __attribute__((always_inline)) {
    if (++cy.k == nt(cy.r)) {
      cy.k = 0; ++cy.r;
      int m0 = 0, n0 = 0;
      const bool live = cy.r < n_my;
      if (live) tile_mn(cy.r, m0, n0);
      rsAy = make_a(m0, live, cy.r);
      rsBy = make_b(n0, live, cy.r);
    }
  };

  u32x4 ga[2][8], gb[2][NB];
  u32x4 vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero[0]));
  vzero[1] = vzero[2] = vzero[3] = vzero[0];
  asm volatile("s_nop 7" ::: "memory");
  if constexpr (DBG & 4) {
#pragma unroll
    for (int u = 0; u < 8; ++u) ga[0][u] = ga[1][u] = vzero;
#pragma unroll
    for (int u = 0; u < NB; ++u) gb[0][u] = gb[1][u] = vzero;
  }
  // fetch piece q of the group cursor's K step into register set `set`
  auto gload = [&](auto set_tag, auto grp_tag, int q) __attribute__((always_inline)) {
    if constexpr (DBG & 4) return;
    constexpr int set = decltype(set_tag)::value;
    constexpr bool GX = decltype(grp_tag)::value == 0;
    const int k = GX ? cx.k : cy.k;
    // (FOLD, m-contiguous operands: pieces 0..3 are the hi plane's k-rows, 4..7 the lo plane's rows of the same 32 positions; the
    // k-contiguous ones carry the plane in their per-thread offset)
    const int u = q - 8;
    const int pa = (FOLD && AMM) ? (q & 3) * strideA + (q >= 4 ? (int)(a_lo * 2) : 0) : q * strideA;
    if (q < 8) ga[set][q] = __builtin_amdgcn_raw_buffer_load_b128(GX ? rsAx : rsAy, voA, k * stepA + pa, 0);
    else if constexpr (BMM) gb[set][u] = __builtin_amdgcn_raw_buffer_load_b128(GX ? rsBx : rsBy, voB3[u % 3], k * stepB + (u / 3) * halfB, 0);
    else gb[set][u] = __builtin_amdgcn_raw_buffer_load_b128(GX ? rsBx : rsBy, voB, k * stepB + u * strideB, 0);
  };
  // piece q: register set -> LDS buffer
  auto lwrite = [&](auto set_tag, auto buf_tag, int q) __attribute__((always_inline)) {
    if constexpr (DBG & 8) return;
    constexpr int set = decltype(set_tag)::value;
    char *a = smem + decltype(buf_tag)::value * BUF_BYTES;
    if (q < 8) {
      if constexpr (!AMM) *reinterpret_cast<u32x4 *>(a + wA + q * 4096) = ga[set][q];
      else *reinterpret_cast<u32x4 *>(a + ((q & 1) ? wA1 : wA0) + q * 1024) = ga[set][q];
    } else {
      const int u = q - 8;
      if constexpr (!BMM) *reinterpret_cast<u32x4 *>(a + A_BYTES + wB0 + u * 4096) = gb[set][u];
      else *reinterpret_cast<u32x4 *>(a + A_BYTES + wB3[u % 3] + (u / 3) * 4096) = gb[set][u];
    }
  };

  f32x4 acc[8][6];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) zero_pinned(acc[i][j], vzero);
  // wgrad: column sums of A (the bias gradient that goes with dW = dy^T x) ride along: every wave multiplies its A fragments with a
  // "ones" fragment (every column of the 16 x 16 result = the row sums) -- eight bf16 1.0 in the workgroups of column tile 0, zeros
  // everywhere else -- in BOTH k-steps, unconditionally; the waves wn == 0 of column tile 0 write the partial row [z][M] of
  // p.colsum_slab.  (Until round 4 only column tile 0 did this, one k-step per wave, behind a uniform branch per row tile: sixteen
  // branches per K step in EVERY workgroup cost more than the MFMAs they skipped -- weight gradients 96 -> 91 us.  CS = false: instances
  // without any of it, for launches that want no column sums.)
  constexpr bool CSM = AMM && CS;
  bool colsum = CSM && p.colsum_slab != nullptr && n_cur == 0;      // (SK: re-evaluated when the workgroup moves on to its next tile)
  f32x4 accb[CSM ? 8 : 1];
  const unsigned one2 = colsum ? 0x3F803F80u : 0u;
  u32x4 ones = {one2, one2, one2, one2};     // eight bf16 1.0 (column tile 0) or zeros
  if constexpr (CSM) {
    asm volatile("" : "+v"(ones));
#pragma unroll
    for (int i = 0; i < 8; ++i) zero_pinned(accb[i], vzero);
  }
  // Fragments: B double-buffered (the next k-step's six tiles load while this one's are in use); A rotates in place (row tile i's
  // fragment is dead after its six MFMAs and is reloaded for the next k-step at once) -- the arch-VGPR budget is 256.
  u32x4 fa[8], fb[2][6];
  if constexpr (DBG & 16) {
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = vzero;
#pragma unroll
    for (int j = 0; j < 6; ++j) fb[0][j] = fb[1][j] = vzero;
  }

  auto read_a = [&](int i, auto ks_tag, auto buf_tag) __attribute__((always_inline)) {
    constexpr int ks = decltype(ks_tag)::value;
    if constexpr (DBG & 16) return fa[i];
    const char *a = smem + decltype(buf_tag)::value * BUF_BYTES;
    if constexpr (!AMM) {
      return *reinterpret_cast<const u32x4 *>(a + i * 2048 + (ks ? offA1 : offA0));
    } else {
      const u32x2 lo = dm_ds_read_tr16(a + xsA[i & 3] + (i >> 2) * 8192 + ks * 4096);
      const u32x2 hi = dm_ds_read_tr16(a + xsA[i & 3] + (i >> 2) * 8192 + ks * 4096 + 512);
      return (u32x4){lo[0], lo[1], hi[0], hi[1]};
    }
  };
  // B fragment of column tile j, k-step ks of LDS buffer buf, into fragment set dst
  auto load_bx = [&](int j, auto ks_tag, auto buf_tag, auto dst_tag) __attribute__((always_inline)) {
    constexpr int ks = decltype(ks_tag)::value, dst = decltype(dst_tag)::value;
    if constexpr (DBG & 16) return;
    const char *b = smem + decltype(buf_tag)::value * BUF_BYTES + A_BYTES;
    if constexpr (!BMM) {
      fb[dst][j] = *reinterpret_cast<const u32x4 *>(b + j * 2048 + offB[ks]);
    } else {
      const u32x2 lo = dm_ds_read_tr16(b + offB[j] + ks * 4096);
      const u32x2 hi = dm_ds_read_tr16(b + offB[j] + ks * 4096 + 512);
      fb[dst][j] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
    }
  };
  auto load_b1 = [&](int j, auto ks_tag, auto buf_tag) __attribute__((always_inline)) { load_bx(j, ks_tag, buf_tag, ks_tag); };
  // 48 MFMAs of k-step KS, row tile by row tile.  A wave issues in order, so everything else is placed BETWEEN the MFMAs (one
  // piece after each, scheduling fences in between): the matrix pipe takes 16 cycles per MFMA, the pieces fit in its shadow.
  //   after MFMA 0: read the row tile's A fragment of the NEXT k-step (NKS of buffer NBUF; it replaces fa[i] after the row)
  //   after MFMA 2: staging piece i of this phase's group (GRP: 0 = X, 1 = Y): register set SSET -> LDS buffer SBUF
  //   after MFMA 3: the freed set entry is refilled from global memory
  //   after MFMA 5: B fragment i of the next k-step (rows 0..5)
  auto mm = [&](f32x4 &c, const u32x4 &a, const u32x4 &b) __attribute__((always_inline)) {
    if constexpr (!(DBG & 32)) mma_pinned(c, a, b);
  };
  auto mfmas = [&](auto ks_tag, auto nks_tag, auto nbuf_tag, auto grp_tag, auto sset_tag, auto sbuf_tag) __attribute__((always_inline)) {
    constexpr int ks = decltype(ks_tag)::value;
    constexpr int q0 = decltype(grp_tag)::value == 0 ? 0 : PH;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      mm(acc[i][0], fa[i], fb[ks][0]);
      const u32x4 na = read_a(i, nks_tag, nbuf_tag);
      __builtin_amdgcn_sched_barrier(0);
      mm(acc[i][1], fa[i], fb[ks][1]);
      __builtin_amdgcn_sched_barrier(0);
      mm(acc[i][2], fa[i], fb[ks][2]);
      if (i < PH) lwrite(sset_tag, sbuf_tag, q0 + i);
      __builtin_amdgcn_sched_barrier(0);
      mm(acc[i][3], fa[i], fb[ks][3]);
      if (i < PH) gload(sset_tag, grp_tag, q0 + i);
      __builtin_amdgcn_sched_barrier(0);
      mm(acc[i][4], fa[i], fb[ks][4]);
      __builtin_amdgcn_sched_barrier(0);
      mm(acc[i][5], fa[i], fb[ks][5]);
      if (i < 6) load_b1(i, nks_tag, nbuf_tag);
      if constexpr (CSM) mm(accb[i], fa[i], ones);
      fa[i] = na;
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (decltype(grp_tag)::value == 0) advance_x(); else advance_y();
  };

  // FOLD: one MFMA set of the three of a K step: 48 MFMAs of the A fragments in fa (hi or lo piece) with fragment set FB of B.
  //   ANX:  after MFMA 0 of a row tile, read the A fragment that replaces fa[i] after the row (k-step AKS of buffer ABUF), or nothing;
  //   GRP:  staging group (0 = X, 1 = Y, -1 = none) moved register set SSET -> LDS buffer SBUF after MFMA 2 and refilled after MFMA 3;
  //   BNX:  after MFMA 5, B fragment i of k-step BKS of buffer BBUF into fragment set BDST, or nothing;
  //   CSW:  the waves wn == CSW add this set's A piece to the column sums (weight gradients), -1 = nobody.
  auto mfmas3 = [&](auto fb_tag, auto anx_tag, auto aks_tag, auto abuf_tag, auto grp_tag, auto sset_tag, auto sbuf_tag, auto bnx_tag, auto bks_tag,
                    auto bbuf_tag, auto bdst_tag, auto csw_tag) __attribute__((always_inline)) {
    constexpr int FB = decltype(fb_tag)::value, GRP = decltype(grp_tag)::value, CSW = decltype(csw_tag)::value;
    constexpr bool ANX = decltype(anx_tag)::value != 0, BNX = decltype(bnx_tag)::value != 0;
    constexpr int q0 = GRP == 1 ? PH : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      mm(acc[i][0], fa[i], fb[FB][0]);
      u32x4 na = fa[i];
      if constexpr (ANX) na = read_a(i, aks_tag, abuf_tag);
      __builtin_amdgcn_sched_barrier(0);
      mm(acc[i][1], fa[i], fb[FB][1]);
      __builtin_amdgcn_sched_barrier(0);
      mm(acc[i][2], fa[i], fb[FB][2]);
      if constexpr (GRP >= 0) { if (i < PH) lwrite(sset_tag, sbuf_tag, q0 + i); }
      __builtin_amdgcn_sched_barrier(0);
      mm(acc[i][3], fa[i], fb[FB][3]);
      if constexpr (GRP >= 0) { if (i < PH) gload(sset_tag, IC<(GRP > 0 ? 1 : 0)>{}, q0 + i); }
      __builtin_amdgcn_sched_barrier(0);
      mm(acc[i][4], fa[i], fb[FB][4]);
      __builtin_amdgcn_sched_barrier(0);
      mm(acc[i][5], fa[i], fb[FB][5]);
      if constexpr (BNX) { if (i < 6) load_bx(i, bks_tag, bbuf_tag, bdst_tag); }
      if constexpr (CSM && CSW >= 0) mm(accb[i], fa[i], ones);
      if constexpr (ANX) fa[i] = na;
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (GRP == 0) advance_x();
    if constexpr (GRP == 1) advance_y();
  };

  int kt = 0, r = 0;
  const int lane_outer = lane;
  // Two K steps before a tile ends, one 4-byte load per 128-byte line pulls the tile of the epilogue's READ operands (fp32 residual;
  // aux of the multiply / GELU' epilogues) towards L2: the epilogue walks rows with few bytes in flight and would otherwise pay an
  // HBM round trip per pass (fc2 forward: 39 us of epilogue for 67 us of K loop).  The loads are younger than every staged piece
  // that is waited for in the meantime, so the in-order vmcnt of the operand pipeline never waits for them.
  float tv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // the touch loads' destinations: not read before the epilogue
  auto touch_epilogue_operands = [&]() __attribute__((always_inline)) {
    const int m = m_cur + t;                            // one row per thread
    if (m >= p.M) return;
    const DmGemmRow rb = dm_gemm_row(p, m);
    if (p.residual) {
#pragma unroll
      for (int j = 0; j < 6; ++j) tv[j] = p.residual[rb.r + n_cur + j * 32];
    } else if (p.aux && (p.epilogue == DM_EPI_DGELU || p.epilogue == DM_EPI_MUL)) {
      if (p.aux_dtype == DM_F32) {
#pragma unroll
        for (int j = 0; j < 6; ++j) tv[j] = reinterpret_cast<const float *>(p.aux)[rb.x + n_cur + j * 32];
      } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) tv[j] = __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short *>(p.aux)[rb.x + n_cur + j * 64]);
      }
    }
  };

  // Tile finished: fused epilogue.  The two waves that share the tile's rows (wn = 0 / 1) transpose 16 rows x 192 columns through a
  // common LDS slab and then each walks 8 of the rows whole: a row is 24 lanes x 8 columns, so bias / residual / aux reads and the C
  // stores are complete 128-byte lines (a wave's own 96 columns = 192 B of bf16 would end in half lines, which the memory system
  // pays for with read-modify-writes: measured 2.6 TB/s of stores).  K-step buffers untouched: the staging of the next tile goes on.
  // Round 4: the same epilogue in the lean form of dm_gemm_common.h (dm_epilogue_rows_lean): three buffer descriptors per tile, per-lane
  // offsets of the three passes computed once per tile, a row tile only advances a scalar offset, bias loaded once per tile, and the
  // read operands of item (i, q + 1) are requested BEFORE the stores of item (i, q) -- in the generic form below every pass pays
  // ~25 quarter-rate integer multiplies (64-bit row offsets, item / 24) and its loads wait for the previous pass's stores (vmcnt
  // retires in order): a memory round trip per pass, 24 per tile.  RT / RES / YL / C32 / XS: the structure of an item, as there.
  auto epilogue_lean = [&](auto rt_tag, auto res_tag, auto yl_tag, auto c32_tag, auto xs_tag) __attribute__((always_inline)) {
    constexpr bool RT = decltype(rt_tag)::value != 0, RES = decltype(res_tag)::value != 0, C32 = decltype(c32_tag)::value != 0;
    constexpr int YL = decltype(yl_tag)::value, XS = decltype(xs_tag)::value;
    char *slab = smem + 2 * BUF_BYTES + wm * (2 * EPI_WAVE);
    int lane = lane_outer;
    asm volatile("" : "+v"(lane) : "v"(tv[0]), "v"(tv[1]), "v"(tv[2]), "v"(tv[3]), "v"(tv[4]), "v"(tv[5]));
    const int g = lane >> 4, li = lane & 15;
    const bool split = SK || p.split_k > 1;
    const bool c32 = (AMM || split) ? true : (RT ? (p.c_dtype == DM_F32) : C32);
    const bool x32 = RT ? (p.aux_dtype == DM_F32) : (YL == 3 || XS == 2);
    const bool plain = AMM || split;                       // no fused epilogue: partial tile / weight gradient
    const bool has_res = !plain && (RT ? (p.residual != nullptr) : RES);
    const bool has_acc = !split && (RT ? (c32 && p.accumulate) : (YL == 1));
    const bool aux_load = !plain && (RT ? (p.aux && (p.epilogue == DM_EPI_DGELU || p.epilogue == DM_EPI_MUL)) : (YL >= 2));
    const bool aux_store = !plain && (RT ? (p.aux && (p.epilogue == DM_EPI_GELU || p.epilogue == DM_EPI_GELU_GRAD)) : (XS != 0));
    const int csz = c32 ? 4 : 2, xsz = x32 ? 4 : 2;
    const int m_base = m_cur + wm * 128, n_base = n_cur;            // wave-uniform (wave id through readfirstlane)
    const bool live = m_base < p.M;
    const long long rows_below = SK ? (long long)(TM - 1 - wm * 128) : (long long)(p.M - 1 - m_base), cols_right = SK ? (long long)TN : (long long)(p.N - n_base);
    const long long ldc_eff = SK ? (long long)TN : split ? (long long)p.N : p.ldc;
    char *cptr = SK ? reinterpret_cast<char *>(p.workspace + ((long long)(sk.slot0 + r) * TM + wm * 128) * TN)      // tile-local slab slot of this segment
               : split ? reinterpret_cast<char *>(p.workspace + ((long long)zslice * p.M + m_base) * p.N + n_base)
                       : reinterpret_cast<char *>(p.C) + ((long long)m_base * p.ldc + n_base) * csz;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(cptr, 0, live ? dm_epi_records((rows_below * ldc_eff + cols_right) * csz) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.residual) + (has_res ? (long long)m_base * p.ldr + n_base : 0), 0,
        (live && has_res) ? dm_epi_records((rows_below * p.ldr + cols_right) * 4) : 0, 0x00020000);
    const bool has_aux = aux_load || aux_store;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(p.aux) + (has_aux ? ((long long)m_base * p.ldaux + n_base) * xsz : 0), 0,
        (live && has_aux) ? dm_epi_records((rows_below * p.ldaux + cols_right) * xsz) : 0, 0x00020000);
    const int stepC = 16 * (int)ldc_eff * csz, stepR = 16 * (int)p.ldr * 4, stepX = 16 * (int)p.ldaux * xsz;      // one row tile
    unsigned oC[3], oR[3], oX[3];
    int sl0[3], sl1[3];
    f32x4 b_lo[3], b_hi[3];
#pragma unroll
    for (int q3 = 0; q3 < 3; ++q3) {
      const int item = q3 * 64 + lane;             // 8 rows x 24 groups of 8 columns
      const int rr = item / 24, cg = item - rr * 24;
      const int row = wn * 8 + rr;
      const unsigned kill = (n_base + cg * 8 < p.N) ? 0u : 0x80000000u;
      oC[q3] = (unsigned)((row * (int)ldc_eff + cg * 8) * csz) | kill;
      oR[q3] = (unsigned)((row * (int)p.ldr + cg * 8) * 4) | kill;
      oX[q3] = (unsigned)((row * (int)p.ldaux + cg * 8) * xsz) | kill;
      sl0[q3] = row * 768 + (((2 * cg) ^ (row & 7)) << 4);
      sl1[q3] = row * 768 + (((2 * cg + 1) ^ (row & 7)) << 4);
      b_lo[q3] = b_hi[q3] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (!plain && p.bias && !kill) { b_lo[q3] = dm_load4(p.bias + n_base + cg * 8); b_hi[q3] = dm_load4(p.bias + n_base + cg * 8 + 4); }
    }
    auto prefetch = [&](DmEpiPre &pre, int i, int q3) __attribute__((always_inline)) {
      if (has_res) {
        pre.r0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, oR[q3], i * stepR, DM_EPI_LOAD_POLICY));
        pre.r1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, oR[q3] + 16, i * stepR, DM_EPI_LOAD_POLICY));
      }
      if (has_acc) {
        pre.y0 = __builtin_amdgcn_raw_buffer_load_b128(rsC, oC[q3], i * stepC, 0);
        pre.y1 = __builtin_amdgcn_raw_buffer_load_b128(rsC, oC[q3] + 16, i * stepC, 0);
      }
      if (aux_load) {
        pre.y0 = __builtin_amdgcn_raw_buffer_load_b128(rsX, oX[q3], i * stepX, DM_EPI_LOAD_POLICY);
        if (x32) pre.y1 = __builtin_amdgcn_raw_buffer_load_b128(rsX, oX[q3] + 16, i * stepX, DM_EPI_LOAD_POLICY);
      }
    };
    auto pack8 = [](const f32x4 &a, const f32x4 &b) __attribute__((always_inline)) {
      const bf16x8 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
      return __builtin_bit_cast(u32x4, o);
    };
    auto emit = [&](f32x4 lo, f32x4 hi, const DmEpiPre &pre, int i, int q3) __attribute__((always_inline)) {
      lo += b_lo[q3]; hi += b_hi[q3];
      if (!plain && p.epilogue == DM_EPI_GELU) {
        if (aux_store) {
          if (x32) {
            DM_EPI_BSTORE(__builtin_bit_cast(u32x4, lo), rsX, oX[q3], i * stepX, DM_EPI_AUX_POLICY);
            DM_EPI_BSTORE(__builtin_bit_cast(u32x4, hi), rsX, oX[q3] + 16, i * stepX, DM_EPI_AUX_POLICY);
          } else {
            DM_EPI_BSTORE(pack8(lo, hi), rsX, oX[q3], i * stepX, DM_EPI_AUX_POLICY);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = dm_gelu_fast(lo[e]); hi[e] = dm_gelu_fast(hi[e]); }
      } else if (!plain && p.epilogue == DM_EPI_GELU_GRAD) {
        f32x4 dl, dh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float cdf, pdf;
          dm_gelu_parts_fast(lo[e], cdf, pdf);
          dl[e] = fmaf(lo[e], pdf, cdf);
          lo[e] = lo[e] * cdf;
          dm_gelu_parts_fast(hi[e], cdf, pdf);
          dh[e] = fmaf(hi[e], pdf, cdf);
          hi[e] = hi[e] * cdf;
        }
        if (aux_store) {
          if (x32) {
            DM_EPI_BSTORE(__builtin_bit_cast(u32x4, dl), rsX, oX[q3], i * stepX, DM_EPI_AUX_POLICY);
            DM_EPI_BSTORE(__builtin_bit_cast(u32x4, dh), rsX, oX[q3] + 16, i * stepX, DM_EPI_AUX_POLICY);
          } else {
            DM_EPI_BSTORE(pack8(dl, dh), rsX, oX[q3], i * stepX, DM_EPI_AUX_POLICY);
          }
        }
      } else if (aux_load) {
        f32x4 ul, uh;
        if (x32) { ul = __builtin_bit_cast(f32x4, pre.y0); uh = __builtin_bit_cast(f32x4, pre.y1); }
        else {
          const bf16x8 u = __builtin_bit_cast(bf16x8, pre.y0);
          ul = (f32x4){(float)u[0], (float)u[1], (float)u[2], (float)u[3]};
          uh = (f32x4){(float)u[4], (float)u[5], (float)u[6], (float)u[7]};
        }
        if (p.epilogue == DM_EPI_MUL) { lo *= ul; hi *= uh; }
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { lo[e] *= dm_dgelu_fast(ul[e]); hi[e] *= dm_dgelu_fast(uh[e]); }
        }
      }
      if (has_res) { lo += pre.r0; hi += pre.r1; }
      if (c32) {
        if (has_acc) { lo += __builtin_bit_cast(f32x4, pre.y0); hi += __builtin_bit_cast(f32x4, pre.y1); }
        DM_EPI_BSTORE(__builtin_bit_cast(u32x4, lo), rsC, oC[q3], i * stepC, 0);
        DM_EPI_BSTORE(__builtin_bit_cast(u32x4, hi), rsC, oC[q3] + 16, i * stepC, 0);
      } else {
        DM_EPI_BSTORE(pack8(lo, hi), rsC, oC[q3], i * stepC, 0);
      }
    };
    DmEpiPre pre[2];
    prefetch(pre[0], 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        asm volatile("" : "+a"(acc[i][j]));
        *reinterpret_cast<f32x4 *>(slab + li * 768 + (((wn * 24 + j * 4 + g) ^ (li & 7)) << 4)) = acc[i][j];
        zero_pinned(acc[i][j], vzero);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q3 = 0; q3 < 3; ++q3) {
        const int idx = i * 3 + q3;
        if (idx + 1 < 24) prefetch(pre[(idx + 1) & 1], (idx + 1) / 3, (idx + 1) % 3);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(slab + sl0[q3]);
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(slab + sl1[q3]);
        emit(lo, hi, pre[idx & 1], i, q3);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next row tile overwrites the slab
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (CSM) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(accb[i]));
      if (colsum && wn == 0 && g == 0) {
        float *row = SK ? p.colsum_slab + (long long)(sk.slot0 + r) * TM - m_cur : p.colsum_slab + (long long)zslice * p.M;
#pragma unroll
        for (int i = 0; i < 8; ++i) row[m_cur + wm * 128 + i * 16 + li] = accb[i][0];
      }
    }
    kt = 0;
    ++r;
    if (r < n_my) tile_mn(r, m_cur, n_cur);
    if constexpr (SK && CSM) {      // the next segment belongs to another tile: fresh column sums, and only column tile 0 takes them
      colsum = p.colsum_slab != nullptr && n_cur == 0 && r < n_my;
      const unsigned o2 = colsum ? 0x3F803F80u : 0u;
      ones = (u32x4){o2, o2, o2, o2};
      asm volatile("" : "+v"(ones));
#pragma unroll
      for (int i = 0; i < 8; ++i) zero_pinned(accb[i], vzero);
    }
  };

  auto epilogue_generic = [&]() __attribute__((always_inline)) {
    char *slab = smem + 2 * BUF_BYTES + wm * (2 * EPI_WAVE);           // 16 rows x 768 B, XOR-swizzled 16-byte chunks
    const int m_pair = m_cur + wm * 128;
    // the lane-dependent addresses of the epilogue are recomputed here from an opaque copy of the lane id: hoisted out of the K
    // loop they would occupy ~30 registers that the fragment / staging sets need
    int lane = lane_outer;
    asm volatile("" : "+v"(lane) : "v"(tv[0]), "v"(tv[1]), "v"(tv[2]), "v"(tv[3]), "v"(tv[4]), "v"(tv[5]));   // (the use that keeps the touch loads alive)
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        asm volatile("" : "+a"(acc[i][j]));             // (no stale register copy of an accumulator can be stored: see accb below)
        *reinterpret_cast<f32x4 *>(slab + li * 768 + (((wn * 24 + j * 4 + g) ^ (li & 7)) << 4)) = acc[i][j];
        zero_pinned(acc[i][j], vzero);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      auto pass = [&](int q3) __attribute__((always_inline)) {               // (unrolled: the three bodies' residual / aux loads are in flight together)
        const int item = q3 * 64 + lane;             // 8 rows x 24 groups of 8 columns
        const int rr = item / 24, cg = item - rr * 24;
        const int row = wn * 8 + rr;
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(slab + row * 768 + (((2 * cg) ^ (row & 7)) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(slab + row * 768 + (((2 * cg + 1) ^ (row & 7)) << 4));
        int m = m_pair + i * 16 + row;
        if (AMM || p.split_k > 1) {
          // wgrad: fp32 partial tile into slab z of the split-K workspace (summed in slice order by splitk_reduce_kernel), or,
          // unsplit, straight into the gradient (a sliced forward / dgrad product always writes the slab)
          if (m < p.M) {
            const int n = n_cur + cg * 8;
            if (p.split_k > 1) {
              float *d = p.workspace + ((long long)zslice * p.M + m) * p.N + n;
              dm_store4(d, lo);
              dm_store4(d + 4, hi);
            } else {
              float *d = reinterpret_cast<float *>(p.C) + (long long)m * p.ldc + n;
              f32x4 l2 = lo, h2 = hi;
              if (p.accumulate) { l2 += dm_load4(d); h2 += dm_load4(d + 4); }
              dm_store4(d, l2);
              dm_store4(d + 4, h2);
            }
          }
        } else {
        if constexpr (DBG & 256) m = wm * 128 + i * 16 + row;         // (ablation: every workgroup stores to the same 256 rows)
        if constexpr (DBG & 128) { if (lo[0] == 12345.678f) dm_gemm_emit8(p, lo, hi, dm_gemm_row(p, m), n_cur + cg * 8); }
        else if (m < p.M) dm_gemm_emit8(p, lo, hi, dm_gemm_row(p, m), (DBG & 256) ? cg * 8 : n_cur + cg * 8);
        }
      };
      if constexpr (EPIU) {        // unrolled: the three bodies' residual / aux loads are in flight together
        pass(0); pass(1); pass(2);
      } else {
#pragma nounroll
        for (int q3 = 0; q3 < 3; ++q3) pass(q3);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next row tile overwrites the slab
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (CSM) {
      // The compiler does not know the asm statements are MFMAs: it copies accb[i] out of the AGPRs right behind the last MFMA
      // (no wait states -> the copy misses the last update) and would store that copy here.  Re-reading through an asm operand forces
      // a fresh copy, long after the last MFMA has retired.
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(accb[i]));
      if (colsum && wn == 0 && g == 0) {        // every column of an accb tile holds the same sum: lanes g == 0 write element 0
        float *row = p.colsum_slab + (long long)zslice * p.M;
#pragma unroll
        for (int i = 0; i < 8; ++i) row[m_cur + wm * 128 + i * 16 + li] = accb[i][0];
      }
    }
    kt = 0;
    ++r;
    if (r < n_my) tile_mn(r, m_cur, n_cur);
  };

  // One K step.  PAR = step & 1 = its LDS buffer; step s + 1 waits in register set (s + 1) & 1 = 1 - PAR and goes to buffer 1 - PAR;
  // the freed set then fetches step s + 3.
  auto body = [&](auto par_tag) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    mfmas(IC<0>{}, IC<1>{}, IC<PAR>{}, IC<1>{}, IC<1 - PAR>{}, IC<1 - PAR>{});     // k-step 0 (+ fragments of k-step 1; group Y of step s + 1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    mfmas(IC<1>{}, IC<0>{}, IC<1 - PAR>{}, IC<0>{}, IC<PAR>{}, IC<PAR>{});         // k-step 1 (+ fragments of step s + 1; group X of step s + 2)
  };

  // FOLD: one K step (32 contraction positions, both pieces of both operands).  PAR = step & 1 = its LDS buffer AND the fragment set that
  // holds its B hi piece (the lo piece sits in set 1 - PAR: the roles alternate, so the next step's hi fragments can load while this
  // step's are still in use).  hi.hi [B lo fragments; group Y of step s + 1 -> buffer 1 - PAR] | hi.lo [A lo fragments] | barrier |
  // lo.hi [fragments of step s + 1; group X of step s + 2 -> buffer PAR].
  // s_memtime stamps of the three sets (tools/w4_stamps.py, -DDM_W4_STAMP; weight gradient 768 x 3072 x 3*16384, cycles per set for 768
  // cycles of MFMAs): 1780 / 972 / 1492, + 330 for the wait and the barrier; with BOTH staging groups in front of the barrier (buffer
  // 1 - PAR is free from the previous barrier on) 1440 / 1452 / 1244: a staging group costs ~500-700 cycles wherever it sits, the 14
  // transposed fragment reads of the next step ~450 -- moving the pieces does not shorten the step.  It is not the LDS array either
  // (tools/pmc_gemm_lds.sh: SQ_LDS_IDX_ACTIVE 22 % of the kernel's cycles per CU, bank conflicts 1 % of those); what the counters do
  // show is the operand delivery: L2 hit rates of 73 / 75 / 80 % (profiles/r04_gemm_attn_mfma_util.md) are exactly 1 - (panels an XCD's 32
  // workgroups share) / (panels they load) for 2 x 16, 2 x 16 and 8 x 4 tiles per XCD, i.e. every fourth or fifth line comes from the
  // fabric at less than half the L2's rate per CU (MI355X_MICROARCH.md 'Indexed rows': 33 vs 70 GB/s per CU).
  auto body3 = [&](auto par_tag) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    [[maybe_unused]] const int sstep = kt + PAR;
    DMW4_T(0);
    mfmas3(IC<PAR>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<1>{}, IC<1 - PAR>{}, IC<1 - PAR>{}, IC<1>{}, IC<1>{}, IC<PAR>{}, IC<1 - PAR>{}, IC<0>{});
    DMW4_T(1);
    mfmas3(IC<1 - PAR>{}, IC<1>{}, IC<1>{}, IC<PAR>{}, IC<-1>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<-1>{});
    DMW4_T(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DMW4_T(3);
    __builtin_amdgcn_s_barrier();
    DMW4_T(4);
    mfmas3(IC<PAR>{}, IC<1>{}, IC<0>{}, IC<1 - PAR>{}, IC<0>{}, IC<PAR>{}, IC<PAR>{}, IC<1>{}, IC<0>{}, IC<1 - PAR>{}, IC<1 - PAR>{}, IC<1>{});
    DMW4_T(5);
  };

  // ---- prologue: the state the schedule above expects at step 0 (buffer 0 complete, X(1) in buffer 1, Y(1) / X(2) / Y(2) / X(3) in flight) ----
  DMW4_K(1);
#pragma unroll
  for (int q = 0; q < PH; ++q) gload(IC<0>{}, IC<0>{}, q);                 // X(0) -> set 0
  advance_x();
#pragma unroll
  for (int q = PH; q < 2 * PH; ++q) gload(IC<0>{}, IC<1>{}, q);            // Y(0) -> set 0
  advance_y();
#pragma unroll
  for (int q = 0; q < PH; ++q) gload(IC<1>{}, IC<0>{}, q);                 // X(1) -> set 1
  advance_x();
#pragma unroll
  for (int q = PH; q < 2 * PH; ++q) gload(IC<1>{}, IC<1>{}, q);            // Y(1) -> set 1
  advance_y();
#pragma unroll
  for (int q = 0; q < PH; ++q) { lwrite(IC<0>{}, IC<0>{}, q); gload(IC<0>{}, IC<0>{}, q); }               // X(0) -> buffer 0; X(2) -> set 0
  advance_x();
#pragma unroll
  for (int q = PH; q < 2 * PH; ++q) { lwrite(IC<0>{}, IC<0>{}, q); gload(IC<0>{}, IC<1>{}, q); }          // Y(0) -> buffer 0; Y(2) -> set 0
  advance_y();
#pragma unroll
  for (int q = 0; q < PH; ++q) { lwrite(IC<1>{}, IC<1>{}, q); gload(IC<1>{}, IC<0>{}, q); }               // X(1) -> buffer 1; X(3) -> set 1
  advance_x();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < 6; ++j) load_b1(j, IC<0>{}, IC<0>{});
#pragma unroll
  for (int i = 0; i < 8; ++i) fa[i] = read_a(i, IC<0>{}, IC<0>{});

  DMW4_K(2);
  // K % 128 == 0 (plan): a tile is an even number of K steps, so tiles end after an odd step only
  for (int step = 0; step < total; step += 2) {
    if constexpr (!AMM) { if (kt + 2 == ntile && p.split_k <= 1) touch_epilogue_operands(); }
    if constexpr (FOLD) { body3(IC<0>{}); body3(IC<1>{}); }
    else { body(IC<0>{}); body(IC<1>{}); }
    kt += 2;
    if (kt == nt(r)) {
      if (r == 0) DMW4_K(3);
      if constexpr (DBG & 1) { kt = 0; ++r; if (r < n_my) tile_mn(r, m_cur, n_cur); } else
      if constexpr (EK == 0) epilogue_generic();
      else epilogue_lean(IC<0>{}, IC<(EK - 1) & 1>{}, IC<((EK - 1) >> 1) & 3>{}, IC<((EK - 1) >> 3) & 1>{}, IC<((EK - 1) >> 4) & 3>{});
      if (r == 1) DMW4_K(4);
      // The next step's k-step-0 fragments were prefetched during the last MFMAs; holding their 56 registers across the epilogue
      // (on top of the 88 staging registers in flight) overflows the register file, so they are read again here instead.
#pragma unroll
      for (int j = 0; j < 6; ++j) load_b1(j, IC<0>{}, IC<0>{});
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = read_a(i, IC<0>{}, IC<0>{});
    }
  }
}

template <int LAYOUT, int DBG = 0, int EK = 0, bool FOLD = false, bool CS = true>
__global__ __launch_bounds__(256) void gemm_w4_kernel(const GemmParams p) {
  gemm_w4_body<LAYOUT, DBG, EK, FOLD, CS>(p, (int)gridDim.x, dm_xcd_remap(blockIdx.x, gridDim.x));
}

// Several independent weight gradients (DM_TN, one K slice each) in ONE launch: workgroups first[i] .. first[i + 1] - 1 (after the XCD
// remap of the whole grid) are the tiles of product i.  The small stages' weight gradients have 12 .. 48 tiles each: alone they
// either leave most CUs idle or pay K slices + a slab + a reduction launch; the four of a block together are 144 tiles.
constexpr int GROUP_MAX = 8;
struct GemmGroup {
  GemmParams p[GROUP_MAX];
  int first[GROUP_MAX + 1];
  int n;
};
// Stream-K form of the grouped launch (long contractions: the 16384-token blocks, ViT's 50 k tokens): product i owns workgroups first[i] ..
// first[i + 1] - 1, each of which takes q[i] consecutive K steps of the product's tile-major step space (W4Sk), so that every CU does the same
// number of K steps whatever the tile count -- where K slices per product leave 181 MB of slabs per block (up to 16 partial copies of a gradient)
// and a reduction launch per product, a tile here has 2-3 partial pieces (~65 MB per block) and ONE fix-up launch sums them in workgroup order.
// MEASURED AND NOT USED (round 5; DM_GEMM_GROUPED=3 selects it, the tests cover it): exact, but the four weight gradients of a 16384-token
// block take 287 us in this kernel + 32 us of fix-up against 283 us for the four sliced launches with their reductions (8192 tokens: 195 vs
// 186 us).  The step count per CU is what was planned (146 vs 4 x ~37-52); the step itself is 1.96 us instead of 1.27: workgroups that run
// at the same time sit at DIFFERENT K offsets of their tiles, so no two of them ever want the same operand panel at the same time, and the
// launch fetches (256 + 192) x 64 x 2 B per workgroup and step = 2.1 GB at 7.3 TB/s -- where the tiles of one K slice, walking K in lockstep,
// share every A panel four ways and every B panel nine ways through L2 (10.7 KB per workgroup and step instead of 57).  An L2 of 4 MB holds
// two steps of an XCD's panels, so an offset in time is as bad as no reuse.  What would keep both properties -- equal step counts AND
// lockstep -- is a rectangular decomposition (all tiles x K[0, q) on T workgroups, the remainders on the others); priced at <= 25 us per
// block over today's launches (their K loops sum to 199 us against an ideal 183; the rest is 4 x prologue / epilogue and 29 us of
// reductions) and not built.
struct GemmGroupSk {
  GemmGroup g;
  int q[GROUP_MAX];          // K steps per workgroup (even)
  int steps[GROUP_MAX];      // K steps per tile (even)
  int tile0[GROUP_MAX + 1];  // first tile of product i in the group's tile numbering (fix-up grid)
  int chunks;                // fix-up workgroups per tile
  int slots;                 // partial pieces a workgroup can leave: 2 (q <= steps: tail of a tile, head of the next) or 3 (q <= 2 * steps)
  float *cs_out[GROUP_MAX];  // column sums of A wanted for product i (NULL: not), summed by the fix-up from the pieces' partial rows
  int cs_acc[GROUP_MAX];     // ... added to what is there
};
template <int EK, bool FOLD, bool CS>
__global__ __launch_bounds__(256) void gemm_w4_streamk_kernel(const GemmGroupSk grp) {
  const int L = dm_xcd_remap(blockIdx.x, gridDim.x);
  int i = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < grp.g.n && L >= grp.g.first[k]) i = k;
  i = __builtin_amdgcn_readfirstlane(i);
  const GemmParams &p = grp.g.p[i];
  const int w = L - grp.g.first[i];
  const int total = p.tiles_m * p.tiles_n * grp.steps[i];
  W4Sk sk;
  sk.g0 = w * grp.q[i];
  sk.count = min(grp.q[i], total - sk.g0);
  sk.steps = grp.steps[i];
  sk.slot0 = L * grp.slots;
  if (sk.count <= 0) return;          // (the plan gives every workgroup work; uniform exit otherwise)
  gemm_w4_body<DM_TN, 0, EK, FOLD, CS, true>(p, 1, 0, sk);
}

// Sums the partial pieces of every tile of a stream-K launch in workgroup order and stores (accumulate: adds to) the gradient; the
// workgroups of a column-0 tile do the same for the partial column sums.  grid = tiles x chunks; a workgroup handles 256 / chunks rows.
__global__ __launch_bounds__(256) void streamk_fixup_kernel(const GemmGroupSk grp) {
  const int tile_g = blockIdx.x / grp.chunks, chunk = blockIdx.x - tile_g * grp.chunks;
  int i = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < grp.g.n && tile_g >= grp.tile0[k]) i = k;
  const GemmParams &p = grp.g.p[i];
  const int t = tile_g - grp.tile0[i], q = grp.q[i], steps = grp.steps[i];
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int w_lo = (t * steps) / q, w_hi = ((t + 1) * steps - 1) / q;
  const int rows = TM / grp.chunks, row0 = chunk * rows;
  auto ntl = [](const float *x) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(x)); };
  float *C = reinterpret_cast<float *>(p.C);
  for (int e = threadIdx.x; e < rows * (TN / 4); e += blockDim.x) {
    const int rr = row0 + e / (TN / 4), c4 = (e % (TN / 4)) * 4;
    float *o = C + (long long)(tm * TM + rr) * p.ldc + tn * TN + c4;
    f32x4 v = p.accumulate ? *reinterpret_cast<const f32x4 *>(o) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int w = w_lo; w <= w_hi; ++w) {
      const int slot = (grp.g.first[i] + w) * grp.slots + (t - (w * q) / steps);
      v += ntl(p.workspace + ((long long)slot * TM + rr) * TN + c4);
    }
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(o));
  }
  if (tn == 0 && grp.cs_out[i] != nullptr && threadIdx.x < rows) {
    const int rr = row0 + threadIdx.x;
    float *out = grp.cs_out[i] + tm * TM + rr;
    float v = grp.cs_acc[i] ? *out : 0.f;
    for (int w = w_lo; w <= w_hi; ++w) {
      const int slot = (grp.g.first[i] + w) * grp.slots + (t - (w * q) / steps);
      v += p.colsum_slab[(long long)slot * TM + rr];
    }
    *out = v;
  }
}

template <int EK, bool FOLD, bool CS>
__global__ __launch_bounds__(256) void gemm_w4_grouped_kernel(const GemmGroup grp) {
  const int L = dm_xcd_remap(blockIdx.x, gridDim.x);
  int i = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < grp.n && L >= grp.first[k]) i = k;
  i = __builtin_amdgcn_readfirstlane(i);
  const int tiles = grp.first[i + 1] - grp.first[i];
  gemm_w4_body<DM_TN, 0, EK, FOLD, CS>(grp.p[i], tiles, L - grp.first[i]);
}

}  // namespace dmw4

namespace {
template <int LAYOUT, int DBG = 0, int EK = 0, bool FOLD = false, bool CS = true> bool w4_set_lds_limit() {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(dmw4::gemm_w4_kernel<LAYOUT, DBG, EK, FOLD, CS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             dmw4::LDS_BYTES) == hipSuccess;
}
int w4_cu_count() {
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

// Decides whether the 4-wave persistent kernel runs this product (bf16 NT / NN); fills p.tiles_m / tiles_n and returns the grid
// size (0 = not taken).  `aligned8`: the 8-column epilogue (dm_gemm_emit8) is legal for C / aux / grouped rows.
int dm_gemm_w4_plan(GemmParams &p, int layout, int ab_dtype, bool aligned8, bool can_split, long long workspace_bytes) {
  using namespace dmw4;
  const char *env = getenv("DM_GEMM_W4");         // 0 = off, 1 = routing rule, 2 = every legal product, 3 = every whole-round shape (read per call: tests flip it)
  const int mode = env ? atoi(env) : 1;
  if (mode == 0 || ab_dtype != DM_BF16 || !aligned8) return 0;
  // FOLD instances: K steps of 32 contraction positions on both pieces of both operands (see the kernel): the standard plane-pair
  // pattern only -- left operand (hi, hi, lo), right operand (hi, lo, hi) -- and an even number of steps per tile
  const bool fold = p.k_fold > 0;
  const int k_eff = fold ? p.k_fold : p.K, bk_eff = fold ? 32 : BK;
  if (fold) {
    if (p.k_fold % (2 * bk_eff) != 0) return 0;
    if (p.a_fold[0] != 0 || p.a_fold[1] != 0 || p.a_fold[2] <= 0 || p.b_fold[0] != 0 || p.b_fold[2] != 0 || p.b_fold[1] <= 0) return 0;
    if (p.a_fold[2] * 2 >= (1LL << 30) || p.b_fold[1] * 2 >= (1LL << 30)) return 0;      // (byte offsets live in 32-bit lane offsets)
    static const bool attr_fold = w4_set_lds_limit<DM_TN, 0, 0, true>() && w4_set_lds_limit<DM_TN, 0, 9, true>() && w4_set_lds_limit<DM_TN, 0, 11, true>() &&
                                  w4_set_lds_limit<DM_TN, 0, 0, true, false>() && w4_set_lds_limit<DM_TN, 0, 9, true, false>() && w4_set_lds_limit<DM_TN, 0, 11, true, false>() &&
                                  w4_set_lds_limit<DM_NT, 0, 0, true>() && w4_set_lds_limit<DM_NT, 0, 10, true>() && w4_set_lds_limit<DM_NN, 0, 0, true>();
    if (!attr_fold) return 0;
  }
  if (layout == DM_TN) {
    // wgrad: one (256 x 192 tile, K slice) per workgroup; slices of an even number of K steps chosen so that tiles x slices fills the CUs
    const char *tenv = getenv("DM_GEMM_W4_TN");     // 0 = off, 1 = routing rule (default), 2 = every legal product
    const int tmode = tenv ? atoi(tenv) : 1;
    if (tmode == 0 || p.M % TM != 0 || p.N % TN != 0 || k_eff % (2 * bk_eff) != 0 || p.M % 8 != 0 || p.N % 8 != 0) return 0;
    if (!can_split || p.c_dtype != DM_F32) return 0;
    if (((long long)k_eff * p.lda + (fold ? p.a_fold[2] : 0)) * 2 >= (1LL << 31) || ((long long)k_eff * p.ldb + (fold ? p.b_fold[1] : 0)) * 2 >= (1LL << 31)) return 0;
    const int cus = w4_cu_count();
    const long long tiles = (long long)(p.M / TM) * (p.N / TN);
    if (cus <= 0 || tiles > cus) return 0;
    const int steps = k_eff / bk_eff;
    int split = (int)(cus / tiles);
    if (split > 16) split = 16;
    int per = (steps + split - 1) / split;
    per += per & 1;
    if (per < 16) per = 16;
    split = (steps + per - 1) / per;
    while (split > 1 && (long long)split * p.M * p.N * 4 > workspace_bytes) {       // slab must fit the caller's workspace
      per += 2;
      split = (steps + per - 1) / per;
    }
    if (split > 1 && (long long)split * p.M * p.N * 4 > workspace_bytes) return 0;
    // in-step per launch (tools/prof_shapes.py): K = 16384: 100 -> 84, 92 -> 83, 80 -> 67, 41 -> 38 us; K = 4096 with 48 tiles x 4 slices:
    // 26 -> 24 us; fewer workgroups than 0.7 of the CUs, or the 1024-token stage, lose
    // (round 5, tools/routing_check.py: 2304 x 768 x 4096 -- 36 tiles x 4 slices = 0.56 of the CUs -- 36.0 us here against 41.1 on the 256 x 256
    // pipeline: the fill bound is 0.55 for >= 24 tiles)
    if (tmode == 1 && ((double)(tiles * split) / cus < (tiles >= 24 ? 0.55 : 0.7) || tiles < 12 || (tiles < 24 && k_eff < 8192))) return 0;
    static const bool attr_tn = w4_set_lds_limit<DM_TN>() && w4_set_lds_limit<DM_TN, 0, 9>() && w4_set_lds_limit<DM_TN, 0, 11>() &&
                                w4_set_lds_limit<DM_TN, 0, 0, false, false>() && w4_set_lds_limit<DM_TN, 0, 9, false, false>() && w4_set_lds_limit<DM_TN, 0, 11, false, false>();
    if (!attr_tn) return 0;
    p.tiles_m = p.M / TM;
    p.tiles_n = p.N / TN;
    p.split_k = split;
    p.k_per_split = per * bk_eff;       // (FOLD: in contraction positions of ONE piece)
    return (int)(tiles * split);
  }
  if (layout != DM_NT && layout != DM_NN) return 0;
  if (k_eff < 2 * bk_eff || k_eff % (2 * bk_eff) != 0 || p.N % TN != 0) return 0;
  const long long spanA = 256LL * p.lda * 2 + 2LL * k_eff + (fold ? 2 * p.a_fold[2] : 0);
  const long long spanB = ((layout == DM_NN) ? (long long)k_eff * p.ldb * 2 : 192LL * p.ldb * 2 + 2LL * k_eff) + (fold ? 2 * p.b_fold[1] : 0);
  if (spanA >= (1LL << 31) || spanB >= (1LL << 31)) return 0;
  const int tiles_m = (p.M + TM - 1) / TM, tiles_n = p.N / TN;
  const long long tiles = (long long)tiles_m * tiles_n;
  const int cus = w4_cu_count();
  if (cus <= 0) return 0;
  {
    // Round 4: few tiles and a long contraction (the 4096-token stage's fc2 forward, fc1 / qkv dgrad: 64 tiles, K = 2304 .. 4096):
    // one (tile, K slice) per workgroup as for the weight gradients, <= 4 slices of an even number (>= 8) of K steps; dm_gemm sums
    // the slab and applies the fused epilogue (splitk_epilogue_kernel).  `can_split`: the caller allows it (automatic slice count,
    // 8-column epilogue legal, workspace present).
    // Measured in the step (tools/prof_shapes.py, same box): 4096 x 768 x 3072 54 us sliced against 47 us on unsplit 64 x 64 tiles (fc1
    // dgrad 51 / 44, qkv dgrad 42 / 35): 256 workgroups x 12 K steps pay the kernel's fill and a 192 KiB fp32 slab each, then the 63 MB
    // reduction -- OFF by default (DM_GEMM_W4_SLICES=1 for A/B runs); the M <= 1024 products gain from slices on 64 x 64 tiles instead.
    static const bool slices_on = [] { const char *e = getenv("DM_GEMM_W4_SLICES"); return e && atoi(e) == 1; }();
    constexpr long long LIM = (1LL << 31) / (128LL * 4);
    if (slices_on && p.k_fold == 0 && can_split && mode != 0 && tiles * 2 <= cus && p.K >= 1536 && p.N < LIM && !(p.debug & 0x400)) {
      const int steps = p.K / BK;
      int split = (int)(cus / tiles);
      if (split > 4) split = 4;
      int per = (steps + split - 1) / split;
      per += per & 1;
      if (per < 8) per = 8;
      split = (steps + per - 1) / per;
      if (split > 1 && (long long)split * p.M * p.N * 4 <= workspace_bytes && (double)(tiles * split) / cus >= 0.7) {
        static const bool attr_sl = w4_set_lds_limit<DM_NT, 0, 9>() && w4_set_lds_limit<DM_NN, 0, 9>();
        if (attr_sl) {
          p.tiles_m = tiles_m;
          p.tiles_n = tiles_n;
          p.split_k = split;
          p.k_per_split = per * BK;
          return (int)(tiles * split);
        }
      }
    }
  }
  if (mode == 1 || mode == 4 || mode == 5) {
    // One workgroup per CU, all of them in lockstep: a tile's stores (25 MB per round of the chip) are not hidden by anybody's MFMAs,
    // ~10 us per round (tools/mb_w4_loop.py).  With ONE tile per workgroup that is paid once and the deep operand pipeline wins
    // (dgrads into N = 768: -13 .. -15 %, fc2 forward: -3 %); with 3-4 tiles per workgroup the older kernels, whose 2-3 workgroups
    // per CU overlap each other's epilogues, stay ahead (+14 .. +19 %).  mode 3 = every whole-round shape (for measurements).
    // Round 5 (tools/routing_check.py at token counts the routing had not been tuned on, profiles/r05_routing_check.txt): the data
    // gradients into N <= 768 with a long contraction (qkv / fc1 dgrad: K = 2304 / 3072) win on this kernel at ANY number of rounds once
    // there are >= 96 tiles -- 12288 tokens 47.5 / 63.5 us against 62.8 / 83.3 on 128x128 tiles, 32768: 107 / 138 against 143 / 189,
    // 61440 (config 5's stage 0): 206 / 271 against 269 / 360, 50432 (ViT-B, 3.08 rounds): 195 / 257 against 219 / 297 -- their four
    // column tiles keep a round's stores at a third of the N = 2304 / 3072 products'; below 96 tiles (the 4096-token stage) 64x64 tiles win.
    // The fp32-residual forward products (fc2) keep the one-round rule, from 0.72 of the CUs on (12288 tokens: 80 against 92 us).
    // (folded products too -- tools/mb_fold.py with TOKENS=61440 / 50432: qkv / fc1 / proj dgrad 551 / 734 / 226 us here against 712 / 927 / 273 on the
    // 256 x 256 pipeline, 519 / 672 / 216 against 691 / 915 / 248)
    const bool deep_dgrad = layout == DM_NN && p.K >= 1536 && tiles_n <= 4;
    if (deep_dgrad) {
      if (tiles < 96) return 0;
    } else if (tiles > cus || (double)tiles / (double)cus < (layout == DM_NT ? 0.72 : 0.85)) {
      return 0;
    }
    // Inside the training step (tools/prof_shapes.py, per launch, same box): dgrad 16384 x 768 x 3072 90 -> 77 us, x 2304 71 -> 64 us,
    // fc2 forward (fp32 rows + residual, touched towards L2 two K steps ahead) 103 -> 99 us; the K = 768 products are all fill and
    // epilogue (49 -> 57 us) and stay on the older kernels.
    if (mode != 5 && p.K < 1536) return 0;        // (5: every one-round shape)
  } else if (mode == 3) {
    if (tiles < cus) return 0;
    const long long rounds = (tiles + cus - 1) / cus;
    if ((double)tiles / (double)(rounds * cus) < 0.85) return 0;
  }
  static const bool attr_ok = w4_set_lds_limit<DM_NT>() && w4_set_lds_limit<DM_NN>() && w4_set_lds_limit<DM_NT, 0, 1>() && w4_set_lds_limit<DM_NT, 0, 10>() &&
                              w4_set_lds_limit<DM_NT, 0, 17>() && w4_set_lds_limit<DM_NN, 0, 1>() && w4_set_lds_limit<DM_NN, 0, 5>();
  if (!attr_ok) return 0;
  p.tiles_m = tiles_m;
  p.tiles_n = tiles_n;
  p.split_k = 1;
  p.k_per_split = p.K;
  return (int)(tiles < cus ? tiles : cus);
}

void dm_gemm_w4_launch(const GemmParams &p, int layout, int grid, hipStream_t s) {
#ifdef DM_W4_ABLATE
  if (layout == DM_NT) {
    const char *denv = getenv("DM_W4_DEBUG");
    const int dbg = denv ? atoi(denv) : 0;
#define W4_CASE(D) case D: { static const bool ok = w4_set_lds_limit<DM_NT, D>(); (void)ok; \
      hipLaunchKernelGGL((dmw4::gemm_w4_kernel<DM_NT, D>), dim3(grid), dim3(256), dmw4::LDS_BYTES, s, p); return; }
    switch (dbg) { W4_CASE(1) W4_CASE(5) W4_CASE(13) W4_CASE(29) W4_CASE(33) W4_CASE(17) W4_CASE(9) W4_CASE(61) W4_CASE(45) W4_CASE(65) W4_CASE(73) W4_CASE(3) W4_CASE(128) W4_CASE(256) W4_CASE(320) W4_CASE(64) default: break; }
#undef W4_CASE
  }
#endif
  // Which epilogue: the lean form (dm_gemm_common.h) where its preconditions hold -- plain rows, 32-bit offsets inside a wave pair's
  // 128 rows -- and the item structure is one of the instantiated ones; the generic form otherwise (and with DM_GEMM_EPI_LEAN=0).
  constexpr long long LIM = (1LL << 31) / (128LL * 4);
  const bool lean_ok = !(p.debug & 0x400) && p.rows_per_group == 0 && p.ldc < LIM && p.ldr < LIM && p.ldaux < LIM && p.N < LIM && p.c_dtype != DM_BF16_PAIR;
  const bool c32 = p.c_dtype == DM_F32, x32 = p.aux_dtype == DM_F32;
  const bool aux_read = p.aux && (p.epilogue == DM_EPI_DGELU || p.epilogue == DM_EPI_MUL);
  const bool aux_write = p.aux && (p.epilogue == DM_EPI_GELU || p.epilogue == DM_EPI_GELU_GRAD);
  const int yl = (c32 && p.accumulate && p.split_k <= 1) ? 1 : aux_read ? (x32 ? 3 : 2) : 0;
  const int xs = aux_write ? (x32 ? 2 : 1) : 0;
  const int key = (p.residual ? 1 : 0) | (yl << 1) | ((c32 ? 1 : 0) << 3) | (xs << 4);
  GemmParams q = p;
  {
    const char *senv = getenv("DM_W4_STAGGER");
    const int stag = senv ? atoi(senv) : 0;
    if (stag > 0 && layout != DM_TN && q.tiles_m * q.tiles_n > grid) q.debug |= (stag & 0x7fff) << 16;      // multi-tile forward / dgrad launches only
  }
#define W4_GO(LAY, EKV) hipLaunchKernelGGL((dmw4::gemm_w4_kernel<LAY, 0, EKV>), dim3(grid), dim3(256), dmw4::LDS_BYTES, s, q)
#define W4_GOF(LAY, EKV) hipLaunchKernelGGL((dmw4::gemm_w4_kernel<LAY, 0, EKV, true>), dim3(grid), dim3(256), dmw4::LDS_BYTES, s, q)
// weight gradients: the instance with the column-sum MFMAs only when the launch wants column sums
#define W4_GOT(EKV, FOLDV) do { if (q.colsum_slab) hipLaunchKernelGGL((dmw4::gemm_w4_kernel<DM_TN, 0, EKV, FOLDV, true>), dim3(grid), dim3(256), dmw4::LDS_BYTES, s, q); \
    else hipLaunchKernelGGL((dmw4::gemm_w4_kernel<DM_TN, 0, EKV, FOLDV, false>), dim3(grid), dim3(256), dmw4::LDS_BYTES, s, q); } while (0)
  if (p.k_fold > 0) {                                      // hi / lo plane pairs ("bf16x3"): fp32 outputs
    if (layout == DM_TN) {
      if (lean_ok && key == 8) W4_GOT(9, true);
      else if (lean_ok && key == 10) W4_GOT(11, true);
      else W4_GOT(0, true);
    } else if (layout == DM_NT) {
      if (lean_ok && key == 9) W4_GOF(DM_NT, 10);          // fp32 C + fp32 residual: fc2 / proj forward
      else W4_GOF(DM_NT, 0);
    } else {
      W4_GOF(DM_NN, 0);
    }
    return;
  }
  if (layout == DM_TN) {
    if (lean_ok && key == 8) W4_GOT(9, false);            // fp32 slab / gradient written
    else if (lean_ok && key == 10) W4_GOT(11, false);     // fp32 gradient accumulated in place
    else W4_GOT(0, false);
  } else if (p.split_k > 1) {                             // sliced forward / dgrad: fp32 partial tile into the slab
    if (layout == DM_NT) W4_GO(DM_NT, 9); else W4_GO(DM_NN, 9);
  } else if (layout == DM_NT) {
    if (lean_ok && key == 0) W4_GO(DM_NT, 1);             // bf16 C
    else if (lean_ok && key == 9) W4_GO(DM_NT, 10);       // fp32 C + fp32 residual: fc2 / proj forward
    else if (lean_ok && key == 16) W4_GO(DM_NT, 17);      // bf16 C + bf16 aux written: fc1 forward
    else W4_GO(DM_NT, 0);
  } else {
    if (lean_ok && key == 0) W4_GO(DM_NN, 1);             // bf16 C: dgrads
    else if (lean_ok && key == 4) W4_GO(DM_NN, 5);        // bf16 C x saved GELU' (bf16 aux read): dgrad of fc2
    else W4_GO(DM_NN, 0);
  }
#undef W4_GO
#undef W4_GOF
#undef W4_GOT
}

// One launch for n independent weight gradients (dm_gemm_grouped): every product a bf16 DM_TN (plain operands or hi / lo plane pairs) with
// fp32 C and whole 256 x 192 tiles.  Two forms:
//   1  ONE K slice per tile (short contractions, all tiles within one round of the CUs): the gradient is stored / accumulated in place, no
//      slab; column sums (if wanted) written to ps[i].colsum_slab as ONE row [M]; the same `accumulate` flag for every product.
//      Measured (tools/mb_grouped_estimate.py): the four weight gradients of a block 78 -> 34 us at 1024 tokens, 142 -> 85 us at 4096.
//   2  stream-K (long contractions): see gemm_w4_streamk_kernel; needs x.ws (dm_gemm_w4_grouped_ws_bytes), column sums to x.cs_out.
// Returns the form taken, 0 = none (nothing launched).  launch = false: decide only.
long long dm_gemm_w4_grouped_ws_bytes() {
  const int cus = w4_cu_count();
  return (long long)(cus > 0 ? cus : 256) * 3 * (dmw4::TM * dmw4::TN + dmw4::TM) * 4;
}
int dm_gemm_w4_grouped(GemmParams *ps, int n, hipStream_t s, bool launch, const DmGroupedExtra &x) {
  using namespace dmw4;
  // DM_GEMM_GROUPED: 0 = off, 1 = rule (default), 2 = the one-slice form for every legal group, 3 = stream-K for every legal group
  // (read per call: the tests flip it)
  const char *menv = getenv("DM_GEMM_GROUPED");
  const int mode = menv ? atoi(menv) : 1;
  if (mode == 0 || n < 1 || n > GROUP_MAX) return 0;
  const int cus = w4_cu_count();
  if (cus <= 0) return 0;
  GemmGroupSk sk{};
  GemmGroup &grp = sk.g;
  long long tiles = 0, work = 0;
  int k_max = 0;
  bool any_cs = false, same_acc = true;
  constexpr long long LIM = (1LL << 31) / (128LL * 4);
  const bool fold = ps[0].k_fold > 0;        // hi / lo plane pairs ("bf16x3"): all members or none; the standard pattern, as in dm_gemm_w4_plan
  for (int i = 0; i < n; ++i) {
    GemmParams &p = ps[i];
    if ((p.k_fold > 0) != fold) return 0;
    const int k_eff = fold ? p.k_fold : p.K, bk_eff = fold ? 32 : BK;
    if (fold) {
      if (p.a_fold[0] != 0 || p.a_fold[1] != 0 || p.a_fold[2] <= 0 || p.b_fold[0] != 0 || p.b_fold[2] != 0 || p.b_fold[1] <= 0) return 0;
      if (p.a_fold[2] * 2 >= (1LL << 30) || p.b_fold[1] * 2 >= (1LL << 30)) return 0;
    }
    if (p.M % TM != 0 || p.N % TN != 0 || k_eff % (2 * bk_eff) != 0 || k_eff < 2 * bk_eff || p.ldc % 4 != 0) return 0;
    if (p.c_dtype != DM_F32 || p.epilogue != DM_EPI_NONE || p.bias || p.residual || p.aux || p.rows_per_group != 0) return 0;
    if (p.ldc >= LIM || p.N >= LIM) return 0;
    same_acc = same_acc && p.accumulate == ps[0].accumulate;
    if (((long long)k_eff * p.lda + (fold ? p.a_fold[2] : 0)) * 2 >= (1LL << 31) || ((long long)k_eff * p.ldb + (fold ? p.b_fold[1] : 0)) * 2 >= (1LL << 31)) return 0;
    p.tiles_m = p.M / TM;
    p.tiles_n = p.N / TN;
    p.split_k = 1;
    p.k_per_split = k_eff;       // (FOLD: in contraction positions of ONE piece)
    grp.first[i] = (int)tiles;
    sk.tile0[i] = (int)tiles;
    sk.steps[i] = k_eff / bk_eff;
    tiles += (long long)p.tiles_m * p.tiles_n;
    work += (long long)p.tiles_m * p.tiles_n * sk.steps[i];
    k_max = k_eff > k_max ? k_eff : k_max;
    any_cs = any_cs || p.colsum_slab != nullptr || x.cs_out[i] != nullptr;
  }
  if (tiles >= (1 << 20) || work >= (1LL << 30)) return 0;
  for (int i = n; i <= GROUP_MAX; ++i) { grp.first[i] = (int)tiles; sk.tile0[i] = (int)tiles; }
  grp.n = n;
  auto lds_ok = [](const void *f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess; };
  // ---- form 1: one K slice per tile.  (The contraction bound: in the step, grouping the 16384-token blocks' gradients is neutral for the
  // headline and costs config 5 0.1-0.7 % (15360 tokens) -- there every product fills the chip with its own slices; 12288 is the last
  // length the microbenchmark shows a clear gain for: 250 -> 206 us per block.)
  const bool one_ok = n >= 2 && same_acc && tiles <= cus;
  const bool sk_wanted = mode == 3;      // stream-K: measured SLOWER than the separate sliced launches at every length -- see gemm_w4_streamk_kernel
  if (one_ok && mode != 3 && (mode == 2 || ((double)tiles / cus >= 0.4 && k_max <= 12288))) {
#define DM_GRP_ATTR(EKV, FOLDV) (lds_ok(reinterpret_cast<const void *>(gemm_w4_grouped_kernel<EKV, FOLDV, true>)) && lds_ok(reinterpret_cast<const void *>(gemm_w4_grouped_kernel<EKV, FOLDV, false>)))
    static const bool attr = DM_GRP_ATTR(9, false) && DM_GRP_ATTR(11, false) && DM_GRP_ATTR(9, true) && DM_GRP_ATTR(11, true);
#undef DM_GRP_ATTR
    if (!attr) return 0;
    if (!launch) return 1;          // (plan only: the caller opens its profiler scope around the real launch)
    for (int i = 0; i < n; ++i) grp.p[i] = ps[i];
    const dim3 grid((unsigned)tiles), block(256);
    // lean epilogue keys as in dm_gemm_w4_launch: 8 = fp32 C written (EK 9), 10 = fp32 C accumulated in place (EK 11)
#define DM_GRP_GO(EKV, FOLDV) do { if (any_cs) hipLaunchKernelGGL((gemm_w4_grouped_kernel<EKV, FOLDV, true>), grid, block, LDS_BYTES, s, grp); \
    else hipLaunchKernelGGL((gemm_w4_grouped_kernel<EKV, FOLDV, false>), grid, block, LDS_BYTES, s, grp); } while (0)
    if (ps[0].accumulate) { if (fold) DM_GRP_GO(11, true); else DM_GRP_GO(11, false); }
    else { if (fold) DM_GRP_GO(9, true); else DM_GRP_GO(9, false); }
#undef DM_GRP_GO
    return 1;
  }
  // ---- form 3: the same K slices for every product of the group, (tile, slice) per workgroup, partial tiles to each product's own slab [S][M][N]
  // (the caller runs the products' splitk_reduce launches).  For products that are too small to be sliced well ALONE: the 768 x 768 proj
  // gradient of a 16384-token block has 12 tiles -- 16 slices of 16 K steps, 38 MB of slabs for a 2.4 MB result, 36 us -- but next to the
  // qkv gradient's 36 tiles the pair takes 5 slices of 52 steps on 240 workgroups, like the fc1 / fc2 gradients do alone.
  // DM_GEMM_GROUPED=4 forces it for every legal group.
  {
    int steps = sk.steps[0];
    bool same_k = true;
    for (int i = 1; i < n; ++i) same_k = same_k && sk.steps[i] == steps;
    int S = (int)(cus / (tiles > 0 ? tiles : 1));
    if (S > 16) S = 16;
    int per = S > 0 ? (steps + S - 1) / S : steps;
    per += per & 1;
    if (per < 16) per = 16;
    S = (steps + per - 1) / per;
    bool fits = same_k && n >= 2 && S >= 2 && mode != 3 && mode != 2;
    for (int i = 0; fits && i < n; ++i)
      fits = x.slab[i] != nullptr && (long long)S * ps[i].M * ps[i].N * 4 <= x.slab_bytes[i] && (!x.cs_out[i] || x.cs_region[i] != nullptr);
    if (fits && (mode == 4 || (k_max > 12288 && (double)(tiles * S) / cus >= 0.85))) {
#define DM_GRP_ATTR(EKV, FOLDV) (lds_ok(reinterpret_cast<const void *>(gemm_w4_grouped_kernel<EKV, FOLDV, true>)) && lds_ok(reinterpret_cast<const void *>(gemm_w4_grouped_kernel<EKV, FOLDV, false>)))
      static const bool attr3 = DM_GRP_ATTR(9, false) && DM_GRP_ATTR(9, true);
#undef DM_GRP_ATTR
      if (!attr3) return 0;
      int first = 0;
      for (int i = 0; i < n; ++i) {
        ps[i].split_k = S;
        ps[i].k_per_split = per * (fold ? 32 : BK);
        ps[i].workspace = x.slab[i];
        ps[i].colsum_slab = x.cs_out[i] ? x.cs_region[i] : nullptr;
        grp.first[i] = first;
        first += ps[i].tiles_m * ps[i].tiles_n * S;
      }
      for (int i = n; i <= GROUP_MAX; ++i) grp.first[i] = first;
      if (!launch) return 3;
      for (int i = 0; i < n; ++i) grp.p[i] = ps[i];
      const dim3 grid((unsigned)first), block(256);
      if (any_cs) { if (fold) hipLaunchKernelGGL((gemm_w4_grouped_kernel<9, true, true>), grid, block, LDS_BYTES, s, grp); else hipLaunchKernelGGL((gemm_w4_grouped_kernel<9, false, true>), grid, block, LDS_BYTES, s, grp); }
      else { if (fold) hipLaunchKernelGGL((gemm_w4_grouped_kernel<9, true, false>), grid, block, LDS_BYTES, s, grp); else hipLaunchKernelGGL((gemm_w4_grouped_kernel<9, false, false>), grid, block, LDS_BYTES, s, grp); }
      return 3;
    }
  }
  // ---- form 2: stream-K.  Workgroups are dealt to the products in proportion to their K steps; a product's workgroups take q (even) steps each.
  if (!sk_wanted || x.ws == nullptr) return 0;
  int G[GROUP_MAX], given = 0;
  for (int i = 0; i < n; ++i) {
    const long long w_i = (long long)ps[i].tiles_m * ps[i].tiles_n * sk.steps[i];
    G[i] = (int)(w_i * cus / work);
    if (G[i] < 1) G[i] = 1;
    given += G[i];
  }
  for (int guard = 0; given < cus && guard < 4 * cus; ++guard) {      // the remainder: to the product whose workgroups carry the most steps
    int best = 0;
    double worst = -1;
    for (int i = 0; i < n; ++i) {
      const double per = (double)ps[i].tiles_m * ps[i].tiles_n * sk.steps[i] / G[i];
      if (per > worst) { worst = per; best = i; }
    }
    ++G[best]; ++given;
  }
  if (given > cus) return 0;
  int grid = 0, slots = 2;
  for (int i = 0; i < n; ++i) {
    const long long w_i = (long long)ps[i].tiles_m * ps[i].tiles_n * sk.steps[i];
    int q = (int)((w_i + G[i] - 1) / G[i]);
    q += q & 1;
    if (q < 16 || q > 2 * sk.steps[i]) return 0;       // pieces too short to pay for their epilogues / more than three pieces per workgroup
    if (q > sk.steps[i]) slots = 3;
    sk.q[i] = q;
    grp.first[i] = grid;
    grid += (int)((w_i + q - 1) / q);
  }
  for (int i = n; i <= GROUP_MAX; ++i) grp.first[i] = grid;
  sk.slots = slots;
  sk.chunks = 4;
  const long long slab_floats = (long long)grid * slots * TM * TN, cs_floats = (long long)grid * slots * TM;
  if ((slab_floats + cs_floats) * 4 > x.ws_bytes) return 0;
#define DM_SK_ATTR(FOLDV) (lds_ok(reinterpret_cast<const void *>(gemm_w4_streamk_kernel<9, FOLDV, true>)) && lds_ok(reinterpret_cast<const void *>(gemm_w4_streamk_kernel<9, FOLDV, false>)))
  static const bool attr_sk = DM_SK_ATTR(false) && DM_SK_ATTR(true);
#undef DM_SK_ATTR
  if (!attr_sk) return 0;
  if (!launch) return 2;
  float *slab = reinterpret_cast<float *>(x.ws), *cs_slab = slab + slab_floats;
  for (int i = 0; i < n; ++i) {
    grp.p[i] = ps[i];
    grp.p[i].workspace = slab;
    grp.p[i].colsum_slab = x.cs_out[i] ? cs_slab : nullptr;
    sk.cs_out[i] = x.cs_out[i];
    sk.cs_acc[i] = x.cs_acc[i];
  }
  const dim3 block(256);
  if (any_cs) { if (fold) hipLaunchKernelGGL((gemm_w4_streamk_kernel<9, true, true>), dim3(grid), block, LDS_BYTES, s, sk); else hipLaunchKernelGGL((gemm_w4_streamk_kernel<9, false, true>), dim3(grid), block, LDS_BYTES, s, sk); }
  else { if (fold) hipLaunchKernelGGL((gemm_w4_streamk_kernel<9, true, false>), dim3(grid), block, LDS_BYTES, s, sk); else hipLaunchKernelGGL((gemm_w4_streamk_kernel<9, false, false>), dim3(grid), block, LDS_BYTES, s, sk); }
  hipLaunchKernelGGL(streamk_fixup_kernel, dim3((unsigned)(tiles * sk.chunks)), block, 0, s, sk);
  return 2;
}
