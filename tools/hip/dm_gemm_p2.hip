// STATUS (end of round 4): SHELVED EXPERIMENT, not built into libdeepmerge_hip.so (it plugged into dm_gemm through dm_gemm_p2_plan /
// dm_gemm_p2_launch; tools/dbg_p2.py, tools/dbg_p2b.py are its drivers).
//   * first form (TWO phases of 16 MFMAs per K tile, bias fetched by inline-asm loads into registers): exact on integer data on the first
//     run for every shape tried (plain, + bias, GELU + saved GELU'; ragged M; 1..6 tiles per workgroup) and SLOWER than the shipped routing:
//     16384 x 3072 x 768 plain 145 us against 82, + GELU' 260 against 107, 16384 x 2304 x 768 119 against 69.  Ablation (same shape):
//     no epilogue / no DMA / no MFMA = 73 us -- the barrier + LDS-round-trip skeleton of 72 K tiles alone costs what the whole product
//     should; + MFMA 94, + DMA 95, both 119 (they add), + epilogue passes 166.  With 64 x 64 outputs per wave a phase has too little
//     matrix work between two barriers: the 8-wave ping-pong structure needs the 128 x 64 outputs per wave that leave no room for a
//     second accumulator set.
//   * this file is the second form (ONE phase of 32 MFMAs per K tile): 170 us, 56-124 spilled registers, and WRONG for >= 3 tiles per
//     workgroup -- under that pressure the compiler spills the asm-loaded bias registers before the loads have landed (the risk named
//     in the header below).  A correct version would bring the bias in by LDS-DMA; not pursued, the structure is not competitive.
//
// Persistent 256x128x64 bf16 GEMM with DOUBLE-BUFFERED ACCUMULATORS (gfx950, round 4): the epilogue of tile r runs inside the K loop
// of tile r + 1.
//
// Why.  Round 4's measurements (profiles/r04_gemm_experiments.md): every GEMM kernel of this library moves operands
// L2 -> LDS at ~13 TB/s chip-wide (~52 GB/s per CU: the 128x128 kernel's K loops alone take 90 us for 16384 x 3072 x 768 = 1.18 GB,
// the 256x256 pipeline's 43 us = 0.59 GB), and a tile then spends about as long again storing its result with the matrix pipe idle
// (a per-CU cost: staggering CUs or staging the next tile early does not move it).  So: the biggest tile that leaves room for a
// SECOND accumulator set, and the stores of tile r issued from the load segments of tile r + 1's K loop:
//   * 8 waves = 4 (M) x 2 (N), 64 x 64 outputs per wave = 64 accumulator registers, two sets (tile parity) = 128 of the 256;
//   * operands by LDS-DMA into THREE 48 KiB K-tile buffers (A 256 x 128 B, B 128 x 128 B): K tile f + 2 is staged during K tile f
//     into the buffer last read during K tile f - 1, so no piece-level hazard table is needed and a DMA has two K tiles to land;
//     3 x 48 KiB + 8 x 2 KiB of epilogue staging = exactly 160 KiB;
//   * a K tile is two phases (the wave's columns 0..31 / 32..63, 16 MFMAs each); as in dm_gemm256.hip the two wave columns run one
//     barrier apart, so on every SIMD one wave issues MFMAs while its partner reads LDS, issues DMA -- and does epilogue work;
//   * epilogue pass p (rows 16 p .. 16 p + 15 of the wave's block, p = 0..3) of the PREVIOUS tile sits in phase 1 of K tile 4 + p of
//     the current tile: GELU / GELU' in the MFMA layout, bf16 into a wave-private swizzled 2 KiB block, whole 128-byte rows out by two
//     buffer stores (four with the saved GELU'); the drained accumulators are re-initialised with the BIAS of tile r + 2's columns
//     (the bias is an accumulator initial value here, not an epilogue addend: no bias registers beside two accumulator sets);
//   * DMA completion is a COUNTED vmcnt per K tile that includes the epilogue's stores issued in that K tile (vmcnt retires in order:
//     the stores are younger than the pieces waited for, so they are never waited for inside the loop).
// Scope of this first version: forward (NT) products whose epilogue reads nothing but the bias -- bf16 C, optional GELU with the saved
// derivative (bf16 aux): qkv and fc1 forward.  No VGPR-destination load may sit in the loop (the compiler would drain the DMA queue at
// the loop head): the bias of tile r + 2 is fetched at the start of tile r + 1 by inline-asm loads whose completion the counted waits
// of the four K tiles before its first use imply.
#include <cstdlib>
#include <type_traits>

#include "dm_common.h"
#include "dm_gemm_common.h"
#include "dm_mfma.h"

namespace dmp2 {

constexpr int TM = 256, TN = 128, BK = 64;
constexpr int A_BYTES = TM * 128;                 // 32 KiB
constexpr int B_BYTES = TN * 128;                 // 16 KiB
constexpr int BUF = A_BYTES + B_BYTES;            // 48 KiB
constexpr int STG = 16 * 128;                     // one wave's staging block: 16 rows x 64 bf16
constexpr int LDS_BYTES = 3 * BUF + 8 * STG;      // 163840

#define DMP2_DMA(rsrc, dst, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst), 16, voff, soff, 0, 0)

template <int V> using IC = std::integral_constant<int, V>;
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// XS: 0 = bf16 C only; 1 = bf16 C + bf16 aux (GELU with the saved derivative / pre-activation)
template <int XS>
__global__ __launch_bounds__(512) void gemm_p2_kernel(const GemmParams p) {
  constexpr int S = 2 + 2 * XS;                   // buffer stores of one epilogue pass
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave & 3, wc = wave >> 2;        // wave row (64 rows), wave column (64 columns) = stagger group
  const int g = lane >> 4, li = lane & 15;

  const int G = gridDim.x;
  const int L = dm_xcd_remap(blockIdx.x, G);
  const int tiles = p.tiles_m * p.tiles_n;
  const int ntile = p.K / BK;
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
    m0 = tm * TM;
    n0 = tn * TN;
  };

  // ---- DMA source offsets (k-contiguous operands; slot s of image row r holds the operand's 16-byte chunk s ^ (r & 7)) ----------
  const int srow = 8 * wave + (lane >> 3);        // image row inside a 64-row piece
  const int chunk = (lane & 7) ^ (lane >> 3);
  unsigned voA[4], voB[2];
#pragma unroll
  for (int u = 0; u < 4; ++u) voA[u] = (unsigned)(((long long)(64 * u + srow) * p.lda) * 2 + chunk * 16);
#pragma unroll
  for (int u = 0; u < 2; ++u) {                   // B image rows are ordered [column half][wave column][32]
    const int col = (srow >> 5) * 64 + u * 32 + (srow & 31);
    voB[u] = (unsigned)(((long long)col * p.ldb) * 2 + chunk * 16);
  }
  const int pieceoff = (8 * wave) * 128;          // + 64 * u * 128

  // ---- fragment offsets ---------------------------------------------------------------------------------------------------------
  const int sw0 = (g ^ (li & 7)) << 4, sw1 = ((4 + g) ^ (li & 7)) << 4;
  const int fragA = (wr * 64 + li) * 128;         // + i * 2048, + sw
  const int fragB = (wc * 32 + li) * 128;         // + NI * 8192 + jj * 2048, + sw

  const bf16_t *Ab = reinterpret_cast<const bf16_t *>(p.A), *Bb = reinterpret_cast<const bf16_t *>(p.B);
  auto make_a = [&](int m0) {
    const long long bytes = ((long long)(min(TM, p.M - m0) - 1) * p.lda + p.K) * 2;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(Ab + (long long)m0 * p.lda), 0, (int)min(bytes, 0x7fffffffLL), 0x00020000);
  };
  auto make_b = [&](int n0) {
    const long long bytes = ((long long)(TN - 1) * p.ldb + p.K) * 2;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(Bb + (long long)n0 * p.ldb), 0, (int)min(bytes, 0x7fffffffLL), 0x00020000);
  };
  int m_cur, n_cur, m_nxt = 0, n_nxt = 0;
  tile_mn(0, m_cur, n_cur);
  __amdgpu_buffer_rsrc_t rsAc = make_a(m_cur), rsBc = make_b(n_cur), rsAn = rsAc, rsBn = rsBc;
  if (n_my > 1) { tile_mn(1, m_nxt, n_nxt); rsAn = make_a(m_nxt); rsBn = make_b(n_nxt); }
  int kt = 0, r = 0, flat = 0;
  // piece u (0..3 = A, 4..5 = B) of the K tile d steps ahead
  auto stage = [&](int d, int u) {
    int kk = kt + d;
    const bool nx = kk >= ntile;
    if (nx) kk -= ntile;
    char *buf = smem + ((flat + d) % 3) * BUF;
    if (u < 4) {
      if (nx) DMP2_DMA(rsAn, buf + 64 * u * 128 + pieceoff, (int)voA[u], kk * (BK * 2));
      else DMP2_DMA(rsAc, buf + 64 * u * 128 + pieceoff, (int)voA[u], kk * (BK * 2));
    } else {
      if (nx) DMP2_DMA(rsBn, buf + A_BYTES + 64 * (u - 4) * 128 + pieceoff, (int)voB[u - 4], kk * (BK * 2));
      else DMP2_DMA(rsBc, buf + A_BYTES + 64 * (u - 4) * 128 + pieceoff, (int)voB[u - 4], kk * (BK * 2));
    }
  };

  // ---- epilogue state -------------------------------------------------------------------------------------------------------------
  char *stg = smem + 3 * BUF + wave * STG;
  const int rl = lane >> 3, c8 = lane & 7;
  int stw[4];                                      // staging write offsets of column tile j (row li): 8 bytes at chunk (2 j + (g >> 1)) ^ ((li >> 1) & 7)
#pragma unroll
  for (int j = 0; j < 4; ++j) stw[j] = li * 128 + (((2 * j + (g >> 1)) ^ ((li >> 1) & 7)) << 4) + (g & 1) * 8;
  const int str0 = rl * 128 + ((c8 ^ ((rl >> 1) & 7)) << 4);              // read offsets: rows rl and rl + 8, 16-byte chunk c8
  const int str1 = (rl + 8) * 128 + ((c8 ^ (((rl + 8) >> 1) & 7)) << 4);
  const unsigned voC = (unsigned)((rl * (int)p.ldc + c8 * 8) * 2), voC8 = (unsigned)(((rl + 8) * (int)p.ldc + c8 * 8) * 2);
  const unsigned voX = (unsigned)((rl * (int)p.ldaux + c8 * 8) * 2), voX8 = (unsigned)(((rl + 8) * (int)p.ldaux + c8 * 8) * 2);
  const int stepC = 16 * (int)p.ldc * 2, stepX = 16 * (int)p.ldaux * 2;
  auto make_c = [&](int m0, int n0) {
    const int mw = m0 + wr * 64, nw = n0 + wc * 64;
    const long long bytes = mw < p.M ? ((long long)(p.M - 1 - mw) * p.ldc + (p.N - nw)) * 2 : 0;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16_t *>(p.C) + (long long)mw * p.ldc + nw, 0, dm_epi_records(bytes), 0x00020000);
  };
  auto make_x = [&](int m0, int n0) {
    const int mw = m0 + wr * 64, nw = n0 + wc * 64;
    const long long bytes = (XS && mw < p.M) ? ((long long)(p.M - 1 - mw) * p.ldaux + (p.N - nw)) * 2 : 0;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16_t *>(p.aux) + (XS ? (long long)mw * p.ldaux + nw : 0), 0, dm_epi_records(bytes), 0x00020000);
  };
  __amdgpu_buffer_rsrc_t rsCp = make_c(m_cur, n_cur), rsXp = make_x(m_cur, n_cur);      // (previous tile's, set at the tile switch)
  // bias of the lane's columns (4 per column tile j), fetched by asm loads the compiler does not see (see the file header)
  i32x4 rsBias;                                    // (descriptor words in scalar registers for the asm loads; a null bias reads zeros)
  {
    const uintptr_t base = reinterpret_cast<uintptr_t>(p.bias);
    rsBias[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base & 0xffffffffu));
    rsBias[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((base >> 32) & 0xffffu));
    rsBias[2] = __builtin_amdgcn_readfirstlane(p.bias ? dm_epi_records((long long)p.N * 4) : 0);
    rsBias[3] = 0x00020000;
    asm volatile("s_nop 4" : "+s"(rsBias));
  }
  f32x4 bias_n[4];                                 // bias of the lane's columns for the tile AFTER the current one (set at the tile switch)
  auto load_bias = [&](int n0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned off = (unsigned)((n0 + wc * 64 + j * 16 + 4 * g) * 4);
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(bias_n[j]) : "v"(off), "s"(rsBias) : "memory");
    }
  };

  f32x4 cur[4][4], prv[4][4];                      // the tile being accumulated; the previous tile's results being drained (see the tile switch)
  u32x4 fa[8], fb[4], fb1[4];

  auto load_a = [&](const char *img) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[2 * i] = *reinterpret_cast<const u32x4 *>(img + fragA + i * 2048 + sw0);
      fa[2 * i + 1] = *reinterpret_cast<const u32x4 *>(img + fragA + i * 2048 + sw1);
    }
  };
  auto load_b = [&](u32x4 (&f)[4], const char *imgB, int ni) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      f[2 * jj] = *reinterpret_cast<const u32x4 *>(imgB + ni * 8192 + fragB + jj * 2048 + sw0);
      f[2 * jj + 1] = *reinterpret_cast<const u32x4 *>(imgB + ni * 8192 + fragB + jj * 2048 + sw1);
    }
  };
  // pass PS of the previous tile's epilogue from accumulator set `acc` (zeroed afterwards)
  auto pack4 = [](const f32x4 &v) __attribute__((always_inline)) {
    const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    return __builtin_bit_cast(u32x2, o);
  };
  auto epi_pass = [&](auto ps_tag) __attribute__((always_inline)) {
    constexpr int ps = decltype(ps_tag)::value;    // (a run-time row-tile index would put the accumulator arrays in scratch memory)
    u32x2 pk[4];                                   // the C values of the pass, packed (kept while the saved operand goes out first)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = prv[ps][j];
      if (p.epilogue == DM_EPI_GELU_GRAD) {
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float cdf, pdf;
          dm_gelu_parts_fast(v[e], cdf, pdf);
          d[e] = fmaf(v[e], pdf, cdf);
          v[e] = v[e] * cdf;
        }
        if constexpr (XS) *reinterpret_cast<u32x2 *>(stg + stw[j]) = pack4(d);
      } else if (p.epilogue == DM_EPI_GELU) {
        if constexpr (XS) *reinterpret_cast<u32x2 *>(stg + stw[j]) = pack4(v);       // aux = the pre-activation
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dm_gelu_fast(v[e]);
      }
      pk[j] = pack4(v);
    }
    if constexpr (XS) {                           // the saved operand first (non-temporal stores: not read before the backward pass)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 x0 = *reinterpret_cast<const u32x4 *>(stg + str0), x1 = *reinterpret_cast<const u32x4 *>(stg + str1);
      DM_EPI_BSTORE(x0, rsXp, voX, ps * stepX, DM_EPI_AUX_POLICY);
      DM_EPI_BSTORE(x1, rsXp, voX8, ps * stepX, DM_EPI_AUX_POLICY);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x2 *>(stg + stw[j]) = pk[j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const u32x4 c0 = *reinterpret_cast<const u32x4 *>(stg + str0), c1 = *reinterpret_cast<const u32x4 *>(stg + str1);
    DM_EPI_BSTORE(c0, rsCp, voC, ps * stepC, 0);
    DM_EPI_BSTORE(c1, rsCp, voC8, ps * stepC, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

#define DMP2_MMA(ACC, NI, FB)                                                              \
  do {                                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                          \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                       \
        mma<bf16_t>(ACC[i][(NI) * 2 + jj], fa[2 * i + ks], FB[2 * jj + ks]);               \
  } while (0)
#define DMP2_SYNC()                                      \
  do {                                                   \
    __builtin_amdgcn_s_barrier();                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
#define DMP2_END()                     \
  do {                                 \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();      \
  } while (0)

  // one tile's K loop into CUR while PREV (the previous tile's results, if `pending`) is drained
  // bias4: four bias loads were issued at the tile switch in front of this tile
  auto run_tile = [&](bool pending, bool bias4) __attribute__((always_inline)) {
    for (kt = 0; kt < ntile; ++kt, ++flat) {
      const char *imgA = smem + (flat % 3) * BUF;
      const char *imgB = imgA + A_BYTES;
      const bool st = flat + 2 < total && !(p.debug & 2);      // (debug bits, DM_P2_DEBUG: 1 no epilogue passes, 2 no DMA in the loop, 4 no MFMAs)
      // ONE phase per K tile (a first version with two -- 16 MFMAs between barriers -- spent 1.0 us per K tile in barriers and LDS round
      // trips alone, tools/dbg_p2b.py): all fragments, the six pieces of K tile flat + 2, one epilogue pass of the previous tile, the
      // counted wait that retires K tile flat + 1, then 32 MFMAs
      load_b(fb, imgB, 0);
      load_b(fb1, imgB, 1);
      __builtin_amdgcn_sched_barrier(0);
      load_a(imgA);
      if (st) { stage(2, 0); stage(2, 1); stage(2, 2); stage(2, 3); stage(2, 4); stage(2, 5); }
      const bool ep = pending && kt >= 4 && kt < 8 && !(p.debug & 1);
      if (ep) {
        if (kt == 4) epi_pass(IC<0>{});
        else if (kt == 5) epi_pass(IC<1>{});
        else if (kt == 6) epi_pass(IC<2>{});
        else epi_pass(IC<3>{});
      }
      if (flat + 1 < total) {                     // (the last K tile of the launch has nothing to wait for)
        // operations younger than the pieces of K tile flat + 1: this K tile's 6 pieces, its epilogue stores, and -- in the first K
        // tile of a tile -- the 4 bias loads of the tile switch
        if (st) { if (ep) wait_vm<6 + S>(); else if (kt == 0 && bias4) wait_vm<10>(); else wait_vm<6>(); }
        else { if (ep) wait_vm<S>(); else wait_vm<0>(); }
      }
      DMP2_SYNC();
      if (!(p.debug & 4)) {
        __builtin_amdgcn_s_setprio(1);
        DMP2_MMA(cur, 0, fb);
        DMP2_MMA(cur, 1, fb1);
        __builtin_amdgcn_s_setprio(0);
      }
      DMP2_END();
    }
  };

  // ---- prologue: the first tile's bias into the accumulators, the second tile's requested; K tiles 0 and 1 -------------------------
  load_bias(n_cur);
  wait_vm<0>();
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bias_n[j]));
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { cur[i][j] = bias_n[j]; prv[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  if (n_my > 1) load_bias(n_nxt);                  // (used at the first tile switch, >= 8 counted waits away)
#pragma unroll
  for (int u = 0; u < 6; ++u) stage(0, u);
  if (total > 1) {
#pragma unroll
    for (int u = 0; u < 6; ++u) stage(1, u);
    wait_vm<6>();
  } else {
    wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();
  if (wc == 1) __builtin_amdgcn_s_barrier();       // the second wave column runs one barrier behind the first

  bool pending = false, bias4 = false;
  while (r < n_my) {
    run_tile(pending, bias4);
    // tile switch.  The finished tile moves to the second accumulator set (64 register copies per wave and tile: ~300 cycles) and is
    // drained during the next tile; the first set restarts from the next tile's bias.  ONE copy of the K loop: with the two sets
    // swapping roles in two inlined copies the accumulators meet in phi nodes and the allocator spills hundreds of registers (tried).
    rsCp = make_c(m_cur, n_cur);
    rsXp = make_x(m_cur, n_cur);
    pending = true;
    ++r;
    bias4 = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bias_n[j]));      // (the asm-loaded values: requested a whole tile ago)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { prv[i][j] = cur[i][j]; cur[i][j] = bias_n[j]; }
    if (r < n_my) {
      m_cur = m_nxt; n_cur = n_nxt;
      rsAc = rsAn; rsBc = rsBn;
      if (r + 1 < n_my) {
        tile_mn(r + 1, m_nxt, n_nxt);
        rsAn = make_a(m_nxt); rsBn = make_b(n_nxt);
        load_bias(n_nxt);                          // for the NEXT tile switch
        bias4 = true;
      }
    }
  }
  if (wc == 0) __builtin_amdgcn_s_barrier();       // balance the stagger
  // the last tile's epilogue (nothing left to hide it under)
  if (!(p.debug & 1)) { epi_pass(IC<0>{}); epi_pass(IC<1>{}); epi_pass(IC<2>{}); epi_pass(IC<3>{}); }
}

}  // namespace dmp2

namespace {
template <int XS> bool p2_set_lds_limit() {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(dmp2::gemm_p2_kernel<XS>), hipFuncAttributeMaxDynamicSharedMemorySize, dmp2::LDS_BYTES) == hipSuccess;
}
int p2_cu_count() {
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
}  // namespace

// Decides whether the double-accumulator kernel runs this product; fills p.tiles_m / tiles_n and returns the grid size (0 = no).
// DM_GEMM_P2: 0 = off, 1 = routing rule, 2 = every legal product (read per call: tests flip it).
int dm_gemm_p2_plan(GemmParams &p, int layout, int ab_dtype) {
  using namespace dmp2;
  const char *env = getenv("DM_GEMM_P2");
  const int mode = env ? atoi(env) : 0;
  if (mode == 0 || layout != DM_NT || ab_dtype != DM_BF16) return 0;
  if (p.c_dtype != DM_BF16 || p.residual || p.accumulate || p.rows_per_group != 0 || p.split_k > 1) return 0;
  const bool xs = p.aux != nullptr;
  if (xs && !(p.aux_dtype == DM_BF16 && (p.epilogue == DM_EPI_GELU || p.epilogue == DM_EPI_GELU_GRAD))) return 0;
  if (!xs && !(p.epilogue == DM_EPI_NONE || p.epilogue == DM_EPI_GELU)) return 0;
  if (p.K % BK != 0 || p.K < 8 * BK || p.N % TN != 0 || p.ldc % 8 != 0 || (xs && p.ldaux % 8 != 0)) return 0;
  constexpr long long LIM = (1LL << 31) / (64LL * 4);
  if (p.ldc >= LIM || p.ldaux >= LIM || 256LL * p.lda * 2 >= (1LL << 31) || 128LL * p.ldb * 2 >= (1LL << 31)) return 0;
  const int tiles_m = (p.M + TM - 1) / TM, tiles_n = p.N / TN;
  const long long tiles = (long long)tiles_m * tiles_n;
  const int cus = p2_cu_count();
  if (cus <= 0) return 0;
  if (mode == 1 && tiles < 2LL * cus) return 0;       // the overlap needs several tiles per workgroup
  static const bool attr_ok = p2_set_lds_limit<0>() && p2_set_lds_limit<1>();
  if (!attr_ok) return 0;
  p.tiles_m = tiles_m;
  p.tiles_n = tiles_n;
  p.split_k = 1;
  p.k_per_split = p.K;
  { const char *d = getenv("DM_P2_DEBUG"); p.debug = (p.debug & ~7) | (d ? (atoi(d) & 7) : 0); }
  return (int)(tiles < cus ? tiles : cus);
}

void dm_gemm_p2_launch(const GemmParams &p, int grid, hipStream_t s) {
  if (p.aux) hipLaunchKernelGGL(dmp2::gemm_p2_kernel<1>, dim3(grid), dim3(512), dmp2::LDS_BYTES, s, p);
  else hipLaunchKernelGGL(dmp2::gemm_p2_kernel<0>, dim3(grid), dim3(512), dmp2::LDS_BYTES, s, p);
}
