// Attention backward, dQ pass, with 32 query rows per wave on the 32x32x16 bf16 MFMA (head dim 64, 128 < N <= 256), gfx950.
// Same structure as the forward kernel of dm_attention_q32.hip (read its header first): the query sits on the MFMA lane, so the
// per-row constants of the backward pass -- the forward's log-sum-exp and delta = rowsum(dO . O) -- are per-LANE scalars:
//   S^T[key][query]   = K . Q^T  (+ bias / scale as the C operand)      A = K rows (LDS), B = Q rows from global memory
//   dP^T[key][query]  = V . dO^T                                        A = V rows (LDS), B = dO rows from global memory
//   P^T = exp2(scale2 S^T - lse log2e),  dS^T = P^T (dP^T - delta)      4 VALU instructions per score, no maximum, no row sum
//   dQ^T[d][query]   += K^T . dS^T                                      A = K^T by transposed LDS reads, B = dS^T packed from registers
// 12 MFMAs per 32-key tile.  K is read by rows AND transposed from ONE image: 16-byte chunk index XOR
// x(key) = (((key >> 1) & 1) << 2) | ((key >> 2) & 3), conflict-free for both access patterns in the bank model of
// MI355X_MICROARCH.md (checked by script); V uses the same layout (rows only).
// Register classes: with a bias its 128 per-lane values AND the S^T tiles live in accumulator registers (the MFMA chain runs C = bias,
// D = scores there; 16 v_accvgpr_read per tile bring a finished tile to the VALU), dP^T tiles are VGPRs; Q^T / dO^T fragments and dQ^T are
// accumulator registers.  Without a bias (8-wave instances, 256 registers) everything the VALU touches is a VGPR.
// delta is computed here from the O / dO rows (v_dot2c_f32_bf16 + one lane exchange) and stored for the dK / dV pass.
#include "dm_attention_q32.h"

#ifndef DMQ_ABL
#define DMQ_ABL 0
#endif

namespace dmq32 {

// scores in accumulator registers (the bias instances): C = bias (a), D (a); A = K fragment (v), B = Q^T fragment (a)
__device__ __forceinline__ void qk_first_acc(f32x16 &d, const u32x4 &k, const u32x4 &q, const f32x16 &c) {
  asm volatile(DMQ_MFMA " %0, %1, %2, %3" : "=&a"(d) : "v"(k), "a"(q), "a"(c));
}
__device__ __forceinline__ void qk_acc_acc(f32x16 &d, const u32x4 &k, const u32x4 &q) {
  asm volatile(DMQ_MFMA " %0, %1, %2, %0" : "+a"(d) : "v"(k), "a"(q));
}
__device__ __forceinline__ void park_acc16(f32x16 &v) { asm volatile("" : "+a"(v)); }
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}

// BM: 0 no bias, 1 dense bias rows in accumulator registers (4 waves), 2 the head's relative-position table in LDS (8 waves; see the
// forward kernel: eight ds_read2_b32 per tile put bias / scale into the score registers, the S^T chain accumulates onto them).
template <int NKT, bool RAGGED, int BM, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void attn_bwd_dq_q32_kernel(const AttnPipeBwdParams p, int bchunk, int nblk, int chunks) {
  constexpr bool BIAS = BM == 1, TAB = BM == 2;
  constexpr bool DIRECT = TAB && NW == 8;                           // no LDS left for the write-back blocks: rows leave from registers
  constexpr bool RUNS = NW == 8;                                    // (head, sample) units dealt out as contiguous runs (forward kernel)
  static_assert(!TAB || (!RAGGED && NKT % 2 == 0 && NW == 8), "table form: N = 64 x scales, 8 waves");
  constexpr int ROWS = 32 * NW;
  constexpr int NP = NKT * 32;
  const int N = RAGGED ? p.N : NP;
  constexpr int IMG = NP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 buffers][K image | V image] | NW x write-back block
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int H = p.H;
  int rb = 0, u0, u1;
  if constexpr (RUNS) {
    const int units = p.B * H, G = gridDim.x, base = units / G, rem = units - base * G, w = blockIdx.x;
    u0 = w * base + min(w, rem);
    u1 = u0 + base + (w < rem ? 1 : 0);
  } else {
    int h0, chunk;
    if (!coords(nblk, H, chunks, h0, rb, chunk)) return;
    u0 = h0 * p.B + chunk * bchunk;
    u1 = h0 * p.B + min(p.B, chunk * bchunk + bchunk);
  }
  if (u0 >= u1) return;
  int h = u0 / p.B;                                                 // head of the current unit
  const int q_wave = rb * ROWS + wave * 32;
  const int q = q_wave + r;
  const bool wave_live = q_wave < N;
  const bool row_ok = q < N;
  const long long tok_stride = 3LL * H * HD;
  const bf16_t *qkv = reinterpret_cast<const bf16_t *>(p.qkv);
  const bf16_t *outp = reinterpret_cast<const bf16_t *>(p.out);
  const bf16_t *dout = reinterpret_cast<const bf16_t *>(p.dout);
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = p.scale * LOG2E;
  constexpr bool QA = BIAS;                                         // Q^T / dO^T fragments in accumulator registers
  constexpr bool PAD = NW == 8;
  static_assert(!(BIAS && NW == 8), "the bias rows need the 512-register budget of one wave per SIMD");

  // ---- C operands of each tile's first S^T MFMA: bias / scale (accumulator registers), or the key mask of the last tile ------------
  constexpr int NC = BIAS ? NKT : (RAGGED ? 1 : 0);
  f32x16 cinit[NC > 0 ? NC : 1];
  if constexpr (BIAS) {
    const float inv_scale = 1.f / p.scale;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int key = 32 * kt + 8 * c + 4 * hh;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (wave_live && row_ok && !(DMQ_ABL & 1)) {
          const float *brow = p.bias + ((long long)h * N + q) * N + key;
          if (!RAGGED) {
            v = dm_load4(brow);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (key + e < N) v[e] = brow[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) cinit[kt][4 * c + e] = (RAGGED && key + e >= N) ? NEG_BIG : v[e] * inv_scale;
      }
      park_acc16(cinit[kt]);
    }
  } else if constexpr (RAGGED) {
#pragma unroll
    for (int i = 0; i < 16; ++i) cinit[0][i] = (32 * (NKT - 1) + 8 * (i >> 2) + 4 * hh + (i & 3) >= N) ? NEG_BIG : 0.f;
  }

  // ---- DMA (waves 0..3; see the forward kernel): both images with the dual-use swizzle x(key), applied on the source chunk --------------
  const int dkey = lane >> 3;
  const unsigned rowoff0 = (unsigned)((8 * wave + dkey) * tok_stride * 2);
  const unsigned src_swz = (unsigned)(((lane & 7) ^ ((((dkey >> 1) & 1) << 2) | ((((wave & 1) << 1) | (dkey >> 2)) & 3))) * 16);
  const unsigned voffK = rowoff0 + (unsigned)(1 * H * HD * 2) + src_swz;
  const unsigned voffV = rowoff0 + (unsigned)(2 * H * HD * 2) + src_swz;
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(DM_LDS char *)smem);
  unsigned step_bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)(32 * tok_stride * 2));
  asm volatile("s_nop 4" : "+s"(step_bytes));
  auto sample_rsrc = [&](int u) -> i32x4 {
    const int hd = u / p.B, b = u - hd * p.B;
    const uintptr_t base = reinterpret_cast<uintptr_t>(qkv + (long long)b * N * tok_stride + (long long)hd * HD);
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base & 0xffffffffu));
    rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((base >> 32) & 0xffffu));
    rs[2] = __builtin_amdgcn_readfirstlane((int)(N * tok_stride * 2));
    rs[3] = 0x00020000;
    asm volatile("s_nop 4" : "+s"(rs));
    return rs;
  };
  const bool dma_wave = wave < 4;
  auto stage_piece = [&](const i32x4 &rs, int buf, int j) {
    if ((DMQ_ABL & 2) || !dma_wave) return;
    const unsigned kimg = lds0 + (unsigned)(buf * (2 * IMG)) + (unsigned)wave * 1024u, vimg = kimg + (unsigned)IMG;
    lds_dma(rs, kimg + (unsigned)j * 4096u, voffK, (unsigned)j * step_bytes);
    lds_dma(rs, vimg + (unsigned)j * 4096u, voffV, (unsigned)j * step_bytes);
  };
  auto stage_all = [&](int u, int buf) {
    if (!dma_wave) return;
    const i32x4 rs = sample_rsrc(u);
#pragma unroll
    for (int j = 0; j < NKT; ++j) stage_piece(rs, buf, j);
  };
  // this lane's rows of the next sample: Q and dO fragments (B operands: d = 16 ks + 8 hh .. + 7), the O fragment for delta, lse
  auto load_rows = [&](int u, u32x4 (&fq)[4], u32x4 (&fdo)[4], u32x4 (&fo)[4], float &lse) {
    const int hd = u / p.B, b = u - hd * p.B;
    const bool ok = wave_live && row_ok && !(DMQ_ABL & 64);
    const bf16_t *qrow = qkv + ((long long)b * N + q) * tok_stride + (long long)hd * HD + 8 * hh;
    const long long orow = ((long long)b * N + q) * H * HD + (long long)hd * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fq[ks] = ok ? *reinterpret_cast<const u32x4 *>(qrow + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
      fdo[ks] = ok ? *reinterpret_cast<const u32x4 *>(dout + orow + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
      fo[ks] = ok ? *reinterpret_cast<const u32x4 *>(outp + orow + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
    }
    lse = ok ? p.lse[((long long)b * H + hd) * N + q] : 0.f;
  };
  // table form (see the forward kernel): rows p = (dz + S - 1) * 15 + (dy + 7) of 16 floats, entry 7 - dx, pre-divided by the scale
  constexpr int TAB_MAXC = 15 * ((NKT - 1) >> 1) + 7;
  float *tab = reinterpret_cast<float *>(smem + 4 * IMG + (DIRECT ? 0 : NW * WB_WAVE));
  const float *tabl = tab;
  // (two phases, as in the forward kernel: the first table's loads fly under the first unit's transfers)
  constexpr int TAB_N = (NKT - 1) * 225, TAB_IT = (TAB_N + 64 * NW - 1) / (64 * NW);
  float tv[TAB ? TAB_IT : 1];
  auto load_table = [&](int hd) {
#pragma unroll
    for (int k = 0; k < TAB_IT; ++k) {
      const int i = t + k * 64 * NW;
      tv[k] = i < TAB_N ? p.table[(long long)i * H + hd] : 0.f;
    }
  };
  auto store_table = [&]() {                                        // every thread; a barrier must follow before the table is read
    const float inv_scale = 1.f / p.scale;
#pragma unroll
    for (int k = 0; k < TAB_IT; ++k) {
      const int i = t + k * 64 * NW;
      const int pz = i / 225, rem = i - pz * 225, py = rem / 15, px = rem - py * 15;
      if (i < TAB_N) tab[(pz * 15 + py) * TAB_PITCH + (14 - px)] = tv[k] * inv_scale;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto fill_table = [&](int hd) { load_table(hd); store_table(); };
  if constexpr (TAB) {
    const int qz = q >> 6, qy = (q >> 3) & 7, qx = q & 7;
    tabl = tab + ((qz + NKT / 2 - 1) * 15 + qy + 7 - TAB_MAXC) * TAB_PITCH + 7 - qx + 4 * hh;
  }
  auto read_bias = [&](int kt, f32x16 &d) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) d[4 * c + e] = tabl[TAB_PITCH * (TAB_MAXC - (15 * (kt >> 1) + 4 * (kt & 1) + c)) + e];
  };

  // ---- fragment offsets ---------------------------------------------------------------------------------------------------------
  const int xr = (((r >> 1) & 1) << 2) | ((r >> 2) & 3);            // x(key) of row r of a 32-key tile (32 kt does not change it)
  int roff[4];                                                      // K / V rows: + 4096 kt
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) roff[ks] = r * 128 + (((2 * ks + hh) ^ xr) << 4);
  // K^T by transposed reads: lane 4 qd + pp of a 16-lane group addresses key row 4 hh + qd (+ 8 j2 + 16 s + 32 kt), d columns
  // 32 dt + 16 e + 4 pp .. + 3; x(key) = ((qd >> 1) << 2) | ((2 j2 + hh) & 3)
  const int ve = (lane >> 4) & 1, qd = (lane >> 2) & 3, pp = lane & 3;
  int toff[2][2];                                                   // [dt][j2]: + (32 kt + 16 s) * 128
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int x = ((qd >> 1) << 2) | ((2 * j2 + hh) & 3);
      toff[dt][j2] = (8 * j2 + 4 * hh + qd) * 128 + (((4 * dt + 2 * ve + (pp >> 1)) ^ x) << 4) + 8 * (pp & 1);
    }

  char *wb = smem + 4 * IMG + wave * WB_WAVE;
  u32x4 fl_v = {0u, 0u, 0u, 0u};
  auto flush_read = [&](int k) {
    if (!wave_live || (DMQ_ABL & 32)) return;
    fl_v = *reinterpret_cast<const u32x4 *>(wb + ((lane >> 3) + 8 * k) * WB_PITCH + (lane & 7) * 16);
  };
  auto flush_store = [&](int u, int k) {                            // dQ rows of unit u: dqkv[b][q][0][h][:]
    if (!wave_live || (DMQ_ABL & 32)) return;
    const int hd = u / p.B, b = u - hd * p.B;
    bf16_t *drow0 = reinterpret_cast<bf16_t *>(p.dqkv) + ((long long)b * N + q_wave) * tok_stride + (long long)hd * HD;
    const int rr = lane >> 3, cc = lane & 7;
    if (q_wave + rr + 8 * k < N) *reinterpret_cast<u32x4 *>(drow0 + (long long)(rr + 8 * k) * tok_stride + cc * 8) = fl_v;
  };

  u32x4 qf[4], dof[4], qld[4], dold[4], old[4];
  float lse_ld = 0.f;
  stage_all(u0, 0);
  load_rows(u0, qld, dold, old, lse_ld);
  if constexpr (TAB) load_table(h);
  for (int b = u0; b < u1; ++b) {                                   // b: the unit (head * B + sample)
    const int b0 = u0, b1 = u1;
    const int buf = (b - b0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (TAB && b == u0) store_table();
    __builtin_amdgcn_s_barrier();
    if constexpr (RUNS) {
      const int hn = b / p.B;
      if (TAB && hn != h) {                                         // the run crossed into the next head
        fill_table(hn);
        __builtin_amdgcn_s_barrier();
      }
      h = hn;
    }
    const int smp = b - h * p.B;
    const bool more = b + 1 < b1;
    // ---- per-row constants of this sample: delta = rowsum(dO . O) over both lane halves, -lse in log2 units ---------------------------
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int w = 0; w < 4; ++w) dl = dot2(dold[ks][w], old[ks][w], dl);
    const float delta = half_sum(dl);
    const float nl = -lse_ld * LOG2E;
    if (wave_live && row_ok && hh == 0) p.delta[((long long)smp * H + h) * N + q] = delta;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = qld[ks];
      dof[ks] = dold[ks];
      if constexpr (QA) { park_acc(qf[ks]); park_acc(dof[ks]); }
    }
    i32x4 rs_next = {0, 0, 0, 0};
    if (more) rs_next = sample_rsrc(b + 1);
    const char *kimg = smem + buf * (2 * IMG), *vimg = kimg + IMG;

    if (wave_live) {
      f32x16 s0, s1, dp0, dp1;                                       // S^T and dP^T tiles of even / odd key tiles
      u32x4 dsb0[2], dsb1[2];                                        // packed dS^T of even / odd tiles, k-steps 0 / 1
      u32x4 kf[4], vf[4];
      u32x2 tf[8];
      f32x16 dq0, dq1;
      auto read_k = [&](int kt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const u32x4 *>(kimg + kt * 4096 + roff[ks]);
      };
      auto read_v = [&](int kt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) vf[ks] = *reinterpret_cast<const u32x4 *>(vimg + kt * 4096 + roff[ks]);
      };
      auto read_t = [&](int kt) {
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const char *a = kimg + (32 * kt + 16 * sx) * 128;
            tf[4 * sx + 2 * dt] = dm_ds_read_tr16(a + toff[dt][0]);
            tf[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(a + toff[dt][1]);
          }
      };
      auto tfrag = [&](int sx, int dt) { return (u32x4){tf[4 * sx + 2 * dt][0], tf[4 * sx + 2 * dt][1], tf[4 * sx + 2 * dt + 1][0], tf[4 * sx + 2 * dt + 1][1]}; };
      auto s_piece = [&](int kt, int ks, f32x16 &d) {
        if ((DMQ_ABL & 16) && kt > 0) return;
        if constexpr (BIAS) {
          if (ks == 0) qk_first_acc(d, kf[0], qf[0], cinit[kt]); else qk_acc_acc(d, kf[ks], qf[ks]);
        } else if constexpr (TAB) {
          qk_acc<QA, PAD>(d, kf[ks], qf[ks]);                        // d holds bias / scale (read_bias)
        } else {
          if (ks == 0) {
            if (RAGGED && kt == NKT - 1) qk_first<QA, PAD>(d, kf[0], qf[0], cinit[0]); else qk_first0<QA, PAD>(d, kf[0], qf[0]);
          } else {
            qk_acc<QA, PAD>(d, kf[ks], qf[ks]);
          }
        }
      };
      auto dp_piece = [&](int kt, int ks, f32x16 &d) {
        if ((DMQ_ABL & 16) && kt > 0) return;
        if (ks == 0) qk_first0<QA, PAD>(d, vf[0], dof[0]); else qk_acc<QA, PAD>(d, vf[ks], dof[ks]);
      };
      auto dq_piece = [&](int kt, int g, const u32x4 (&dsb)[2]) {
        const int sx = g >> 1, dt = g & 1;
        f32x16 &o = dt ? dq1 : dq0;
        if ((DMQ_ABL & 8) && kt > 0) return;
        if (kt == 0 && sx == 0) pv_first<PAD>(o, tfrag(0, dt), dsb[0]); else pv_acc<PAD>(o, tfrag(sx, dt), dsb[sx]);
      };
      // VALU piece k (0..7) of a tile: scores 2k, 2k + 1 -> P, dS = P (dP - delta), one packed pair
      auto ds_piece = [&](const f32x16 &sc, const f32x16 &dp, int k, u32x4 (&dsb)[2]) {
        if (DMQ_ABL & 4) { dsb[k >> 2][k & 3] = 0x3f803f80u; return; }
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * k], scale2, nl));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * k + 1], scale2, nl));
        const unsigned w = pk_bf16(p0 * (dp[2 * k] - delta), p1 * (dp[2 * k + 1] - delta));
        dsb[k >> 2][k & 3] = w;
        asm volatile("" :: "v"(w));                                  // this piece's arithmetic stays in ITS gap
      };

      if constexpr (NW == 4) {
        // ---- prologue: S^T and dP^T of tile 0 -------------------------------------------------------------------------------------------
        read_k(0);
        read_v(0);
        asm volatile("s_nop 1");                                       // the fragment copies into accumulator registers may be fresh
        s_piece(0, 0, s0); s_piece(0, 1, s0); s_piece(0, 2, s0); s_piece(0, 3, s0);
        dp_piece(0, 0, dp0); dp_piece(0, 1, dp0); dp_piece(0, 2, dp0); dp_piece(0, 3, dp0);
        if (NKT > 1) { read_k(1); read_v(1); }
        if constexpr (BIAS) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(s0), "+v"(dp0)); else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s0), "+v"(dp0));
  #pragma unroll
        for (int j = 0; j < NKT; ++j) {
          // iteration j, twelve MFMA gaps: S^T(j + 1) x 4, dP^T(j + 1) x 4, dQ^T(j - 1) x 4, with tile j's VALU pieces in the first eight
          f32x16 &sc = (j & 1) ? s1 : s0;
          f32x16 &sn = (j & 1) ? s0 : s1;
          f32x16 &dpc = (j & 1) ? dp1 : dp0;
          f32x16 &dpn = (j & 1) ? dp0 : dp1;
          u32x4 (&dsc)[2] = (j & 1) ? dsb1 : dsb0;
          u32x4 (&dsp)[2] = (j & 1) ? dsb0 : dsb1;
          // tile j's tiles are read below this point only: their last MFMA was issued at least four MFMAs ago (MFMA D -> VALU distance)
          if constexpr (BIAS) asm volatile("" : "+a"(sc), "+v"(dpc)); else asm volatile("" : "+v"(sc), "+v"(dpc));
          if (j > 0) read_t(j - 1);
          __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
          for (int g = 0; g < 12; ++g) {
            if (g < 4) {
              if (j + 1 < NKT) s_piece(j + 1, g, sn);
            } else if (g < 8) {
              if (j + 1 < NKT) dp_piece(j + 1, g - 4, dpn);
            } else {
              if (j > 0) dq_piece(j - 1, g - 8, dsp);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g < 8) ds_piece(sc, dpc, g, dsc);
            if (g == 3 && j + 2 < NKT) read_k(j + 2);
            if (g == 7 && j + 2 < NKT) read_v(j + 2);
            if (g == 8 && more && j == NKT - 2) load_rows(b + 1, qld, dold, old, lse_ld);       // (all S^T / dP^T MFMAs of this sample are issued)
            if (g == 1 && b > b0 && j >= NKT - 4) flush_read(j - (NKT - 4));
            if (g == 9 && b > b0 && j >= NKT - 4) flush_store(b - 1, j - (NKT - 4));
            if (g == 5 && more) {
              if (j < NKT - 1) stage_piece(rs_next, buf ^ 1, j);
              if (j == 0) stage_piece(rs_next, buf ^ 1, NKT - 1);
            }
            if (g < 4) {
              if (j + 1 < NKT) asm volatile("" :: "v"(kf[g]));
            } else if (g < 8) {
              if (j + 1 < NKT) asm volatile("" :: "v"(vf[g - 4]));
            } else {
              if (j > 0) asm volatile("" :: "v"(tf[2 * (g - 8)]), "v"(tf[2 * (g - 8) + 1]), "v"(dsp[(g - 8) >> 1]));
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // ---- epilogue: dQ^T of the last tile --------------------------------------------------------------------------------------------
        read_t(NKT - 1);
        u32x4 (&dsl)[2] = ((NKT - 1) & 1) ? dsb1 : dsb0;
        asm volatile("s_nop 1");
#pragma unroll
        for (int g = 0; g < 4; ++g) dq_piece(NKT - 1, g, dsl);
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(dq0), "+a"(dq1) : "v"(dsl[0]), "v"(dsl[1]));
      } else {
        // Two waves per SIMD: the partner wave fills this wave's gaps, so a tile runs start to end (S^T, dP^T, the VALU pieces, dQ^T)
        // with ONE score / dP / dS tile live -- the software-pipelined form above needs 276-300 registers, this one fits 256.
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
          if constexpr (TAB) read_bias(j, s0);
          read_k(j);
          read_v(j);
          read_t(j);
          if (j == 0) asm volatile("s_nop 1");
          s_piece(j, 0, s0); s_piece(j, 1, s0); s_piece(j, 2, s0); s_piece(j, 3, s0);
          dp_piece(j, 0, dp0); dp_piece(j, 1, dp0); dp_piece(j, 2, dp0); dp_piece(j, 3, dp0);
          if (!DIRECT && b > b0 && j >= NKT - 4) flush_read(j - (NKT - 4));
          if (more) {
            if (j < NKT - 1) stage_piece(rs_next, buf ^ 1, j);
            if (j == 0) stage_piece(rs_next, buf ^ 1, NKT - 1);
          }
          if (more && j == NKT - 1) load_rows(b + 1, qld, dold, old, lse_ld);
          asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s0), "+v"(dp0) : "v"(kf[3]), "v"(vf[0]), "v"(vf[1]), "v"(vf[2]), "v"(vf[3]));
#pragma unroll
          for (int k = 0; k < 8; ++k) ds_piece(s0, dp0, k, dsb0);
          if (!DIRECT && b > b0 && j >= NKT - 4) flush_store(b - 1, j - (NKT - 4));
#pragma unroll
          for (int g = 0; g < 4; ++g) dq_piece(j, g, dsb0);
          asm volatile("" :: "v"(tf[0]), "v"(tf[1]), "v"(tf[2]), "v"(tf[3]), "v"(tf[4]), "v"(tf[5]), "v"(tf[6]), "v"(tf[7]), "v"(dsb0[0]), "v"(dsb0[1]));
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(dq0), "+a"(dq1));
      }
      // ---- scale, park the rows in the wave's LDS block ---------------------------------------------------------------------------------
      const float sc_out = p.scale;
      if constexpr (DIRECT) {
        // the two halves of a row trade 8-byte pieces so that each lane owns 16 contiguous bytes (forward kernel)
        bf16_t *drow = reinterpret_cast<bf16_t *>(p.dqkv) + ((long long)smp * N + q) * tok_stride + (long long)h * HD + 8 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const f32x16 &o = dt ? dq1 : dq0;
#pragma unroll
          for (int cp = 0; cp < 2; ++cp) {
            const int c0 = 2 * cp, c1 = c0 + 1;
            u32x4 w;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const unsigned a = pk_bf16(o[4 * c0 + 2 * k] * sc_out, o[4 * c0 + 2 * k + 1] * sc_out);
              const unsigned bb = pk_bf16(o[4 * c1 + 2 * k] * sc_out, o[4 * c1 + 2 * k + 1] * sc_out);
              const auto sw = __builtin_amdgcn_permlane32_swap(a, bb, false, false);
              w[k] = (unsigned)sw[0];
              w[2 + k] = (unsigned)sw[1];
            }
            *reinterpret_cast<u32x4 *>(drow + 32 * dt + 16 * cp) = w;
          }
        }
      } else
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const u32x2 w0 = {pk_bf16(dq0[4 * c] * sc_out, dq0[4 * c + 1] * sc_out), pk_bf16(dq0[4 * c + 2] * sc_out, dq0[4 * c + 3] * sc_out)};
        const u32x2 w1 = {pk_bf16(dq1[4 * c] * sc_out, dq1[4 * c + 1] * sc_out), pk_bf16(dq1[4 * c + 2] * sc_out, dq1[4 * c + 3] * sc_out)};
        *reinterpret_cast<u32x2 *>(wb + r * WB_PITCH + (8 * c + 4 * hh) * 2) = w0;
        *reinterpret_cast<u32x2 *>(wb + r * WB_PITCH + (32 + 8 * c + 4 * hh) * 2) = w1;
      }
    } else if (more) {
      stage_all(b + 1, buf ^ 1);
      load_rows(b + 1, qld, dold, old, lse_ld);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (!DIRECT) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { flush_read(k); flush_store(u1 - 1, k); }
  }
}

// ---- dK / dV pass without a bias (ViT: vit_model.py:125-131), 32 KEYS per wave: the key sits on the MFMA lane -------------------------
//   S[query][key]  = Q . K^T - lse / scale        A = Q rows (LDS), B = K rows from global memory (lane = key), C = per-query constants
//   dP[query][key] = dO . V^T - delta             A = dO rows (LDS), B = V rows from global memory,            C = per-query constants
//   P = exp2(scale2 S),  dS = P dP                 3 VALU instructions per score + two packs
//   dV^T[d][key] += dO^T . P,  dK^T[d][key] += Q^T . dS      A by transposed reads of the SAME two images, B = P / dS packed from registers
// The per-query constants (-lse / scale, -delta; -1e30 / 0 for queries >= N) are staged per sample in a small LDS table and reach the
// accumulators as the C operand: 4 + 4 broadcast ds_read_b128 per tile, no VALU.  16 MFMAs per 32-query tile.  Two waves per SIMD
// (8 waves, 256 keys per workgroup) wherever the images fit; a tile runs start to end, the partner wave fills the gaps.
// With a bias the 16-row kernels of dm_attention_pipe.hip keep the pass: the transposed bias rows AND the dS accumulator of the
// bias-table gradient are 128 registers each per 32 keys.
template <int NKT, bool RAGGED, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void attn_bwd_dkv_q32_kernel(const AttnPipeBwdParams p, int bchunk, int nblk, int chunks) {
  constexpr int ROWS = 32 * NW;
  constexpr int NP = NKT * 32;
  const int N = RAGGED ? p.N : NP;
  constexpr int IMG = NP * 128;
  constexpr int STAT = 2 * NP * 4;                                  // [-lse / scale | -delta] per buffer
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 buffers][Q image | dO image] | [2 buffers] stat | NW x write-back block
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hh = lane >> 5;
  int h, rb, chunk;
  if (!coords(nblk, p.H, chunks, h, rb, chunk)) return;
  const int H = p.H;
  const int b0 = chunk * bchunk, b1 = min(p.B, b0 + bchunk);
  if (b0 >= b1) return;
  const int k_wave = rb * ROWS + wave * 32;
  const int key = k_wave + r;                                       // this lane's key row
  const bool wave_live = k_wave < N;
  const bool row_ok = key < N;
  const long long tok_stride = 3LL * H * HD, out_stride = (long long)H * HD;
  const bf16_t *qkv = reinterpret_cast<const bf16_t *>(p.qkv);
  const bf16_t *dout = reinterpret_cast<const bf16_t *>(p.dout);
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = p.scale * LOG2E;
  const float inv_scale = 1.f / p.scale;
  constexpr bool PAD = NW == 8;

  // ---- DMA (waves 0..3): Q rows of qkv and dO rows, both with the dual-use swizzle ---------------------------------------------------------
  const int dkey = lane >> 3;
  const unsigned src_swz = (unsigned)(((lane & 7) ^ ((((dkey >> 1) & 1) << 2) | ((((wave & 1) << 1) | (dkey >> 2)) & 3))) * 16);
  const unsigned voffQ = (unsigned)((8 * wave + dkey) * tok_stride * 2) + src_swz;
  const unsigned voffD = (unsigned)((8 * wave + dkey) * out_stride * 2) + src_swz;
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(DM_LDS char *)smem);
  unsigned stepQ = (unsigned)__builtin_amdgcn_readfirstlane((int)(32 * tok_stride * 2));
  unsigned stepD = (unsigned)__builtin_amdgcn_readfirstlane((int)(32 * out_stride * 2));
  asm volatile("s_nop 4" : "+s"(stepQ), "+s"(stepD));
  auto make_rsrc = [&](const bf16_t *base, long long bytes) -> i32x4 {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base);
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffu));
    rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    rs[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    rs[3] = 0x00020000;
    asm volatile("s_nop 4" : "+s"(rs));
    return rs;
  };
  const bool dma_wave = wave < 4;
  auto stage_piece = [&](const i32x4 &rsq, const i32x4 &rsd, int buf, int j) {
    if (!dma_wave) return;
    const unsigned qimg = lds0 + (unsigned)(buf * (2 * IMG)) + (unsigned)wave * 1024u, dimg = qimg + (unsigned)IMG;
    lds_dma(rsq, qimg + (unsigned)j * 4096u, voffQ, (unsigned)j * stepQ);
    lds_dma(rsd, dimg + (unsigned)j * 4096u, voffD, (unsigned)j * stepD);
  };
  auto rsrc_q = [&](int b) { return make_rsrc(qkv + (long long)b * N * tok_stride + (long long)h * HD, N * tok_stride * 2); };
  auto rsrc_d = [&](int b) { return make_rsrc(dout + (long long)b * N * out_stride + (long long)h * HD, N * out_stride * 2); };
  // this lane's K / V rows of a sample (B operands: d = 16 ks + 8 hh .. + 7) and, for the first 64 * 4 lanes of the workgroup, one
  // query's lse / delta for the stat table
  const int sq = (wave & 3) * 64 + lane;                            // the query whose constants this lane stages (waves 0..3)
  auto load_rows = [&](int b, u32x4 (&fk)[4], u32x4 (&fv)[4], float &lse, float &dl) {
    const bool ok = wave_live && row_ok;
    const bf16_t *krow = qkv + ((long long)b * N + key) * tok_stride + (long long)(H + h) * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fk[ks] = ok ? *reinterpret_cast<const u32x4 *>(krow + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
      fv[ks] = ok ? *reinterpret_cast<const u32x4 *>(krow + (long long)H * HD + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
    }
    const bool sok = dma_wave && sq < N;
    lse = sok ? p.lse[((long long)b * H + h) * N + sq] : 0.f;
    dl = sok ? p.delta[((long long)b * H + h) * N + sq] : 0.f;
  };

  // ---- fragment offsets (the dual-use image of the dQ kernel, rows = queries here) -------------------------------------------------------
  const int xr = (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
  int roff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) roff[ks] = r * 128 + (((2 * ks + hh) ^ xr) << 4);
  const int ve = (lane >> 4) & 1, qd = (lane >> 2) & 3, pp = lane & 3;
  int toff[2][2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int x = ((qd >> 1) << 2) | ((2 * j2 + hh) & 3);
      toff[dt][j2] = (8 * j2 + 4 * hh + qd) * 128 + (((4 * dt + 2 * ve + (pp >> 1)) ^ x) << 4) + 8 * (pp & 1);
    }
  char *stat0 = smem + 4 * IMG;
  char *wb = smem + 4 * IMG + 2 * STAT + wave * WB_WAVE;

  u32x4 kfr[4], vfr[4], kld[4], vld[4];
  float lse_ld = 0.f, dl_ld = 0.f;
  if (dma_wave) {
    const i32x4 rq = rsrc_q(b0), rd = rsrc_d(b0);
#pragma unroll
    for (int j = 0; j < NKT; ++j) stage_piece(rq, rd, 0, j);
  }
  load_rows(b0, kld, vld, lse_ld, dl_ld);
  for (int b = b0; b < b1; ++b) {
    const int buf = (b - b0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // this sample's per-query constants into its stat buffer (queries >= N: no probability, no gradient)
    if (dma_wave && sq < NP) {
      float *st = reinterpret_cast<float *>(stat0 + buf * STAT);
      st[sq] = sq < N ? -lse_ld * inv_scale : NEG_BIG;
      st[NP + sq] = sq < N ? -dl_ld : 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the table entries are in LDS before the barrier releases their readers
    __builtin_amdgcn_s_barrier();
    const bool more = b + 1 < b1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kfr[ks] = kld[ks]; vfr[ks] = vld[ks]; }
    i32x4 rq_next = {0, 0, 0, 0}, rd_next = {0, 0, 0, 0};
    if (more && dma_wave) { rq_next = rsrc_q(b + 1); rd_next = rsrc_d(b + 1); }
    const char *qimg = smem + buf * (2 * IMG), *dimg = qimg + IMG;
    const char *stat = stat0 + buf * STAT;

    if (wave_live) {
      f32x16 sc, dp;
      u32x4 pb[2], dsb[2];
      u32x4 qf[4], df[4];
      u32x2 qt[8], dtf[8];
      f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
      for (int j = 0; j < NKT; ++j) {
        // C operands: -lse / scale and -delta of this lane half's 16 queries of the tile, straight into the accumulators
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 a = *reinterpret_cast<const f32x4 *>(stat + (32 * j + 8 * c + 4 * hh) * 4);
          const f32x4 d = *reinterpret_cast<const f32x4 *>(stat + (NP + 32 * j + 8 * c + 4 * hh) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { sc[4 * c + e] = a[e]; dp[4 * c + e] = d[e]; }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          qf[ks] = *reinterpret_cast<const u32x4 *>(qimg + j * 4096 + roff[ks]);
          df[ks] = *reinterpret_cast<const u32x4 *>(dimg + j * 4096 + roff[ks]);
        }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const int o = (32 * j + 16 * sx) * 128;
            qt[4 * sx + 2 * dt] = dm_ds_read_tr16(qimg + o + toff[dt][0]);
            qt[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(qimg + o + toff[dt][1]);
            dtf[4 * sx + 2 * dt] = dm_ds_read_tr16(dimg + o + toff[dt][0]);
            dtf[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(dimg + o + toff[dt][1]);
          }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qk_acc<false, true>(sc, qf[ks], kfr[ks]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qk_acc<false, true>(dp, df[ks], vfr[ks]);
        if (more && j < NKT - 1) stage_piece(rq_next, rd_next, buf ^ 1, j);
        if (more && j == 0) stage_piece(rq_next, rd_next, buf ^ 1, NKT - 1);
        if (more && j == NKT - 1) load_rows(b + 1, kld, vld, lse_ld, dl_ld);
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sc), "+v"(dp) : "v"(qf[3]), "v"(df[0]), "v"(df[1]), "v"(df[2]), "v"(df[3]));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float p0 = __builtin_amdgcn_exp2f(sc[2 * k] * scale2), p1 = __builtin_amdgcn_exp2f(sc[2 * k + 1] * scale2);
          pb[k >> 2][k & 3] = pk_bf16(p0, p1);
          dsb[k >> 2][k & 3] = pk_bf16(p0 * dp[2 * k], p1 * dp[2 * k + 1]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int sx = g >> 1, dt = g & 1;
          const u32x4 fa = {dtf[4 * sx + 2 * dt][0], dtf[4 * sx + 2 * dt][1], dtf[4 * sx + 2 * dt + 1][0], dtf[4 * sx + 2 * dt + 1][1]};
          const u32x4 fb = {qt[4 * sx + 2 * dt][0], qt[4 * sx + 2 * dt][1], qt[4 * sx + 2 * dt + 1][0], qt[4 * sx + 2 * dt + 1][1]};
          f32x16 &dv = dt ? dv1 : dv0;
          f32x16 &dk = dt ? dk1 : dk0;
          if (j == 0 && sx == 0) { pv_first<true>(dv, fa, pb[0]); pv_first<true>(dk, fb, dsb[0]); }
          else { pv_acc<true>(dv, fa, pb[sx]); pv_acc<true>(dk, fb, dsb[sx]); }
        }
        asm volatile("" :: "v"(qt[0]), "v"(qt[1]), "v"(qt[2]), "v"(qt[3]), "v"(qt[4]), "v"(qt[5]), "v"(qt[6]), "v"(qt[7]),
                     "v"(dtf[0]), "v"(dtf[1]), "v"(dtf[2]), "v"(dtf[3]), "v"(dtf[4]), "v"(dtf[5]), "v"(dtf[6]), "v"(dtf[7]),
                     "v"(pb[0]), "v"(pb[1]), "v"(dsb[0]), "v"(dsb[1]));
      }
      asm volatile("s_nop 15\n\ts_nop 7" : "+a"(dk0), "+a"(dk1), "+a"(dv0), "+a"(dv1));
      // ---- rows of this wave's keys: dK (scaled) then dV through the wave's LDS block, eight whole 128-byte rows per store instruction ----
      const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const f32x16 &a0 = which ? dv0 : dk0;
        const f32x16 &a1 = which ? dv1 : dk1;
        const float f = which ? 1.f : p.scale;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const u32x2 w0 = {pk_bf16(a0[4 * c] * f, a0[4 * c + 1] * f), pk_bf16(a0[4 * c + 2] * f, a0[4 * c + 3] * f)};
          const u32x2 w1 = {pk_bf16(a1[4 * c] * f, a1[4 * c + 1] * f), pk_bf16(a1[4 * c + 2] * f, a1[4 * c + 3] * f)};
          *reinterpret_cast<u32x2 *>(wb + r * WB_PITCH + (8 * c + 4 * hh) * 2) = w0;
          *reinterpret_cast<u32x2 *>(wb + r * WB_PITCH + (32 + 8 * c + 4 * hh) * 2) = w1;
        }
        bf16_t *drow0 = reinterpret_cast<bf16_t *>(p.dqkv) + ((long long)b * N + k_wave) * tok_stride + (long long)((1 + which) * H + h) * HD;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u32x4 v = *reinterpret_cast<const u32x4 *>(wb + (rr + 8 * k) * WB_PITCH + cc * 16);
          if (k_wave + rr + 8 * k < N) *reinterpret_cast<u32x4 *>(drow0 + (long long)(rr + 8 * k) * tok_stride + cc * 8) = v;
        }
      }
    } else if (more) {
      if (dma_wave) {
#pragma unroll
        for (int j = 0; j < NKT; ++j) stage_piece(rq_next, rd_next, buf ^ 1, j);
      }
      load_rows(b + 1, kld, vld, lse_ld, dl_ld);
    }
  }
}

// ---- dK / dV pass WITH the relative-position bias, 32 keys per wave, head's table in LDS, one wave per SIMD ---------------------------------
// Same products as the kernel above with the key on the lane; what the bias adds, and how it is paid for:
//   * bias / scale reaches the scores as the C operand: a lane's 16 queries of a 32-query tile are 4 query rows x 4 consecutive query
//     columns of the token cube, i.e. four runs of 4 consecutive table entries in the table's natural order -- eight ds_read2_b32 per
//     tile into the score registers (the forward / dQ kernels read the x-reversed table for the same reason with the roles swapped);
//     -lse then enters as the addend of the exponent's fma (a per-query value: 16 registers per tile from the stat table), so the
//     VALU count per score is the same as without a bias;
//   * the gradient of the table needs sum_samples dS[query][key]: 128 fp32 values per lane for 256 queries.  They live in ACCUMULATOR
//     registers and are added to by the matrix pipe: H_j += E_s . dS_s (two MFMAs per tile), where dS_s is the packed bf16 dS^T operand
//     the dK product uses anyway and E_s is the 32 x 16 selection matrix that puts k-step s's 16 queries back on their rows.  (The sum
//     therefore sees dS rounded to bf16, like dK does.)  One slab [chunk][head][query][key] per workgroup chunk as before; the
//     deterministic bin reduction (dm_relpos_bias_reduce) is unchanged.
// 18 MFMAs per tile, software-pipelined like the dQ kernel's 4-wave form: iteration j issues S / dP of tile j + 1 and dV / dK / H of
// tile j - 1 with tile j's VALU work (three stages: fma, exp2, products + packs) in the gaps.
template <int NKT>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_tab_kernel(const AttnPipeBwdParams p, int bchunk, int nblk, int chunks) {
  constexpr int NW = 4, ROWS = 128;
  constexpr int NP = NKT * 32, N = NP;
  constexpr int IMG = NP * 128;
  constexpr int STAT = 2 * NP * 4;                                  // [-lse log2e | -delta] per buffer
  constexpr int TAB_ROWS = (NKT - 1) * 15;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][Q image | dO image] | [2] stat | 4 x write-back block | table
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hh = lane >> 5;
  int h, rb, chunk;
  if (!coords(nblk, p.H, chunks, h, rb, chunk)) return;
  const int H = p.H;
  const int b0 = chunk * bchunk, b1 = min(p.B, b0 + bchunk);
  if (b0 >= b1) return;
  const int k_wave = rb * ROWS + wave * 32;
  const int key = k_wave + r;
  const bool wave_live = k_wave < N;                                // (N = 192: the second workgroup's upper two waves own no keys)
  const long long tok_stride = 3LL * H * HD, out_stride = (long long)H * HD;
  const bf16_t *qkv = reinterpret_cast<const bf16_t *>(p.qkv);
  const bf16_t *dout = reinterpret_cast<const bf16_t *>(p.dout);
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = p.scale * LOG2E;

  // ---- DMA: Q rows of qkv and dO rows, both with the dual-use swizzle (kernel above) ------------------------------------------------------
  const int dkey = lane >> 3;
  const unsigned src_swz = (unsigned)(((lane & 7) ^ ((((dkey >> 1) & 1) << 2) | ((((wave & 1) << 1) | (dkey >> 2)) & 3))) * 16);
  const unsigned voffQ = (unsigned)((8 * wave + dkey) * tok_stride * 2) + src_swz;
  const unsigned voffD = (unsigned)((8 * wave + dkey) * out_stride * 2) + src_swz;
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(DM_LDS char *)smem);
  unsigned stepQ = (unsigned)__builtin_amdgcn_readfirstlane((int)(32 * tok_stride * 2));
  unsigned stepD = (unsigned)__builtin_amdgcn_readfirstlane((int)(32 * out_stride * 2));
  asm volatile("s_nop 4" : "+s"(stepQ), "+s"(stepD));
  auto make_rsrc = [&](const bf16_t *base, long long bytes) -> i32x4 {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base);
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffu));
    rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    rs[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    rs[3] = 0x00020000;
    asm volatile("s_nop 4" : "+s"(rs));
    return rs;
  };
  auto stage_piece = [&](const i32x4 &rsq, const i32x4 &rsd, int buf, int j) {
    const unsigned qimg = lds0 + (unsigned)(buf * (2 * IMG)) + (unsigned)wave * 1024u, dimg = qimg + (unsigned)IMG;
    lds_dma(rsq, qimg + (unsigned)j * 4096u, voffQ, (unsigned)j * stepQ);
    lds_dma(rsd, dimg + (unsigned)j * 4096u, voffD, (unsigned)j * stepD);
  };
  auto rsrc_q = [&](int b) { return make_rsrc(qkv + (long long)b * N * tok_stride + (long long)h * HD, (long long)N * tok_stride * 2); };
  auto rsrc_d = [&](int b) { return make_rsrc(dout + (long long)b * N * out_stride + (long long)h * HD, (long long)N * out_stride * 2); };
  const int sq = wave * 64 + lane;                                  // the query whose constants this lane stages
  auto load_rows = [&](int b, u32x4 (&fk)[4], u32x4 (&fv)[4], float &lse, float &dl) {
    const bf16_t *krow = qkv + ((long long)b * N + key) * tok_stride + (long long)(H + h) * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fk[ks] = wave_live ? *reinterpret_cast<const u32x4 *>(krow + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
      fv[ks] = wave_live ? *reinterpret_cast<const u32x4 *>(krow + (long long)H * HD + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
    }
    const bool sok = sq < N;
    lse = sok ? p.lse[((long long)b * H + h) * N + sq] : 0.f;
    dl = sok ? p.delta[((long long)b * H + h) * N + sq] : 0.f;
  };

  // ---- fragment offsets (dual-use images, rows = queries) ------------------------------------------------------------------------------------
  const int xr = (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
  int roff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) roff[ks] = r * 128 + (((2 * ks + hh) ^ xr) << 4);
  const int ve = (lane >> 4) & 1, qd = (lane >> 2) & 3, pp = lane & 3;
  int toff[2][2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int x = ((qd >> 1) << 2) | ((2 * j2 + hh) & 3);
      toff[dt][j2] = (8 * j2 + 4 * hh + qd) * 128 + (((4 * dt + 2 * ve + (pp >> 1)) ^ x) << 4) + 8 * (pp & 1);
    }
  char *stat0 = smem + 4 * IMG;
  char *wb = smem + 4 * IMG + 2 * STAT + wave * WB_WAVE;
  float *tab = reinterpret_cast<float *>(smem + 4 * IMG + 2 * STAT + NW * WB_WAVE);
  // the head's table, natural order: row (dz + S - 1) * 15 + dy + 7, entry dx + 7 -- loaded behind the first sample's transfers, written
  // to LDS in front of the first barrier (in front of the staging the fill cost its memory round trips in full)
  constexpr int TAB_N = (NKT - 1) * 225, TAB_IT = (TAB_N + 255) / 256;
  float tv[TAB_IT];
  const int kz = key >> 6, ky = (key >> 3) & 7, kx = key & 7;
  const float *tabl = tab + (wave_live ? ((NKT / 2 - 1 - kz) * 15 + 7 - ky) * TAB_PITCH + 4 * hh - kx + 7 : 0);
  // selection matrices of the table-gradient accumulation: E_s[row][k] = 1 where row = 16 s + 8 (i >> 2) + 4 hh + (i & 3), k = 8 hh + i
  u32x4 esel[2];
#pragma unroll
  for (int sx = 0; sx < 2; ++sx)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int i0 = 2 * w, i1 = 2 * w + 1;
      const unsigned lo = (r == 16 * sx + 8 * (i0 >> 2) + 4 * hh + (i0 & 3)) ? 0x3f80u : 0u;
      const unsigned hi = (r == 16 * sx + 8 * (i1 >> 2) + 4 * hh + (i1 & 3)) ? 0x3f80u : 0u;
      esel[sx][w] = lo | (hi << 16);
    }
  f32x16 hacc[NKT];                                                 // sum over this workgroup's samples of dS[query tile j][key]: accumulator registers
#pragma unroll
  for (int j = 0; j < NKT; ++j) {
#pragma unroll
    for (int i = 0; i < 16; ++i) hacc[j][i] = 0.f;
    asm volatile("" : "+a"(hacc[j]));
  }

  u32x4 kfr[4], vfr[4], kld[4], vld[4];
  float lse_ld = 0.f, dl_ld = 0.f;
  {
    const i32x4 rq = rsrc_q(b0), rd = rsrc_d(b0);
#pragma unroll
    for (int j = 0; j < NKT; ++j) stage_piece(rq, rd, 0, j);
  }
  load_rows(b0, kld, vld, lse_ld, dl_ld);
#pragma unroll
  for (int k = 0; k < TAB_IT; ++k) {
    const int i = t + k * 256;
    tv[k] = i < TAB_N ? p.table[(long long)i * H + h] : 0.f;
  }
  for (int b = b0; b < b1; ++b) {
    const int buf = (b - b0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (b == b0) {
      const float inv_scale = 1.f / p.scale;
#pragma unroll
      for (int k = 0; k < TAB_IT; ++k) {
        const int i = t + k * 256;
        const int prow = i / 15, px = i - prow * 15;
        if (i < TAB_N) tab[prow * TAB_PITCH + px] = tv[k] * inv_scale;
      }
    }
    if (sq < NP) {
      float *st = reinterpret_cast<float *>(stat0 + buf * STAT);
      st[sq] = -lse_ld * LOG2E;
      st[NP + sq] = -dl_ld;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the stat entries (and, first sample, the table) are in LDS before the barrier
    __builtin_amdgcn_s_barrier();
    const bool more = b + 1 < b1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kfr[ks] = kld[ks]; vfr[ks] = vld[ks];
      park_acc(kfr[ks]); park_acc(vfr[ks]);                         // B operands of S / dP: accumulator registers
    }
    i32x4 rq_next = {0, 0, 0, 0}, rd_next = {0, 0, 0, 0};
    if (more) { rq_next = rsrc_q(b + 1); rd_next = rsrc_d(b + 1); }
    const char *qimg = smem + buf * (2 * IMG), *dimg = qimg + IMG;
    const char *stat = stat0 + buf * STAT;

    if (wave_live) {
      f32x16 s0, s1, dp0, dp1;
      f32x16 nl;
      u32x4 pb0[2], pb1[2], dsb0[2], dsb1[2];
      u32x4 qf[4], df[4];
      u32x2 qt[8], dtf[8];
      f32x16 dk0, dk1, dv0, dv1;
      auto read_a = [&](int jt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          qf[ks] = *reinterpret_cast<const u32x4 *>(qimg + jt * 4096 + roff[ks]);
          df[ks] = *reinterpret_cast<const u32x4 *>(dimg + jt * 4096 + roff[ks]);
        }
      };
      auto read_t = [&](int jt) {
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const int o = (32 * jt + 16 * sx) * 128;
            qt[4 * sx + 2 * dt] = dm_ds_read_tr16(qimg + o + toff[dt][0]);
            qt[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(qimg + o + toff[dt][1]);
            dtf[4 * sx + 2 * dt] = dm_ds_read_tr16(dimg + o + toff[dt][0]);
            dtf[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(dimg + o + toff[dt][1]);
          }
      };
      // query (qz, qy, qx) = (jt >> 1, 4 (jt & 1) + c, 4 hh + e) for register 4 c + e
      auto read_bias = [&](int jt, f32x16 &d) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) d[4 * c + e] = tabl[TAB_PITCH * (15 * (jt >> 1) + 4 * (jt & 1) + c) + e];
      };
      auto read_stat = [&](int jt, int which, f32x16 &d) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 a = *reinterpret_cast<const f32x4 *>(stat + (which * NP + 32 * jt + 8 * c + 4 * hh) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) d[4 * c + e] = a[e];
        }
      };
      auto s_piece = [&](int ks, f32x16 &d) { qk_acc<true, false>(d, qf[ks], kfr[ks]); };
      auto dp_piece = [&](int ks, f32x16 &d) { qk_acc<true, false>(d, df[ks], vfr[ks]); };
      auto tfrag = [&](const u32x2 (&f)[8], int sx, int dt) { return (u32x4){f[4 * sx + 2 * dt][0], f[4 * sx + 2 * dt][1], f[4 * sx + 2 * dt + 1][0], f[4 * sx + 2 * dt + 1][1]}; };
      auto dv_piece = [&](int jt, int g, const u32x4 (&pb)[2]) {
        const int sx = g >> 1, dt = g & 1;
        f32x16 &o = dt ? dv1 : dv0;
        if (jt == 0 && sx == 0) pv_first<false>(o, tfrag(dtf, 0, dt), pb[0]); else pv_acc<false>(o, tfrag(dtf, sx, dt), pb[sx]);
      };
      auto dk_piece = [&](int jt, int g, const u32x4 (&dsb)[2]) {
        const int sx = g >> 1, dt = g & 1;
        f32x16 &o = dt ? dk1 : dk0;
        if (jt == 0 && sx == 0) pv_first<false>(o, tfrag(qt, 0, dt), dsb[0]); else pv_acc<false>(o, tfrag(qt, sx, dt), dsb[sx]);
      };
      // VALU pipeline of a tile: A (exponent), B (exp2), C (dS = P dP, the two packs)
      float fa[8][2], ex[8][2];
      auto stage_a = [&](const f32x16 &sc, int k) {
        fa[k][0] = __builtin_fmaf(sc[2 * k], scale2, nl[2 * k]);
        fa[k][1] = __builtin_fmaf(sc[2 * k + 1], scale2, nl[2 * k + 1]);
        asm volatile("" :: "v"(fa[k][0]), "v"(fa[k][1]));
      };
      auto stage_b = [&](int k) {
        ex[k][0] = __builtin_amdgcn_exp2f(fa[k][0]);
        ex[k][1] = __builtin_amdgcn_exp2f(fa[k][1]);
        asm volatile("" :: "v"(ex[k][0]), "v"(ex[k][1]));
      };
      auto stage_c = [&](const f32x16 &dp, int k, u32x4 (&pb)[2], u32x4 (&dsb)[2]) {
        const unsigned wp = pk_bf16(ex[k][0], ex[k][1]);
        const unsigned wd = pk_bf16(ex[k][0] * dp[2 * k], ex[k][1] * dp[2 * k + 1]);
        pb[k >> 2][k & 3] = wp;
        dsb[k >> 2][k & 3] = wd;
        asm volatile("" :: "v"(wp), "v"(wd));
      };

      // ---- prologue: S and dP of tile 0 on top of bias / scale and -delta ------------------------------------------------------------------
      read_a(0);
      read_bias(0, s0);
      read_stat(0, 1, dp0);
      read_stat(0, 0, nl);
      asm volatile("s_nop 1");                                       // the K / V fragment copies into accumulator registers may be fresh
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s_piece(ks, s0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dp_piece(ks, dp0);
      asm volatile("" :: "v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "v"(df[0]), "v"(df[1]), "v"(df[2]), "v"(df[3]));
      if (NKT > 1) { read_a(1); read_bias(1, s1); read_stat(1, 1, dp1); }
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s0), "+v"(dp0));
#pragma unroll
      for (int j = 0; j < NKT; ++j) {
        f32x16 &sc = (j & 1) ? s1 : s0;
        f32x16 &sn = (j & 1) ? s0 : s1;
        f32x16 &dpc = (j & 1) ? dp1 : dp0;
        f32x16 &dpn = (j & 1) ? dp0 : dp1;
        u32x4 (&pbc)[2] = (j & 1) ? pb1 : pb0;
        u32x4 (&pbp)[2] = (j & 1) ? pb0 : pb1;
        u32x4 (&dsc)[2] = (j & 1) ? dsb1 : dsb0;
        u32x4 (&dsp)[2] = (j & 1) ? dsb0 : dsb1;
        asm volatile("" : "+v"(sc), "+v"(dpc));                      // tile j's tiles are read below this point only (MFMA D -> VALU distance)
        if (j > 0) read_t(j - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
          if (g < 4) {
            if (j + 1 < NKT) s_piece(g, sn);
          } else if (g < 8) {
            if (j + 1 < NKT) dp_piece(g - 4, dpn);
          } else if (g < 12) {
            if (j > 0) dv_piece(j - 1, g - 8, pbp);
          } else if (g < 16) {
            if (j > 0) dk_piece(j - 1, g - 12, dsp);
          } else {
            if (j > 0) pv_acc<false>(hacc[j > 0 ? j - 1 : 0], esel[g - 16], dsp[g - 16]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (g < 8) stage_a(sc, g);
          if (g >= 1 && g < 9) stage_b(g - 1);
          if (g >= 2 && g < 10) stage_c(dpc, g - 2, pbc, dsc);
          if (g == 8 && j + 2 < NKT) { read_a(j + 2); read_bias(j + 2, sc); }       // (tile j's scores were last read in gap 7)
          if (g == 10 && j + 2 < NKT) read_stat(j + 2, 1, dpc);                      // (tile j's dP was last read in gap 9)
          if (g == 10 && j + 1 < NKT) read_stat(j + 1, 0, nl);
          if (g == 13 && more) {
            if (j < NKT - 1) stage_piece(rq_next, rd_next, buf ^ 1, j);
            if (j == 0) stage_piece(rq_next, rd_next, buf ^ 1, NKT - 1);
          }
          if (g == 14 && more && j == NKT - 1) load_rows(b + 1, kld, vld, lse_ld, dl_ld);
          if (g < 4) {
            if (j + 1 < NKT) asm volatile("" :: "v"(qf[g]));
          } else if (g < 8) {
            if (j + 1 < NKT) asm volatile("" :: "v"(df[g - 4]));
          } else if (g < 12) {
            if (j > 0) asm volatile("" :: "v"(dtf[2 * (g - 8)]), "v"(dtf[2 * (g - 8) + 1]), "v"(pbp[(g - 8) >> 1]));
          } else if (g < 16) {
            if (j > 0) asm volatile("" :: "v"(qt[2 * (g - 12)]), "v"(qt[2 * (g - 12) + 1]), "v"(dsp[(g - 12) >> 1]));
          } else {
            if (j > 0) asm volatile("" :: "v"(esel[g - 16]), "v"(dsp[g - 16]));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // ---- epilogue: dV, dK and the table-gradient rows of the last tile -------------------------------------------------------------------
      read_t(NKT - 1);
      {
        u32x4 (&pbl)[2] = ((NKT - 1) & 1) ? pb1 : pb0;
        u32x4 (&dsl)[2] = ((NKT - 1) & 1) ? dsb1 : dsb0;
        asm volatile("s_nop 1");
#pragma unroll
        for (int g = 0; g < 4; ++g) dv_piece(NKT - 1, g, pbl);
#pragma unroll
        for (int g = 0; g < 4; ++g) dk_piece(NKT - 1, g, dsl);
        pv_acc<false>(hacc[NKT - 1], esel[0], dsl[0]);
        pv_acc<false>(hacc[NKT - 1], esel[1], dsl[1]);
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(dk0), "+a"(dk1), "+a"(dv0), "+a"(dv1)
                     : "v"(pbl[0]), "v"(pbl[1]), "v"(dsl[0]), "v"(dsl[1]), "v"(esel[0]), "v"(esel[1]),
                       "v"(qt[0]), "v"(qt[1]), "v"(qt[2]), "v"(qt[3]), "v"(qt[4]), "v"(qt[5]), "v"(qt[6]), "v"(qt[7]),
                       "v"(dtf[0]), "v"(dtf[1]), "v"(dtf[2]), "v"(dtf[3]), "v"(dtf[4]), "v"(dtf[5]), "v"(dtf[6]), "v"(dtf[7]));
      }
      // ---- rows of this wave's keys: dK (scaled) then dV through the wave's LDS block ------------------------------------------------------
      const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const f32x16 &a0 = which ? dv0 : dk0;
        const f32x16 &a1 = which ? dv1 : dk1;
        const float f = which ? 1.f : p.scale;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const u32x2 w0 = {pk_bf16(a0[4 * c] * f, a0[4 * c + 1] * f), pk_bf16(a0[4 * c + 2] * f, a0[4 * c + 3] * f)};
          const u32x2 w1 = {pk_bf16(a1[4 * c] * f, a1[4 * c + 1] * f), pk_bf16(a1[4 * c + 2] * f, a1[4 * c + 3] * f)};
          *reinterpret_cast<u32x2 *>(wb + r * WB_PITCH + (8 * c + 4 * hh) * 2) = w0;
          *reinterpret_cast<u32x2 *>(wb + r * WB_PITCH + (32 + 8 * c + 4 * hh) * 2) = w1;
        }
        bf16_t *drow0 = reinterpret_cast<bf16_t *>(p.dqkv) + ((long long)b * N + k_wave) * tok_stride + (long long)((1 + which) * H + h) * HD;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u32x4 v = *reinterpret_cast<const u32x4 *>(wb + (rr + 8 * k) * WB_PITCH + cc * 16);
          *reinterpret_cast<u32x4 *>(drow0 + (long long)(rr + 8 * k) * tok_stride + cc * 8) = v;
        }
      }
    } else if (more) {
#pragma unroll
      for (int j = 0; j < NKT; ++j) stage_piece(rq_next, rd_next, buf ^ 1, j);
      load_rows(b + 1, kld, vld, lse_ld, dl_ld);
    }
  }
  // ---- the workgroup's slab rows: slab[chunk][h][query][key], 32 consecutive keys per store instruction ---------------------------------------
  if (wave_live && p.slab) {
    float *sl = p.slab + ((long long)(chunk * H + h) * N) * N + key;
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      asm volatile("s_nop 15\n\ts_nop 7" : "+a"(hacc[j]));
#pragma unroll
      for (int i = 0; i < 16; ++i) sl[(long long)(32 * j + 8 * (i >> 2) + 4 * hh + (i & 3)) * N] = hacc[j][i];
    }
  }
}

inline void grid_bwd(int B, int N, int H, int rows, int &nblk, int &chunks, int &bchunk) {
  nblk = (N + rows - 1) / rows;
  chunks = 256 / (H * nblk);
  if (chunks < 1) chunks = 1;
  if (chunks > B) chunks = B;
  bchunk = (B + chunks - 1) / chunks;
  chunks = (B + bchunk - 1) / bchunk;
}

template <int NKT, bool RAGGED, int BM, int NW> bool launch_dq(const AttnPipeBwdParams &p, hipStream_t s) {
  constexpr int LDS = 4 * NKT * 32 * 128 + (BM == 2 ? 0 : NW * WB_WAVE) + (BM == 2 ? (NKT - 1) * 15 * TAB_PITCH * 4 : 0);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dq_q32_kernel<NKT, RAGGED, BM, NW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  if (!ok) return false;
  int nblk, chunks, bchunk;
  grid_bwd(p.B, p.N, p.H, 32 * NW, nblk, chunks, bchunk);
  const int wgs = NW == 8 ? (p.B * p.H < 256 ? p.B * p.H : 256) : grid_size(nblk, p.H, chunks);      // 8 waves: one run of units per CU
  hipLaunchKernelGGL((attn_bwd_dq_q32_kernel<NKT, RAGGED, BM, NW>), dim3(wgs), dim3(64 * NW), LDS, s, p, bchunk, nblk, chunks);
  return true;
}

template <int NKT> bool launch_dq_n(const AttnPipeBwdParams &p, hipStream_t s) {
  const bool ragged = p.N != NKT * 32;
  if constexpr (NKT % 2 == 0) {
    if (p.table && !ragged) return launch_dq<NKT, false, 2, 8>(p, s);
  }
  if (p.bias) return ragged ? launch_dq<NKT, true, 1, 4>(p, s) : launch_dq<NKT, false, 1, 4>(p, s);
  if constexpr (NKT <= 7) {
    static const bool w8 = [] { const char *e = getenv("DM_ATTN_Q32_W8"); return !(e && atoi(e) == 0); }();
    if (w8) return ragged ? launch_dq<NKT, true, 0, 8>(p, s) : launch_dq<NKT, false, 0, 8>(p, s);
  }
  return ragged ? launch_dq<NKT, true, 0, 4>(p, s) : launch_dq<NKT, false, 0, 4>(p, s);
}

template <int NKT, bool RAGGED, int NW> bool launch_dkv(const AttnPipeBwdParams &p, hipStream_t s) {
  constexpr int LDS = 4 * NKT * 32 * 128 + 2 * (2 * NKT * 32 * 4) + NW * WB_WAVE;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dkv_q32_kernel<NKT, RAGGED, NW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  if (!ok) return false;
  int nblk, chunks, bchunk;
  grid_bwd(p.B, p.N, p.H, 32 * NW, nblk, chunks, bchunk);
  hipLaunchKernelGGL((attn_bwd_dkv_q32_kernel<NKT, RAGGED, NW>), dim3(grid_size(nblk, p.H, chunks)), dim3(64 * NW), LDS, s, p, bchunk, nblk, chunks);
  return true;
}

template <int NKT> bool launch_dkv_tab(const AttnPipeBwdParams &p, hipStream_t s) {
  constexpr int LDS = 4 * NKT * 32 * 128 + 2 * (2 * NKT * 32 * 4) + 4 * WB_WAVE + (NKT - 1) * 15 * TAB_PITCH * 4;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dkv_tab_kernel<NKT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  if (!ok) return false;
  int nblk, chunks, bchunk;
  grid_bwd(p.B, p.N, p.H, 128, nblk, chunks, bchunk);            // the same chunks as dm_attn_bwd_pipe_chunks: one slab per chunk
  hipLaunchKernelGGL((attn_bwd_dkv_tab_kernel<NKT>), dim3(grid_size(nblk, p.H, chunks)), dim3(256), LDS, s, p, bchunk, nblk, chunks);
  return true;
}

template <int NKT> bool launch_dkv_n(const AttnPipeBwdParams &p, hipStream_t s) {
  const bool ragged = p.N != NKT * 32;
  if constexpr (NKT <= 7) return ragged ? launch_dkv<NKT, true, 8>(p, s) : launch_dkv<NKT, false, 8>(p, s);
  else return ragged ? launch_dkv<NKT, true, 4>(p, s) : launch_dkv<NKT, false, 4>(p, s);
}

}  // namespace dmq32

// dK / dV pass without a bias (reads p.delta, written by a dQ pass): bf16, head dim 64, 128 < N <= 256; true if it took the call.
bool dm_attn_bwd_dkv_q32(const AttnPipeBwdParams &p, hipStream_t s) {
  static const int mode = [] { const char *e = getenv("DM_ATTN_Q32_BWD"); return e ? atoi(e) : 1; }();
  if (mode == 0 || mode == 3) return false;                       // 3: new dQ only (A/B runs)
  if (p.table && p.N == 64 * p.cube_s && (p.cube_s == 3 || p.cube_s == 4) && (mode == 2 || p.B * p.H >= 96) &&
      (long long)p.N * 3 * p.H * 64 * 2 < (1LL << 31)) {
    static const bool tabkv = [] { const char *e = getenv("DM_ATTN_Q32_TABKV"); return !(e && atoi(e) == 0); }();      // A/B switch
    if (tabkv) return p.cube_s == 3 ? dmq32::launch_dkv_tab<6>(p, s) : dmq32::launch_dkv_tab<8>(p, s);
  }
  if (p.bias || p.slab) return false;
  if (p.N <= 128 || p.N > 256) return false;
  if ((long long)p.N * 3 * p.H * 64 * 2 >= (1LL << 31)) return false;
  if (mode != 2 && p.B * p.H < 96) return false;
  switch ((p.N + 31) / 32) {
    case 5: return dmq32::launch_dkv_n<5>(p, s);
    case 6: return dmq32::launch_dkv_n<6>(p, s);
    case 7: return dmq32::launch_dkv_n<7>(p, s);
    case 8: return dmq32::launch_dkv_n<8>(p, s);
    default: return false;
  }
}

bool dm_attn_bwd_tab_takes(const AttnPipeBwdParams &p) {
  static const int mode = [] { const char *e = getenv("DM_ATTN_Q32_BWD"); return e ? atoi(e) : 1; }();
  static const bool tabkv = [] { const char *e = getenv("DM_ATTN_Q32_TABKV"); return !(e && atoi(e) == 0); }();
  if (mode == 0 || mode == 3 || !tabkv || !p.table) return false;
  if (p.N != 64 * p.cube_s || (p.cube_s != 3 && p.cube_s != 4)) return false;
  if ((long long)p.N * 3 * p.H * 64 * 2 >= (1LL << 31)) return false;
  if (mode != 2 && p.B * p.H < 96) return false;
  return dm_attn_bwd_pipe_ok(p);      // attention_bwd only reaches the 32-row kernels behind this gate (DM_ATTN_PIPE, B * H)
}

// dQ pass (+ delta) of the backward: bf16, head dim 64, 128 < N <= 256.  The dK / dV pass that follows reads p.delta.
// true if it took the call; DM_ATTN_Q32_BWD=0 keeps the 16-row pipelined dQ kernel (A/B runs).
bool dm_attn_bwd_dq_q32(const AttnPipeBwdParams &p, hipStream_t s) {
  static const int mode = [] { const char *e = getenv("DM_ATTN_Q32_BWD"); return e ? atoi(e) : 1; }();
  if (mode == 0) return false;
  if (p.N <= 128 || p.N > 256) return false;
  if ((long long)p.N * 3 * p.H * 64 * 2 >= (1LL << 31)) return false;
  if (mode != 2 && p.B * p.H < 96) return false;
  if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15u)) return false;
  if (p.table && (p.N != 64 * p.cube_s || (p.cube_s != 3 && p.cube_s != 4))) return false;      // (the dense rows take the pass if given)
  switch ((p.N + 31) / 32) {
    case 5: return dmq32::launch_dq_n<5>(p, s);
    case 6: return dmq32::launch_dq_n<6>(p, s);
    case 7: return dmq32::launch_dq_n<7>(p, s);
    case 8: return dmq32::launch_dq_n<8>(p, s);
    default: return false;
  }
}
