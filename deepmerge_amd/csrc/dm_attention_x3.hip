// Attention for the "bf16x3" numerics mode (fp32 tensors, products on the bf16 matrix pipe as split-bf16 triples), head dim 64,
// 128 < N <= 256, gfx950.  Same structure as the 32-rows-per-wave kernels of dm_attention_q32*.hip -- read those headers first.
//
// An fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi); a product a.b is hi.hi + lo.hi + hi.lo (the lo.lo term, 2^-18
// relative, is dropped) -- three MFMAs into one fp32 accumulator, the same rule as dm_split_bf16's [Ah | Ah | Al] . [Bh | Bl | Bh].
// The qkv rows are split ONCE per forward pass by `split2_kernel` into two bf16 tensors of qkv's own layout (the caller keeps them
// for the backward pass), so the LDS-DMA staging of the bf16 kernels applies unchanged, to twice the images:
//   [K hi | K lo | V hi | V lo], 128 KiB at N = 256 -- single-buffered; a (head, sample) unit stages, waits, computes.
// The probabilities are split in registers (three VALU instructions per pair more than the bf16 kernel's pack).
// Scores, softmax statistics and the output are fp32 throughout; the bias enters as the scores' C operand from the head's table in
// LDS (token cubes (3 | 4, 8, 8)) or is absent (ViT); other bias forms stay on the generic fp32 kernels.
// One wave per SIMD (4 waves x 32 query rows): 28 MFMAs per 32-key tile with the tile's VALU work in their gaps.
#include "dm_attention_q32.h"
#include "dm_attention_x3.h"

namespace dmx3 {
using namespace dmq32;

// x -> hi = bf16(x), lo = bf16(x - hi); n4 groups of four
// four consecutive elements of dqkv: fp32, or -- AttnX3BwdParams::dqkv_pair -- as the hi / lo plane pair the qkv weight / data gradient
// products of the "bf16x3" mode read (no fp32 dqkv, no split pass)
__device__ __forceinline__ void x3_store_grad(const AttnX3BwdParams &p, long long off, const f32x4 &v) {
  if (p.dqkv_pair) {
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = (bf16_t)v[e];
      l[e] = (bf16_t)(v[e] - (float)h[e]);
    }
    *reinterpret_cast<bf16x4 *>(p.dqkv_pair + off) = h;
    *reinterpret_cast<bf16x4 *>(p.dqkv_pair + p.pair_plane + off) = l;
  } else {
    dm_store4(p.dqkv + off, v);
  }
}

__global__ __launch_bounds__(256) void split2_kernel(const float *__restrict__ x, bf16_t *__restrict__ hi, bf16_t *__restrict__ lo, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = dm_load4(x + 4 * i);
    u32x2 wh, wl;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned h2 = pk_bf16(v[2 * k], v[2 * k + 1]);
      const float h0 = __builtin_bit_cast(float, h2 << 16), h1 = __builtin_bit_cast(float, h2 & 0xffff0000u);
      wh[k] = h2;
      wl[k] = pk_bf16(v[2 * k] - h0, v[2 * k + 1] - h1);
    }
    *reinterpret_cast<u32x2 *>(hi + 4 * i) = wh;
    *reinterpret_cast<u32x2 *>(lo + 4 * i) = wl;
  }
}

__device__ __forceinline__ unsigned split_lo(float a, float b, unsigned h2) {      // the lo pair of (a, b) given their packed hi pair
  const float h0 = __builtin_bit_cast(float, h2 << 16), h1 = __builtin_bit_cast(float, h2 & 0xffff0000u);
  return pk_bf16(a - h0, b - h1);
}

// NKT: 32-key tiles; RAGGED: N < 32 NKT (no table then); TAB: p.table != NULL (N = 64 cube_s)
// NW = 4: one wave per SIMD, a workgroup owns a 128-row block of one head for a chunk of samples (`coords`).  NW = 8 (round 4): two waves
// per SIMD on the SAME K / V images (the register budget of 242 fits twice; one wave's exp / split VALU work runs under the other's MFMAs,
// and a unit's K / V are staged once instead of once per row block); the (head, sample) units u = head * B + sample are dealt out as
// gridDim.x contiguous runs of equal length +- 1, which may cross a head boundary (the table is reloaded there).
template <int NKT, bool RAGGED, bool TAB, int NW = 4>
__global__ __launch_bounds__(64 * NW, 1) void attn_fwd_x3_kernel(const AttnX3Params p, int bchunk, int nblk, int chunks) {
  static_assert(!(TAB && (RAGGED || NKT % 2)), "table form: N = 64 x scales");
  constexpr int NP = NKT * 32;
  const int N = RAGGED ? p.N : NP;
  constexpr int IMG = NP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // K hi | K lo | V hi | V lo | table
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int H = p.H;
  int rb = 0, u0, u1;
  if constexpr (NW == 8) {
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
  int h = -1;                                                       // head of the unit in progress (its table is in LDS)
  constexpr bool PAD = NW == 8;      // (256-register budget: operands may come fresh from v_accvgpr moves -- every asm MFMA opens with a wait state)
  const int q_wave = rb * (32 * NW) + wave * 32;
  const int q = q_wave + r;
  const bool wave_live = q_wave < N;
  const bool row_ok = q < N;
  const long long tok_stride = 3LL * H * HD;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  const float scale2 = p.scale * LOG2E;

  // ---- table (forward form: x reversed; dm_attention_q32.hip) ---------------------------------------------------------------------------
  constexpr int TAB_MAXC = 15 * ((NKT - 1) >> 1) + 7;
  float *tab = reinterpret_cast<float *>(smem + 4 * IMG);
  const float *tabl = tab;
  if constexpr (TAB) {
    const int qz = q >> 6, qy = (q >> 3) & 7, qx = q & 7;
    tabl = tab + ((qz + NKT / 2 - 1) * 15 + qy + 7 - TAB_MAXC) * TAB_PITCH + 7 - qx + 4 * hh;
  }
  auto load_table = [&](int hd) {                                   // (called between two barriers: nobody reads the old table any more)
    if constexpr (TAB) {
      const float inv_scale = 1.f / p.scale;
      for (int i = t; i < (NKT - 1) * 225; i += 64 * NW) {
        const int prow = i / 15, px = i - prow * 15;
        tab[prow * TAB_PITCH + (14 - px)] = p.table[(long long)i * H + hd] * inv_scale;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  auto init_c = [&](int kt, f32x16 &d) {                            // what the tile's score chain accumulates onto
    if constexpr (TAB) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) d[4 * c + e] = tabl[TAB_PITCH * (TAB_MAXC - (15 * (kt >> 1) + 4 * (kt & 1) + c)) + e];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) d[i] = (RAGGED && kt == NKT - 1 && 32 * kt + 8 * (i >> 2) + 4 * hh + (i & 3) >= N) ? NEG_BIG : 0.f;
      asm volatile("" : "+v"(d));
    }
  };

  // ---- DMA: the forward kernel's K / V piece layout, for the hi and the lo tensor ---------------------------------------------------------
  // (a 32-key tile is four 8-row pieces per image; 4 waves: wave w stages piece w of all four images; 8 waves: waves 0..3 the K images,
  // waves 4..7 the V images)
  const int dkey = lane >> 3;
  const int pw = wave & 3;                                          // the 8-row piece of a tile this wave stages
  const unsigned rowoff0 = (unsigned)((8 * pw + dkey) * tok_stride * 2);
  const unsigned voffK = rowoff0 + (unsigned)(1 * H * HD * 2) + (unsigned)(((lane & 7) ^ (((pw & 1) << 2) | (dkey >> 1))) * 16);
  const unsigned voffV = rowoff0 + (unsigned)(2 * H * HD * 2) + (unsigned)(((lane & 7) ^ (((dkey >> 1) & 1) << 2)) * 16);
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(DM_LDS char *)smem);
  unsigned step_bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)(32 * tok_stride * 2));
  asm volatile("s_nop 4" : "+s"(step_bytes));
  auto sample_rsrc = [&](const bf16_t *src, int b) -> i32x4 {
    const uintptr_t base = reinterpret_cast<uintptr_t>(src + (long long)b * N * tok_stride + (long long)h * HD);
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base & 0xffffffffu));
    rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((base >> 32) & 0xffffu));
    rs[2] = __builtin_amdgcn_readfirstlane((int)(N * tok_stride * 2));
    rs[3] = 0x00020000;
    asm volatile("s_nop 4" : "+s"(rs));
    return rs;
  };
  auto stage_all = [&](int b) {
    const i32x4 rh = sample_rsrc(p.hi, b), rl = sample_rsrc(p.lo, b);
    const unsigned base = lds0 + (unsigned)pw * 1024u;
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      if (NW == 4 || wave < 4) {
        lds_dma(rh, base + 0 * IMG + (unsigned)j * 4096u, voffK, (unsigned)j * step_bytes);
        lds_dma(rl, base + 1 * IMG + (unsigned)j * 4096u, voffK, (unsigned)j * step_bytes);
      }
      if (NW == 4 || wave >= 4) {
        lds_dma(rh, base + 2 * IMG + (unsigned)j * 4096u, voffV, (unsigned)j * step_bytes);
        lds_dma(rl, base + 3 * IMG + (unsigned)j * 4096u, voffV, (unsigned)j * step_bytes);
      }
    }
  };
  auto load_q = [&](int b, u32x4 (&fh)[4], u32x4 (&fl)[4]) {
    const long long off = ((long long)b * N + q) * tok_stride + (long long)h * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fh[ks] = (wave_live && row_ok) ? *reinterpret_cast<const u32x4 *>(p.hi + off + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
      fl[ks] = (wave_live && row_ok) ? *reinterpret_cast<const u32x4 *>(p.lo + off + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
    }
  };
  const int kx = (r >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((2 * ks + hh) ^ kx) << 4);
  const int ve = (lane >> 4) & 1, qd = (lane >> 2) & 3, pp = lane & 3;
  int voff[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) voff[dt] = (4 * hh + qd) * 128 + ((dt ^ ((qd >> 1) & 1)) << 6) + ve * 32 + pp * 8;

  constexpr float RESCALE_LOG2 = 16.f;
  u32x4 qh[4], ql[4];
  for (int u = u0; u < u1; ++u) {
    const int hu = u / p.B, b = u - hu * p.B;
    __builtin_amdgcn_s_barrier();                                   // everyone is done with the previous unit's images and table
    if (hu != h) { h = hu; load_table(h); }                         // (uniform: the first unit, or the run crossed into the next head)
    stage_all(b);
    load_q(b, qh, ql);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { park_acc(qh[ks]); park_acc(ql[ks]); }
    if (!wave_live) continue;
    const char *kh = smem, *kl = smem + IMG, *vh = smem + 2 * IMG, *vl = smem + 3 * IMG;

    f32x16 s0, s1;
    u32x4 pbh0[2], pbh1[2], pbl0[2], pbl1[2];                        // packed P^T hi / lo of even / odd tiles, k-steps 0 / 1
    u32x4 kfh[4], kfl[4];
    u32x2 vfh[8], vfl[8];
    f32x16 o0, o1, la;
    u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    asm volatile("" : "+v"(ones));
    float m = 0.f, msc = 0.f, alpha = 1.f;
    auto read_k = [&](int kt) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kfh[ks] = *reinterpret_cast<const u32x4 *>(kh + kt * 4096 + koff[ks]);
        kfl[ks] = *reinterpret_cast<const u32x4 *>(kl + kt * 4096 + koff[ks]);
      }
    };
    auto read_v = [&](int kt) {
#pragma unroll
      for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int o = (32 * kt + 16 * sx) * 128 + voff[dt];
          vfh[4 * sx + 2 * dt] = dm_ds_read_tr16(vh + o);
          vfh[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(vh + o + 8 * 128);
          vfl[4 * sx + 2 * dt] = dm_ds_read_tr16(vl + o);
          vfl[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(vl + o + 8 * 128);
        }
    };
    auto vfrag = [&](const u32x2 (&f)[8], int sx, int dt) { return (u32x4){f[4 * sx + 2 * dt][0], f[4 * sx + 2 * dt][1], f[4 * sx + 2 * dt + 1][0], f[4 * sx + 2 * dt + 1][1]}; };
    // piece pi (0..11) of a tile's score chain: k-step pi / 3, term pi % 3 = (K hi, Q hi), (K lo, Q hi), (K hi, Q lo)
    auto qk_piece = [&](int pi, f32x16 &d) {
      const int ks = pi / 3, term = pi % 3;
      qk_acc<true, PAD>(d, term == 1 ? kfl[ks] : kfh[ks], term == 2 ? ql[ks] : qh[ks]);
    };
    // piece pi (0..11) of a tile's P.V: (k-step, d tile) = (pi / 6, (pi / 3) % 2), term pi % 3 = (V hi, P hi), (V lo, P hi), (V hi, P lo)
    auto pv_piece = [&](int kt, int pi, const u32x4 (&ph)[2], const u32x4 (&pl)[2]) {
      const int sx = pi / 6, dt = (pi / 3) % 2, term = pi % 3;
      f32x16 &o = dt ? o1 : o0;
      const u32x4 a = term == 1 ? vfrag(vfl, sx, dt) : vfrag(vfh, sx, dt);
      const u32x4 &bq = term == 2 ? pl[sx] : ph[sx];
      if (kt == 0 && sx == 0 && term == 0) pv_first<PAD>(o, a, bq); else pv_acc<PAD>(o, a, bq);
    };
    auto l_piece = [&](int kt, int pi, const u32x4 (&ph)[2], const u32x4 (&pl)[2]) {      // pi 0..3: hi k-steps, lo k-steps
      const u32x4 &bq = pi < 2 ? ph[pi] : pl[pi - 2];
      if (kt == 0 && pi == 0) l_first<PAD>(la, ones, bq); else l_acc<PAD>(la, ones, bq);
    };
    float fa[8][2], ex[8][2];
    unsigned wh[8];
    auto stage_f = [&](const f32x16 &sc, int k) {
      fa[k][0] = __builtin_fmaf(sc[2 * k], scale2, msc);
      fa[k][1] = __builtin_fmaf(sc[2 * k + 1], scale2, msc);
      asm volatile("" :: "v"(fa[k][0]), "v"(fa[k][1]));
    };
    auto stage_e = [&](int k) {
      ex[k][0] = __builtin_amdgcn_exp2f(fa[k][0]);
      ex[k][1] = __builtin_amdgcn_exp2f(fa[k][1]);
      asm volatile("" :: "v"(ex[k][0]), "v"(ex[k][1]));
    };
    auto stage_c = [&](int k, u32x4 (&ph)[2]) {
      wh[k] = pk_bf16(ex[k][0], ex[k][1]);
      ph[k >> 2][k & 3] = wh[k];
      asm volatile("" :: "v"(wh[k]));
    };
    auto stage_l = [&](int k, u32x4 (&pl)[2]) {
      const unsigned w = split_lo(ex[k][0], ex[k][1], wh[k]);
      pl[k >> 2][k & 3] = w;
      asm volatile("" :: "v"(w));
    };
    float t0 = 0.f, t1 = 0.f;
    auto stage_m = [&](f32x16 &sn, int c) {
      if (c == 0) {
        asm volatile("" : "+v"(sn));
        t0 = max3(sn[0], sn[1], sn[2]); t1 = max3(sn[3], sn[4], sn[5]);
      } else if (c == 1) {
        t0 = max3(t0, sn[6], sn[7]); t1 = max3(t1, sn[8], sn[9]);
      } else if (c == 2) {
        t0 = max3(t0, sn[10], sn[11]); t1 = max3(t1, sn[12], sn[13]);
      } else {
        t0 = max3(t0, sn[14], sn[15]);
      }
      asm volatile("" :: "v"(t0), "v"(t1));
    };
    auto tile_max = [&](int kt) -> bool {
      const float th = __builtin_fmaxf(t0, t1);
      if (kt == 0) {
        m = half_max(th);
        msc = -m * scale2;
        return false;
      }
      const bool grow = (th - m) * scale2 > RESCALE_LOG2;
      if (__builtin_amdgcn_ballot_w64(grow) == 0) return false;
      const float mn = __builtin_fmaxf(m, half_max(th));
      alpha = __builtin_amdgcn_exp2f((m - mn) * scale2);
      m = mn;
      msc = -m * scale2;
      return true;
    };
    auto rescale_o = [&]() {
      asm volatile("s_nop 15\n\ts_nop 7" : "+a"(o0), "+a"(o1), "+a"(la));
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; la[i] *= alpha; }
      asm volatile("s_nop 3" : "+a"(o0), "+a"(o1), "+a"(la));
    };

    // ---- prologue: tile 0's scores and their maximum ----------------------------------------------------------------------------------------
    init_c(0, s0);
    if (NKT > 1) init_c(1, s1);
    read_k(0);
    asm volatile("s_nop 1");
#pragma unroll
    for (int pi = 0; pi < 12; ++pi) qk_piece(pi, s0);
    asm volatile("" :: "v"(kfh[0]), "v"(kfh[1]), "v"(kfh[2]), "v"(kfh[3]), "v"(kfl[0]), "v"(kfl[1]), "v"(kfl[2]), "v"(kfl[3]));
    if (NKT > 1) read_k(1);
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s0));
    stage_m(s0, 0); stage_m(s0, 1); stage_m(s0, 2); stage_m(s0, 3);
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      // iteration j, 28 MFMA gaps: scores(j + 1) x 12, row sums(j - 1) x 4, P.V(j - 1) x 12, with tile j's VALU pipeline between them
      f32x16 &sc = (j & 1) ? s1 : s0;
      f32x16 &sn = (j & 1) ? s0 : s1;
      u32x4 (&phc)[2] = (j & 1) ? pbh1 : pbh0;
      u32x4 (&plc)[2] = (j & 1) ? pbl1 : pbl0;
      u32x4 (&php)[2] = (j & 1) ? pbh0 : pbh1;
      u32x4 (&plp)[2] = (j & 1) ? pbl0 : pbl1;
      const bool resc = tile_max(j);
      if (j > 0) read_v(j - 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 28; ++g) {
        if (g < 12) {
          if (j + 1 < NKT) qk_piece(g, sn);
        } else if (g < 16) {
          if (j > 0) l_piece(j - 1, g - 12, php, plp);
        } else {
          if (j > 0) pv_piece(j - 1, g - 16, php, plp);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g < 16 && (g & 1) == 0) stage_f(sc, g >> 1);
        if (g < 16 && (g & 1) == 1) stage_e(g >> 1);
        if (g >= 3 && g < 19 && (g & 1) == 1) stage_c((g - 3) >> 1, phc);
        if (g >= 4 && g < 20 && (g & 1) == 0) stage_l((g - 4) >> 1, plc);
        if (g >= 20 && g < 24 && j + 1 < NKT) stage_m(sn, g - 20);
        if (g == 12 && j + 2 < NKT) read_k(j + 2);
        if (g == 15 && j + 2 < NKT) init_c(j + 2, sc);                // (tile j's scores were last read in gap 14)
        if (g < 12) {
          if (j + 1 < NKT) asm volatile("" :: "v"(kfh[g / 3]), "v"(kfl[g / 3]));
        } else if (g < 16) {
          if (j > 0) asm volatile("" :: "v"(php[0]), "v"(php[1]), "v"(plp[0]), "v"(plp[1]), "v"(ones));
        } else {
          if (j > 0) {
            const int pi = g - 16, sx = pi / 6, dt = (pi / 3) % 2;
            asm volatile("" :: "v"(vfh[4 * sx + 2 * dt]), "v"(vfh[4 * sx + 2 * dt + 1]), "v"(vfl[4 * sx + 2 * dt]), "v"(vfl[4 * sx + 2 * dt + 1]),
                         "v"(php[sx]), "v"(plp[sx]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (resc) rescale_o();
    }
    // ---- epilogue: row sums and P.V of the last tile, normalise, store ------------------------------------------------------------------------
    read_v(NKT - 1);
    {
      u32x4 (&phl)[2] = ((NKT - 1) & 1) ? pbh1 : pbh0;
      u32x4 (&pll)[2] = ((NKT - 1) & 1) ? pbl1 : pbl0;
      asm volatile("s_nop 1");
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) l_piece(NKT - 1, pi, phl, pll);
#pragma unroll
      for (int pi = 0; pi < 12; ++pi) pv_piece(NKT - 1, pi, phl, pll);
      asm volatile("s_nop 15\n\ts_nop 7" : "+a"(o0), "+a"(o1), "+a"(la)
                   : "v"(phl[0]), "v"(phl[1]), "v"(pll[0]), "v"(pll[1]), "v"(ones),
                     "v"(vfh[0]), "v"(vfh[1]), "v"(vfh[2]), "v"(vfh[3]), "v"(vfh[4]), "v"(vfh[5]), "v"(vfh[6]), "v"(vfh[7]),
                     "v"(vfl[0]), "v"(vfl[1]), "v"(vfl[2]), "v"(vfl[3]), "v"(vfl[4]), "v"(vfl[5]), "v"(vfl[6]), "v"(vfl[7]));
    }
    const float l = la[0];
    const float inv = 1.f / l;
    if (row_ok) {
      float *orow = p.out + ((long long)b * N + q) * H * HD + (long long)h * HD + 4 * hh;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const f32x16 &o = dt ? o1 : o0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v = {o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv};
          dm_store4(orow + 32 * dt + 8 * c, v);
          if (p.out_pair) {      // the projection's left operand ("bf16x3" plane pairs): no split pass over `out`
            bf16x4 h4, l4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              h4[e] = (bf16_t)v[e];
              l4[e] = (bf16_t)(v[e] - (float)h4[e]);
            }
            bf16_t *pr = p.out_pair + (orow - p.out) + 32 * dt + 8 * c;
            *reinterpret_cast<bf16x4 *>(pr) = h4;
            *reinterpret_cast<bf16x4 *>(pr + (long long)p.B * N * H * HD) = l4;
          }
        }
      }
      if (hh == 0) p.lse[((long long)b * H + h) * N + q] = (m * scale2 + __builtin_amdgcn_logf(l)) * LN2;
    }
  }
}

// ---- backward, dQ pass (+ delta): the dQ kernel of dm_attention_q32_bwd.hip with every product a triple --------------------------------------
//   S^T = K . Q^T (+ bias / scale),  dP^T = V . dO^T,  P^T = exp2(scale2 S^T - lse log2e),  dS^T = P^T (dP^T - delta),  dQ^T += K^T . dS^T
// Images [K hi | K lo | V hi | V lo] with the dual-use swizzle (K is read by rows and transposed); Q / dO fragments of the lane's row
// come from the split tensors in global memory, dS^T is split in registers.  36 MFMAs per 32-key tile; a tile runs start to end
// (score chains, wait, VALU, dQ chain): the simple form already is several times the fp32-MFMA kernels' rate.
// NW: 4 = one wave per SIMD on 128-row blocks; 8 = two waves per SIMD over whole (head, sample) units dealt out as runs (see attn_fwd_x3_kernel)
template <int NKT, bool RAGGED, bool TAB, int NW = 4>
__global__ __launch_bounds__(64 * NW, 1) void attn_bwd_dq_x3_kernel(const AttnX3BwdParams p, int bchunk, int nblk, int chunks) {
  static_assert(!(TAB && (RAGGED || NKT % 2)), "table form: N = 64 x scales");
  constexpr int NP = NKT * 32;
  const int N = RAGGED ? p.N : NP;
  constexpr int IMG = NP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // K hi | K lo | V hi | V lo | table
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int H = p.H;
  int rb = 0, u0, u1;
  if constexpr (NW == 8) {
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
  int h = -1;                                                       // head of the unit in progress (its table is in LDS)
  constexpr bool PAD = NW == 8;      // (256-register budget: operands may come fresh from v_accvgpr moves -- every asm MFMA opens with a wait state)
  const int q_wave = rb * (32 * NW) + wave * 32;
  const int q = q_wave + r;
  const bool wave_live = q_wave < N;
  const bool row_ok = q < N;
  const long long tok_stride = 3LL * H * HD, out_stride = (long long)H * HD;
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = p.scale * LOG2E;

  constexpr int TAB_MAXC = 15 * ((NKT - 1) >> 1) + 7;
  float *tab = reinterpret_cast<float *>(smem + 4 * IMG);
  const float *tabl = tab;
  if constexpr (TAB) {
    const int qz = q >> 6, qy = (q >> 3) & 7, qx = q & 7;
    tabl = tab + ((qz + NKT / 2 - 1) * 15 + qy + 7 - TAB_MAXC) * TAB_PITCH + 7 - qx + 4 * hh;
  }
  auto load_table = [&](int hd) {                                   // (between two barriers: nobody reads the old table any more)
    if constexpr (TAB) {
      const float inv_scale = 1.f / p.scale;
      for (int i = t; i < (NKT - 1) * 225; i += 64 * NW) {
        const int prow = i / 15, px = i - prow * 15;
        tab[prow * TAB_PITCH + (14 - px)] = p.table[(long long)i * H + hd] * inv_scale;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  auto init_c = [&](int kt, f32x16 &d) {
    if constexpr (TAB) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) d[4 * c + e] = tabl[TAB_PITCH * (TAB_MAXC - (15 * (kt >> 1) + 4 * (kt & 1) + c)) + e];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) d[i] = (RAGGED && kt == NKT - 1 && 32 * kt + 8 * (i >> 2) + 4 * hh + (i & 3) >= N) ? NEG_BIG : 0.f;
      asm volatile("" : "+v"(d));
    }
  };

  const int dkey = lane >> 3;
  const int pw = wave & 3;                                          // the 8-row piece of a tile this wave stages (8 waves: 0..3 K, 4..7 V)
  const unsigned rowoff0 = (unsigned)((8 * pw + dkey) * tok_stride * 2);
  const unsigned src_swz = (unsigned)(((lane & 7) ^ ((((dkey >> 1) & 1) << 2) | ((((pw & 1) << 1) | (dkey >> 2)) & 3))) * 16);
  const unsigned voffK = rowoff0 + (unsigned)(1 * H * HD * 2) + src_swz;
  const unsigned voffV = rowoff0 + (unsigned)(2 * H * HD * 2) + src_swz;
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(DM_LDS char *)smem);
  unsigned step_bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)(32 * tok_stride * 2));
  asm volatile("s_nop 4" : "+s"(step_bytes));
  auto sample_rsrc = [&](const bf16_t *src, int b) -> i32x4 {
    const uintptr_t base = reinterpret_cast<uintptr_t>(src + (long long)b * N * tok_stride + (long long)h * HD);
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base & 0xffffffffu));
    rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((base >> 32) & 0xffffu));
    rs[2] = __builtin_amdgcn_readfirstlane((int)(N * tok_stride * 2));
    rs[3] = 0x00020000;
    asm volatile("s_nop 4" : "+s"(rs));
    return rs;
  };
  auto stage_all = [&](int b) {
    const i32x4 rh = sample_rsrc(p.hi, b), rl = sample_rsrc(p.lo, b);
    const unsigned base = lds0 + (unsigned)pw * 1024u;
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      if (NW == 4 || wave < 4) {
        lds_dma(rh, base + 0 * IMG + (unsigned)j * 4096u, voffK, (unsigned)j * step_bytes);
        lds_dma(rl, base + 1 * IMG + (unsigned)j * 4096u, voffK, (unsigned)j * step_bytes);
      }
      if (NW == 4 || wave >= 4) {
        lds_dma(rh, base + 2 * IMG + (unsigned)j * 4096u, voffV, (unsigned)j * step_bytes);
        lds_dma(rl, base + 3 * IMG + (unsigned)j * 4096u, voffV, (unsigned)j * step_bytes);
      }
    }
  };
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

  for (int u = u0; u < u1; ++u) {
    const int hu = u / p.B, b = u - hu * p.B;
    __builtin_amdgcn_s_barrier();
    if (hu != h) { h = hu; load_table(h); }                         // (uniform: the first unit, or the run crossed into the next head)
    stage_all(b);
    // this lane's row: Q and dO fragments (hi / lo), delta = rowsum(dO . O) in fp32, -lse in log2 units
    u32x4 qh[4], ql[4], dh[4], dl[4];
    float dsum = 0.f, lse = 0.f;
    {
      const bool ok = wave_live && row_ok;
      const long long qoff = ((long long)b * N + q) * tok_stride + (long long)h * HD + 8 * hh;
      const long long ooff = ((long long)b * N + q) * out_stride + (long long)h * HD + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qh[ks] = ok ? *reinterpret_cast<const u32x4 *>(p.hi + qoff + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
        ql[ks] = ok ? *reinterpret_cast<const u32x4 *>(p.lo + qoff + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
        dh[ks] = ok ? *reinterpret_cast<const u32x4 *>(p.dohi + ooff + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
        dl[ks] = ok ? *reinterpret_cast<const u32x4 *>(p.dolo + ooff + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
        if (ok) {
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const f32x4 ov = dm_load4(p.out + ooff + 16 * ks + 4 * w), gv = dm_load4(p.dout + ooff + 16 * ks + 4 * w);
            dsum += ov[0] * gv[0] + ov[1] * gv[1] + ov[2] * gv[2] + ov[3] * gv[3];
          }
        }
      }
      lse = ok ? p.lse[((long long)b * H + h) * N + q] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float delta = half_sum(dsum);
    const float nl = -lse * LOG2E;
    if (wave_live && row_ok && hh == 0) p.delta[((long long)b * H + h) * N + q] = delta;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { park_acc(qh[ks]); park_acc(ql[ks]); park_acc(dh[ks]); park_acc(dl[ks]); }
    if (!wave_live) continue;
    const char *kh = smem, *kl = smem + IMG, *vh = smem + 2 * IMG, *vl = smem + 3 * IMG;

    f32x16 sc, dp;
    u32x4 dsh[2], dsl[2];
    u32x4 kfh[4], kfl[4], vfh[4], vfl[4];
    u32x2 tfh[8], tfl[8];
    f32x16 dq0, dq1;
    auto tfrag = [&](const u32x2 (&f)[8], int sx, int dt) { return (u32x4){f[4 * sx + 2 * dt][0], f[4 * sx + 2 * dt][1], f[4 * sx + 2 * dt + 1][0], f[4 * sx + 2 * dt + 1][1]}; };
    asm volatile("s_nop 1");
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      init_c(j, sc);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kfh[ks] = *reinterpret_cast<const u32x4 *>(kh + j * 4096 + roff[ks]);
        kfl[ks] = *reinterpret_cast<const u32x4 *>(kl + j * 4096 + roff[ks]);
        vfh[ks] = *reinterpret_cast<const u32x4 *>(vh + j * 4096 + roff[ks]);
        vfl[ks] = *reinterpret_cast<const u32x4 *>(vl + j * 4096 + roff[ks]);
      }
#pragma unroll
      for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int o = (32 * j + 16 * sx) * 128;
          tfh[4 * sx + 2 * dt] = dm_ds_read_tr16(kh + o + toff[dt][0]);
          tfh[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(kh + o + toff[dt][1]);
          tfl[4 * sx + 2 * dt] = dm_ds_read_tr16(kl + o + toff[dt][0]);
          tfl[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(kl + o + toff[dt][1]);
        }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qk_acc<true, PAD>(sc, kfh[ks], qh[ks]);
        qk_acc<true, PAD>(sc, kfl[ks], qh[ks]);
        qk_acc<true, PAD>(sc, kfh[ks], ql[ks]);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks == 0) qk_first0<true, PAD>(dp, vfh[0], dh[0]); else qk_acc<true, PAD>(dp, vfh[ks], dh[ks]);
        qk_acc<true, PAD>(dp, vfl[ks], dh[ks]);
        qk_acc<true, PAD>(dp, vfh[ks], dl[ks]);
      }
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sc), "+v"(dp)
                   : "v"(kfh[0]), "v"(kfh[1]), "v"(kfh[2]), "v"(kfh[3]), "v"(kfl[0]), "v"(kfl[1]), "v"(kfl[2]), "v"(kfl[3]),
                     "v"(vfh[0]), "v"(vfh[1]), "v"(vfh[2]), "v"(vfh[3]), "v"(vfl[0]), "v"(vfl[1]), "v"(vfl[2]), "v"(vfl[3]));
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * k], scale2, nl));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * k + 1], scale2, nl));
        const float d0 = p0 * (dp[2 * k] - delta), d1 = p1 * (dp[2 * k + 1] - delta);
        const unsigned w = pk_bf16(d0, d1);
        dsh[k >> 2][k & 3] = w;
        dsl[k >> 2][k & 3] = split_lo(d0, d1, w);
      }
      asm volatile("s_nop 1" : "+v"(dsh[0]), "+v"(dsh[1]), "+v"(dsl[0]), "+v"(dsl[1]));      // VALU write -> MFMA operand
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int sx = g >> 1, dt = g & 1;
        f32x16 &o = dt ? dq1 : dq0;
        if (j == 0 && sx == 0) pv_first<PAD>(o, tfrag(tfh, 0, dt), dsh[0]); else pv_acc<PAD>(o, tfrag(tfh, sx, dt), dsh[sx]);
        pv_acc<PAD>(o, tfrag(tfl, sx, dt), dsh[sx]);
        pv_acc<PAD>(o, tfrag(tfh, sx, dt), dsl[sx]);
      }
      asm volatile("" :: "v"(tfh[0]), "v"(tfh[1]), "v"(tfh[2]), "v"(tfh[3]), "v"(tfh[4]), "v"(tfh[5]), "v"(tfh[6]), "v"(tfh[7]),
                   "v"(tfl[0]), "v"(tfl[1]), "v"(tfl[2]), "v"(tfl[3]), "v"(tfl[4]), "v"(tfl[5]), "v"(tfl[6]), "v"(tfl[7]),
                   "v"(dsh[0]), "v"(dsh[1]), "v"(dsl[0]), "v"(dsl[1]));
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(dq0), "+a"(dq1));
    if (row_ok) {
      const long long roff = ((long long)b * N + q) * tok_stride + (long long)h * HD + 4 * hh;
      const float f = p.scale;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const f32x16 &o = dt ? dq1 : dq0;
#pragma unroll
        for (int c = 0; c < 4; ++c) x3_store_grad(p, roff + 32 * dt + 8 * c, (f32x4){o[4 * c] * f, o[4 * c + 1] * f, o[4 * c + 2] * f, o[4 * c + 3] * f});
      }
    }
  }
}

// ---- backward, dK / dV pass (+ the table-gradient slab): the key on the lane (dm_attention_q32_bwd.hip), every product a triple -------------
//   S = Q . K^T (+ bias / scale from the table, natural order),  dP = dO . V^T - delta (C operand from the stat table)
//   P = exp2(scale2 S - lse log2e) (-lse: the fma's addend, one value per query register),  dS = P dP
//   dV^T += dO^T . P,  dK^T += Q^T . dS,  H_j += E_s . dS_s (hi and lo: the slab sums dS to ~2^-17 relative)
// Images [Q hi | Q lo | dO hi | dO lo] (dual-use swizzle); K / V fragments of the lane's key from the split tensors.  52 MFMAs per tile.
template <int NKT, bool RAGGED, bool TAB>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_x3_kernel(const AttnX3BwdParams p, int bchunk, int nblk, int chunks) {
  static_assert(!(TAB && (RAGGED || NKT % 2)), "table form: N = 64 x scales");
  constexpr int NP = NKT * 32;
  const int N = RAGGED ? p.N : NP;
  constexpr int IMG = NP * 128;
  constexpr int STAT = 2 * NP * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // Q hi | Q lo | dO hi | dO lo | stat | table
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hh = lane >> 5;
  int h, rb, chunk;
  if (!coords(nblk, p.H, chunks, h, rb, chunk)) return;
  const int H = p.H;
  const int b0 = chunk * bchunk, b1 = min(p.B, b0 + bchunk);
  if (b0 >= b1) return;
  const int k_wave = rb * 128 + wave * 32;
  const int key = k_wave + r;
  const bool wave_live = k_wave < N;
  const bool row_ok = key < N;
  const long long tok_stride = 3LL * H * HD, out_stride = (long long)H * HD;
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = p.scale * LOG2E;

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
  auto stage_all = [&](int b) {
    const long long qb = (long long)b * N * tok_stride + (long long)h * HD, db = (long long)b * N * out_stride + (long long)h * HD;
    const i32x4 rqh = make_rsrc(p.hi + qb, (long long)N * tok_stride * 2), rql = make_rsrc(p.lo + qb, (long long)N * tok_stride * 2);
    const i32x4 rdh = make_rsrc(p.dohi + db, (long long)N * out_stride * 2), rdl = make_rsrc(p.dolo + db, (long long)N * out_stride * 2);
    const unsigned base = lds0 + (unsigned)wave * 1024u;
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      lds_dma(rqh, base + 0 * IMG + (unsigned)j * 4096u, voffQ, (unsigned)j * stepQ);
      lds_dma(rql, base + 1 * IMG + (unsigned)j * 4096u, voffQ, (unsigned)j * stepQ);
      lds_dma(rdh, base + 2 * IMG + (unsigned)j * 4096u, voffD, (unsigned)j * stepD);
      lds_dma(rdl, base + 3 * IMG + (unsigned)j * 4096u, voffD, (unsigned)j * stepD);
    }
  };
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
  float *st = reinterpret_cast<float *>(smem + 4 * IMG);
  float *tab = reinterpret_cast<float *>(smem + 4 * IMG + STAT);
  const float *tabl = tab;
  if constexpr (TAB) {                                              // natural order: row (dz + S - 1) * 15 + dy + 7, entry dx + 7
    const float inv_scale = 1.f / p.scale;
    for (int i = t; i < (NKT - 1) * 225; i += 256) {
      const int prow = i / 15, px = i - prow * 15;
      tab[prow * TAB_PITCH + px] = p.table[(long long)i * H + h] * inv_scale;
    }
    const int kz = key >> 6, ky = (key >> 3) & 7, kx = key & 7;
    tabl = tab + (wave_live ? ((NKT / 2 - 1 - kz) * 15 + 7 - ky) * TAB_PITCH + 4 * hh - kx + 7 : 0);
  }
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
  constexpr int NH = TAB ? NKT : 1;
  f32x16 hacc[NH];
  if constexpr (TAB) {
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
#pragma unroll
      for (int i = 0; i < 16; ++i) hacc[j][i] = 0.f;
      asm volatile("" : "+a"(hacc[j]));
    }
  }
  const bool want_slab = TAB && p.slab != nullptr;

  const int sq = wave * 64 + lane;
  for (int b = b0; b < b1; ++b) {
    __builtin_amdgcn_s_barrier();
    stage_all(b);
    u32x4 kh[4], kl[4], vh[4], vl[4];
    {
      const bool ok = wave_live && row_ok;
      const long long koff = ((long long)b * N + key) * tok_stride + (long long)(H + h) * HD + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kh[ks] = ok ? *reinterpret_cast<const u32x4 *>(p.hi + koff + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
        kl[ks] = ok ? *reinterpret_cast<const u32x4 *>(p.lo + koff + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
        vh[ks] = ok ? *reinterpret_cast<const u32x4 *>(p.hi + koff + (long long)H * HD + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
        vl[ks] = ok ? *reinterpret_cast<const u32x4 *>(p.lo + koff + (long long)H * HD + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
      }
      if (sq < NP) {                                                // per-query constants (queries >= N: no probability, no gradient)
        const bool sok = sq < N;
        st[sq] = sok ? -p.lse[((long long)b * H + h) * N + sq] * LOG2E : NEG_BIG;
        st[NP + sq] = sok ? -p.delta[((long long)b * H + h) * N + sq] : 0.f;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { park_acc(kh[ks]); park_acc(kl[ks]); park_acc(vh[ks]); park_acc(vl[ks]); }
    if (!wave_live) continue;
    const char *qih = smem, *qil = smem + IMG, *dih = smem + 2 * IMG, *dil = smem + 3 * IMG;

    f32x16 sc, dp, nl;
    u32x4 pbh[2], pbl[2], dsh[2], dsl[2];
    u32x4 afh[4], afl[4];
    u32x2 th[8], tl[8];
    f32x16 dk0, dk1, dv0, dv1;
    auto tfrag = [&](const u32x2 (&f)[8], int sx, int dt) { return (u32x4){f[4 * sx + 2 * dt][0], f[4 * sx + 2 * dt][1], f[4 * sx + 2 * dt + 1][0], f[4 * sx + 2 * dt + 1][1]}; };
    auto read_rows = [&](const char *ih, const char *il, int j) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        afh[ks] = *reinterpret_cast<const u32x4 *>(ih + j * 4096 + roff[ks]);
        afl[ks] = *reinterpret_cast<const u32x4 *>(il + j * 4096 + roff[ks]);
      }
    };
    auto read_tr = [&](const char *ih, const char *il, int j) {
#pragma unroll
      for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int o = (32 * j + 16 * sx) * 128;
          th[4 * sx + 2 * dt] = dm_ds_read_tr16(ih + o + toff[dt][0]);
          th[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(ih + o + toff[dt][1]);
          tl[4 * sx + 2 * dt] = dm_ds_read_tr16(il + o + toff[dt][0]);
          tl[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(il + o + toff[dt][1]);
        }
    };
    auto pin_rows = [&]() {
      asm volatile("" :: "v"(afh[0]), "v"(afh[1]), "v"(afh[2]), "v"(afh[3]), "v"(afl[0]), "v"(afl[1]), "v"(afl[2]), "v"(afl[3]));
    };
    auto pin_tr = [&]() {
      asm volatile("" :: "v"(th[0]), "v"(th[1]), "v"(th[2]), "v"(th[3]), "v"(th[4]), "v"(th[5]), "v"(th[6]), "v"(th[7]),
                   "v"(tl[0]), "v"(tl[1]), "v"(tl[2]), "v"(tl[3]), "v"(tl[4]), "v"(tl[5]), "v"(tl[6]), "v"(tl[7]));
    };
    asm volatile("s_nop 1");
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      // C operands: bias / scale (or 0) for the scores, -delta for dP; -lse log2e as the exponent's addend
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(st + 32 * j + 8 * c + 4 * hh);
        const f32x4 d = *reinterpret_cast<const f32x4 *>(st + NP + 32 * j + 8 * c + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          nl[4 * c + e] = a[e];
          dp[4 * c + e] = d[e];
          if constexpr (TAB) sc[4 * c + e] = tabl[TAB_PITCH * (15 * (j >> 1) + 4 * (j & 1) + c) + e];
          else sc[4 * c + e] = 0.f;
        }
      }
      if constexpr (!TAB) asm volatile("" : "+v"(sc));
      read_rows(qih, qil, j);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qk_acc<true, false>(sc, afh[ks], kh[ks]);
        qk_acc<true, false>(sc, afl[ks], kh[ks]);
        qk_acc<true, false>(sc, afh[ks], kl[ks]);
      }
      pin_rows();
      read_rows(dih, dil, j);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qk_acc<true, false>(dp, afh[ks], vh[ks]);
        qk_acc<true, false>(dp, afl[ks], vh[ks]);
        qk_acc<true, false>(dp, afh[ks], vl[ks]);
      }
      pin_rows();
      read_tr(dih, dil, j);                                          // dO^T for dV (lands during the wait below)
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sc), "+v"(dp));
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * k], scale2, nl[2 * k]));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * k + 1], scale2, nl[2 * k + 1]));
        const float d0 = p0 * dp[2 * k], d1 = p1 * dp[2 * k + 1];
        const unsigned wp = pk_bf16(p0, p1), wd = pk_bf16(d0, d1);
        pbh[k >> 2][k & 3] = wp;
        pbl[k >> 2][k & 3] = split_lo(p0, p1, wp);
        dsh[k >> 2][k & 3] = wd;
        dsl[k >> 2][k & 3] = split_lo(d0, d1, wd);
      }
      asm volatile("s_nop 1" : "+v"(pbh[0]), "+v"(pbh[1]), "+v"(pbl[0]), "+v"(pbl[1]), "+v"(dsh[0]), "+v"(dsh[1]), "+v"(dsl[0]), "+v"(dsl[1]));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int sx = g >> 1, dt = g & 1;
        f32x16 &o = dt ? dv1 : dv0;
        if (j == 0 && sx == 0) pv_first<false>(o, tfrag(th, 0, dt), pbh[0]); else pv_acc<false>(o, tfrag(th, sx, dt), pbh[sx]);
        pv_acc<false>(o, tfrag(tl, sx, dt), pbh[sx]);
        pv_acc<false>(o, tfrag(th, sx, dt), pbl[sx]);
      }
      pin_tr();
      read_tr(qih, qil, j);                                          // Q^T for dK
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int sx = g >> 1, dt = g & 1;
        f32x16 &o = dt ? dk1 : dk0;
        if (j == 0 && sx == 0) pv_first<false>(o, tfrag(th, 0, dt), dsh[0]); else pv_acc<false>(o, tfrag(th, sx, dt), dsh[sx]);
        pv_acc<false>(o, tfrag(tl, sx, dt), dsh[sx]);
        pv_acc<false>(o, tfrag(th, sx, dt), dsl[sx]);
      }
      pin_tr();
      if constexpr (TAB) {
        if (want_slab) {
          pv_acc<false>(hacc[j], esel[0], dsh[0]);
          pv_acc<false>(hacc[j], esel[1], dsh[1]);
          pv_acc<false>(hacc[j], esel[0], dsl[0]);
          pv_acc<false>(hacc[j], esel[1], dsl[1]);
        }
      }
      asm volatile("" :: "v"(pbh[0]), "v"(pbh[1]), "v"(pbl[0]), "v"(pbl[1]), "v"(dsh[0]), "v"(dsh[1]), "v"(dsl[0]), "v"(dsl[1]), "v"(esel[0]), "v"(esel[1]));
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(dk0), "+a"(dk1), "+a"(dv0), "+a"(dv1));
    if (row_ok) {
      const long long koff = ((long long)b * N + key) * tok_stride + (long long)(H + h) * HD + 4 * hh;
      const long long voff = koff + (long long)H * HD;
      const float f = p.scale;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const f32x16 &a = dt ? dk1 : dk0;
        const f32x16 &v = dt ? dv1 : dv0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          x3_store_grad(p, koff + 32 * dt + 8 * c, (f32x4){a[4 * c] * f, a[4 * c + 1] * f, a[4 * c + 2] * f, a[4 * c + 3] * f});
          x3_store_grad(p, voff + 32 * dt + 8 * c, (f32x4){v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]});
        }
      }
    }
  }
  if constexpr (TAB) {
    if (wave_live && want_slab) {
      float *sl = p.slab + ((long long)(chunk * H + h) * N) * N + key;
#pragma unroll
      for (int j = 0; j < NKT; ++j) {
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(hacc[j]));
#pragma unroll
        for (int i = 0; i < 16; ++i) sl[(long long)(32 * j + 8 * (i >> 2) + 4 * hh + (i & 3)) * N] = hacc[j][i];
      }
    }
  }
}

inline void grid(int B, int N, int H, int &nblk, int &chunks, int &bchunk) {
  nblk = (N + 127) / 128;
  chunks = 256 / (H * nblk);
  if (chunks < 1) chunks = 1;
  if (chunks > B) chunks = B;
  bchunk = (B + chunks - 1) / chunks;
  chunks = (B + bchunk - 1) / bchunk;
}

// DM_ATTN_X3_W8=0: the one-wave-per-SIMD forms of the forward / dQ kernels (A/B runs)
inline bool x3_w8() {
  static const bool on = [] { const char *e = getenv("DM_ATTN_X3_W8"); return !(e && atoi(e) == 0); }();
  return on;
}

template <int NKT, bool RAGGED, bool TAB, int NW> bool launch_fwd_nw(const AttnX3Params &p, hipStream_t s) {
  constexpr int LDS = 4 * NKT * 32 * 128 + (TAB ? (NKT - 1) * 15 * TAB_PITCH * 4 : 0);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_x3_kernel<NKT, RAGGED, TAB, NW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  if (!ok) return false;
  int nblk, chunks, bchunk;
  grid(p.B, p.N, p.H, nblk, chunks, bchunk);
  const int wgs = NW == 8 ? (p.B * p.H < 256 ? p.B * p.H : 256) : grid_size(nblk, p.H, chunks);      // 8 waves: one run of units per CU
  hipLaunchKernelGGL((attn_fwd_x3_kernel<NKT, RAGGED, TAB, NW>), dim3(wgs), dim3(64 * NW), LDS, s, p, bchunk, nblk, chunks);
  return true;
}
template <int NKT, bool RAGGED, bool TAB> bool launch_fwd(const AttnX3Params &p, hipStream_t s) {
  return x3_w8() ? launch_fwd_nw<NKT, RAGGED, TAB, 8>(p, s) : launch_fwd_nw<NKT, RAGGED, TAB, 4>(p, s);
}

template <int NKT> bool launch_fwd_n(const AttnX3Params &p, hipStream_t s) {
  const bool ragged = p.N != NKT * 32;
  if constexpr (NKT % 2 == 0) {
    if (p.table) return !ragged && launch_fwd<NKT, false, true>(p, s);
  }
  if (p.table) return false;
  return ragged ? launch_fwd<NKT, true, false>(p, s) : launch_fwd<NKT, false, false>(p, s);
}

template <int NKT, bool RAGGED, bool TAB, int NW> bool launch_dq_nw(const AttnX3BwdParams &p, hipStream_t s) {
  constexpr int LDS = 4 * NKT * 32 * 128 + (TAB ? (NKT - 1) * 15 * TAB_PITCH * 4 : 0);
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dq_x3_kernel<NKT, RAGGED, TAB, NW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  if (!ok) return false;
  int nblk, chunks, bchunk;
  grid(p.B, p.N, p.H, nblk, chunks, bchunk);
  const int wgs = NW == 8 ? (p.B * p.H < 256 ? p.B * p.H : 256) : grid_size(nblk, p.H, chunks);
  hipLaunchKernelGGL((attn_bwd_dq_x3_kernel<NKT, RAGGED, TAB, NW>), dim3(wgs), dim3(64 * NW), LDS, s, p, bchunk, nblk, chunks);
  return true;
}
template <int NKT, bool RAGGED, bool TAB> bool launch_dq(const AttnX3BwdParams &p, hipStream_t s) {
  return x3_w8() ? launch_dq_nw<NKT, RAGGED, TAB, 8>(p, s) : launch_dq_nw<NKT, RAGGED, TAB, 4>(p, s);
}

template <int NKT> bool launch_dq_n(const AttnX3BwdParams &p, hipStream_t s) {
  const bool ragged = p.N != NKT * 32;
  if constexpr (NKT % 2 == 0) {
    if (p.table) return !ragged && launch_dq<NKT, false, true>(p, s);
  }
  if (p.table) return false;
  return ragged ? launch_dq<NKT, true, false>(p, s) : launch_dq<NKT, false, false>(p, s);
}

template <int NKT, bool RAGGED, bool TAB> bool launch_dkv(const AttnX3BwdParams &p, hipStream_t s) {
  constexpr int LDS = 4 * NKT * 32 * 128 + 2 * NKT * 32 * 4 + (TAB ? (NKT - 1) * 15 * TAB_PITCH * 4 : 0);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dkv_x3_kernel<NKT, RAGGED, TAB>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  if (!ok) return false;
  int nblk, chunks, bchunk;
  grid(p.B, p.N, p.H, nblk, chunks, bchunk);
  hipLaunchKernelGGL((attn_bwd_dkv_x3_kernel<NKT, RAGGED, TAB>), dim3(grid_size(nblk, p.H, chunks)), dim3(256), LDS, s, p, bchunk, nblk, chunks);
  return true;
}

template <int NKT> bool launch_dkv_n(const AttnX3BwdParams &p, hipStream_t s) {
  const bool ragged = p.N != NKT * 32;
  if constexpr (NKT % 2 == 0) {
    if (p.table) return !ragged && launch_dkv<NKT, false, true>(p, s);
  }
  if (p.table) return false;
  return ragged ? launch_dkv<NKT, true, false>(p, s) : launch_dkv<NKT, false, false>(p, s);
}

}  // namespace dmx3

bool dm_attn_x3_shape(int B, int N, int H, bool has_table, int cube_s) {
  static const bool on = [] { const char *e = getenv("DM_ATTN_X3"); return !(e && atoi(e) == 0); }();
  if (!on || N <= 128 || N > 256 || B <= 0 || H <= 0) return false;
  if ((long long)N * 3 * H * 64 * 2 >= (1LL << 31)) return false;
  if (has_table && (N != 64 * cube_s || (cube_s != 3 && cube_s != 4))) return false;
  return true;
}

void dm_attn_x3_split(const float *x, void *hi, void *lo, long long n, hipStream_t s) {
  const long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dmx3::split2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, reinterpret_cast<bf16_t *>(hi), reinterpret_cast<bf16_t *>(lo), n4);
}

bool dm_attn_fwd_x3(const AttnX3Params &p, hipStream_t s) {
  if (!dm_attn_x3_shape(p.B, p.N, p.H, p.table != nullptr, p.cube_s)) return false;
  switch ((p.N + 31) / 32) {
    case 5: return dmx3::launch_fwd_n<5>(p, s);
    case 6: return dmx3::launch_fwd_n<6>(p, s);
    case 7: return dmx3::launch_fwd_n<7>(p, s);
    case 8: return dmx3::launch_fwd_n<8>(p, s);
    default: return false;
  }
}

bool dm_attn_bwd_dq_x3(const AttnX3BwdParams &p, hipStream_t s) {
  if (!dm_attn_x3_shape(p.B, p.N, p.H, p.table != nullptr, p.cube_s)) return false;
  switch ((p.N + 31) / 32) {
    case 5: return dmx3::launch_dq_n<5>(p, s);
    case 6: return dmx3::launch_dq_n<6>(p, s);
    case 7: return dmx3::launch_dq_n<7>(p, s);
    case 8: return dmx3::launch_dq_n<8>(p, s);
    default: return false;
  }
}

bool dm_attn_bwd_dkv_x3(const AttnX3BwdParams &p, hipStream_t s) {
  if (!dm_attn_x3_shape(p.B, p.N, p.H, p.table != nullptr, p.cube_s)) return false;
  switch ((p.N + 31) / 32) {
    case 5: return dmx3::launch_dkv_n<5>(p, s);
    case 6: return dmx3::launch_dkv_n<6>(p, s);
    case 7: return dmx3::launch_dkv_n<7>(p, s);
    case 8: return dmx3::launch_dkv_n<8>(p, s);
    default: return false;
  }
}

// slab chunks of the dK / dV pass (first dimension of AttnX3BwdParams::slab)
int dm_attn_x3_chunks(int B, int N, int H) {
  int nblk, chunks, bchunk;
  dmx3::grid(B, N, H, nblk, chunks, bchunk);
  return chunks;
}
