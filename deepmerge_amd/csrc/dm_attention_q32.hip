// Attention forward with 32 query rows per wave on the 32x32x16 bf16 MFMA (head dim 64, 128 < N <= 256), gfx950.
//
// Why (profiles/r02_attn_ablations.txt): in the 8-wave x 16-row kernel of dm_attention_pipe.hip every wave reads ALL of K and V from
// LDS for 16 query rows -- 11 of its 42 us per stage-0 launch are fragment reads at the LDS peak, the softmax costs 7.8 VALU
// instructions per score, and nothing overlaps.  Here a wave owns 32 query rows, so a K / V fragment feeds twice the MFMA work
// (LDS fragment bytes per score halve), and the products are turned round so the QUERY sits on the MFMA lane:
//   S^T[key][query] = K . Q^T     A = K rows from LDS (ds_read_b128), B = Q rows straight from global memory (lane = query)
//   O^T[d][query]   = V^T . P^T   A = V^T by hardware-transposed LDS reads, B = P^T = the score registers themselves, packed
// A lane then holds 16 keys per 32-key tile of ITS query row: row maximum and row sum are in-lane reductions plus one exchange with
// lane ^ 32 (v_permlane32_swap), and the probabilities feed the second product without leaving the registers (MI355X guide, "an
// accumulator tile as the next MFMA's operand": registers 8s .. 8s+7 of a 32x32 tile are the B fragment of k-step s, key order
// 16 s + 8 (j >> 2) + 4 h + (j & 3) -- V^T is fetched in that order).
// The relative-position bias costs NO instruction per score: the rows of the workgroup's queries (pre-divided by the scale) are
// loaded once per workgroup and are the C operand of each tile's first MFMA (D != C).  Masked keys (ragged N) are a C operand of
// -1e30 the same way.  Softmax per score: max3 (1/2), fma, exp2, add, cvt_pk (1/2) = 4 VALU issues.
//
// Register classes decide the structure.  C and D of an MFMA share one class (one acc_cd bit), and the VALU cannot read accumulator
// registers, so scores AND bias must be architectural VGPRs (<= 256): 128 bias + 128 scores of an exact whole-row softmax do not fit.
// Hence an ONLINE softmax over 32-key tiles with a deferred maximum: the reference maximum is the first tile's; a later tile only
// forces a rescale when its maximum exceeds the reference by more than 2^16 (then l, and O after the pending P.V, are multiplied by
// 2^(old - new): exact powers-of-two bookkeeping, so results do not depend on whether a rescale happened beyond fp32 rounding).
// Live at any time: two score tiles (32 VGPRs), two packed P tiles (16), one K and one V fragment set (32), the bias (128).
// O^T (32) and the Q^T fragments of this and the next sample (32) live in accumulator registers: the MFMAs are inline asm with
// explicit register classes ("v" / "a"), so the wait states hipcc would pad are placed by hand:
//   * VALU write -> MFMA operand: packed P is written at least four gaps before the MFMAs that read it (stage C of the pipeline
//     below), fragments come from LDS behind hipcc's own waits; `s_nop 1` stands where an operand may be fresh (the Q copies at a
//     sample's first MFMA, the last pairs before the epilogue's MFMAs, after a rescale); tools/isa_hazards.py checks the listing;
//   * MFMA D -> VALU read: a tile's scores are first read after the NEXT tile's four MFMAs have been issued (>= 96 cycles), behind an
//     opaque `asm volatile("" : "+v")` that the reads depend on; O is read behind `s_nop 15` twice.
// Per tile the wave issues 4 QK^T MFMAs (tile j + 1) and 4 P.V MFMAs (tile j - 1) with the VALU work of tile j placed between them
// piece by piece (sched_barrier fences): one wave per SIMD has no partner wave to fill its matrix pipe's gaps.
//
// One wave per SIMD (4 waves x 32 rows = 128-row block, one workgroup per CU); persistent over a chunk of samples like the pipelined
// kernels: K / V of sample i + 1 arrive by LDS-DMA into the other buffer during the first tiles of sample i, Q fragments of sample
// i + 1 are prefetched, one barrier per sample, results leave one sample late through a wave-private LDS block as whole 128-byte rows.
// LDS images (bank model of MI355X_MICROARCH.md, both conflict-free for the access patterns used here):
//   K [keys][128 B], 16-byte chunk index XOR ((key >> 1) & 7)   (ds_read_b128 of 32 consecutive rows, same chunk)
//   V [keys][128 B], 64-byte half index XOR ((key >> 1) & 1)    (ds_read_b64_tr_b16 of 4 rows x 64 B per half-wave)
#include "dm_attention_q32.h"

// Timing ablations (tools/gpu_q32_abl.sh builds with EXTRA=-DDMQ_ABL=<bits>; results are then wrong by design):
// 1 no bias loads, 2 no K / V DMA, 4 no exp pieces, 8 no P.V MFMAs, 16 no QK^T MFMAs, 32 no write-back, 64 no Q loads, 128 no fragment reads
#ifndef DMQ_ABL
#define DMQ_ABL 0
#endif
#ifndef DMQ_PRIO_FLIP
#define DMQ_PRIO_FLIP 0
#endif
// Half-unit stagger of the 8-wave instances (see the kernel); -DDMQ_STAGGER=0 for A/B builds
#ifndef DMQ_STORES_IN_FLIGHT
#define DMQ_STORES_IN_FLIGHT 0
#endif
#ifndef DMQ_STAGGER
#define DMQ_STAGGER 0
#endif

// -DDMQ_STAMP: wave 0 of every workgroup stamps s_memtime at six points of each of its first 8 samples into a __device__ array
// (diagnostic build only: tools/q32_stamps.py reads it through dm_debug_q32_stamps)
#ifdef DMQ_STAMP
__device__ unsigned long long dmq_stamps[512 * 8 * 8];
#define DMQ_T(i) do { if (wave == 0 && lane == 0 && (b - b0) < 8 && blockIdx.x < 512) dmq_stamps[(blockIdx.x * 8 + (b - b0)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// chip-wide 100 MHz clock at kernel entry (k = 0), in front of the unit loop (1) and at kernel exit (2): slot 7 of units 0 / 1 / 2
#define DMQ_RT(k) do { if (wave == 0 && lane == 0 && blockIdx.x < 512) dmq_stamps[(blockIdx.x * 8 + (k)) * 8 + 7] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int dm_debug_q32_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dmq_stamps), sizeof(dmq_stamps)); }
#else
#define DMQ_T(i) do { } while (0)
#define DMQ_RT(k) do { } while (0)
#endif

namespace dmq32 {

// NKT: 32-key tiles; RAGGED: N < 32 NKT (keys >= N are zero-filled by the DMA descriptor and masked).
// BM: 0 no bias; 1 dense bias rows p.bias in registers; 2 the head's relative-position TABLE in LDS (p.table; tokens are a
// (NKT / 2, 8, 8) cube, scale-major then row-major): a lane's 16 keys of a tile are 4 key rows x 4 consecutive key columns, i.e. four
// runs of 4 consecutive table entries once the table is stored with its x axis reversed -- eight ds_read2_b32 per tile straight into
// the score registers, which the tile's MFMAs then accumulate onto (C = D).  No bias registers: the 8-wave form applies.
template <int NKT, bool RAGGED, int BM, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void attn_fwd_q32_kernel(const AttnPipeParams p, int bchunk, int nblk, int chunks) {
  constexpr bool BIAS = BM == 1, TAB = BM == 2;
  constexpr bool DIRECT = TAB && NW == 8;                           // no LDS left for the write-back blocks: rows leave from registers
  static_assert(!TAB || (!RAGGED && NKT % 2 == 0), "table form: N = 64 x scales");
  constexpr int ROWS = 32 * NW;
  constexpr int NP = NKT * 32;
  const int N = RAGGED ? p.N : NP;
  constexpr int IMG = NP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 buffers][K image | V image] | 4 x write-back block
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  DMQ_RT(0);
  const int r = lane & 31, hh = lane >> 5;
  // Work units are (head, sample) pairs, u = head * B + sample.  4 waves: a workgroup owns one row block of one head for a chunk of
  // samples (`coords`).  8 waves (RUNS: the workgroup covers all rows of a unit, nothing is shared between workgroups): the units are
  // dealt out as gridDim.x contiguous runs of equal length +- 1, which may cross a head boundary -- 64 samples x 12 heads on 256 CUs
  // are 3 units each, where whole-sample chunks per head would be 192 workgroups of 4.
  constexpr bool RUNS = NW == 8;
  // STAG (8 waves, round 5): waves 4-7 run HALF A UNIT behind waves 0-3.  The stamps of the lockstep form (profiles/r05_attn_fwd_ablations_and_
  // stamps.md) show 43 % of a unit outside its tile loop -- barrier wait, pipeline fill, epilogue -- with both waves of a SIMD in those phases
  // together.  Every wave now passes TWO barriers per unit (top, and in front of tile HT = NKT / 2); waves 4-7 pass one extra barrier before their
  // first unit and waves 0-3 one after their last, so that barrier k of the older half (top of unit u) is barrier k of the younger half (middle
  // of unit u - 1): one half's fill / epilogue runs under the other half's tiles.  What that obliges:
  //   * K / V of unit u + 1 go into the buffer the younger half still reads (unit u - 1, tiles >= HT) while the older half is in tiles
  //     < HT of unit u: there only pieces the younger half is done with are staged -- K tiles < HT + 2 (tile j + 2's fragments are read in
  //     tile j), V tiles < HT - 1 (tile j - 1's in tile j) -- the rest during the older half's tiles >= HT, when the younger half is in unit u;
  //   * the DMA waves (0-3) wait for their transfers in front of THEIR top barrier; the younger half reads the unit one barrier later;
  //   * the head's table has two slots (head parity): a run that crosses into the next head finds the older half filling the new
  //     head's slot in front of its top barrier while the younger half still reads the old one.
  constexpr bool STAG = RUNS && (DMQ_STAGGER != 0);
  constexpr int HT = NKT / 2;
  const bool older = wave < 4;
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
  const int q = q_wave + r;                                         // this lane's query row
  const bool wave_live = q_wave < N;                                // a wave without rows only takes part in the DMA / barriers
  const bool row_ok = q < N;
  const long long tok_stride = 3LL * H * HD;
  const bf16_t *qkv = reinterpret_cast<const bf16_t *>(p.qkv);
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  const float scale2 = p.scale * LOG2E;                             // scores are kept in raw q.k units; exp2(scale2 * (s - m))

  // ---- C operands of each tile's first MFMA: bias / scale (per workgroup, all samples), or the key mask of the last tile ------------
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
          if (!RAGGED || key + 4 <= N) {
            if (!RAGGED) v = dm_load4(brow);                        // N % 32 == 0: 16-byte aligned rows
            else { v[0] = brow[0]; v[1] = brow[1]; v[2] = brow[2]; v[3] = brow[3]; }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (key + e < N) v[e] = brow[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) cinit[kt][4 * c + e] = (RAGGED && key + e >= N) ? NEG_BIG : v[e] * inv_scale;
      }
    }
  } else if constexpr (RAGGED) {
#pragma unroll
    for (int i = 0; i < 16; ++i) cinit[0][i] = (32 * (NKT - 1) + 8 * (i >> 2) + 4 * hh + (i & 3) >= N) ? NEG_BIG : 0.f;
  }
  // table form: rows p = (dz + S - 1) * 15 + (dy + 7) of 16 floats, entry j = 7 - dx (x reversed), pre-divided by the scale
  constexpr int TAB_ROWS = (NKT - 1) * 15;
  constexpr int TAB_MAXC = 15 * ((NKT - 1) >> 1) + 7;               // largest (15 kz + ky) of a key
  float *tab = reinterpret_cast<float *>(smem + 4 * IMG + (DIRECT ? 0 : NW * WB_WAVE));
  const float *tabl = tab;
  // Two phases: the global loads of the FIRST table are issued behind the first unit's K / V / Q transfers and land under them (as one
  // loop in front of the staging the fill was three dependent memory round trips: 3.9 us from kernel entry to the unit loop, r05 stamps)
  constexpr int TAB_N = (NKT - 1) * 225, TAB_FLOATS = TAB_ROWS * TAB_PITCH;
  constexpr int TAB_T = STAG ? 256 : 64 * NW;                       // threads that fill a table (STAG: the older half)
  constexpr int TAB_IT = (TAB_N + TAB_T - 1) / TAB_T;
  float tv[TAB ? TAB_IT : 1];
  auto load_table = [&](int hd) {
    if (STAG && !older) return;
#pragma unroll
    for (int k = 0; k < TAB_IT; ++k) {
      const int i = t + k * TAB_T;
      tv[k] = i < TAB_N ? p.table[(long long)i * H + hd] : 0.f;
    }
  };
  auto store_table = [&](int hd) {                                  // a barrier must follow before the table is read (slot = head parity under STAG)
    if (STAG && !older) return;
    const float inv_scale = 1.f / p.scale;
    float *dst = tab + (STAG ? (hd & 1) * TAB_FLOATS : 0);
#pragma unroll
    for (int k = 0; k < TAB_IT; ++k) {
      const int i = t + k * TAB_T;
      const int pz = i / 225, rem = i - pz * 225, py = rem / 15, px = rem - py * 15;
      if (i < TAB_N) dst[(pz * 15 + py) * TAB_PITCH + (14 - px)] = tv[k] * inv_scale;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto fill_table = [&](int hd) { load_table(hd); store_table(hd); };
  int tabl_off = 0;
  auto point_table = [&](int hd) { tabl = tab + (STAG ? (hd & 1) * TAB_FLOATS : 0) + tabl_off; };
  if constexpr (TAB) {
    const int qz = q >> 6, qy = (q >> 3) & 7, qx = q & 7;
    tabl_off = ((qz + NKT / 2 - 1) * 15 + qy + 7 - TAB_MAXC) * TAB_PITCH + 7 - qx + 4 * hh;
    point_table(h);
  }
  // bias / scale of tile kt into its score registers: key (kz, ky, kx) = (kt >> 1, 4 (kt & 1) + c, 4 hh + e) for register 4 c + e
  auto read_bias = [&](int kt, f32x16 &d) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) d[4 * c + e] = tabl[TAB_PITCH * (TAB_MAXC - (15 * (kt >> 1) + 4 * (kt & 1) + c)) + e];
  };

  // ---- DMA: one wave-instruction = 8 keys x 128 B (1 KiB of an image); every wave stages NKT instructions of K and of V ------------
  const int dkey = lane >> 3;                                       // key row of the instruction this lane fills
  // the swizzles are applied on the SOURCE chunk (a wave-instruction's LDS destinations are lane-linear); instruction `inst` of a
  // wave has inst & 1 == wave & 1, so (key >> 1) & 7 = ((wave & 1) << 2) | (dkey >> 1)
  const unsigned rowoff0 = (unsigned)((8 * wave + dkey) * tok_stride * 2);
  const unsigned voffK = rowoff0 + (unsigned)(1 * H * HD * 2) + (unsigned)(((lane & 7) ^ (((wave & 1) << 2) | (dkey >> 1))) * 16);
  const unsigned voffV = rowoff0 + (unsigned)(2 * H * HD * 2) + (unsigned)(((lane & 7) ^ (((dkey >> 1) & 1) << 2)) * 16);
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(DM_LDS char *)smem);
  unsigned step_bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)(32 * tok_stride * 2));          // instruction j + 1 of a wave: 32 keys on
  asm volatile("s_nop 4" : "+s"(step_bytes));
  // buffer descriptor of sample b's rows of this head (out-of-range rows read zero): wave-uniform words in scalar registers
  auto sample_rsrc = [&](int u) -> i32x4 {
    const int hd = u / p.B, b = u - hd * p.B;
    const uintptr_t base = reinterpret_cast<uintptr_t>(qkv + (long long)b * N * tok_stride + (long long)hd * HD);
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base & 0xffffffffu));
    rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((base >> 32) & 0xffffu));
    rs[2] = __builtin_amdgcn_readfirstlane((int)(N * tok_stride * 2));
    rs[3] = 0x00020000;
    asm volatile("s_nop 4" : "+s"(rs));                             // v_readfirstlane -> descriptor read by a buffer instruction: 5 wait states, paid here once
    return rs;
  };
  // The staging is done by waves 0..3 whatever NW is: instruction inst = wave + 4 j covers keys 8 inst .. 8 inst + 7, j < NKT.  In the
  // 8-wave form the older wave of each SIMD wins the issue arbitration and otherwise waits ~1900 cycles per sample at the barrier for
  // its partner (stamps): the DMA issue (~100 cycles a piece) goes where that slack is.
  const bool dma_wave = wave < 4;
  auto stage_piece = [&](const i32x4 &rs, int buf, int j) {
    if ((DMQ_ABL & 2) || !dma_wave) return;
    const unsigned kimg = lds0 + (unsigned)(buf * (2 * IMG)) + (unsigned)wave * 1024u, vimg = kimg + (unsigned)IMG;
    lds_dma(rs, kimg + (unsigned)j * 4096u, voffK, (unsigned)j * step_bytes);
    lds_dma(rs, vimg + (unsigned)j * 4096u, voffV, (unsigned)j * step_bytes);
  };
  auto stage_k = [&](const i32x4 &rs, int buf, int j) {
    if ((DMQ_ABL & 2) || !dma_wave) return;
    lds_dma(rs, lds0 + (unsigned)(buf * (2 * IMG)) + (unsigned)wave * 1024u + (unsigned)j * 4096u, voffK, (unsigned)j * step_bytes);
  };
  auto stage_v = [&](const i32x4 &rs, int buf, int j) {
    if ((DMQ_ABL & 2) || !dma_wave) return;
    lds_dma(rs, lds0 + (unsigned)(buf * (2 * IMG)) + (unsigned)IMG + (unsigned)wave * 1024u + (unsigned)j * 4096u, voffV, (unsigned)j * step_bytes);
  };
  // STAG: the pieces of the NEXT unit that tile j's gap stages.  First half (j < HT): K 0 .. HT + 1 and V 0 .. HT - 2 (what the lagging half
  // no longer reads), alternating K, V, dealt evenly over the HT tiles; second half: the rest over the NKT - HT tiles.
  auto stage_stag = [&](const i32x4 &rs, int buf, int j) {
    constexpr int NK1 = (HT + 2 < NKT) ? HT + 2 : NKT, NV1 = HT - 1 > 0 ? HT - 1 : 0, N1 = NK1 + NV1, N2 = 2 * NKT - N1;
    if (j < HT) {
#pragma unroll
      for (int i = 0; i < N1; ++i) {
        if (i * HT / N1 != j) continue;
        // order: K0 V0 K1 V1 ... while V pieces last, then the remaining K pieces
        if (i < 2 * NV1) { if (i & 1) stage_v(rs, buf, i >> 1); else stage_k(rs, buf, i >> 1); }
        else stage_k(rs, buf, i - NV1);
      }
    } else {
#pragma unroll
      for (int i = 0; i < N2; ++i) {
        if (HT + i * (NKT - HT) / N2 != j) continue;
        // order: V(NV1) K(NK1) V(NV1 + 1) K(NK1 + 1) ... while K pieces last, then the remaining V pieces
        constexpr int NK2 = NKT - NK1;
        if (i < 2 * NK2) { if (i & 1) stage_k(rs, buf, NK1 + (i >> 1)); else stage_v(rs, buf, NV1 + (i >> 1)); }
        else stage_v(rs, buf, NV1 + i - NK2);
      }
    }
  };
  auto stage_all = [&](int u, int buf) {
    if (!dma_wave) return;
    const i32x4 rs = sample_rsrc(u);
#pragma unroll
    for (int j = 0; j < NKT; ++j) stage_piece(rs, buf, j);
  };
  // Q^T fragments (B operand): lane (query r, half hh) holds d = 16 ks + 8 hh .. + 7 of its row for k-step ks
  auto load_q = [&](int u, u32x4 (&f)[4]) {
    const int hd = u / p.B, b = u - hd * p.B;
    const bf16_t *qrow = qkv + ((long long)b * N + q) * tok_stride + (long long)hd * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      f[ks] = (wave_live && row_ok && !(DMQ_ABL & 64)) ? *reinterpret_cast<const u32x4 *>(qrow + 16 * ks) : (u32x4){0u, 0u, 0u, 0u};
  };
  // ---- fragment offsets inside the images --------------------------------------------------------------------------------------
  const int kx = (r >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((2 * ks + hh) ^ kx) << 4);                 // + 4096 kt
  // V^T by transposed reads: lane 4 qd + pp of a 16-lane group addresses key row qd, d columns 4 pp .. 4 pp + 3 of the group's 16
  const int ve = (lane >> 4) & 1, qd = (lane >> 2) & 3, pp = lane & 3;
  int voff[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) voff[dt] = (4 * hh + qd) * 128 + ((dt ^ ((qd >> 1) & 1)) << 6) + ve * 32 + pp * 8;   // + (32 kt + 16 s + 8 j2) * 128

  char *wb = smem + 4 * IMG + wave * WB_WAVE;                       // (NW blocks behind the four images)
  float lse_prev = 0.f;
  // results of sample b sit in the wave's LDS block (bf16 rows) until the top of the next iteration: lane L stores the 16-byte
  // chunk L & 7 of rows (L >> 3) + 8 k, i.e. every store instruction covers eight whole 128-byte rows
  // part k (0..3): rows rr + 8 k of the wave's block (one LDS read + one store per lane); the parts are spread over the first tiles
  // of the next sample: four stores issued together at its top queue behind the other waves' (the store path, not the bytes)
  // (read and store sit several MFMA gaps apart: the store then waits for ITS LDS read only, not for the fragment reads behind it)
  u32x4 fl_v = {0u, 0u, 0u, 0u};
  auto flush_read = [&](int k) {
    if (!wave_live || (DMQ_ABL & 32)) return;
    fl_v = *reinterpret_cast<const u32x4 *>(wb + ((lane >> 3) + 8 * k) * WB_PITCH + (lane & 7) * 16);
  };
  auto flush_store = [&](int u, int k) {
    if (!wave_live || (DMQ_ABL & 32)) return;
    const int hd = u / p.B, b = u - hd * p.B;
    bf16_t *orow0 = reinterpret_cast<bf16_t *>(p.out) + ((long long)b * N + q_wave) * H * HD + (long long)hd * HD;
    const int rr = lane >> 3, cc = lane & 7;
    if (q_wave + rr + 8 * k < N) *reinterpret_cast<u32x4 *>(orow0 + (long long)(rr + 8 * k) * H * HD + cc * 8) = fl_v;
    if (k == 0 && hh == 0 && row_ok) p.lse[((long long)b * H + hd) * N + q] = lse_prev;
  };
  auto flush = [&](int u) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { flush_read(k); flush_store(u, k); }
  };

  constexpr float RESCALE_LOG2 = 16.f;                              // a later tile may exceed the reference maximum by 2^16 before l / O are rescaled
  constexpr bool QA = BIAS;                                         // Q^T fragments in accumulator registers
  constexpr bool PAD = NW == 8;
  static_assert(!(BIAS && NW == 8), "the bias rows need the 512-register budget of one wave per SIMD");
  u32x4 qf[4], qld[4];                                              // qld: the next sample's rows, requested late in this sample
  stage_all(u0, 0);
  load_q(u0, qld);
  if constexpr (TAB) load_table(h);
  DMQ_RT(1);
  if (STAG && !older) __builtin_amdgcn_s_barrier();                 // the lagging half's extra barrier (the older half's top of its first unit)
  for (int b = u0; b < u1; ++b) {                                   // b: the unit (head * B + sample)
    const int b0 = u0, b1 = u1;
    const int buf = (b - b0) & 1;
    DMQ_T(0);
    if (DMQ_PRIO_FLIP && NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(0);
    // this sample's K / V / Q have landed (issued during the previous sample).  The table form's waves store their result rows from
    // registers at the end of a unit -- four 16-byte stores and the lse store, the YOUNGEST vector-memory operations at this point: they are
    // left in flight (a store's acknowledgement takes ~1 us: 1 270 cycles per unit at this wait in the stamped skeleton build).
    if (DIRECT && DMQ_STORES_IN_FLIGHT && b != u0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DMQ_T(1);
    if constexpr (STAG) {
      const int hn = b / p.B;
      if constexpr (TAB) {
        if (b == u0) store_table(h);                                // (visible to the other waves behind the barrier)
        else if (hn != h) fill_table(hn);                           // the run crossed into the next head: the OTHER slot, which nobody reads any more
        point_table(hn);
      }
      h = hn;
    } else {
      if (TAB && b == u0) store_table(h);
    }
    __builtin_amdgcn_s_barrier();                                   // ... for every wave; and everyone is done with the other buffer
    DMQ_T(2);
    if constexpr (RUNS && !STAG) {
      const int hn = b / p.B;
      if (TAB && hn != h) {                                         // the run crossed into the next head: everyone is past the old table's last read
        fill_table(hn);
        __builtin_amdgcn_s_barrier();
      }
      h = hn;
    }
    const bool more = b + 1 < b1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = qld[ks];
      if constexpr (QA) park_acc(qf[ks]);
    }
    i32x4 rs_next = {0, 0, 0, 0};
    if (more) rs_next = sample_rsrc(b + 1);
    const char *kimg = smem + buf * (2 * IMG), *vimg = kimg + IMG;

    if (wave_live) {
      f32x16 s0, s1;                                                 // score tiles j (even) / j (odd): named, never indexed at run time
      u32x4 pb0[2], pb1[2];                                          // packed P^T of tiles j even / odd, k-steps 0 / 1
      u32x4 kf[4];
      u32x2 vf[8];
      f32x16 o0, o1, la;                                             // O^T (two 32-row d tiles) and the row sums: accumulator registers
      u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
      asm volatile("" : "+v"(ones));
      float m = 0.f, msc = 0.f, alpha = 1.f;
      auto read_k = [&](int kt) {
        if ((DMQ_ABL & 128) && kt > 0) return;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const u32x4 *>(kimg + kt * 4096 + koff[ks]);
      };
      auto read_v = [&](int kt) {
        if ((DMQ_ABL & 128) && kt > 0) return;
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const char *a = vimg + (32 * kt + 16 * sx) * 128 + voff[dt];
            vf[4 * sx + 2 * dt] = dm_ds_read_tr16(a);
            vf[4 * sx + 2 * dt + 1] = dm_ds_read_tr16(a + 8 * 128);
          }
      };
      auto vfrag = [&](int sx, int dt) { return (u32x4){vf[4 * sx + 2 * dt][0], vf[4 * sx + 2 * dt][1], vf[4 * sx + 2 * dt + 1][0], vf[4 * sx + 2 * dt + 1][1]}; };
      // piece `ks` of tile kt's QK^T chain into `d`
      auto qk_piece = [&](int kt, int ks, f32x16 &d) {
        if ((DMQ_ABL & 16) && kt > 0) return;
        if (ks == 0) {
          if constexpr (TAB) qk_acc<QA, PAD>(d, kf[0], qf[0]);       // d holds bias / scale (read_bias)
          else if constexpr (BIAS) qk_first<QA, PAD>(d, kf[0], qf[0], cinit[kt]);
          else if (RAGGED && kt == NKT - 1) qk_first<QA, PAD>(d, kf[0], qf[0], cinit[0]);
          else qk_first0<QA, PAD>(d, kf[0], qf[0]);
        } else {
          qk_acc<QA, PAD>(d, kf[ks], qf[ks]);
        }
      };
      // piece g (0..3) of tile kt's P.V: (k-step, d tile) = (g >> 1, g & 1); packed P of a k-step may come fresh from the VALU
      auto pv_piece = [&](int kt, int g, const u32x4 (&pb)[2]) {
        const int sx = g >> 1, dt = g & 1;
        f32x16 &o = dt ? o1 : o0;
        if ((DMQ_ABL & 8) && kt > 0) return;
        if (kt == 0 && sx == 0) pv_first<PAD>(o, vfrag(0, dt), pb[0]);
        else pv_acc<PAD>(o, vfrag(sx, dt), pb[sx]);
      };
      auto l_piece = [&](int kt, int sx, const u32x4 (&pb)[2]) {
        if ((DMQ_ABL & 8) && kt > 0) return;
        if (kt == 0 && sx == 0) l_first<PAD>(la, ones, pb[0]);
        else l_acc<PAD>(la, ones, pb[sx]);
      };
      // ---- the VALU work of a tile as a three-stage pipeline over the MFMA gaps ---------------------------------------------------------
      // One wave per SIMD has nobody to cover a dependent VALU result's latency (tools/hip/mb_coissue.hip: fma, fma, exp, exp, cvt on
      // each other's results = 55 cycles instead of the 28 they take to issue; the MFMA itself hides completely beside them).  So a
      // gap holds stage F of score pair k (s * scale2 - m * scale2), stage E of pair k - 1 (exp2) and stage C of pair k - 2 (pack to
      // bf16): nothing in a gap depends on anything computed in the same or the previous instruction.
      float fa[8][2], ex[8][2];
      auto stage_f = [&](const f32x16 &sc, int k) {
        if (DMQ_ABL & 4) return;
        fa[k][0] = __builtin_fmaf(sc[2 * k], scale2, msc);
        fa[k][1] = __builtin_fmaf(sc[2 * k + 1], scale2, msc);
        asm volatile("" :: "v"(fa[k][0]), "v"(fa[k][1]));            // (stays in ITS gap: hipcc otherwise sinks it to the use)
      };
      auto stage_e = [&](int k) {
        if (DMQ_ABL & 4) return;
        ex[k][0] = __builtin_amdgcn_exp2f(fa[k][0]);
        ex[k][1] = __builtin_amdgcn_exp2f(fa[k][1]);
        asm volatile("" :: "v"(ex[k][0]), "v"(ex[k][1]));
      };
      auto stage_c = [&](int k, u32x4 (&pb)[2]) {
        if (DMQ_ABL & 4) { pb[k >> 2][k & 3] = 0x3f803f80u; return; }
        const unsigned w = pk_bf16(ex[k][0], ex[k][1]);
        pb[k >> 2][k & 3] = w;
        asm volatile("" :: "v"(w));
      };
      // running maximum of the NEXT tile's scores, two v_max3 per call (calls c = 0..3), started once its MFMAs are two gaps behind
      float t0 = 0.f, t1 = 0.f;
      auto stage_m = [&](f32x16 &sn, int c) {
        if (c == 0) {
          asm volatile("" : "+v"(sn));                               // the scores are read below this point only (MFMA D -> VALU distance)
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
      // reference maximum: tile 0 sets it (row maximum over both lane halves); a later tile only moves it when one of its scores is more
      // than 2^RESCALE_LOG2 above (checked per half: no exchange on the common path).  t0 / t1 hold the tile's running maxima.
      auto tile_max = [&](int kt) -> bool {
        const float th = __builtin_fmaxf(t0, t1);
        if (kt == 0) {
          m = half_max(th);
          msc = -m * scale2;
          return false;
        }
        const bool grow = (th - m) * scale2 > RESCALE_LOG2;
        if (__builtin_amdgcn_ballot_w64(grow) == 0) return false;    // wave-uniform: almost always
        const float mn = __builtin_fmaxf(m, half_max(th));
        alpha = __builtin_amdgcn_exp2f((m - mn) * scale2);           // 1 for the rows whose maximum did not grow
        m = mn;
        msc = -m * scale2;
        return true;
      };
      auto rescale_o = [&]() {                                       // after the pending P.V / row-sum MFMAs of tile j - 1, before tile j's
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(o0), "+a"(o1), "+a"(la));
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; la[i] *= alpha; }
        asm volatile("s_nop 3" : "+a"(o0), "+a"(o1), "+a"(la));
      };

      // ---- prologue: tile 0's scores and their maximum ------------------------------------------------------------------------------------
      if constexpr (TAB) {
        read_bias(0, s0);
        if (NKT > 1) read_bias(1, s1);
      }
      read_k(0);
      asm volatile("s_nop 1");                                       // Q^T copies into accumulator registers may be fresh (VALU write -> MFMA operand)
      qk_piece(0, 0, s0); qk_piece(0, 1, s0); qk_piece(0, 2, s0); qk_piece(0, 3, s0);
      if (NKT > 1) read_k(1);
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s0));
      stage_m(s0, 0); stage_m(s0, 1); stage_m(s0, 2); stage_m(s0, 3);
      DMQ_T(3);
#pragma unroll
      for (int j = 0; j < NKT; ++j) {
        if (j == 4) DMQ_T(4);
        // Two waves per SIMD: the older one (waves 0-3) wins the issue arbitration and reached the unit's barrier ~2700 cycles before its
        // partner (stamps, profiles/r05_attn_fwd_ablations_and_stamps.md).  With -DDMQ_PRIO_FLIP=1 the younger half takes priority for the second half of the tiles so the
        // two finish together: measured inside the noise (30.2 / 30.5 us with, 31.4 / 30.6 without), OFF by default.
        if (DMQ_PRIO_FLIP && NW == 8 && j == NKT / 2 && wave >= 4) __builtin_amdgcn_s_setprio(1);
        if (STAG && j == HT) __builtin_amdgcn_s_barrier();          // the unit's second barrier (the other half's top)
        // iteration j, ten MFMA gaps: QK^T(j + 1) x 4, row sums(j - 1) x 2, P.V(j - 1) x 4, with tile j's VALU pipeline between them
        f32x16 &sc = (j & 1) ? s1 : s0;                              // tile j's scores; tile j + 1 accumulates into sn
        f32x16 &sn = (j & 1) ? s0 : s1;
        u32x4 (&pbc)[2] = (j & 1) ? pb1 : pb0;
        u32x4 (&pbp)[2] = (j & 1) ? pb0 : pb1;
        const bool resc = tile_max(j);
        if (j > 0) read_v(j - 1);
        __builtin_amdgcn_sched_barrier(0);
        // The operands of an MFMA stay LIVE to the end of its gap (the empty asm statements): to hipcc an asm MFMA has read its
        // operands when it is issued, so it would reuse a fragment's registers for the very next VALU results.
#pragma unroll
        for (int g = 0; g < 10; ++g) {
          if (g < 4) {
            if (j + 1 < NKT) qk_piece(j + 1, g, sn);
          } else if (g < 6) {
            if (j > 0) l_piece(j - 1, g - 4, pbp);
          } else {
            if (j > 0) pv_piece(j - 1, g - 6, pbp);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (g < 8) stage_f(sc, g);
          if (g >= 1 && g < 9) stage_e(g - 1);
          if (g >= 2) stage_c(g - 2, pbc);
          if (g >= 6 && j + 1 < NKT) stage_m(sn, g - 6);
          if (g == 4) {
            if (j + 2 < NKT) read_k(j + 2);
            if (QA && more && j == NKT - 2) load_q(b + 1, qld);      // bias instances: into the K fragment registers, free from here on
            if (!QA && more && j == 2) load_q(b + 1, qld);           // next sample's Q rows (registers to spare without a bias)
          }
          // previous sample's rows: a quarter per tile over the LAST four tiles, stored six gaps after its LDS read.  (The vector-memory
          // instructions of a sample -- K / V pieces, Q rows, these stores -- are spread over its tiles: bunched into the first
          // four they ran into the CU's ~11 B/clk memory path, +1200 cycles per sample in the 8-wave form.)
          if constexpr (!DIRECT) {
            if (g == 1 && b > b0 && j >= NKT - 4) flush_read(j - (NKT - 4));
            if (g == 7 && b > b0 && j >= NKT - 4) flush_store(b - 1, j - (NKT - 4));
          }
          if (TAB && g == 8 && j + 2 < NKT) read_bias(j + 2, sc);     // tile j's scores were last read in gap 7; tile j + 2 accumulates onto these
          if (g == 5 && more) {                                      // next sample's K / V: one (K, V) pair of pieces per tile, all before the last
            if constexpr (STAG) {
              stage_stag(rs_next, buf ^ 1, j);
            } else {
              if (j < NKT - 1) stage_piece(rs_next, buf ^ 1, j);
              if (j == 0) stage_piece(rs_next, buf ^ 1, NKT - 1);
            }
          }
          if (g < 4) {
            if (j + 1 < NKT) asm volatile("" :: "v"(kf[g]));
          } else if (g < 6) {
            if (j > 0) asm volatile("" :: "v"(pbp[g - 4]), "v"(ones));
          } else {
            if (j > 0) asm volatile("" :: "v"(vf[2 * (g - 6)]), "v"(vf[2 * (g - 6) + 1]), "v"(pbp[(g - 6) >> 1]));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (resc) rescale_o();
      }
      DMQ_T(5);
      // ---- epilogue: P.V and row sums of the last tile, normalise, park the rows in the wave's LDS block ---------------------------------
      read_v(NKT - 1);
      {
        u32x4 (&pbl)[2] = ((NKT - 1) & 1) ? pb1 : pb0;
        asm volatile("s_nop 1");                                     // the last packed pairs were written in the last gaps (VALU write -> MFMA operand)
        l_piece(NKT - 1, 0, pbl);
        l_piece(NKT - 1, 1, pbl);
#pragma unroll
        for (int g = 0; g < 4; ++g) pv_piece(NKT - 1, g, pbl);
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(o0), "+a"(o1), "+a"(la) : "v"(pbl[0]), "v"(pbl[1]), "v"(ones));
      }
      const float l = la[0];                                         // every row of `la` is the row sum of this lane's query
      const float inv = 1.f / l;
      lse_prev = (m * scale2 + __builtin_amdgcn_logf(l)) * LN2;      // natural-log units for the backward kernels
      // lane (query r, half hh) holds d = 32 dt + 8 c + 4 hh + e in register 4 c + e of o<dt>
      if constexpr (DIRECT) {
        // the two halves of a row trade 8-byte pieces so that each lane owns 16 contiguous bytes: half 0 keeps the even c, half 1 the odd
        const int smp = b - h * p.B;
        bf16_t *orow = reinterpret_cast<bf16_t *>(p.out) + ((long long)smp * N + q) * H * HD + (long long)h * HD + 8 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const f32x16 &o = dt ? o1 : o0;
#pragma unroll
          for (int cp = 0; cp < 2; ++cp) {
            const int c0 = 2 * cp, c1 = c0 + 1;
            u32x4 w;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const unsigned a = pk_bf16(o[4 * c0 + 2 * k] * inv, o[4 * c0 + 2 * k + 1] * inv);
              const unsigned bb = pk_bf16(o[4 * c1 + 2 * k] * inv, o[4 * c1 + 2 * k + 1] * inv);
              const auto sw = __builtin_amdgcn_permlane32_swap(a, bb, false, false);
              w[k] = (unsigned)sw[0];
              w[2 + k] = (unsigned)sw[1];
            }
            *reinterpret_cast<u32x4 *>(orow + 32 * dt + 16 * cp) = w;
          }
        }
        if (hh == 0) p.lse[((long long)smp * H + h) * N + q] = lse_prev;
      } else
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const u32x2 w0 = {pk_bf16(o0[4 * c] * inv, o0[4 * c + 1] * inv), pk_bf16(o0[4 * c + 2] * inv, o0[4 * c + 3] * inv)};
        const u32x2 w1 = {pk_bf16(o1[4 * c] * inv, o1[4 * c + 1] * inv), pk_bf16(o1[4 * c + 2] * inv, o1[4 * c + 3] * inv)};
        *reinterpret_cast<u32x2 *>(wb + r * WB_PITCH + (8 * c + 4 * hh) * 2) = w0;
        *reinterpret_cast<u32x2 *>(wb + r * WB_PITCH + (32 + 8 * c + 4 * hh) * 2) = w1;
      }
      DMQ_T(6);
    } else {
      if (STAG) __builtin_amdgcn_s_barrier();                        // (a wave without rows: the unit's second barrier; such waves are never DMA waves)
      if (more) stage_all(b + 1, buf ^ 1);                           // a wave without rows still stages its share
    }
  }
  if (STAG && older) __builtin_amdgcn_s_barrier();                  // the older half's extra barrier (the lagging half's middle of its last unit)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (!DIRECT) flush(u1 - 1);
#ifdef DMQ_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DMQ_RT(2);
#endif
}

inline void grid(int B, int N, int H, int rows, int &nblk, int &chunks, int &bchunk) {
  nblk = (N + rows - 1) / rows;
  chunks = 256 / (H * nblk);
  if (chunks < 1) chunks = 1;
  if (chunks > B) chunks = B;
  bchunk = (B + chunks - 1) / chunks;
  chunks = (B + bchunk - 1) / bchunk;
}

template <int NKT, bool RAGGED, int BM, int NW> bool launch(const AttnPipeParams &p, hipStream_t s) {
  constexpr int LDS = 4 * NKT * 32 * 128 + (BM == 2 && NW == 8 ? 0 : NW * WB_WAVE) + (BM == 2 ? (NKT - 1) * 15 * TAB_PITCH * 4 * (NW == 8 ? 2 : 1) : 0);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_q32_kernel<NKT, RAGGED, BM, NW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  if (!ok) return false;
  int nblk, chunks, bchunk;
  grid(p.B, p.N, p.H, 32 * NW, nblk, chunks, bchunk);
  const int wgs = NW == 8 ? (p.B * p.H < 256 ? p.B * p.H : 256) : grid_size(nblk, p.H, chunks);      // 8 waves: one run of units per CU
  hipLaunchKernelGGL((attn_fwd_q32_kernel<NKT, RAGGED, BM, NW>), dim3(wgs), dim3(64 * NW), LDS, s, p, bchunk, nblk, chunks);
  return true;
}

template <int NKT> bool launch_n(const AttnPipeParams &p, hipStream_t s) {
  const bool ragged = p.N != NKT * 32;
  if constexpr (NKT % 2 == 0) {
    if (p.table) {                               // DM_ATTN_Q32_TABW=4: the table form with one wave per SIMD (A/B runs)
      static const int tabw = [] { const char *e = getenv("DM_ATTN_Q32_TABW"); return e ? atoi(e) : 8; }();
      return tabw == 4 ? launch<NKT, false, 2, 4>(p, s) : launch<NKT, false, 2, 8>(p, s);
    }
  }
  if (p.bias) return ragged ? launch<NKT, true, 1, 4>(p, s) : launch<NKT, false, 1, 4>(p, s);
  if constexpr (NKT <= 7) {                     // 8 waves: two per SIMD, K / V staged once per (sample, head); DM_ATTN_Q32_W8=0 for A/B runs
    static const bool w8 = [] { const char *e = getenv("DM_ATTN_Q32_W8"); return !(e && atoi(e) == 0); }();
    if (w8) return ragged ? launch<NKT, true, 0, 8>(p, s) : launch<NKT, false, 0, 8>(p, s);
  }
  return ragged ? launch<NKT, true, 0, 4>(p, s) : launch<NKT, false, 0, 4>(p, s);
}

}  // namespace dmq32

// bf16, head dim 64, 128 < N <= 256 (5 .. 8 key tiles of 32).  DM_ATTN_Q32=0 keeps the 16-row pipelined kernels (A/B runs).
bool dm_attn_fwd_q32_takes(const AttnPipeParams &p) {
  static const int mode = [] { const char *e = getenv("DM_ATTN_Q32"); return e ? atoi(e) : 1; }();
  if (mode == 0) return false;
  if (p.N <= 128 || p.N > 256) return false;
  if ((long long)p.N * 3 * p.H * 64 * 2 >= (1LL << 31)) return false;          // one sample's rows must fit a 32-bit DMA offset
  if (mode != 2 && p.B * p.H < 96) return false;                              // too little work for persistent workgroups
  if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15u)) return false;
  if (p.table && (p.bias || p.N != 64 * p.cube_s || (p.cube_s != 3 && p.cube_s != 4))) return false;
  return true;
}

bool dm_attn_fwd_q32(const AttnPipeParams &p, hipStream_t s) {
  if (!dm_attn_fwd_q32_takes(p)) return false;
  switch ((p.N + 31) / 32) {
    case 5: return dmq32::launch_n<5>(p, s);
    case 6: return dmq32::launch_n<6>(p, s);
    case 7: return dmq32::launch_n<7>(p, s);
    case 8: return dmq32::launch_n<8>(p, s);
    default: return false;
  }
}
