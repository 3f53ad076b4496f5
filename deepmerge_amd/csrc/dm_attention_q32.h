// Shared pieces of the 32-rows-per-wave attention kernels (dm_attention_q32.hip: forward; dm_attention_q32_bwd.hip: dQ, dK / dV):
// inline-asm MFMAs with explicit register classes, the asm LDS-DMA, lane-exchange helpers, the workgroup -> (head, block, chunk) map.
// Why the MFMAs are asm and what that obliges the kernels to do by hand is explained at the top of dm_attention_q32.hip.
#pragma once
#include <cstdlib>

#include "dm_attention_pipe.h"
#include "dm_common.h"
#include "dm_mfma.h"

namespace dmq32 {

constexpr int HD = 64;
// query rows per workgroup = 32 NW.  NW = 4: one wave per SIMD and up to 512 registers (what the bias instances need).  NW = 8 (no
// bias, <= 7 key tiles: 8 waves x 240 registers, K / V for all rows of a (sample, head) staged once): two waves per SIMD -- a single
// wave issues a VALU instruction every 4 cycles, two waves share the SIMD's full rate of one per 2, and one wave's MFMAs run under the
// other's VALU work (the 4-wave form is ISSUE-bound: ~610 cycles of instruction issue per tile against 320 of the matrix pipe).
constexpr int WB_PITCH = 144;                   // write-back staging: 128-byte row + 16 bytes (keeps rows 16-byte aligned)
constexpr int WB_WAVE = 32 * WB_PITCH;
constexpr float NEG_BIG = -1.0e30f;
// Row pitch (floats) of the head's relative-position table in LDS: 15 entries per (dz, dy) row.  A lane's address is
// pitch * (its token's y) -+ (its token's x) + const, and a 32-lane group of a ds_read2_b32 covers 4 y x 8 x: with a pitch of 16 the rows
// y and y + 2 fall on the same 8 banks (2-way conflict on every bias read: the 50 k conflict cycles of profiles/r04_attn_mfma_util.md);
// 24 puts the four rows on banks 0-7 / 24-31 / 16-23 / 8-15.
constexpr int TAB_PITCH = 24;

// ---- MFMAs with explicit register classes (see the header: hipcc pads nothing inside or around these) ----------------------------
#define DMQ_MFMA "v_mfma_f32_32x32x16_bf16"
// scores: D, C in VGPRs; A = K fragment (VGPR), B = Q^T fragment: an accumulator register (QA, the bias instances: their 128 bias
// registers leave no room in the architectural file) or a VGPR
// PAD (the 8-wave instances): under their 256-register budget hipcc parks values in accumulator registers and brings them back with
// v_accvgpr_read directly in front of an asm MFMA that reads them (tools/isa_hazards.py found it) -- every MFMA there opens with
// `s_nop 1`; with two waves per SIMD a wave's issue slots are not the limit.
#define DMQ_PAD "s_nop 1\n\t"
template <bool QA, bool PAD> __device__ __forceinline__ void qk_first(f32x16 &d, const u32x4 &k, const u32x4 &q, const f32x16 &c) {
  if constexpr (QA && PAD) asm volatile(DMQ_PAD DMQ_MFMA " %0, %1, %2, %3" : "=&v"(d) : "v"(k), "a"(q), "v"(c));
  else if constexpr (QA) asm volatile(DMQ_MFMA " %0, %1, %2, %3" : "=&v"(d) : "v"(k), "a"(q), "v"(c));
  else if constexpr (PAD) asm volatile(DMQ_PAD DMQ_MFMA " %0, %1, %2, %3" : "=&v"(d) : "v"(k), "v"(q), "v"(c));
  else asm volatile(DMQ_MFMA " %0, %1, %2, %3" : "=&v"(d) : "v"(k), "v"(q), "v"(c));
}
template <bool QA, bool PAD> __device__ __forceinline__ void qk_first0(f32x16 &d, const u32x4 &k, const u32x4 &q) {
  if constexpr (QA && PAD) asm volatile(DMQ_PAD DMQ_MFMA " %0, %1, %2, 0" : "=&v"(d) : "v"(k), "a"(q));
  else if constexpr (QA) asm volatile(DMQ_MFMA " %0, %1, %2, 0" : "=&v"(d) : "v"(k), "a"(q));
  else if constexpr (PAD) asm volatile(DMQ_PAD DMQ_MFMA " %0, %1, %2, 0" : "=&v"(d) : "v"(k), "v"(q));
  else asm volatile(DMQ_MFMA " %0, %1, %2, 0" : "=&v"(d) : "v"(k), "v"(q));
}
template <bool QA, bool PAD> __device__ __forceinline__ void qk_acc(f32x16 &d, const u32x4 &k, const u32x4 &q) {
  if constexpr (QA && PAD) asm volatile(DMQ_PAD DMQ_MFMA " %0, %1, %2, %0" : "+v"(d) : "v"(k), "a"(q));
  else if constexpr (QA) asm volatile(DMQ_MFMA " %0, %1, %2, %0" : "+v"(d) : "v"(k), "a"(q));
  else if constexpr (PAD) asm volatile(DMQ_PAD DMQ_MFMA " %0, %1, %2, %0" : "+v"(d) : "v"(k), "v"(q));
  else asm volatile(DMQ_MFMA " %0, %1, %2, %0" : "+v"(d) : "v"(k), "v"(q));
}
// O^T: C = D in accumulator registers; A = V^T fragment, B = packed P^T (both VGPRs).  PAD: an operand may come fresh from the VALU.
template <bool PAD> __device__ __forceinline__ void pv_acc(f32x16 &o, const u32x4 &v, const u32x4 &pb) {
  if constexpr (PAD) asm volatile(DMQ_PAD DMQ_MFMA " %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pb));
  else asm volatile(DMQ_MFMA " %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pb));
}
// row sums on the matrix pipe: A = ones, B = packed P^T -> every row of D holds, per query column, the sum over the k-step's 16 keys of
// BOTH lane halves (what the P.V product sees, bf16-rounded); 2 MFMAs per tile replace 16 v_add_f32 per lane and the final exchange
template <bool PAD> __device__ __forceinline__ void l_first(f32x16 &l, const u32x4 &ones, const u32x4 &pb) {
  if constexpr (PAD) asm volatile(DMQ_PAD DMQ_MFMA " %0, %1, %2, 0" : "=&a"(l) : "v"(ones), "v"(pb));
  else asm volatile(DMQ_MFMA " %0, %1, %2, 0" : "=&a"(l) : "v"(ones), "v"(pb));
}
template <bool PAD> __device__ __forceinline__ void l_acc(f32x16 &l, const u32x4 &ones, const u32x4 &pb) {
  if constexpr (PAD) asm volatile(DMQ_PAD DMQ_MFMA " %0, %1, %2, %0" : "+a"(l) : "v"(ones), "v"(pb));
  else asm volatile(DMQ_MFMA " %0, %1, %2, %0" : "+a"(l) : "v"(ones), "v"(pb));
}
template <bool PAD> __device__ __forceinline__ void pv_first(f32x16 &o, const u32x4 &v, const u32x4 &pb) {
  if constexpr (PAD) asm volatile(DMQ_PAD DMQ_MFMA " %0, %1, %2, 0" : "=&a"(o) : "v"(v), "v"(pb));
  else asm volatile(DMQ_MFMA " %0, %1, %2, 0" : "=&a"(o) : "v"(v), "v"(pb));
}
// A 128-bit value parked in accumulator registers: the "+a" operand makes hipcc copy it into ONE contiguous a[n:n+3] tuple here
// (four v_accvgpr_write of its own, padded by itself), and from here on the value lives in that class -- the "a" operands of the
// MFMAs below then need no copies.  (Four scalar "=a" outputs instead gave scattered registers that hipcc gathered with
// v_accvgpr_mov directly in front of every MFMA: extra VALU work and, with no wait state before an asm MFMA, stale operands.)
__device__ __forceinline__ void park_acc(u32x4 &v) { asm volatile("" : "+a"(v)); }
// This file is built with -fno-slp-vectorize -ffinite-math-only (Makefile): adjacent f32 adds / fmas stay single instructions
// (v_pk_*_f32 beside MFMAs cost more than two scalar ops) and fmaxf on MFMA results gets no NaN canonicalisation in front.
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  const bf16x2 r = {(bf16_t)lo, (bf16_t)hi};
  return __builtin_bit_cast(unsigned, r);
}
// maximum / sum over the two half-waves (lane and lane ^ 32 hold the two key halves of one query row)
__device__ __forceinline__ float half_max(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
// LDS-DMA as inline asm: hipcc orders a `buffer_load ... lds` builtin against every later ds_read with s_waitcnt vmcnt(0) (it sees an
// LDS store), i.e. a wave that issues the next sample's K / V would wait for them at once.  In asm the transfer is invisible to that
// bookkeeping; the kernel's own vmcnt(0) + barrier at the top of the next sample orders it (the destination is the OTHER buffer).
// lds = wave-uniform LDS byte address of the 1 KiB piece, voff = per-lane byte offset, soff = wave-uniform byte offset.
__device__ __forceinline__ void lds_dma(const i32x4 &rsrc, unsigned lds, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds), "s"(rsrc), "s"(soff) : "memory");
}

// Workgroup -> (head, row block, sample chunk): ids L, L + 8, ... share an XCD (round-robin dispatch) and are the row blocks of one
// (head, chunk) group, so the second reader of a K / V row hits that XCD's L2 (same mapping as dm_attention_pipe.hip).
__device__ __forceinline__ bool coords(int nblk, int H, int chunks, int &h, int &rb, int &chunk) {
  const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
  rb = j % nblk;
  const int group = xcd + 8 * (j / nblk);
  if (group >= H * chunks) return false;
  h = group % H;
  chunk = group / H;
  return true;
}
inline int grid_size(int nblk, int H, int chunks) { return (H * chunks + 7) / 8 * 8 * nblk; }

}  // namespace dmq32
