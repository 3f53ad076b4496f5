// MFMA wrappers shared by the GEMM and attention kernels (gfx950).
#pragma once
#include "dm_common.h"

// A "fragment" is 16 bytes per lane: lane group g = lane>>4 owns the g-th 16-byte chunk of a
// 64-byte k-block (k = 8g..8g+7 for bf16, 4g..4g+3 for fp32); lane&15 selects the row.
// ---- MFMA (operands swapped: lane ends with 4 consecutive n of row m = lane&15) ---------------
template <typename T> __device__ __forceinline__ void mma(f32x4 &acc, const u32x4 &a, const u32x4 &b);
template <> __device__ __forceinline__ void mma<bf16_t>(f32x4 &acc, const u32x4 &a, const u32x4 &b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<float>(f32x4 &acc, const u32x4 &a, const u32x4 &b) {
  const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[j], acc, 0, 0, 0);
}


// ds_read_b64_tr_b16 (hardware transpose read).  Within each group of 16 consecutive lanes,
// lane 4q+p supplies the address of row q, columns 4p..4p+3 (8 bytes) of a 4-row x 16-column
// block of 16-bit elements; lane i of the group receives column i of the 4 rows (row q in
// element q).  EXEC must be all ones at the call site.
__device__ __forceinline__ u32x2 dm_ds_read_tr16(const char *lds_addr) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((DM_LDS s16x4 *)(lds_addr));
  return __builtin_bit_cast(u32x2, v);
}
