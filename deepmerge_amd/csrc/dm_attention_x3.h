// Attention kernels of the "bf16x3" numerics mode (dm_attention_x3.hip): fp32 tensors, split-bf16 products on the matrix pipe.
#pragma once
#include <hip/hip_runtime.h>

#include "dm_common.h"

struct AttnX3Params {
  const bf16_t *hi, *lo;   // the split images of qkv [B, N, 3, H, 64] (dm_attn_x3_split)
  const float *table;      // [bins, H] relative-position table of a (cube_s, 8, 8) token cube, or NULL (no bias)
  int cube_s;
  float *out;              // [B, N, H*64] fp32
  float *lse;              // [B, H, N]
  int B, N, H;
  float scale;
  bf16_t *out_pair;        // or NULL: the hi / lo plane pair of `out` on the side (lo plane B*N*H*64 elements behind the hi plane)
};

// shapes the kernels take: head dim 64, 128 < N <= 256, bias absent or a (3 | 4, 8, 8) cube's table; DM_ATTN_X3=0 switches them off
bool dm_attn_x3_shape(int B, int N, int H, bool has_table, int cube_s);
// x [n] fp32 -> hi = bf16(x), lo = bf16(x - hi); n % 4 == 0
void dm_attn_x3_split(const float *x, void *hi, void *lo, long long n, hipStream_t s);
bool dm_attn_fwd_x3(const AttnX3Params &p, hipStream_t s);

struct AttnX3BwdParams {
  const bf16_t *hi, *lo;       // split images of qkv (from the forward pass)
  const bf16_t *dohi, *dolo;   // split images of dout [B, N, H*64]
  const float *out, *dout;     // [B, N, H*64] fp32 (delta = rowsum(dO . O))
  const float *lse;            // [B, H, N]
  float *delta;                // [B, H, N]: written by the dQ pass, read by the dK / dV pass
  float *dqkv;                 // [B, N, 3, H, 64] fp32, fully written by the two passes (unless dqkv_pair)
  bf16_t *dqkv_pair;           // or NULL: the result as a hi / lo plane pair instead (lo plane pair_plane elements behind the hi plane)
  long long pair_plane;
  float *slab;                 // [chunks, H, N, N] or NULL
  const float *table;
  int cube_s;
  int B, N, H;
  float scale;
};
bool dm_attn_bwd_dq_x3(const AttnX3BwdParams &p, hipStream_t s);
bool dm_attn_bwd_dkv_x3(const AttnX3BwdParams &p, hipStream_t s);
int dm_attn_x3_chunks(int B, int N, int H);
