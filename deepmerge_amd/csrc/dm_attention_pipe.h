// Persistent, LDS-DMA-pipelined attention kernels (dm_attention_pipe.hip); dm_attention.hip routes to them.
#pragma once
#include <hip/hip_runtime.h>

struct AttnPipeParams {
  const void *qkv;      // [B, N, 3, H, 64] bf16
  const float *bias;    // [H, N, N] fp32 or NULL
  void *out;            // [B, N, H*64] bf16
  float *lse;           // [B, H, N]
  int B, N, H;
  float scale;
};

// true if the pipelined forward kernel took the call (bf16, N a multiple of 16 in [128, 256]); false: nothing launched
bool dm_attn_fwd_pipe(const AttnPipeParams &p, hipStream_t s);
