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
  const float *table = nullptr;   // [bins, H] fp32 relative-position table of a (cube_s, 8, 8) token cube, instead of `bias` (q32 kernels only)
  int cube_s = 0;
};

// true if the pipelined forward kernel took the call (bf16, N a multiple of 16 in [128, 256]); false: nothing launched
bool dm_attn_fwd_pipe(const AttnPipeParams &p, hipStream_t s);

// Forward with 32 query rows per wave on the 32x32x16 MFMA, online softmax (dm_attention_q32.hip): bf16, 128 < N <= 256.
// true if it took the call; DM_ATTN_Q32=0 disables it (A/B runs against the kernel above).
bool dm_attn_fwd_q32(const AttnPipeParams &p, hipStream_t s);
bool dm_attn_fwd_q32_takes(const AttnPipeParams &p);      // the same decision without launching

struct AttnPipeBwdParams {
  const void *qkv;      // [B, N, 3, H, 64] bf16
  const float *bias;    // [H, N, N] fp32 or NULL
  const void *out;      // [B, N, H*64] bf16 (forward output)
  const void *dout;     // [B, N, H*64] bf16
  const float *lse;     // [B, H, N]
  float *delta;         // [B, H, N] scratch: written by the dQ kernel, read by the dK/dV kernel
  void *dqkv;           // [B, N, 3, H, 64] bf16, fully written
  float *slab;          // [chunks, H, N, N] fp32 or NULL: sum over the chunk's samples of dS
  int B, N, H;
  float scale;
  const float *table = nullptr;   // [bins, H] relative-position table of a (cube_s, 8, 8) token cube: the q32 dQ kernel reads it instead of `bias`
  int cube_s = 0;
};

// Number of batch chunks the pipelined backward uses (first dimension of `slab`); 0 if it does not take this shape.
int dm_attn_bwd_pipe_chunks(int B, int N, int H, int dtype_is_bf16);
// true if the pipelined dQ and dK/dV kernels take this shape (`dm_attn_bwd_pipe` would return true)
bool dm_attn_bwd_pipe_ok(const AttnPipeBwdParams &p);
// true if the pipelined kernels took the call; dq_done: dQ and delta are already written (dm_attn_bwd_dq_q32), only dK / dV (+ slab) run
bool dm_attn_bwd_pipe(const AttnPipeBwdParams &p, hipStream_t s, bool dq_done = false);
// dQ + delta with 32 query rows per wave (dm_attention_q32_bwd.hip): bf16, 128 < N <= 256; true if it took the call.  Call it only
// where `dm_attn_bwd_pipe_ok` holds (the generic dQ kernel also produces the bias-gradient slab, this one does not).
bool dm_attn_bwd_dq_q32(const AttnPipeBwdParams &p, hipStream_t s);
// dK / dV with 32 keys per wave, bias-free shapes only (reads p.delta: run a dQ pass first); true if it took the call
bool dm_attn_bwd_dkv_q32(const AttnPipeBwdParams &p, hipStream_t s);
// true if BOTH table-reading backward kernels (dQ and dK / dV with the head's table in LDS) take this shape under the current
// switches -- the only case in which a backward pass may run without the dense bias rows (the decision, without launching)
bool dm_attn_bwd_tab_takes(const AttnPipeBwdParams &p);
