// A stand-in for a collective's kernels on a one-GPU box: `blocks` workgroups of 256 threads (32 KB of LDS, ~64 registers: the footprint of an
// RCCL channel, one per CU) that hold their CUs for `micros` microseconds.  tools/mb_cu_hog.py launches it on a side stream next to the training
// step to price what a bucket all-reduce in flight costs the one-workgroup-per-CU GEMM grids, and what DM_GEMM_CUS_RESERVED buys back.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void cu_hog_kernel(unsigned long long ticks, unsigned int *sink) {
  __shared__ unsigned int pad[8192];
  pad[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  unsigned int acc = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    acc += pad[(threadIdx.x * 7 + acc) & 8191];
    __builtin_amdgcn_s_sleep(8);
  }
  if (acc == 0xdeadbeefu) sink[0] = acc;
}
extern "C" int cu_hog_launch(void *stream, int blocks, int micros, unsigned int *sink) {
  hipLaunchKernelGGL(cu_hog_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (unsigned long long)micros * 100ull, sink);
  return (int)hipGetLastError();
}
