// Micro-benchmark (tuning aid, not part of the library): how fast can a CU fill LDS by `buffer_load_dwordx4 ... lds` when the
// access pattern is a GEMM operand's -- P row panels of `rows` rows x `pitch` bytes, walked `row_bytes` at a time along k?
// Every workgroup (256 threads) streams panel (blockIdx.x / 8) % panels; a wave instruction covers 1024 / row_bytes rows.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DMA(rsrc, dst, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst), 16, voff, soff, 0, 0)

template <int DEPTH, int PIECES>   // stages in flight, pieces (1 KiB) per wave and stage
__global__ __launch_bounds__(256) void fill_kernel(const char *src, int panels, int rows, int pitch, int row_bytes, int ksteps, int *sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int panel = (blockIdx.x >> 3) % panels;
  const char *base = src + (long long)panel * rows * pitch;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, rows * pitch, 0x00020000);
  const int lpr = row_bytes / 16, rpp = 1024 / row_bytes;      // lanes per row, rows per piece
  int vo[PIECES];
#pragma unroll
  for (int u = 0; u < PIECES; ++u) {
    const int row = ((wave * PIECES + u) * rpp + lane / lpr) % rows;
    vo[u] = row * pitch + (lane % lpr) * 16;
  }
  const int kmax = pitch / row_bytes;
  for (int kt = 0; kt < ksteps; ++kt) {
    const int slot = kt % DEPTH;
#pragma unroll
    for (int u = 0; u < PIECES; ++u) DMA(rs, smem + (slot * PIECES * 4 + wave * PIECES + u) * 1024, (int)vo[u], (int)((kt % kmax) * row_bytes));
    if constexpr (DEPTH == 2) { if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
    else if constexpr (DEPTH == 3) { if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
    else { if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<int *>(smem + 16 * (blockIdx.x & 63));
}

extern "C" int fill_bench(const void *src, int panels, int rows, int pitch, int row_bytes, int ksteps, int depth, int pieces, int grid, void *sink, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const int lds = depth * pieces * 4 * 1024;
#define LAUNCH(D, P)                                                                                                          \
  if (depth == D && pieces == P) {                                                                                            \
    hipFuncSetAttribute(reinterpret_cast<const void *>(fill_kernel<D, P>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);  \
    hipLaunchKernelGGL((fill_kernel<D, P>), dim3(grid), dim3(256), lds, s, (const char *)src, panels, rows, pitch, row_bytes, ksteps, (int *)sink); \
    return (int)hipGetLastError();                                                                                            \
  }
  LAUNCH(2, 6) LAUNCH(3, 6) LAUNCH(4, 6) LAUNCH(2, 12) LAUNCH(3, 12)
  return -1;
}
