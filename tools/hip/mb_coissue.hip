// Does the VALU run beside an MFMA issued by the SAME wave (one wave per SIMD), and does that depend on whether the MFMA's operands
// are architectural VGPRs or accumulator registers?  hipcc --offload-arch=gfx950 -O2 mb_coissue.hip -o mb_coissue && ./mb_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define VALU6 "v_fma_f32 v40, v40, v46, v47\n\tv_fma_f32 v41, v41, v46, v47\n\tv_exp_f32 v42, v42\n\tv_exp_f32 v43, v43\n\tv_fma_f32 v44, v44, v46, v47\n\tv_cvt_pk_bf16_f32 v45, v40, v41\n\t"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23", \
  "v40","v41","v42","v43","v44","v45","v46","v47", \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23", \
  "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47"

template <int V> __global__ __launch_bounds__(256, 1) void k(unsigned long long *out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (V == 0) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], a[16:19], a[20:23], a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[32:47], a[16:19], a[20:23], a[32:47]" ::: CLOB);
    if constexpr (V == 1) asm volatile(VALU6 VALU6 ::: CLOB);
    if constexpr (V == 2) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], a[16:19], a[20:23], a[0:15]\n\t" VALU6 "v_mfma_f32_32x32x16_bf16 a[32:47], a[16:19], a[20:23], a[32:47]\n\t" VALU6 ::: CLOB);
    if constexpr (V == 3) asm volatile("v_mfma_f32_32x32x16_bf16 v[0:15], a[16:19], a[20:23], v[0:15]\n\t" VALU6 "v_mfma_f32_32x32x16_bf16 v[0:15], a[16:19], a[20:23], v[0:15]\n\t" VALU6 ::: CLOB);
    if constexpr (V == 4) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], v[16:19], a[20:23], a[0:15]\n\t" VALU6 "v_mfma_f32_32x32x16_bf16 a[32:47], v[16:19], a[20:23], a[32:47]\n\t" VALU6 ::: CLOB);
    if constexpr (V == 5) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], v[16:19], v[20:23], a[0:15]\n\t" VALU6 "v_mfma_f32_32x32x16_bf16 a[32:47], v[16:19], v[20:23], a[32:47]\n\t" VALU6 ::: CLOB);
    if constexpr (V == 6) asm volatile("v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]\n\t" VALU6 "v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]\n\t" VALU6 ::: CLOB);
    // dependent chain on ONE accumulator, all AGPR (the QK^T chain shape)
    if constexpr (V == 7) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], a[16:19], a[20:23], a[0:15]\n\t" VALU6 "v_mfma_f32_32x32x16_bf16 a[0:15], a[16:19], a[20:23], a[0:15]\n\t" VALU6 ::: CLOB);
    // VGPR D with a DIFFERENT C (the bias form), B in AGPR
    if constexpr (V == 8) asm volatile("v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], a[20:23], v[24:39]\n\t" VALU6 "v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], a[20:23], v[0:15]\n\t" VALU6 ::: CLOB, "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39");
    // 3 VALU only per gap (fits the 24-cycle budget)
    if constexpr (V == 9) asm volatile("v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]\n\tv_fma_f32 v40, v40, v46, v47\n\tv_exp_f32 v42, v42\n\tv_fma_f32 v41, v41, v46, v47\n\t"
                                       "v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]\n\tv_fma_f32 v40, v40, v46, v47\n\tv_exp_f32 v42, v42\n\tv_fma_f32 v41, v41, v46, v47\n\t" ::: CLOB);
    if constexpr (V == 10) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], a[16:19], a[20:23], a[0:15]\n\tv_fma_f32 v40, v40, v46, v47\n\tv_exp_f32 v42, v42\n\tv_fma_f32 v41, v41, v46, v47\n\t"
                                        "v_mfma_f32_32x32x16_bf16 a[32:47], a[16:19], a[20:23], a[32:47]\n\tv_fma_f32 v40, v40, v46, v47\n\tv_exp_f32 v42, v42\n\tv_fma_f32 v41, v41, v46, v47\n\t" ::: CLOB);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V> void run(const char *name, unsigned long long *d, int iters, int grid) {
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-62s %7.1f cycles per (MFMA + VALU piece)\n", name, s / h.size() / iters / 2);
}

int main() {
  unsigned long long *d;
  const int grid = 256, iters = 2000;
  hipMalloc(&d, grid * 4 * 8);
  run<0>("MFMA only (all accumulator registers)", d, iters, grid);
  run<1>("VALU only (2 fma, 2 exp, fma, cvt_pk)", d, iters, grid);
  run<2>("MFMA all-AGPR + VALU", d, iters, grid);
  run<7>("MFMA all-AGPR, ONE dependent chain + VALU", d, iters, grid);
  run<3>("MFMA D/C = VGPR, A/B AGPR + VALU", d, iters, grid);
  run<4>("MFMA A = VGPR, B/C/D AGPR + VALU", d, iters, grid);
  run<5>("MFMA A/B = VGPR, C/D AGPR + VALU", d, iters, grid);
  run<6>("MFMA all VGPR + VALU", d, iters, grid);
  run<8>("MFMA D = VGPR, C = other VGPRs / then chain, B AGPR + VALU", d, iters, grid);
  run<9>("MFMA all VGPR + 3 VALU (fma, exp, fma)", d, iters, grid);
  run<10>("MFMA all AGPR + 3 VALU (fma, exp, fma)", d, iters, grid);
  return 0;
}
