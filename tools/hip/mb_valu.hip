// Issue cost and dependent latency of the softmax's VALU instructions for ONE wave per SIMD (gfx950), alone and beside an MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
  "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23"
#define MFMA "v_mfma_f32_32x32x16_bf16 a[0:15], a[16:19], a[20:23], a[0:15]\n\t"
#define F(d, s) "v_fma_f32 v" #d ", v" #s ", v62, v63\n\t"
#define E(d, s) "v_exp_f32 v" #d ", v" #s "\n\t"
#define C(d, s, t) "v_cvt_pk_bf16_f32 v" #d ", v" #s ", v" #t "\n\t"
#define M(d, s, t) "v_max3_f32 v" #d ", v" #d ", v" #s ", v" #t "\n\t"
template <int V> __global__ __launch_bounds__(256, 1) void k(unsigned long long *out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (V == 0) asm volatile(F(40,40) F(41,41) F(42,42) F(43,43) F(44,44) F(45,45) F(46,46) F(47,47) ::: CLOB);          // 8 independent fma chains
    if constexpr (V == 1) asm volatile(F(40,40) F(40,40) F(40,40) F(40,40) F(40,40) F(40,40) F(40,40) F(40,40) ::: CLOB);          // one dependent fma chain
    if constexpr (V == 2) asm volatile(E(40,40) E(41,41) E(42,42) E(43,43) E(44,44) E(45,45) E(46,46) E(47,47) ::: CLOB);          // 8 independent exp
    if constexpr (V == 3) asm volatile(E(40,40) E(40,40) E(40,40) E(40,40) E(40,40) E(40,40) E(40,40) E(40,40) ::: CLOB);          // dependent exp chain
    if constexpr (V == 4) asm volatile(C(40,48,49) C(41,48,49) C(42,48,49) C(43,48,49) C(44,48,49) C(45,48,49) C(46,48,49) C(47,48,49) ::: CLOB);
    if constexpr (V == 5) asm volatile(M(40,48,49) M(41,48,49) M(42,48,49) M(43,48,49) M(44,48,49) M(45,48,49) M(46,48,49) M(47,48,49) ::: CLOB);
    // the pipelined gap: fma pair k, exp pair k-1, cvt pair k-2, one max3; inputs produced one full gap earlier (4 gaps unrolled)
    if constexpr (V == 6) asm volatile(
        F(40,0) F(41,1) E(48,44) E(49,45) C(56,52,53) M(60,2,3)
        F(42,4) F(43,5) E(50,40) E(51,41) C(57,48,49) M(61,6,7)
        F(44,8) F(45,9) E(52,42) E(53,43) C(58,50,51) M(60,10,11)
        F(46,12) F(47,13) E(54,44) E(55,45) C(59,52,53) M(61,14,15) ::: CLOB);
    if constexpr (V == 7) asm volatile(
        MFMA F(40,0) F(41,1) E(48,44) E(49,45) C(56,52,53) M(60,2,3)
        MFMA F(42,4) F(43,5) E(50,40) E(51,41) C(57,48,49) M(61,6,7)
        MFMA F(44,8) F(45,9) E(52,42) E(53,43) C(58,50,51) M(60,10,11)
        MFMA F(46,12) F(47,13) E(54,44) E(55,45) C(59,52,53) M(61,14,15) ::: CLOB);
    if constexpr (V == 8) asm volatile(MFMA MFMA MFMA MFMA ::: CLOB);
    // exp consumes the fma result of the SAME gap (the unpipelined form)
    if constexpr (V == 9) asm volatile(
        MFMA F(40,0) F(41,1) E(48,40) E(49,41) C(56,48,49) M(60,2,3)
        MFMA F(42,4) F(43,5) E(50,42) E(51,43) C(57,50,51) M(61,6,7)
        MFMA F(44,8) F(45,9) E(52,44) E(53,45) C(58,52,53) M(60,10,11)
        MFMA F(46,12) F(47,13) E(54,46) E(55,47) C(59,54,55) M(61,14,15) ::: CLOB);
    // with s_nop 1 in front of the MFMA and one ds_read_b128 per gap
    if constexpr (V == 10) asm volatile(
        "s_nop 1\n\t" MFMA F(40,0) F(41,1) E(48,44) E(49,45) C(56,52,53) M(60,2,3) "ds_read_b128 v[16:19], v31\n\t"
        "s_nop 1\n\t" MFMA F(42,4) F(43,5) E(50,40) E(51,41) C(57,48,49) M(61,6,7) "ds_read_b128 v[20:23], v31\n\t"
        "s_nop 1\n\t" MFMA F(44,8) F(45,9) E(52,42) E(53,43) C(58,50,51) M(60,10,11) "ds_read_b128 v[24:27], v31\n\t"
        "s_nop 1\n\t" MFMA F(46,12) F(47,13) E(54,44) E(55,45) C(59,52,53) M(61,14,15) "ds_read_b128 v[16:19], v31\n\ts_waitcnt lgkmcnt(2)\n\t" ::: CLOB);
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int V> void run(const char *name, unsigned long long *d, int iters, int grid, double per) {
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 4096, 0, d, iters); hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-78s %7.2f cycles per %s\n", name, s / h.size() / iters / per, per == 8 ? "instruction" : "gap");
}
int main() {
  unsigned long long *d;
  const int grid = 256, iters = 2000;
  hipMalloc(&d, grid * 4 * 8);
  run<0>("v_fma_f32, 8 independent chains", d, iters, grid, 8);
  run<1>("v_fma_f32, one dependent chain", d, iters, grid, 8);
  run<2>("v_exp_f32, 8 independent", d, iters, grid, 8);
  run<3>("v_exp_f32, one dependent chain", d, iters, grid, 8);
  run<4>("v_cvt_pk_bf16_f32, independent", d, iters, grid, 8);
  run<5>("v_max3_f32, 8 independent chains", d, iters, grid, 8);
  run<8>("MFMA 32x32x16 only", d, iters, grid, 4);
  run<6>("gap VALU only: 2 fma, 2 exp, cvt, max3 (inputs one gap old)", d, iters, grid, 4);
  run<7>("MFMA + that gap", d, iters, grid, 4);
  run<9>("MFMA + gap whose exp / cvt consume results of the same gap", d, iters, grid, 4);
  run<10>("s_nop 1 + MFMA + gap + one ds_read_b128", d, iters, grid, 4);
  return 0;
}
