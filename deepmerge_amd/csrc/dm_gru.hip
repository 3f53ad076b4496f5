// GRU cell (PyTorch gate order r, z, n; reference Nets.py:60-66 uses nn.GRU as a 4-layer bidirectional MNIST sandbox net):
//   r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) * n + z * h
// gi = x W_ih^T + b_ih for every time step comes from one GEMM per layer and direction, gh = h W_hh^T + b_hh from one small GEMM per
// step; these two elementwise kernels are everything between the GEMMs.  HBM-bound, trivially small (B x 80 per step).
#include "dm_common.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ void gru_cell_fwd_kernel(const float *__restrict__ gi, long long gi_stride, const float *__restrict__ gh,
                                    const float *__restrict__ h, float *__restrict__ h_new, float *__restrict__ saved, int B, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx - b * H;
  const float *gib = gi + (long long)b * gi_stride, *ghb = gh + (long long)b * 3 * H;
  const float r = sigm(gib[j] + ghb[j]);
  const float z = sigm(gib[H + j] + ghb[H + j]);
  const float ghn = ghb[2 * H + j];
  const float n = tanhf(gib[2 * H + j] + r * ghn);
  const float hp = h[idx];
  h_new[idx] = (1.f - z) * n + z * hp;
  float *s = saved + (long long)b * 4 * H;
  s[j] = r; s[H + j] = z; s[2 * H + j] = n; s[3 * H + j] = ghn;
}

__global__ void gru_cell_bwd_kernel(const float *__restrict__ dh_new, const float *__restrict__ saved, const float *__restrict__ h,
                                    float *__restrict__ dgi, float *__restrict__ dgh, float *__restrict__ dh, int B, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx - b * H;
  const float *s = saved + (long long)b * 4 * H;
  const float r = s[j], z = s[H + j], n = s[2 * H + j], ghn = s[3 * H + j];
  const float g = dh_new[idx], hp = h[idx];
  const float dn_pre = g * (1.f - z) * (1.f - n * n);
  const float dz_pre = g * (hp - n) * z * (1.f - z);
  const float dr_pre = dn_pre * ghn * r * (1.f - r);
  float *a = dgi + (long long)b * 3 * H, *c = dgh + (long long)b * 3 * H;
  a[j] = dr_pre; a[H + j] = dz_pre; a[2 * H + j] = dn_pre;
  c[j] = dr_pre; c[H + j] = dz_pre; c[2 * H + j] = dn_pre * r;
  dh[idx] = g * z;
}

}  // namespace

extern "C" int dm_gru_cell_fwd(const float *gi, int64_t gi_stride, const float *gh, const float *h, float *h_new, float *saved,
                               int32_t B, int32_t H, void *stream) {
  DM_REQUIRE(B > 0 && H > 0 && gi_stride >= 3LL * H, DM_ERR_BAD_SHAPE, "dm_gru_cell_fwd: B=%d H=%d gi_stride=%lld", B, H, (long long)gi_stride);
  DM_REQUIRE(gi && gh && h && h_new && saved, DM_ERR_BAD_SHAPE, "dm_gru_cell_fwd: null pointer");
  hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), gi, (long long)gi_stride,
                     gh, h, h_new, saved, B, H);
  DM_LAUNCH_CHECK("dm_gru_cell_fwd");
  return DM_OK;
}

extern "C" int dm_gru_cell_bwd(const float *dh_new, const float *saved, const float *h, float *dgi, float *dgh, float *dh, int32_t B,
                               int32_t H, void *stream) {
  DM_REQUIRE(B > 0 && H > 0, DM_ERR_BAD_SHAPE, "dm_gru_cell_bwd: B=%d H=%d", B, H);
  DM_REQUIRE(dh_new && saved && h && dgi && dgh && dh, DM_ERR_BAD_SHAPE, "dm_gru_cell_bwd: null pointer");
  hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dh_new, saved, h, dgi,
                     dgh, dh, B, H);
  DM_LAUNCH_CHECK("dm_gru_cell_bwd");
  return DM_OK;
}
