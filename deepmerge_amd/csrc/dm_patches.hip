// Multi-scale patch pyramid gather (gfx950): for every sample point, crop an L x L window of a uint8
// multi-band tile (zero-padded outside the raster), resize it to t x t with an EXACT integer area average and
// emit float32 in [0,1] -- the GPU counterpart of the per-item GDAL read + per-band cv2.resize that the
// reference's data loaders run on one host thread (MyUtils1.py:116-223).  The resize rule is the build's own
// spec (OpenCV parity is unpinned, see oracle/patches.py); it is integer arithmetic, so results are bit-exact
// against the oracle.  HBM/L2-bound byte gather: the window is staged in LDS once, then each thread produces
// output pixels from LDS.
#include "dm_common.h"

namespace {

constexpr int MAX_WINDOW = 384;   // L*L bytes of LDS (147 KB)

// TLOG2 >= 0: the target side is 2^TLOG2 (32 / 64 / 128 in the reference's config.py:32): every division by T is a shift.
// TLOG2 < 0: any T.  The rounded quotient num / L^2 is taken with one float reciprocal and an exact integer fix-up
// (num < 256 L^2, so the estimate is off by at most one).
template <int TLOG2>
__global__ __launch_bounds__(256) void patch_pyramid_kernel(const unsigned char *__restrict__ tile, int bands, int H, int W,
                                                            const int *__restrict__ xy, const int *__restrict__ wins,
                                                            int Trt, float *__restrict__ out) {
  extern __shared__ unsigned char win[];
  const int T = TLOG2 >= 0 ? (1 << TLOG2) : Trt;
  auto divT = [&](int v) { return TLOG2 >= 0 ? (v >> TLOG2) : v / T; };
  const int p = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
  const int L = wins[p];
  const int mx = xy[2 * p], my = xy[2 * p + 1];
  const int x0 = (2 * mx - L) / 2, y0 = (2 * my - L) / 2;     // int(mid - L/2): truncation toward zero
  const unsigned char *band = tile + (long long)c * H * W;
  // window -> LDS, row by row: a wave walks along a row (coalesced bytes), no per-element division
  for (int j = t >> 6; j < L; j += 4) {
    const int gy = y0 + j;
    const bool rowin = gy >= 0 && gy < H;
    const unsigned char *src = band + (long long)gy * W;
    for (int i = t & 63; i < L; i += 64) {
      const int gx = x0 + i;
      win[j * L + i] = (rowin && gx >= 0 && gx < W) ? src[gx] : (unsigned char)0;
    }
  }
  __syncthreads();
  const int den = L * L;
  const float rden = 1.0f / (float)den;
  float *dst = out + ((long long)p * bands + c) * T * T;
  for (int o = t; o < T * T; o += 256) {
    const int oy = divT(o), ox = o - oy * T;
    const int ylo = oy * L, yhi = ylo + L, xlo = ox * L, xhi = xlo + L;       // footprints in 1/T input-pixel units
    const int iy0 = divT(ylo), iy1 = divT(yhi + T - 1), ix0 = divT(xlo), ix1 = divT(xhi + T - 1);
    int num = 0;
    for (int iy = iy0; iy < iy1; ++iy) {
      const int ovy = min(yhi, (iy + 1) * T) - max(ylo, iy * T);
      int row = 0;
      for (int ix = ix0; ix < ix1; ++ix) {
        const int ovx = min(xhi, (ix + 1) * T) - max(xlo, ix * T);
        row += ovx * (int)win[iy * L + ix];
      }
      num += ovy * row;
    }
    int q = (int)((float)num * rden);                          // floor(num / den) up to +-1 ...
    int r = num - q * den;
    if (r < 0) { --q; r += den; } else if (r >= den) { ++q; r -= den; }       // ... made exact
    if (2 * r > den || (2 * r == den && (q & 1))) ++q;        // round half to even
    dst[o] = (float)q / 255.0f;
  }
}

}  // namespace

extern "C" int dm_patch_pyramid(const uint8_t *tile, int32_t bands, int32_t H, int32_t W, const int32_t *xy, const int32_t *windows,
                                int32_t max_window, int32_t P, int32_t target, float *out, void *stream) {
  DM_REQUIRE(tile && xy && windows && out && bands > 0 && H > 0 && W > 0 && P > 0 && target > 0, DM_ERR_BAD_SHAPE,
             "dm_patch_pyramid: bad arguments");
  DM_REQUIRE(max_window > 0 && max_window <= MAX_WINDOW, DM_ERR_UNSUPPORTED,
             "dm_patch_pyramid: window side %d outside 1..%d", max_window, MAX_WINDOW);
  DM_REQUIRE(bands <= 65535, DM_ERR_BAD_SHAPE, "dm_patch_pyramid: too many bands");
  const size_t lds = (size_t)max_window * max_window;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(P, bands);
  switch (target) {
    case 32: hipLaunchKernelGGL(patch_pyramid_kernel<5>, grid, dim3(256), lds, s, tile, bands, H, W, xy, windows, target, out); break;
    case 64: hipLaunchKernelGGL(patch_pyramid_kernel<6>, grid, dim3(256), lds, s, tile, bands, H, W, xy, windows, target, out); break;
    case 128: hipLaunchKernelGGL(patch_pyramid_kernel<7>, grid, dim3(256), lds, s, tile, bands, H, W, xy, windows, target, out); break;
    case 256: hipLaunchKernelGGL(patch_pyramid_kernel<8>, grid, dim3(256), lds, s, tile, bands, H, W, xy, windows, target, out); break;
    default: hipLaunchKernelGGL(patch_pyramid_kernel<-1>, grid, dim3(256), lds, s, tile, bands, H, W, xy, windows, target, out); break;
  }
  DM_LAUNCH_CHECK("dm_patch_pyramid");
  return DM_OK;
}
