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

__global__ __launch_bounds__(256) void patch_pyramid_kernel(const unsigned char *__restrict__ tile, int bands, int H, int W,
                                                            const int *__restrict__ xy, const int *__restrict__ wins,
                                                            int T, float *__restrict__ out) {
  extern __shared__ unsigned char win[];
  const int p = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
  const int L = wins[p];
  const int mx = xy[2 * p], my = xy[2 * p + 1];
  const int x0 = (2 * mx - L) / 2, y0 = (2 * my - L) / 2;     // int(mid - L/2): truncation toward zero
  const unsigned char *band = tile + (long long)c * H * W;
  for (int idx = t; idx < L * L; idx += 256) {
    const int j = idx / L, i = idx - j * L;
    const int gy = y0 + j, gx = x0 + i;
    unsigned char v = 0;
    if (gx >= 0 && gx < W && gy >= 0 && gy < H) v = band[(long long)gy * W + gx];
    win[idx] = v;
  }
  __syncthreads();
  const int den = L * L;
  float *dst = out + ((long long)p * bands + c) * T * T;
  for (int o = t; o < T * T; o += 256) {
    const int oy = o / T, ox = o - oy * T;
    const int ylo = oy * L, yhi = ylo + L, xlo = ox * L, xhi = xlo + L;       // footprints in 1/T input-pixel units
    const int iy0 = ylo / T, iy1 = (yhi + T - 1) / T, ix0 = xlo / T, ix1 = (xhi + T - 1) / T;
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
    int q = num / den;
    const int r = num - q * den;
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
  hipLaunchKernelGGL(patch_pyramid_kernel, dim3(P, bands), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), tile, bands, H, W,
                     xy, windows, target, out);
  DM_LAUNCH_CHECK("dm_patch_pyramid");
  return DM_OK;
}
