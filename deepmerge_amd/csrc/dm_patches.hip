// Multi-scale patch pyramid gather (gfx950): for every sample point, crop an L x L window of a uint8
// multi-band tile (zero-padded outside the raster), resize it to t x t and emit float32 in [0,1] -- the GPU counterpart of
// the per-item GDAL read + per-band cv2.resize(..., INTER_AREA) that the reference's data loaders run on one host thread
// (MyUtils1.py:116-223).  Two resize rules (oracle/patches.py states both; the kernel is bit-exact against it for either):
//   DM_RESIZE_OPENCV (default)  cv::resize's published INTER_AREA algorithm for uint8, branch by branch (integer-ratio fast path,
//                               float area tables for other shrinks, 11-bit fixed-point bilinear with area coordinates when enlarging);
//                               float32 operations in OpenCV's order with explicit roundings (this file is built with -ffp-contract=off)
//   DM_RESIZE_EXACT_AREA        the exact rational area average (integer arithmetic), the rule of rounds 1-2
// HBM/L2-bound byte gather: the window is staged in LDS once, then each thread produces output pixels from LDS.
#include "dm_common.h"

namespace {

constexpr int MAX_WINDOW = 384;   // L*L bytes of LDS (147 KB)

// TLOG2 >= 0: the target side is 2^TLOG2 (32 / 64 / 128 in the reference's config.py:32): every division by T is a shift.
// TLOG2 < 0: any T.  The rounded quotient num / L^2 is taken with one float reciprocal and an exact integer fix-up
// (num < 256 L^2, so the estimate is off by at most one).
// OUT = float: planar patches [P, bands, T, T] (the tensor contract of the reference's loaders).
// OUT = bf16_t / float with COLS: the patch-embed GEMM's operand rows directly -- row (p * G + py) * G + px, column
// (c * ps + dy) * ps + dx with ps = T / G (the im2col order of Conv2d(k = ps, stride = ps), nets/ShfitScaleFormer.py:28-37) --
// so the fp32 patch tensor and the separate im2col pass (dm_patchify) never exist.
// ---- cv::resize(INTER_AREA), uint8, one output pixel (oy, ox) of a T x T image from the L x L window in LDS ---------------------------
// computeResizeAreaTab's entries for destination index d: at most ceil(scale) + 2 (source index, float weight) pairs.
struct AreaTab { int si0, n; float first, mid, last; bool has_first, has_last; int sx1, sx2; };
__device__ __forceinline__ AreaTab area_tab(int d, int L, double scale) {
  AreaTab t;
  const double fsx1 = (double)d * scale, fsx2 = fsx1 + scale;
  const double cell = fmin(scale, (double)L - fsx1);
  int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
  sx2 = min(sx2, L - 1);
  sx1 = min(sx1, sx2);
  t.sx1 = sx1; t.sx2 = sx2;
  t.has_first = (double)sx1 - fsx1 > 1e-3;
  t.first = (float)(((double)sx1 - fsx1) / cell);
  t.mid = (float)(1.0 / cell);
  t.has_last = fsx2 - (double)sx2 > 1e-3;
  t.last = (float)(fmin(fmin(fsx2 - (double)sx2, 1.0), cell) / cell);
  t.si0 = 0; t.n = 0;
  return t;
}
__device__ __forceinline__ int cv_area_pixel(const unsigned char *win, int L, int T, int oy, int ox) {
  const double inv_scale = (double)T / (double)L, scale = 1.0 / inv_scale;
  // cv::resize's is_area_fast: |scale - cvRound(scale)| < DBL_EPSILON on the DOUBLE quotient -- for some integer ratios
  // (k = 49, 93, 98, ...: 1 / (1 / k) != k in binary64) it fails and the float table path below runs (oracle/patches.py cv_is_area_fast)
  if (L % T == 0 && fabs(scale - rint(scale)) < 2.220446049250313e-16) {
    const int k = L / T;
    if (k == 1) return win[oy * L + ox];
    int sum = 0;
    for (int iy = 0; iy < k; ++iy)
      for (int ix = 0; ix < k; ++ix) sum += win[(oy * k + iy) * L + ox * k + ix];
    if (k == 2) return (sum + 2) >> 2;
    const float scale = 1.f / (float)(k * k);
    return min(255, max(0, __float2int_rn(__fmul_rn((float)sum, scale))));
  }
  if (L > T) {                                                  // area tables, float accumulation in table order
    const AreaTab tx = area_tab(ox, L, scale), ty = area_tab(oy, L, scale);
    auto fold_row = [&](int sy) {
      const unsigned char *S = win + sy * L;
      float buf = 0.f;
      if (tx.has_first) buf = __fadd_rn(buf, __fmul_rn((float)S[tx.sx1 - 1], tx.first));
      for (int sx = tx.sx1; sx < tx.sx2; ++sx) buf = __fadd_rn(buf, __fmul_rn((float)S[sx], tx.mid));
      if (tx.has_last) buf = __fadd_rn(buf, __fmul_rn((float)S[tx.sx2], tx.last));
      return buf;
    };
    float sum = 0.f;
    bool first = true;
    auto add_row = [&](int sy, float beta) {
      const float term = __fmul_rn(beta, fold_row(sy));
      sum = first ? term : __fadd_rn(sum, term);
      first = false;
    };
    if (ty.has_first) add_row(ty.sx1 - 1, ty.first);
    for (int sy = ty.sx1; sy < ty.sx2; ++sy) add_row(sy, ty.mid);
    if (ty.has_last) add_row(ty.sx2, ty.last);
    return min(255, max(0, __float2int_rn(sum)));
  }
  // enlarging: the bilinear code with area-style coordinates, INTER_RESIZE_COEF_BITS = 11
  auto coeff = [&](int d, bool zero_at_border, int &s0, int &c0, int &c1, bool &inside) {
    int sx = (int)floor((double)d * scale);
    float fx = (float)((double)(d + 1) - (double)(sx + 1) * inv_scale);
    fx = fx <= 0.f ? 0.f : __fsub_rn(fx, floorf(fx));
    inside = sx + 1 < L;
    if (!inside && zero_at_border && sx >= L - 1) { fx = 0.f; sx = L - 1; }
    s0 = sx;
    c0 = __float2int_rn(__fmul_rn(__fsub_rn(1.f, fx), 2048.f));
    c1 = __float2int_rn(__fmul_rn(fx, 2048.f));
  };
  int sx, a0, a1, sy, b0, b1;
  bool xin, yin;
  coeff(ox, true, sx, a0, a1, xin);
  coeff(oy, false, sy, b0, b1, yin);
  const int r0 = min(sy, L - 1), r1 = min(sy + 1, L - 1);
  auto hpass = [&](int row) {
    const unsigned char *S = win + row * L;
    return xin ? (int)S[sx] * a0 + (int)S[sx + 1] * a1 : (int)S[min(sx, L - 1)] * 2048;
  };
  const int h0 = hpass(r0), h1 = hpass(r1);
  const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
  return min(255, max(0, v));
}

// The sample table of a training batch (dm_pair_batch_gather; all pointers null for the single-tile entry points): the tile a sample
// is cut from, its inner / object window sides -- the four window sides of MyUtils1.py:130-156 are derived here, `scale_index` picks
// one -- and, in the blocks of band 0, the designed-feature row [15 region attributes | 4 scale factors] (MyUtils1.py:60-77).  A sample
// whose tile id or window is out of range gets zeros and raises `err[0]` (read back by the host every few steps, not per step).
struct BatchTable {
  const int *tile_id, *inner, *obj;
  const float *region;
  float *designed;
  int *err;
  int n_tiles, scale_index, max_window;
};

template <int TLOG2, typename OUT, bool COLS>
__global__ __launch_bounds__(256) void patch_pyramid_kernel(const unsigned char *__restrict__ tile, int bands, int H, int W,
                                                            const int *__restrict__ xy, const int *__restrict__ wins,
                                                            int Trt, int G, OUT *__restrict__ out, int rule, const BatchTable bt) {
  extern __shared__ unsigned char win[];
  const int T = TLOG2 >= 0 ? (1 << TLOG2) : Trt;
  auto divT = [&](int v) { return TLOG2 >= 0 ? (v >> TLOG2) : v / T; };
  const int p = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
  int L, tid = 0;
  if (bt.inner) {
    const int in = bt.inner[p], ob = bt.obj[p], iv = ob - in;
    L = bt.scale_index == 0 ? in : ob + (bt.scale_index - 1) * iv;
    tid = bt.tile_id ? bt.tile_id[p] : 0;
    if (c == 0 && bt.designed && t < 19) {
      const int k = t - 15;
      const float wk = (float)(k == 0 ? in : ob + (k - 1) * iv);
      const float sk = k == 0 ? 32.f : k == 1 ? 64.f : k == 2 ? 128.f : 1.f;          // config.py:32 (patches.CONFIG_SCALES)
      bt.designed[(long long)p * 19 + t] = t < 15 ? bt.region[(long long)p * 15 + t] : __fdiv_rn(wk, sk);
    }
    if (L <= 0 || L > bt.max_window || tid < 0 || tid >= bt.n_tiles) {                 // (uniform per block)
      if (t == 0 && bt.err) atomicOr(bt.err, 1);
      const int ps0 = COLS ? T / G : 1;
      const long long Kc0 = (long long)bands * ps0 * ps0;
      const int zA = (int)blockIdx.z * T / (int)gridDim.z, zB = ((int)blockIdx.z + 1) * T / (int)gridDim.z;
      for (int o = zA * T + t; o < zB * T; o += 256) {
        if constexpr (COLS) {
          const int oy = divT(o), ox = o - oy * T, py = oy / ps0, dy = oy - py * ps0, px = ox / ps0, dx = ox - px * ps0;
          out[(((long long)p * G + py) * G + px) * Kc0 + ((long long)c * ps0 + dy) * ps0 + dx] = (OUT)0.f;
        } else {
          out[((long long)p * bands + c) * T * T + o] = (OUT)0.f;
        }
      }
      return;
    }
  } else {
    L = wins[p];
  }
  const int mx = xy[2 * p], my = xy[2 * p + 1];
  const int x0 = (2 * mx - L) / 2, y0 = (2 * my - L) / 2;     // int(mid - L/2): truncation toward zero
  const unsigned char *band = tile + ((long long)tid * bands + c) * H * W;
  // gridDim.z > 1 (dm_pair_batch_gather): block z produces the output rows [oyA, oyB) of its (sample, band) and stages only the window
  // rows those pixels read (every resize rule reads source rows floor(oy L / T) - 1 .. ceil((oy + 1) L / T) + 1 at most) -- a 256 x 256
  // target is 256 pixels per thread, and a training batch has only 2B x bands (sample, band) pairs to fill 256 CUs with
  const int oyA = (int)blockIdx.z * T / (int)gridDim.z, oyB = ((int)blockIdx.z + 1) * T / (int)gridDim.z;
  const int jA = gridDim.z > 1 ? max(0, (int)((long long)oyA * L / T) - 1) : 0;
  const int jB = gridDim.z > 1 ? min(L, (int)(((long long)oyB * L + T - 1) / T) + 2) : L;
  // window -> LDS, row by row: a wave walks along a row (coalesced bytes), no per-element division
  for (int j = jA + (t >> 6); j < jB; j += 4) {
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
  OUT *dst = COLS ? out : out + ((long long)p * bands + c) * T * T;
  const int ps = COLS ? T / G : 1;
  const long long Kc = (long long)bands * ps * ps;
  for (int o = oyA * T + t; o < oyB * T; o += 256) {
    const int oy = divT(o), ox = o - oy * T;
    if (rule == DM_RESIZE_OPENCV) {
      const float v = (float)cv_area_pixel(win, L, T, oy, ox) / 255.0f;
      if constexpr (COLS) {
        const int py = oy / ps, dy = oy - py * ps, px = ox / ps, dx = ox - px * ps;
        dst[(((long long)p * G + py) * G + px) * Kc + ((long long)c * ps + dy) * ps + dx] = (OUT)v;
      } else {
        dst[o] = (OUT)v;
      }
      continue;
    }
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
    const float v = (float)q / 255.0f;
    if constexpr (COLS) {
      const int py = oy / ps, dy = oy - py * ps, px = ox / ps, dx = ox - px * ps;
      dst[(((long long)p * G + py) * G + px) * Kc + ((long long)c * ps + dy) * ps + dx] = (OUT)v;
    } else {
      dst[o] = (OUT)v;
    }
  }
}

}  // namespace

namespace {
template <typename OUT, bool COLS>
void launch_pyramid(dim3 grid, size_t lds, hipStream_t s, const uint8_t *tile, int bands, int H, int W, const int32_t *xy, const int32_t *windows,
                    int target, int G, OUT *out, int rule, const BatchTable bt = BatchTable{}) {
  switch (target) {
    case 32: hipLaunchKernelGGL((patch_pyramid_kernel<5, OUT, COLS>), grid, dim3(256), lds, s, tile, bands, H, W, xy, windows, target, G, out, rule, bt); break;
    case 64: hipLaunchKernelGGL((patch_pyramid_kernel<6, OUT, COLS>), grid, dim3(256), lds, s, tile, bands, H, W, xy, windows, target, G, out, rule, bt); break;
    case 128: hipLaunchKernelGGL((patch_pyramid_kernel<7, OUT, COLS>), grid, dim3(256), lds, s, tile, bands, H, W, xy, windows, target, G, out, rule, bt); break;
    case 256: hipLaunchKernelGGL((patch_pyramid_kernel<8, OUT, COLS>), grid, dim3(256), lds, s, tile, bands, H, W, xy, windows, target, G, out, rule, bt); break;
    default: hipLaunchKernelGGL((patch_pyramid_kernel<-1, OUT, COLS>), grid, dim3(256), lds, s, tile, bands, H, W, xy, windows, target, G, out, rule, bt); break;
  }
}
}  // namespace

// One scale of a TRAINING batch straight from resident tiles and a device sample table: no host-side window arithmetic, no per-tile
// launches, nothing to synchronise on (Train_SMT.py:212-262 draws a batch from the loaders of MyUtils1.py:41-77 on the host).
extern "C" int dm_pair_batch_gather(const uint8_t *tiles, int32_t n_tiles, int32_t bands, int32_t H, int32_t W, const int32_t *tile_id,
                                    const int32_t *xy, const int32_t *inner, const int32_t *obj, int32_t scale_index, int32_t max_window,
                                    int32_t P, int32_t target, int32_t grid, int32_t resize_rule, void *out, int32_t dtype,
                                    const float *region_features, float *designed, int32_t *error_flag, void *stream) {
  DM_REQUIRE(resize_rule == DM_RESIZE_OPENCV || resize_rule == DM_RESIZE_EXACT_AREA, DM_ERR_UNSUPPORTED, "dm_pair_batch_gather: unknown resize rule %d", resize_rule);
  DM_REQUIRE(tiles && xy && inner && obj && out && n_tiles > 0 && bands > 0 && H > 0 && W > 0 && P > 0 && target > 0, DM_ERR_BAD_SHAPE,
             "dm_pair_batch_gather: bad arguments");
  DM_REQUIRE(scale_index >= 0 && scale_index < 4, DM_ERR_BAD_SHAPE, "dm_pair_batch_gather: scale index %d outside 0..3", scale_index);
  DM_REQUIRE(grid >= 0 && (grid == 0 || target % grid == 0), DM_ERR_BAD_SHAPE, "dm_pair_batch_gather: target %d is not a multiple of the token grid %d", target, grid);
  DM_REQUIRE(max_window > 0 && max_window <= MAX_WINDOW, DM_ERR_UNSUPPORTED, "dm_pair_batch_gather: window bound %d outside 1..%d", max_window, MAX_WINDOW);
  DM_REQUIRE(bands <= 65535, DM_ERR_BAD_SHAPE, "dm_pair_batch_gather: too many bands");
  DM_REQUIRE((designed == nullptr) == (region_features == nullptr), DM_ERR_BAD_SHAPE, "dm_pair_batch_gather: designed rows need the region features (and vice versa)");
  BatchTable bt;
  bt.tile_id = tile_id; bt.inner = inner; bt.obj = obj; bt.region = region_features; bt.designed = designed; bt.err = error_flag;
  bt.n_tiles = n_tiles; bt.scale_index = scale_index; bt.max_window = max_window;
  // <= 32 output pixels per thread while the grid stays small (a batch of 64 samples x 4 bands at target 256: 8 blocks per pair)
  int zn = target * target / (256 * 32);
  while (zn > 1 && (long long)P * bands * zn > 8192) zn >>= 1;
  const dim3 g(P, bands, zn < 1 ? 1 : zn);
  const size_t lds = (size_t)max_window * max_window;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (grid == 0) {
    DM_REQUIRE(dtype == DM_F32, DM_ERR_BAD_DTYPE, "dm_pair_batch_gather: planar patches are fp32");
    launch_pyramid<float, false>(g, lds, s, tiles, bands, H, W, xy, nullptr, target, 1, reinterpret_cast<float *>(out), resize_rule, bt);
  } else if (dtype == DM_BF16) {
    launch_pyramid<bf16_t, true>(g, lds, s, tiles, bands, H, W, xy, nullptr, target, grid, reinterpret_cast<bf16_t *>(out), resize_rule, bt);
  } else if (dtype == DM_F32) {
    launch_pyramid<float, true>(g, lds, s, tiles, bands, H, W, xy, nullptr, target, grid, reinterpret_cast<float *>(out), resize_rule, bt);
  } else {
    DM_REQUIRE(false, DM_ERR_BAD_DTYPE, "dm_pair_batch_gather: bad dtype %d", dtype);
  }
  DM_LAUNCH_CHECK("dm_pair_batch_gather");
  return DM_OK;
}

extern "C" int dm_patch_pyramid(const uint8_t *tile, int32_t bands, int32_t H, int32_t W, const int32_t *xy, const int32_t *windows,
                                int32_t max_window, int32_t P, int32_t target, int32_t resize_rule, float *out, void *stream) {
  DM_REQUIRE(resize_rule == DM_RESIZE_OPENCV || resize_rule == DM_RESIZE_EXACT_AREA, DM_ERR_UNSUPPORTED, "dm_patch_pyramid: unknown resize rule %d", resize_rule);
  DM_REQUIRE(tile && xy && windows && out && bands > 0 && H > 0 && W > 0 && P > 0 && target > 0, DM_ERR_BAD_SHAPE,
             "dm_patch_pyramid: bad arguments");
  DM_REQUIRE(max_window > 0 && max_window <= MAX_WINDOW, DM_ERR_UNSUPPORTED,
             "dm_patch_pyramid: window side %d outside 1..%d", max_window, MAX_WINDOW);
  DM_REQUIRE(bands <= 65535, DM_ERR_BAD_SHAPE, "dm_patch_pyramid: too many bands");
  launch_pyramid<float, false>(dim3(P, bands), (size_t)max_window * max_window, reinterpret_cast<hipStream_t>(stream), tile, bands, H, W, xy, windows,
                               target, 1, out, resize_rule);
  DM_LAUNCH_CHECK("dm_patch_pyramid");
  return DM_OK;
}

extern "C" int dm_patch_pyramid_cols(const uint8_t *tile, int32_t bands, int32_t H, int32_t W, const int32_t *xy, const int32_t *windows,
                                     int32_t max_window, int32_t P, int32_t target, int32_t grid, int32_t resize_rule, void *cols, int32_t dtype,
                                     void *stream) {
  DM_REQUIRE(resize_rule == DM_RESIZE_OPENCV || resize_rule == DM_RESIZE_EXACT_AREA, DM_ERR_UNSUPPORTED, "dm_patch_pyramid_cols: unknown resize rule %d", resize_rule);
  DM_REQUIRE(tile && xy && windows && cols && bands > 0 && H > 0 && W > 0 && P > 0 && target > 0, DM_ERR_BAD_SHAPE,
             "dm_patch_pyramid_cols: bad arguments");
  DM_REQUIRE(grid > 0 && target % grid == 0, DM_ERR_BAD_SHAPE, "dm_patch_pyramid_cols: target %d is not a multiple of the token grid %d", target, grid);
  DM_REQUIRE(max_window > 0 && max_window <= MAX_WINDOW, DM_ERR_UNSUPPORTED,
             "dm_patch_pyramid_cols: window side %d outside 1..%d", max_window, MAX_WINDOW);
  DM_REQUIRE(bands <= 65535, DM_ERR_BAD_SHAPE, "dm_patch_pyramid_cols: too many bands");
  const dim3 g(P, bands);
  const size_t lds = (size_t)max_window * max_window;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DM_BF16) launch_pyramid<bf16_t, true>(g, lds, s, tile, bands, H, W, xy, windows, target, grid, reinterpret_cast<bf16_t *>(cols), resize_rule);
  else if (dtype == DM_F32) launch_pyramid<float, true>(g, lds, s, tile, bands, H, W, xy, windows, target, grid, reinterpret_cast<float *>(cols), resize_rule);
  else DM_REQUIRE(false, DM_ERR_BAD_DTYPE, "dm_patch_pyramid_cols: bad dtype %d", dtype);
  DM_LAUNCH_CHECK("dm_patch_pyramid_cols");
  return DM_OK;
}
