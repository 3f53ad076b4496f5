// Region-adjacency graph and per-superpixel statistics from a label raster (gfx950).  SURVEY 8f rank 2: the step either
// side of the ExtractFeatures sweep -- the reference reads the edge list (`LEFT_FID`, `RIGHT_FID` of lines.shp,
// MyUtils2.py:155-193) and the 15 designed attributes (MyUtils1.py:79-114) from files written by external GIS software;
// this build derives both on the device from the segmentation's label raster and the image tile.  The definitions are
// the build's own (oracle/rag.py states them; every quantity is an exact integer or a fixed double-precision formula
// of exact integers, so the GPU and the oracle agree bit for bit).
//
// HBM-bound integer work: one pass over the raster per kernel.  A thread walks a 16-pixel strip of one row (one 16-byte
// load per band, 64 bytes of labels), run-length merges what it finds and flushes once per run, so the number of atomics
// is ~10x below one per pixel; all atomics are on integers (exact, order-independent => deterministic).
#include <climits>

#include "dm_common.h"

namespace {

constexpr int STRIP = 16;
constexpr long long EMPTY_KEY = -1;
constexpr int TSLOTS_LOG2 = 6, TSLOTS = 1 << TSLOTS_LOG2;      // labels per 64x64 tile kept in LDS (more: global atomics)
constexpr int ESLOTS_LOG2 = 7, ESLOTS = 1 << ESLOTS_LOG2;      // label pairs per tile kept in LDS

__global__ void rag_init_kernel(long long *count, long long *sum, long long *sumsq, int *bbox, long long *peri, int S, int bands3) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  count[s] = 0;
  for (int b = 0; b < bands3; ++b) { sum[(long long)s * bands3 + b] = 0; sumsq[(long long)s * bands3 + b] = 0; }
  bbox[4 * s + 0] = INT_MAX; bbox[4 * s + 1] = INT_MAX; bbox[4 * s + 2] = -1; bbox[4 * s + 3] = -1;
  peri[2 * s + 0] = 0; peri[2 * s + 1] = 0;
}

__device__ __forceinline__ void atomic_add64(long long *p, long long v) {
  atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v);
}

// Per label: pixel count, per-band sum and sum of squares (first NB <= 3 bands), bounding box, perimeter in pixel edges
// (peri[2s] = edges shared with another label, peri[2s+1] = edges on the raster border).
template <int NB, bool VEC>
__global__ __launch_bounds__(256) void label_stats_kernel(const int *__restrict__ labels, const unsigned char *__restrict__ tile,
                                                          int H, int W, int S, long long *__restrict__ count,
                                                          long long *__restrict__ sum, long long *__restrict__ sumsq,
                                                          int *__restrict__ bbox, long long *__restrict__ peri) {
  // A workgroup owns a 64x64-pixel tile (thread = one 16-pixel strip of one row).  A tile meets only a handful of superpixels,
  // so the per-run statistics are first folded in an LDS-private table (integer LDS atomics) and every label of the tile then
  // costs ONE set of global atomics: ~25x fewer global atomics than flushing every run.
  __shared__ int t_key[TSLOTS];
  __shared__ unsigned t_cnt[TSLOTS], t_sum[TSLOTS][NB], t_sq[TSLOTS][NB], t_per[TSLOTS][2];
  __shared__ int t_box[TSLOTS][4];
  for (int i = threadIdx.x; i < TSLOTS; i += blockDim.x) {
    t_key[i] = -1; t_cnt[i] = 0; t_per[i][0] = 0; t_per[i][1] = 0;
    t_box[i][0] = INT_MAX; t_box[i][1] = INT_MAX; t_box[i][2] = -1; t_box[i][3] = -1;
#pragma unroll
    for (int b = 0; b < NB; ++b) { t_sum[i][b] = 0; t_sq[i][b] = 0; }
  }
  __syncthreads();
  const int tiles_x = (W + 63) / 64;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  {
    const int y = ty * 64 + (threadIdx.x >> 2), x0 = tx * 64 + (threadIdx.x & 3) * STRIP;
    const bool live = y < H && x0 < W;
    const int n = live ? min(STRIP, W - x0) : 0;
    const int *row = labels + (long long)(live ? y : 0) * W;
    // strip registers: this row's labels with one neighbour on each side, the rows above / below, the bands' bytes
    int lab[STRIP + 2], up[STRIP], dn[STRIP];
    unsigned char px[NB][STRIP];
    if (VEC && n == STRIP) {
#pragma unroll
      for (int v = 0; v < STRIP / 4; ++v) {
        const i32x4 a = *reinterpret_cast<const i32x4 *>(row + x0 + 4 * v);
        const i32x4 u = (y > 0) ? *reinterpret_cast<const i32x4 *>(row - W + x0 + 4 * v) : (i32x4){-2, -2, -2, -2};
        const i32x4 d = (y + 1 < H) ? *reinterpret_cast<const i32x4 *>(row + W + x0 + 4 * v) : (i32x4){-2, -2, -2, -2};
#pragma unroll
        for (int e = 0; e < 4; ++e) { lab[1 + 4 * v + e] = a[e]; up[4 * v + e] = u[e]; dn[4 * v + e] = d[e]; }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const u32x4 q = *reinterpret_cast<const u32x4 *>(tile + ((long long)b * H + y) * W + x0);
#pragma unroll
        for (int e = 0; e < STRIP; ++e) px[b][e] = (unsigned char)(q[e >> 2] >> (8 * (e & 3)));
      }
    } else {
#pragma unroll
      for (int i = 0; i < STRIP; ++i) {
        const bool in = i < n;
        lab[1 + i] = in ? row[x0 + i] : -2;
        up[i] = (in && y > 0) ? row[x0 + i - W] : -2;
        dn[i] = (in && y + 1 < H) ? row[x0 + i + W] : -2;
#pragma unroll
        for (int b = 0; b < NB; ++b) px[b][i] = in ? tile[((long long)b * H + y) * W + x0 + i] : 0;
      }
    }
    lab[0] = (live && x0 > 0) ? row[x0 - 1] : -2;             // -2 = outside the raster
    lab[STRIP + 1] = (live && x0 + STRIP < W) ? row[x0 + STRIP] : -2;
    if (n < STRIP) lab[1 + n] = (live && x0 + n < W) ? row[x0 + n] : -2;
    int cur = -1, run_x0 = 0;
    long long c = 0, sm[NB], sq[NB], pin = 0, pbd = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) { sm[b] = 0; sq[b] = 0; }
    auto flush = [&](int xend) {
      if (cur < 0 || cur >= S || c == 0) return;
      // slot of this label in the tile's table (open addressing); a full table falls back to global atomics
      unsigned slot = ((unsigned)cur * 2654435761u) >> (32 - TSLOTS_LOG2);
      bool found = false;
      for (int probe = 0; probe < TSLOTS; ++probe) {
        const int seen = atomicCAS(&t_key[slot], -1, cur);
        if (seen == -1 || seen == cur) { found = true; break; }
        slot = (slot + 1) & (TSLOTS - 1);
      }
      if (found) {
        atomicAdd(&t_cnt[slot], (unsigned)c);
#pragma unroll
        for (int b = 0; b < NB; ++b) { atomicAdd(&t_sum[slot][b], (unsigned)sm[b]); atomicAdd(&t_sq[slot][b], (unsigned)sq[b]); }
        atomicMin(&t_box[slot][0], run_x0); atomicMin(&t_box[slot][1], y);
        atomicMax(&t_box[slot][2], xend); atomicMax(&t_box[slot][3], y);
        if (pin) atomicAdd(&t_per[slot][0], (unsigned)pin);
        if (pbd) atomicAdd(&t_per[slot][1], (unsigned)pbd);
        return;
      }
      atomic_add64(count + cur, c);
#pragma unroll
      for (int b = 0; b < NB; ++b) { atomic_add64(sum + (long long)cur * NB + b, sm[b]); atomic_add64(sumsq + (long long)cur * NB + b, sq[b]); }
      atomicMin(bbox + 4 * cur + 0, run_x0); atomicMin(bbox + 4 * cur + 1, y);
      atomicMax(bbox + 4 * cur + 2, xend); atomicMax(bbox + 4 * cur + 3, y);
      if (pin) atomic_add64(peri + 2 * cur, pin);
      if (pbd) atomic_add64(peri + 2 * cur + 1, pbd);
    };
#pragma unroll
    for (int i = 0; i < STRIP; ++i) {
      if (i >= n) break;
      const int x = x0 + i;
      const int l = lab[1 + i];
      if (l != cur) {
        flush(x - 1);
        cur = l; run_x0 = x; c = 0; pin = 0; pbd = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) { sm[b] = 0; sq[b] = 0; }
      }
      ++c;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const long long v = px[b][i];
        sm[b] += v; sq[b] += v * v;
      }
      // the four pixel edges: raster border (-2), or a different label on the other side
      const int nbr[4] = {lab[i], lab[2 + i], up[i], dn[i]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (nbr[e] == -2) ++pbd; else if (nbr[e] != l) ++pin;
      }
    }
    if (live) flush(x0 + n - 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TSLOTS; i += blockDim.x) {      // one set of global atomics per label of the tile
    const int l = t_key[i];
    if (l < 0) continue;
    atomic_add64(count + l, (long long)t_cnt[i]);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      atomic_add64(sum + (long long)l * NB + b, (long long)t_sum[i][b]);
      atomic_add64(sumsq + (long long)l * NB + b, (long long)t_sq[i][b]);
    }
    atomicMin(bbox + 4 * l + 0, t_box[i][0]); atomicMin(bbox + 4 * l + 1, t_box[i][1]);
    atomicMax(bbox + 4 * l + 2, t_box[i][2]); atomicMax(bbox + 4 * l + 3, t_box[i][3]);
    if (t_per[i][0]) atomic_add64(peri + 2 * l, (long long)t_per[i][0]);
    if (t_per[i][1]) atomic_add64(peri + 2 * l + 1, (long long)t_per[i][1]);
  }
}

// The 15 designed attributes (order of MyUtils1.py:79-114) from the exact integer statistics, in double precision:
//   area, peri, len, width, smooth, std0, std1, std2, mean0, mean1, mean2, shapeness, compact, bright, border
__global__ void label_features_kernel(const long long *__restrict__ count, const long long *__restrict__ sum,
                                      const long long *__restrict__ sumsq, const int *__restrict__ bbox,
                                      const long long *__restrict__ peri, int S, int nb, float *__restrict__ feat) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  float *f = feat + (long long)s * 15;
  const double area = (double)count[s];
  if (count[s] == 0) {
    for (int i = 0; i < 15; ++i) f[i] = 0.f;
    return;
  }
  const double pin = (double)peri[2 * s], pbd = (double)peri[2 * s + 1], per = pin + pbd;
  const double bw = (double)(bbox[4 * s + 2] - bbox[4 * s + 0] + 1), bh = (double)(bbox[4 * s + 3] - bbox[4 * s + 1] + 1);
  double mean[3] = {0.0, 0.0, 0.0}, sd[3] = {0.0, 0.0, 0.0};
  for (int b = 0; b < nb; ++b) {
    const double m = (double)sum[(long long)s * nb + b] / area;
    const double var = (double)sumsq[(long long)s * nb + b] / area - m * m;
    mean[b] = m;
    sd[b] = sqrt(var > 0.0 ? var : 0.0);
  }
  f[0] = (float)area;
  f[1] = (float)per;
  f[2] = (float)(bw > bh ? bw : bh);
  f[3] = (float)(bw > bh ? bh : bw);
  f[4] = (float)(per / (2.0 * (bw + bh)));
  f[5] = (float)sd[0]; f[6] = (float)sd[1]; f[7] = (float)sd[2];
  f[8] = (float)mean[0]; f[9] = (float)mean[1]; f[10] = (float)mean[2];
  f[11] = (float)(per / (4.0 * sqrt(area)));
  f[12] = (float)(area / (bw * bh));
  f[13] = (float)((mean[0] + mean[1] + mean[2]) / (double)(nb > 0 ? nb : 1));
  f[14] = (float)pin;
}

// ---- adjacency: open-addressing table keyed by a * S + b (a < b), value = number of shared pixel edges --------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}
__device__ __forceinline__ void table_add(long long *keys, int *cnt, unsigned mask, long long key, int c, int *overflow) {
  unsigned slot = (unsigned)mix64((unsigned long long)key) & mask;
  for (unsigned probe = 0; probe <= mask; ++probe) {
    const long long seen = (long long)atomicCAS(reinterpret_cast<unsigned long long *>(keys + slot), (unsigned long long)EMPTY_KEY,
                                                (unsigned long long)key);
    if (seen == EMPTY_KEY || seen == key) {
      atomicAdd(cnt + slot, c);
      return;
    }
    slot = (slot + 1) & mask;
    if (probe > 4096) break;
  }
  atomicExch(overflow, 1);
}

__global__ void table_clear_kernel(long long *keys, int *cnt, long long n, int *overflow, int *n_out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    keys[i] = EMPTY_KEY;
    cnt[i] = 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { *overflow = 0; *n_out = 0; }
}

template <bool VEC>
__global__ __launch_bounds__(256) void rag_edges_kernel(const int *__restrict__ labels, int H, int W, int S, long long *__restrict__ keys,
                                                        int *__restrict__ cnt, unsigned mask, int *__restrict__ overflow) {
  // same 64x64 tiling as label_stats_kernel: pairs are first counted in an LDS-private table, then every distinct pair of the
  // tile is added to the global table once
  __shared__ long long e_key[ESLOTS];
  __shared__ int e_cnt[ESLOTS];
  for (int i = threadIdx.x; i < ESLOTS; i += blockDim.x) { e_key[i] = EMPTY_KEY; e_cnt[i] = 0; }
  __syncthreads();
  const int tiles_x = (W + 63) / 64;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  auto tile_add = [&](long long key, int c) {
    unsigned slot = (unsigned)mix64((unsigned long long)key) & (ESLOTS - 1);
    for (int probe = 0; probe < ESLOTS; ++probe) {
      const long long seen = (long long)atomicCAS(reinterpret_cast<unsigned long long *>(&e_key[slot]), (unsigned long long)EMPTY_KEY,
                                                  (unsigned long long)key);
      if (seen == EMPTY_KEY || seen == key) { atomicAdd(&e_cnt[slot], c); return; }
      slot = (slot + 1) & (ESLOTS - 1);
    }
    table_add(keys, cnt, mask, key, c, overflow);            // tile table full: straight to the global table
  };
  {
    const int y = ty * 64 + (threadIdx.x >> 2), x0 = tx * 64 + (threadIdx.x & 3) * STRIP;
    const bool live = y < H && x0 < W;
    const int n = live ? min(STRIP, W - x0) : 0;
    const int *row = labels + (long long)(live ? y : 0) * W;
    int lab[STRIP + 1], dn[STRIP];
    if (VEC && n == STRIP) {
#pragma unroll
      for (int v = 0; v < STRIP / 4; ++v) {
        const i32x4 a = *reinterpret_cast<const i32x4 *>(row + x0 + 4 * v);
        const i32x4 d = (y + 1 < H) ? *reinterpret_cast<const i32x4 *>(row + W + x0 + 4 * v) : (i32x4){-2, -2, -2, -2};
#pragma unroll
        for (int e = 0; e < 4; ++e) { lab[4 * v + e] = a[e]; dn[4 * v + e] = d[e]; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < STRIP; ++i) {
        lab[i] = (i < n) ? row[x0 + i] : -2;
        dn[i] = (i < n && y + 1 < H) ? row[x0 + i + W] : -2;
      }
    }
    lab[STRIP] = (live && x0 + STRIP < W) ? row[x0 + STRIP] : -2;
    if (n < STRIP) lab[n] = (live && x0 + n < W) ? row[x0 + n] : -2;
    long long run_key = EMPTY_KEY;
    int run_cnt = 0;
    auto emit = [&](int a, int b) {
      if (a == b || a < 0 || b < 0 || a >= S || b >= S) return;
      const long long key = (long long)min(a, b) * S + max(a, b);
      if (key == run_key) { ++run_cnt; return; }
      if (run_cnt) tile_add(run_key, run_cnt);
      run_key = key; run_cnt = 1;
    };
#pragma unroll
    for (int i = 0; i < STRIP; ++i)
      if (i < n) emit(lab[i], dn[i]);                // vertical neighbours first: a boundary running along the row is one run
#pragma unroll
    for (int i = 0; i < STRIP; ++i)
      if (i < n) emit(lab[i], lab[i + 1]);           // (ids < 0, incl. the -2 "outside" marker, are dropped by emit)
    if (run_cnt) tile_add(run_key, run_cnt);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ESLOTS; i += blockDim.x)
    if (e_key[i] != EMPTY_KEY) table_add(keys, cnt, mask, e_key[i], e_cnt[i], overflow);
}

__global__ void table_compact_kernel(const long long *__restrict__ keys, const int *__restrict__ cnt, long long n,
                                     long long *__restrict__ out_keys, int *__restrict__ out_cnt, int *__restrict__ n_out, int max_out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long k = keys[i];
    if (k == EMPTY_KEY) continue;
    const int pos = atomicAdd(n_out, 1);
    if (pos < max_out) { out_keys[pos] = k; out_cnt[pos] = cnt[i]; }
  }
}

inline int grid_for(long long items, int cap = 8192) {
  long long g = (items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int dm_label_stats(const int32_t *labels, const uint8_t *tile, int32_t bands, int32_t H, int32_t W, int32_t S,
                              int64_t *count, int64_t *sum, int64_t *sumsq, int32_t *bbox, int64_t *peri, void *stream) {
  DM_REQUIRE(labels && tile && count && sum && sumsq && bbox && peri, DM_ERR_BAD_SHAPE, "dm_label_stats: null pointer");
  DM_REQUIRE(H > 0 && W > 0 && S > 0 && bands >= 1, DM_ERR_BAD_SHAPE, "dm_label_stats: bad sizes (H=%d W=%d S=%d bands=%d)", H, W, S, bands);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nb = bands < 3 ? bands : 3;
  hipLaunchKernelGGL(rag_init_kernel, dim3((S + 255) / 256), dim3(256), 0, s, (long long *)count, (long long *)sum, (long long *)sumsq, bbox,
                     (long long *)peri, S, nb);
  const dim3 grid((unsigned)(((W + 63) / 64) * ((H + 63) / 64)));     // one workgroup per 64x64-pixel tile
  // 16-byte strip loads need W % 16 == 0 and 16-byte aligned rasters
  const bool vec = (W % STRIP == 0) && dm_aligned16(labels) && dm_aligned16(tile);
#define DM_STATS(NB_)                                                                                                                      \
  do {                                                                                                                                     \
    if (vec) hipLaunchKernelGGL((label_stats_kernel<NB_, true>), grid, dim3(256), 0, s, labels, tile, H, W, S, (long long *)count,        \
                                (long long *)sum, (long long *)sumsq, bbox, (long long *)peri);                                           \
    else hipLaunchKernelGGL((label_stats_kernel<NB_, false>), grid, dim3(256), 0, s, labels, tile, H, W, S, (long long *)count,           \
                            (long long *)sum, (long long *)sumsq, bbox, (long long *)peri);                                               \
  } while (0)
  switch (nb) {
    case 1: DM_STATS(1); break;
    case 2: DM_STATS(2); break;
    default: DM_STATS(3); break;
  }
#undef DM_STATS
  DM_LAUNCH_CHECK("dm_label_stats");
  return DM_OK;
}

extern "C" int dm_label_features(const int64_t *count, const int64_t *sum, const int64_t *sumsq, const int32_t *bbox, const int64_t *peri,
                                 int32_t S, int32_t bands, float *features, void *stream) {
  DM_REQUIRE(count && sum && sumsq && bbox && peri && features && S > 0 && bands >= 1, DM_ERR_BAD_SHAPE, "dm_label_features: bad arguments");
  hipLaunchKernelGGL(label_features_kernel, dim3((S + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const long long *)count,
                     (const long long *)sum, (const long long *)sumsq, bbox, (const long long *)peri, S, bands < 3 ? bands : 3, features);
  DM_LAUNCH_CHECK("dm_label_features");
  return DM_OK;
}

extern "C" int dm_rag_edges(const int32_t *labels, int32_t H, int32_t W, int32_t S, int64_t *table_keys, int32_t *table_counts,
                            int32_t capacity_log2, int64_t *edge_keys, int32_t *edge_counts, int32_t max_edges, int32_t *n_edges,
                            int32_t *overflow, void *stream) {
  DM_REQUIRE(labels && table_keys && table_counts && edge_keys && edge_counts && n_edges && overflow, DM_ERR_BAD_SHAPE, "dm_rag_edges: null pointer");
  DM_REQUIRE(H > 0 && W > 0 && S > 0 && capacity_log2 >= 8 && capacity_log2 <= 30 && max_edges > 0, DM_ERR_BAD_SHAPE,
             "dm_rag_edges: bad sizes (H=%d W=%d S=%d capacity_log2=%d)", H, W, S, capacity_log2);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long cap = 1LL << capacity_log2;
  hipLaunchKernelGGL(table_clear_kernel, dim3(grid_for(cap, 2048)), dim3(256), 0, s, (long long *)table_keys, table_counts, cap, overflow, n_edges);
  const dim3 tgrid((unsigned)(((W + 63) / 64) * ((H + 63) / 64)));
  if (W % STRIP == 0 && dm_aligned16(labels))
    hipLaunchKernelGGL(rag_edges_kernel<true>, tgrid, dim3(256), 0, s, labels, H, W, S, (long long *)table_keys, table_counts,
                       (unsigned)(cap - 1), overflow);
  else
    hipLaunchKernelGGL(rag_edges_kernel<false>, tgrid, dim3(256), 0, s, labels, H, W, S, (long long *)table_keys, table_counts,
                       (unsigned)(cap - 1), overflow);
  hipLaunchKernelGGL(table_compact_kernel, dim3(grid_for(cap, 2048)), dim3(256), 0, s, (const long long *)table_keys, table_counts, cap,
                     (long long *)edge_keys, edge_counts, n_edges, max_edges);
  DM_LAUNCH_CHECK("dm_rag_edges");
  return DM_OK;
}

// ---- merge step of the region-adjacency sweep: connected components over the edges flagged `merge` ------------------------
// (optional: the reference stops at writing `simi` and leaves the merging to external GIS tooling.)
// Min-id union-find: roots are hooked under the smaller root with atomicMin, so every component ends up labelled with its
// smallest member id whatever the order of the updates (deterministic result).  One round = hook over all merging edges +
// full path compression; the host repeats rounds until the `changed` flag stays 0.
namespace {
__device__ __forceinline__ int uf_find(const int *parent, int x) {
  int p = parent[x];
  while (p != x) { x = p; p = parent[x]; }
  return x;
}
__global__ void uf_init_kernel(int *parent, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < S) parent[i] = i;
}
__global__ void uf_hook_kernel(const int *__restrict__ edges, const unsigned char *__restrict__ merge, int E, int S, int *parent,
                               int *changed) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x) {
    if (!merge[e]) continue;
    const int a = edges[2 * e], b = edges[2 * e + 1];
    if (a < 0 || b < 0 || a >= S || b >= S) continue;
    const int ra = uf_find(parent, a), rb = uf_find(parent, b);
    if (ra == rb) continue;
    const int lo = min(ra, rb), hi = max(ra, rb);
    atomicMin(parent + hi, lo);
    *changed = 1;
  }
}
__global__ void uf_compress_kernel(int *parent, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < S) parent[i] = uf_find(parent, i);
}
}  // namespace

extern "C" int dm_merge_round(const int32_t *edges, const uint8_t *merge, int32_t E, int32_t S, int32_t *parent, int32_t *changed,
                              int32_t init, void *stream) {
  DM_REQUIRE(parent && changed && E >= 0 && S > 0 && (E == 0 || (edges && merge)), DM_ERR_BAD_SHAPE, "dm_merge_round: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (init) hipLaunchKernelGGL(uf_init_kernel, dim3((S + 255) / 256), dim3(256), 0, s, parent, S);
  hipMemsetAsync(changed, 0, sizeof(int32_t), s);
  if (E > 0) hipLaunchKernelGGL(uf_hook_kernel, dim3(grid_for(E, 2048)), dim3(256), 0, s, edges, merge, E, S, parent, changed);
  hipLaunchKernelGGL(uf_compress_kernel, dim3((S + 255) / 256), dim3(256), 0, s, parent, S);
  DM_LAUNCH_CHECK("dm_merge_round");
  return DM_OK;
}
