/*
 * Oracle (TEST INFRASTRUCTURE ONLY): the ExtractFeatures sweep with a PINNED float32 evaluation order.
 *
 * Same formulas as oracle/sweep.py, i.e. the reference's
 *   np.mean(rows, axis=0)                         ExtractFeatures.py:211-212
 *   Euclidean_distance on two 1 x p rows, .max()  ExtractFeatures.py:139-147, :215-216
 *   skip edges with LEFT_FID/RIGHT_FID == -1      MyUtils2.py:184-186
 * but every sum is written out in one explicit order, so the GPU kernels (dm_segment_mean,
 * dm_edge_similarity) can be required to match BIT FOR BIT:
 *   - mean over k rows: acc = row0; acc += row1; ...; acc / k      (numpy add.reduce over axis 0 of a
 *     C-contiguous block accumulates row by row; true_divide by the count)
 *   - |x|^2, |y|^2 and x.y over p elements: numpy's pairwise summation for a contiguous float32
 *     vector (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum): for p < 8 a plain loop from 0;
 *     for 8 <= p <= 128 eight accumulators r[j] += t[8i+j], combined as
 *     ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the p%8 tail added in order; above 128 the vector is
 *     split recursively at (p/2 - (p/2)%8).  The products t are rounded to float32 first.
 *     numpy evaluates x.y with BLAS (order is implementation defined); the oracle pins the SAME
 *     pairwise order for it -- tests/test_oracle_sweep.py bounds the difference to np.dot.
 *   - d = (|x|^2 + |y|^2) - 2*(x.y); d < 0 -> 0; simi = sqrtf(d) (correctly rounded).
 * Build with -ffp-contract=off (no FMA contraction): see oracle/Makefile.
 */
#include <math.h>
#include <stdint.h>

static float pairwise(const float *t, int n) {
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res += t[i];
    return res;
  }
  if (n <= 128) {
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = t[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += t[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += t[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise(t, n2) + pairwise(t + n2, n - n2);
}

void dm_oracle_segment_mean(const float *F, const int32_t *ptr, const int32_t *idx, float *pooled, int32_t S, int32_t D) {
  for (int s = 0; s < S; ++s) {
    const int beg = ptr[s], end = ptr[s + 1];
    for (int c = 0; c < D; ++c) {
      float acc = 0.f;
      if (end > beg) {
        acc = F[(int64_t)idx[beg] * D + c];
        for (int k = beg + 1; k < end; ++k) acc += F[(int64_t)idx[k] * D + c];
        acc = acc / (float)(end - beg);
      }
      pooled[(int64_t)s * D + c] = acc;
    }
  }
}

void dm_oracle_edge_similarity(const float *pooled, const int32_t *edges, float *simi, uint8_t *merge, int32_t E, int32_t D,
                               float margin) {
  float tx[4096], ty[4096], txy[4096];
  for (int e = 0; e < E; ++e) {
    const int L = edges[2 * e], R = edges[2 * e + 1];
    if (L < 0 || R < 0 || D > 4096) {
      simi[e] = NAN;
      if (merge) merge[e] = 0;
      continue;
    }
    const float *x = pooled + (int64_t)L * D, *y = pooled + (int64_t)R * D;
    for (int i = 0; i < D; ++i) {
      tx[i] = x[i] * x[i];
      ty[i] = y[i] * y[i];
      txy[i] = x[i] * y[i];
    }
    const float xx = pairwise(tx, D), yy = pairwise(ty, D), xy = pairwise(txy, D);
    float d = (xx + yy) - 2.0f * xy;
    if (d < 0.f) d = 0.f;
    const float sm = sqrtf(d);
    simi[e] = sm;
    if (merge) merge[e] = (sm < margin) ? 1 : 0;
  }
}
