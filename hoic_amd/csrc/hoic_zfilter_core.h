// hoic_zfilter_core.h — the arithmetic of the running observation filter (uhc/khrylib/utils/zfilter.py:8-73) as inline device
// functions, shared by the two-launch form (hoic_zfilter.h, hoic_capi.hip) and the sampler's one-launch form that also writes the
// rollout forward's operand (hoic_mlp.hip hoic_zfilter_tiled_kernel): both forms execute THIS source, so a row normalised by
// either is the same float.
#pragma once
#include <hip/hip_runtime.h>

#define ZF_ROWS 128      // rows per chunk

// mean and sum of squared deviations (float64) of rows r0 .. r1 - 1 of column `col` (two passes, four accumulators each)
__device__ __forceinline__ void zf_chunk_moments(const float* __restrict__ x, int dim, int col, int r0, int r1, double& mean_out, double& m2_out) {
  const float* p = x + (size_t)r0 * dim + col;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int r = r0;
  for (; r + 4 <= r1; r += 4, p += (size_t)4 * dim) {
    s0 += (double)p[0]; s1 += (double)p[dim]; s2 += (double)p[2 * (size_t)dim]; s3 += (double)p[3 * (size_t)dim];
  }
  for (; r < r1; r++, p += dim) s0 += (double)p[0];
  const double mean = ((s0 + s1) + (s2 + s3)) / (double)(r1 - r0);
  p = x + (size_t)r0 * dim + col;
  double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
  for (r = r0; r + 4 <= r1; r += 4, p += (size_t)4 * dim) {
    const double d0 = (double)p[0] - mean, d1 = (double)p[dim] - mean, d2 = (double)p[2 * (size_t)dim] - mean, d3 = (double)p[3 * (size_t)dim] - mean;
    q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2; q3 += d3 * d3;
  }
  for (; r < r1; r++, p += dim) { const double d0 = (double)p[0] - mean; q0 += d0 * d0; }
  mean_out = mean; m2_out = (q0 + q1) + (q2 + q3);
}
// Chan's pairwise update of (count, mean, S) with a chunk of nb rows (mean mb, sum of squared deviations Sb)
__device__ __forceinline__ void zf_merge(double& cnt, double& mean, double& S, double nb, double mb, double Sb) {
  const double tot = cnt + nb, delta = mb - mean;
  S = S + Sb + delta * delta * cnt * nb / tot;
  mean = mean + delta * nb / tot;
  cnt = tot;
}
// 1 / (std + 1e-8) of a column (zfilter.py:35: var = S / (n - 1), mean^2 while n == 1)
__device__ __forceinline__ double zf_rden(double cnt, double mean, double S) {
  const double var = cnt > 1.0 ? S / fmax(cnt - 1.0, 1.0) : mean * mean;
  return 1.0 / (sqrt(var) + 1e-8);
}
__device__ __forceinline__ float zf_apply(float x, double mean, double rden, double lim) {
  const double v = ((double)x - mean) * rden;
  return (float)fmin(fmax(v, -lim), lim);
}
