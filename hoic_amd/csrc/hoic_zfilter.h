// hoic_zfilter.h — the running observation filter on the device (khrylib ZFilter / RunningStat,
// uhc/khrylib/utils/zfilter.py:8-73; the sampler calls it on every observation, agent_handmimic.py:463).
//
// A batch [n, dim] is pushed as a whole: per 128-row chunk the column means and sums of squared deviations (two passes
// over the chunk, float64), then the chunks are merged one after the other into the running (count, mean, S) with
// Chan's pairwise update - the same statistics as pushing the rows one at a time - and every row is normalised with
// the statistics after the batch:  y = clip((x - mean) / (sqrt(S / (count - 1)) + 1e-8), +-clip).
// Lane = column: a wave reads 256 contiguous bytes of a row, nothing is reduced across lanes, the result does not
// depend on the launch geometry.  Two launches replace ~30 small tensor kernels of the host mirror.
#pragma once
#include <hip/hip_runtime.h>
#include "hoic_zfilter_core.h"

#define ZF_NT 64

// partial[(chunk * dim + col) * 2 + {0, 1}] = mean, M2 of the chunk's rows
__global__ __launch_bounds__(ZF_NT) void hoic_zfilter_moments_kernel(const float* __restrict__ x, int n, int dim,
                                                                     double* __restrict__ partial) {
  const int col = blockIdx.x * ZF_NT + threadIdx.x, chunk = blockIdx.y;
  if (col >= dim) return;
  const int r0 = chunk * ZF_ROWS, r1 = min(r0 + ZF_ROWS, n);
  double mean, m2;
  zf_chunk_moments(x, dim, col, r0, r1, mean, m2);
  partial[((size_t)chunk * dim + col) * 2] = mean;
  partial[((size_t)chunk * dim + col) * 2 + 1] = m2;
}

// state = (count, mean[dim], S[dim]).  Every workgroup merges the chunks of its 64 columns itself (a few hundred
// flops), the workgroups of chunk 0 write the new state, all normalise the rows of their chunk.
__global__ __launch_bounds__(ZF_NT) void hoic_zfilter_apply_kernel(const float* __restrict__ x, int n, int dim,
                                                                   const double* __restrict__ partial,
                                                                   const double* __restrict__ state_in, double* __restrict__ state_out,
                                                                   int update, float clip, float* __restrict__ y) {
  const int col = blockIdx.x * ZF_NT + threadIdx.x, chunk = blockIdx.y;
  if (col >= dim) return;
  double cnt = state_in[0], mean = state_in[1 + col], S = state_in[1 + dim + col];
  if (update) {
    const int nchunk = (n + ZF_ROWS - 1) / ZF_ROWS;
    for (int c = 0; c < nchunk; c++) {
      const double nb = (double)(min((c + 1) * ZF_ROWS, n) - c * ZF_ROWS);
      zf_merge(cnt, mean, S, nb, partial[((size_t)c * dim + col) * 2], partial[((size_t)c * dim + col) * 2 + 1]);
    }
    if (chunk == 0) {
      state_out[1 + col] = mean; state_out[1 + dim + col] = S;
      if (col == 0) state_out[0] = cnt;
    }
  }
  if (!y) return;
  const double rden = zf_rden(cnt, mean, S);      // one division per column; the rows multiply (float64: the float32 result sees no difference)
  const int r0 = chunk * ZF_ROWS, r1 = min(r0 + ZF_ROWS, n);
  const double lim = (double)clip;
#pragma unroll 4
  for (int r = r0; r < r1; r++) y[(size_t)r * dim + col] = zf_apply(x[(size_t)r * dim + col], mean, rden, lim);
}

// ---- merge what the per-range forks of the filter saw during a pipelined rollout back into the filter they were forked from
// (hoic_amd/rl.py BatchZFilter.absorb; the reference's sampler threads each run their own copy of the filter, agent.py:64-120):
// thread = column, the forks in order.  The tensor form of this was ~35 float64 tensor operations per fork -- 77 launches of
// 5.6 us back to back at the end of every rollout, 0.45 ms of an iteration.  Every operation is the correctly rounded IEEE one
// in the order of the tensor expression (no fused multiply-adds), so the merged statistics are bit-identical to it.
#define ZF_MAXFORK 8
struct ZfForks { const double* st[ZF_MAXFORK]; int n; };
__global__ __launch_bounds__(ZF_NT) void hoic_zfilter_absorb_kernel(const double* __restrict__ base, ZfForks forks, int dim, double* __restrict__ out) {
  // (the library is built with -ffp-contract=fast, which lets the backend fuse any product into a following sum whatever the
  //  source says -- pragma and __dmul_rn included: fn * fm - n0 * m0 must stay two products and a difference, so the products
  //  that feed a sum pass through an empty asm)
  const int col = blockIdx.x * ZF_NT + threadIdx.x;
  if (col >= dim) return;
  const double n0 = base[0], m0 = base[1 + col], S0 = base[1 + dim + col];
  double n1 = n0, m1 = m0, S1 = S0;
  for (int f = 0; f < forks.n; f++) {
    const double fn = forks.st[f][0], fm = forks.st[f][1 + col], fS = forks.st[f][1 + dim + col];
    const double nb = fn - n0;
    const double safe = fmax(nb, 1.0);
    double p1 = fn * fm, p0 = n0 * m0;
    asm volatile("" : "+v"(p1), "+v"(p0));        // (products that feed a sum are pinned: see above)
    const double mb = (p1 - p0) / safe;
    const double dm = mb - m0;
    const double corr = dm * dm * n0 * nb / fmax(fn, 1.0);
    const double Sb = fmax((fS - S0) - corr, 0.0);
    const double tot = n1 + nb, delta = mb - m1;
    const double w = nb > 0.0 ? 1.0 : 0.0;                     // a fork that saw nothing changes nothing
    const double den = fmax(tot, 1.0);
    const double Sadd = Sb + delta * delta * n1 * nb / den;
    double wS = w * Sadd;
    asm volatile("" : "+v"(wS));
    S1 = S1 + wS;
    const double dmean = w * delta * nb / den;
    m1 = m1 + dmean;
    n1 = tot;
  }
  out[1 + col] = m1; out[1 + dim + col] = S1;
  if (col == 0) out[0] = n1;
}

// ---- generalized advantage estimation over a time-major rollout (khrylib core/common.py:12-19): thread = env, the
// recursion over T runs in registers; every float32 operation is rounded like the tensor expression of the host
// mirror (no fused multiply-adds), so both give bit-identical advantages.
//   delta_t = r_t + gamma V_{t+1} m_t - V_t,   A_t = delta_t + gamma tau A_{t+1} m_t,   returns_t = V_t + A_t
__global__ void hoic_gae_kernel(int T, int N, const float* __restrict__ rewards, const float* __restrict__ masks,
                                const float* __restrict__ values, const float* __restrict__ next_values, float gamma,
                                float gamma_tau, float* __restrict__ adv, float* __restrict__ returns) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float prev_v = next_values ? next_values[n] : 0.f, prev_a = 0.f;
  for (int t = T - 1; t >= 0; t--) {
    const size_t k = (size_t)t * N + n;
    const float r = rewards[k], m = masks[k], v = values[k];
    const float delta = __fsub_rn(__fadd_rn(r, __fmul_rn(__fmul_rn(gamma, prev_v), m)), v);
    prev_a = __fadd_rn(delta, __fmul_rn(__fmul_rn(gamma_tau, prev_a), m));
    adv[k] = prev_a;
    returns[k] = __fadd_rn(v, prev_a);
    prev_v = v;
  }
}
