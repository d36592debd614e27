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

// ---- normalisation of the batch's advantages (core/common.py:22: (A - mean) / std, torch's unbiased std): two launches instead
// of the ~15 tensor kernels (two reductions and a dozen scalar operations, each a 5-6 us launch in the middle of the update's
// first milliseconds).  Sums in float64 in a fixed order -- ADV_BLOCKS chunk sums by a tree per block, then every block of the
// second launch adds the chunk sums in index order -- so the result does not depend on the launch.
#define ADV_BLOCKS 256
__global__ __launch_bounds__(256) void hoic_adv_moments_kernel(const float* __restrict__ a, long long n, double* __restrict__ part) {
  __shared__ double s0[256], s1[256];
  const long long per = (n + ADV_BLOCKS - 1) / ADV_BLOCKS, lo = (long long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double x0 = 0.0, x1 = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) { const double v = (double)a[i]; x0 += v; x1 += v * v; }
  s0[threadIdx.x] = x0; s1[threadIdx.x] = x1;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { s0[threadIdx.x] += s0[threadIdx.x + o]; s1[threadIdx.x] += s1[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s0[0]; part[2 * blockIdx.x + 1] = s1[0]; }
}
__global__ __launch_bounds__(256) void hoic_adv_apply_kernel(float* __restrict__ a, long long n, const double* __restrict__ part) {
  __shared__ double sm[2];
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int b = 0; b < ADV_BLOCKS; b++) { t0 += part[2 * b]; t1 += part[2 * b + 1]; }
    const double mean = t0 / (double)n, var = (t1 - (double)n * mean * mean) / (double)(n - 1);
    sm[0] = mean; sm[1] = 1.0 / sqrt(var);
  }
  __syncthreads();
  const double mean = sm[0], rstd = sm[1];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) a[i] = (float)(((double)a[i] - mean) * rstd);
}

// ---- the logger's statistics of a fixed-horizon rollout (LoggerRL as the reference's sampler fills it step by step,
// agent_handmimic.py:476-482, uhc/khrylib/rl/core/logger_rl.py) and the batch's masks, in ONE launch over the rollout's [T x N]
// storage instead of ~28 tensor kernels at the end of every rollout: c_reward = reward - end bonus on 'end' steps; its sum,
// minimum and maximum, the number of finished episodes and the sums of the nine reward terms, all in float64; masks = 1 - done.
// Fixed order: a tree per block, then the LAST block to finish (a ticket) adds the blocks' partial results in block order.
#define RS_BLOCKS 128
#define RS_MAXINFO 16
__global__ __launch_bounds__(256) void hoic_rollout_stats_kernel(long long n, const float* __restrict__ rewards, const int* __restrict__ flags,
                                                                 const float* __restrict__ rinfo, int n_info, float bonus,
                                                                 float* __restrict__ masks, double* __restrict__ part, unsigned* __restrict__ ticket,
                                                                 double* __restrict__ stats) {
  __shared__ double red[256];
  __shared__ bool last;
  const int tid = threadIdx.x, W = 4 + n_info;
  const long long per = (n + RS_BLOCKS - 1) / RS_BLOCKS, lo = (long long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double acc[4 + RS_MAXINFO];
  acc[0] = 0.0; acc[1] = 1e300; acc[2] = -1e300; acc[3] = 0.0;
  for (int k = 0; k < n_info; k++) acc[4 + k] = 0.0;
  for (long long i = lo + tid; i < hi; i += 256) {
    const int end = flags[4 * i + 1], done = flags[4 * i + 2];
    const double cr = (double)rewards[i] - (double)bonus * (end != 0 ? 1.0 : 0.0);
    acc[0] += cr; acc[1] = cr < acc[1] ? cr : acc[1]; acc[2] = cr > acc[2] ? cr : acc[2]; acc[3] += done != 0 ? 1.0 : 0.0;
    for (int k = 0; k < n_info; k++) acc[4 + k] += (double)rinfo[i * n_info + k];
    if (masks) masks[i] = done != 0 ? 0.f : 1.f;
  }
  for (int q = 0; q < W; q++) {       // one tree per quantity (sum / min / max)
    red[tid] = acc[q];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        const double a = red[tid], b = red[tid + o];
        red[tid] = q == 1 ? (b < a ? b : a) : (q == 2 ? (b > a ? b : a) : a + b);
      }
      __syncthreads();
    }
    if (tid == 0) part[(size_t)blockIdx.x * W + q] = red[0];
    __syncthreads();
  }
  if (tid == 0) {
    __threadfence();
    last = atomicAdd(ticket, 1u) == RS_BLOCKS - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (tid < W) {
    double r = tid == 1 ? 1e300 : (tid == 2 ? -1e300 : 0.0);
    for (int b = 0; b < RS_BLOCKS; b++) {
      const double v = __hip_atomic_load(&part[(size_t)b * W + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (written by other blocks)
      r = tid == 1 ? (v < r ? v : r) : (tid == 2 ? (v > r ? v : r) : r + v);
    }
    stats[tid] = r;
  }
  if (tid == 0) *ticket = 0u;       // ready for the next launch
}
