// hoic_solver.h — constraint rows and the convex constraint solve, one wavefront per env.
//
// Replaces mj_makeConstraint + the Newton solver inside self.sim.step() (uhc/envs/ho_im4.py:545).
// Problem (MuJoCo's primal form, unique optimum):
//     min_a  1/2 (a - a0)' M (a - a0) + sum_r s_r(J_r a - aref_r)
// rows: dof friction loss (Huber), joint limits and pyramidal contact edges (one-sided quadratics).
// MI355X mapping: nv = 32 unknowns = half a wavefront.  The 32x32 Hessian lives one ROW PER LANE in
// registers; the dense Cholesky and both triangular solves broadcast pivots with v_readlane (no LDS round
// trips, no barriers inside the factorisation).  Contact rows are never materialised: each contact stores
// 4 frame-Jacobian rows (n, t1, t2, spin) in LDS and the pyramid edges are formed on the fly.
#pragma once
#include "hoic_types.h"
#include "hoic_math.h"

// ---- 32x32 SPD assemble + solve on the matrix core.
// The matrix lives in ONE v_mfma_f32_32x32x2_f32 accumulator (16 VGPRs per lane; element (row, col) sits in
// lane (col + 32*((row>>2)&1)), register (row&3) + 4*(row>>3)).  Exact f32 (an fma chain), so numerics equal
// the VALU version.
//   A = M (+ diag) restricted to the leading nact x nact block, identity elsewhere
//   (+ sum over active contact rows  curv_r J_r' J_r : two rank-1 terms per MFMA, K = 2)
// Factorisation: right-looking LDL^T, one rank-1 MFMA per pivot (64 cycles) instead of 31 readlane+FMA pairs.
// Row j of the running matrix (= column j of L times d_j) is one value per lane, so the scaled column is also
// the MFMA A-operand.  Columns of L are kept one register each (lane = row); the backward substitution gets
// the transpose through the 32x33 LDS scratch `T`.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

template <int J> HD float hs_row_bcast(const f32x16& acc) {
  constexpr int reg = (J & 3) + 4 * (J >> 3);
  constexpr int half = (J >> 2) & 1;
  const unsigned v = __float_as_uint(acc[reg]);
  const u32x2v r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // .x = low half on both, .y = high half on both
  return __uint_as_float(half ? r.y : r.x);
}

template <int J> struct HsFactor {
  static HD void run(f32x16& acc, float (&lcol)[32], float& dinv, int col, int hi) {
    const float u = hs_row_bcast<J>(acc);           // u[lane&31] = A[J][lane&31] = L[.][J] * d_J
    const float d = fmaxf(rl(u, J), 1e-30f);
    const float inv = 1.f / d;
    const float lj = u * inv;
    if (col == J) dinv = inv;
    lcol[J] = lj;
    if (J < 31) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? 0.f : -lj, hi ? 0.f : u, acc, 0, 0, 0);
    HsFactor<J + 1>::run(acc, lcol, dinv, col, hi);
  }
};
template <> struct HsFactor<32> { static HD void run(f32x16&, float (&)[32], float&, int, int) {} };

// diag: 32 floats in LDS added to the diagonal; use_rows: add the active contact rows through the MFMA (the
// caller folds friction/limit curvature into diag); b (LDS, 32) is overwritten by x.
__device__ __forceinline__ void dev_hsolve(const DevModel& m, Work& w, const float* diag, int nact, bool use_rows, float* b) {
  const int lane = threadIdx.x, col = lane & 31, hi = lane >> 5;
  f32x16 acc;
  const float dg = diag[col];     // diagonal increment of row/col `col`
#pragma unroll
  for (int reg = 0; reg < 16; reg++) {
    const int r = (reg & 3) + 8 * (reg >> 2) + 4 * hi;
    float v = w.M[r * LD + col];
    if (r == col) v += dg;
    if (r >= nact || col >= nact) v = (r == col) ? 1.f : 0.f;
    acc[reg] = v;
  }
  if (use_rows) {
    for (int c = 0; c < w.ncon; c++) {
      const int nr = w.c_nrow[c], r0 = w.c_row0[c];
      const float* J = &w.Jc[(c * NBASIS) * LD];
      const float jn = J[col];
      for (int p = 0; 2 * p < nr; p++) {       // edges 2p (low half of the wave) and 2p+1 (high half)
        const float cu0 = w.r_curv[r0 + 2 * p], cu1 = (2 * p + 1 < nr) ? w.r_curv[r0 + 2 * p + 1] : 0.f;
        if (cu0 == 0.f && cu1 == 0.f) continue;
        float v = jn;
        if (nr > 1) v += (hi ? -1.f : 1.f) * w.c_mu[c][p] * J[(1 + p) * LD + col];
        const float cu = hi ? cu1 : cu0;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cu * v, v, acc, 0, 0, 0);
      }
    }
  }
  PT(15);
  float lcol[32], dinv = 1.f;
  HsFactor<0>::run(acc, lcol, dinv, col, hi);
  PT(16);
  // forward substitution (unit lower), diagonal scaling
  float y = b[col];
#pragma unroll
  for (int k = 0; k < 31; k++) {
    const float yk = rl(y, k);
    if (col > k) y -= lcol[k] * yk;
  }
  y *= dinv;
  PT(17);
  __syncthreads();
  float* T = w.H;
  if (lane < 32) {
#pragma unroll
    for (int k = 0; k < 32; k++) T[lane * LD + k] = lcol[k];     // row `lane` of L
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 32; k++) lcol[k] = T[k * LD + col];        // column `col` of L (valid for k > col)
  float x = y;
#pragma unroll
  for (int k = 31; k > 0; k--) {
    const float xk = rl(x, k);
    if (col < k) x -= lcol[k] * xk;
  }
  PT(18);
  __syncthreads();
  if (lane < 32) b[lane] = x;
  __syncthreads();
  PT(19);
}

// impedance d(r) from solimp [MJ-doc: getimpedance]
HD float dev_impedance(const float* s_in, float pos, float margin) {
  float s0 = fminf(fmaxf(s_in[0], 0.0001f), 0.9999f), s1 = fminf(fmaxf(s_in[1], 0.0001f), 0.9999f);
  float wdt = fmaxf(s_in[2], 0.f), mid = fminf(fmaxf(s_in[3], 0.0001f), 0.9999f), pw = fmaxf(s_in[4], 1.f);
  if (s0 == s1 || wdt <= MINVALF) return 0.5f * (s0 + s1);
  float x = fabsf((pos - margin) / wdt);
  if (x >= 1.f) return s1;
  if (x <= 0.f) return s0;
  float y;
  if (pw == 1.f) y = x;
  else if (x <= mid) y = powf(x, pw) / powf(mid, pw - 1.f);
  else y = 1.f - powf(1.f - x, pw) / powf(1.f - mid, pw - 1.f);
  return s0 + y * (s1 - s0);
}

// ---- u[t] = Jc[t,:] . x for every stored contact-frame row t
__device__ void dev_basis_dot(Work& w, const float* x) {
  const int nb = w.ncon * NBASIS;
  for (int t = threadIdx.x; t < nb; t += NT) {
    float s = 0.f;
    const float* row = &w.Jc[t * LD];
#pragma unroll 8
    for (int d = 0; d < NV; d++) s += row[d] * x[d];
    w.u[t] = s;
  }
  __syncthreads();
}

// J_r . x for constraint row r (u must hold dev_basis_dot(x))
HD float dev_row_times(const DevModel& m, const Work& w, int r, const float* x) {
  if (r < m.nv) return x[r];
  if (r < m.nv + w.nlim) { const int l = r - m.nv; return w.lim_sign[l] * x[w.lim_dof[l]]; }
  const int c = w.r_con[r], e = w.r_edge[r];
  const float un = w.u[c * NBASIS];
  if (w.c_nrow[c] == 1) return un;
  const int k = e >> 1;
  return un + ((e & 1) ? -1.f : 1.f) * w.c_mu[c][k] * w.u[c * NBASIS + 1 + k];
}

// per-row cost pieces: returns cost, sets force = -ds/djar and curvature
HD float dev_row_cost(const DevModel& m, const Work& w, int r, float jar, float& force, float& curv) {
  if (r < m.nv) {
    const float f = m.dof_frictionloss[r], R = m.dof_flR[r], D = 1.f / R;
    if (jar <= -R * f) { force = f; curv = 0.f; return -f * (0.5f * R * f + jar); }
    if (jar >= R * f) { force = -f; curv = 0.f; return -f * (0.5f * R * f - jar); }
    force = -D * jar; curv = D; return 0.5f * D * jar * jar;
  }
  const float D = (r < m.nv + w.nlim) ? w.lim_D[r - m.nv] : w.c_D[w.r_con[r]];
  if (jar < 0.f) { force = -D * jar; curv = D; return 0.5f * D * jar * jar; }
  force = 0.f; curv = 0.f; return 0.f;
}

// ---- constraint rows for the current kinematics / contacts
__device__ void dev_make_constraint(const DevModel& m, Work& w, const float* qpos, const float* qvel) {
  const int tid = threadIdx.x;
  // joint limits (one side per joint can be active: every range is wider than twice the margin)
  {
    bool act = false; float dist = 0.f, sgn = 0.f;
    if (tid < m.njnt && m.jnt_limited[tid] && m.jnt_type[tid] != HOIC_JNT_FREE) {
      const float q = qpos[m.jnt_qposadr[tid]], dl = q - m.jnt_range[tid][0], du = m.jnt_range[tid][1] - q;
      if (dl < m.jnt_margin[tid]) { act = true; dist = dl; sgn = 1.f; }
      else if (du < m.jnt_margin[tid]) { act = true; dist = du; sgn = -1.f; }
    }
    const unsigned long long mask = __ballot(act);
    const int idx = __popcll(mask & ((1ull << tid) - 1ull));
    if (tid == 0) w.nlim = min(__popcll(mask), MAXLIM);
    if (act && idx < MAXLIM) {
      const int d = m.jnt_dofadr[tid];
      const float imp = dev_impedance(m.jnt_solimp[tid], dist, m.jnt_margin[tid]);
      const float R = fmaxf(MINVALF, (1.f - imp) * m.jnt_diag[tid] / imp);
      w.lim_dof[idx] = d; w.lim_sign[idx] = sgn; w.lim_D[idx] = 1.f / R;
      w.r_aref[m.nv + idx] = -m.jnt_B[tid] * (sgn * qvel[d]) - m.jnt_K[tid] * imp * (dist - m.jnt_margin[tid]);
    }
  }
  if (tid < m.nv) w.r_aref[tid] = -m.dof_flB[tid] * qvel[tid];
  __syncthreads();
  // per-contact parameters and row layout
  {
    int nrow = 0;
    const int c = tid;
    if (c < w.ncon) {
      const int p = w.c_pair[c], dim = m.pair_condim[p];
      nrow = dim == 1 ? 1 : 2 * (dim - 1);
      const float incl = m.pair_margin[p] - m.pair_gap[p];
      const float imp = dev_impedance(m.pair_solimp[p], w.c_dist[c], incl);
      const float R = fmaxf(MINVALF, (1.f - imp) / imp) * m.pair_Rscale[p];
      w.c_D[c] = 1.f / fmaxf(R, MINVALF);
      w.c_B[c] = m.pair_B[p];
      w.c_aref0[c] = -m.pair_K[p] * imp * (w.c_dist[c] - incl);
      for (int k = 0; k < 3; k++) w.c_mu[c][k] = m.pair_mu[p][k];
      w.c_nrow[c] = nrow;
    }
    int incl_sum = nrow;
#pragma unroll
    for (int o = 1; o < NT; o <<= 1) { int v = __shfl_up(incl_sum, o); if (tid >= o) incl_sum += v; }
    const int row0 = m.nv + w.nlim + incl_sum - nrow;
    if (c < w.ncon) {
      w.c_row0[c] = row0;
      for (int e = 0; e < nrow; e++) { w.r_con[row0 + e] = (unsigned char)c; w.r_edge[row0 + e] = (unsigned char)e; }
    }
    const int total = __shfl(incl_sum, NT - 1);
    if (tid == 0) w.nrow = m.nv + w.nlim + total;
  }
  __syncthreads();
  // contact-frame Jacobian rows: lane = dof, loop over contacts
  for (int c = 0; c < w.ncon; c++) {
    if (tid < NV) {
      const int p = w.c_pair[c];
      const int b1 = m.geom_bodyid[m.pair_geom1[p]], b2 = m.geom_bodyid[m.pair_geom2[p]];
      float jn = 0.f, jt1 = 0.f, jt2 = 0.f, js = 0.f;
      if (tid < m.nv) {
        const float sg = (float)((m.body_dofmask[b2] >> tid) & 1u) - (float)((m.body_dofmask[b1] >> tid) & 1u);
        if (sg != 0.f) {
          float col[3];
          cross3(w.S[tid], w.c_pos[c], col);
          for (int k = 0; k < 3; k++) col[k] += w.S[tid][3 + k];
          const float* f = w.c_frame[c];
          jn = sg * dot3(f, col); jt1 = sg * dot3(f + 3, col); jt2 = sg * dot3(f + 6, col);
          js = sg * dot3(f, w.S[tid]);
        }
      }
      float* J = &w.Jc[(c * NBASIS) * LD];
      J[tid] = jn; J[LD + tid] = jt1; J[2 * LD + tid] = jt2; J[3 * LD + tid] = js;
    }
  }
  __syncthreads();
  // reference accelerations of the contact rows
  dev_basis_dot(w, qvel);
  for (int r = m.nv + w.nlim + tid; r < w.nrow; r += NT) {
    const int c = w.r_con[r];
    w.r_aref[r] = -w.c_B[c] * dev_row_times(m, w, r, qvel) + w.c_aref0[c];
  }
  __syncthreads();
}

// out[i] = sum_k M[i,k] x[k]
HD float dev_Mrow(const Work& w, int i, const float* x) {
  float s = 0.f;
  const float* row = &w.M[i * LD];
#pragma unroll 8
  for (int k = 0; k < NV; k++) s += row[k] * x[k];
  return s;
}

// jar, force, curvature of every row at acceleration x; returns the constraint cost (wave-reduced)
__device__ float dev_eval_rows(const DevModel& m, Work& w, const float* x) {
  dev_basis_dot(w, x);
  float cost = 0.f;
  for (int r = threadIdx.x; r < w.nrow; r += NT) {
    const float jar = dev_row_times(m, w, r, x) - w.r_aref[r];
    float f, cv;
    cost += dev_row_cost(m, w, r, jar, f, cv);
    w.r_jar[r] = jar; w.r_force[r] = f; w.r_curv[r] = cv;
  }
  __syncthreads();
  return wave_sum(cost);
}

// out[i] = (J^T force)[i] from r_force (uses u as scratch for the per-contact frame forces)
__device__ void dev_jt_force(const DevModel& m, Work& w, float* out) {
  const int tid = threadIdx.x;
  for (int t = tid; t < w.ncon * NBASIS; t += NT) {
    const int c = t / NBASIS, k = t % NBASIS, r0 = w.c_row0[c], nr = w.c_nrow[c];
    float g = 0.f;
    if (k == 0) { for (int e = 0; e < nr; e++) g += w.r_force[r0 + e]; }
    else if (nr > 1 && 2 * k - 1 < nr) g = w.c_mu[c][k - 1] * (w.r_force[r0 + 2 * (k - 1)] - w.r_force[r0 + 2 * (k - 1) + 1]);
    w.u[t] = g;
  }
  __syncthreads();
  if (tid < m.nv) {
    float s = w.r_force[tid];
    for (int l = 0; l < w.nlim; l++) if (w.lim_dof[l] == tid) s += w.lim_sign[l] * w.r_force[m.nv + l];
    const int nb = w.ncon * NBASIS;
    for (int t = 0; t < nb; t++) s += w.Jc[t * LD + tid] * w.u[t];
    out[tid] = s;
  }
  __syncthreads();
}

// ---- Newton with exact line search.  In: M, fsmooth, asmooth, warm, rows.  Out: qacc, fcon.
__device__ void dev_solve(const DevModel& m, Work& w, int maxit) {
  const int tid = threadIdx.x;
  const float scale = 1.f / (m.meaninertia * (float)max(m.nv, 1));
  // warm start choice: cost(warm) vs cost(asmooth)
  float gw = 0.f;
  if (tid < m.nv) { const float Ma = dev_Mrow(w, tid, w.warm); gw = 0.5f * (Ma - w.fsmooth[tid]) * (w.warm[tid] - w.asmooth[tid]); }
  gw = wave_sum(gw);
  const float cw = gw + dev_eval_rows(m, w, w.warm);
  const float cs = dev_eval_rows(m, w, w.asmooth);
  if (tid < NV) w.qacc[tid] = (tid < m.nv) ? ((cw < cs) ? w.warm[tid] : w.asmooth[tid]) : 0.f;
  __syncthreads();
  int it = 0;
  for (; it < maxit; it++) {
    if (tid < m.nv) w.Ma[tid] = dev_Mrow(w, tid, w.qacc);
    dev_eval_rows(m, w, w.qacc);
    dev_jt_force(m, w, w.tv);
    float g2 = 0.f;
    if (tid < NV) {
      const float g = (tid < m.nv) ? (w.Ma[tid] - w.fsmooth[tid] - w.tv[tid]) : 0.f;
      w.grad[tid] = g; w.search[tid] = -g; g2 = g * g;
    }
    g2 = wave_sum(g2);
    __syncthreads();
    if (sqrtf(g2) * scale < 1e-6f) break;
    if (tid < NV) {   // diagonal curvature of the friction-loss and limit rows
      float dg = 0.f;
      if (tid < m.nv) {
        dg = w.r_curv[tid];
        for (int l = 0; l < w.nlim; l++) if (w.lim_dof[l] == tid) dg += w.r_curv[m.nv + l];
      }
      w.tv2[tid] = dg;
    }
    __syncthreads();
    PT(20);
    dev_hsolve(m, w, w.tv2, m.nv, true, w.search);
    // line-search quantities
    float gq = 0.f, hh = 0.f, g0 = 0.f;
    if (tid < m.nv) {
      const float Ms = dev_Mrow(w, tid, w.search);
      gq = (w.Ma[tid] - w.fsmooth[tid]) * w.search[tid]; hh = w.search[tid] * Ms; g0 = w.grad[tid] * w.search[tid];
    }
    gq = wave_sum(gq); hh = wave_sum(hh); g0 = wave_sum(g0);
    dev_basis_dot(w, w.search);
    float jar[NROW / NT], jv[NROW / NT];
#pragma unroll
    for (int k = 0; k < NROW / NT; k++) {
      const int r = tid + k * NT;
      jar[k] = 0.f; jv[k] = 0.f;
      if (r < w.nrow) { jar[k] = w.r_jar[r]; jv[k] = dev_row_times(m, w, r, w.search); }
    }
    float a = 0.f, lo = 0.f, hi = -1.f, alpha = 0.f;
    for (int ls = 0; ls < 10; ls++) {
      float dphi = 0.f, ddphi = 0.f;
#pragma unroll
      for (int k = 0; k < NROW / NT; k++) {
        const int r = tid + k * NT;
        if (r < w.nrow) {
          float f, cv;
          dev_row_cost(m, w, r, jar[k] + a * jv[k], f, cv);
          dphi -= f * jv[k]; ddphi += cv * jv[k] * jv[k];
        }
      }
      dphi = wave_sum(dphi) + gq + a * hh; ddphi = wave_sum(ddphi) + hh;
      alpha = a;
      if (fabsf(dphi) < 1e-5f * fabsf(g0) + 1e-12f) break;
      if (dphi < 0.f) lo = a; else hi = a;
      float an = a - dphi / ddphi;
      if (hi >= 0.f && (an <= lo || an >= hi)) an = 0.5f * (lo + hi);
      if (hi >= 0.f && hi - lo < 1e-6f * (1.f + hi)) break;
      a = an;
    }
    float st = 0.f;
    if (tid < m.nv) { const float dq = alpha * w.search[tid]; w.qacc[tid] += dq; st = fabsf(dq) / (1.f + fabsf(w.qacc[tid])); }
    st = wave_max(st);
    __syncthreads();
    if (st < 1e-7f) { it++; break; }
  }
  // forces at the final acceleration
  dev_eval_rows(m, w, w.qacc);
  dev_jt_force(m, w, w.fcon);
  if (tid == 0) w.solver_iter = it;
  __syncthreads();
}
