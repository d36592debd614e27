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

HD float hs_rcp(float d) {
  float inv = __builtin_amdgcn_rcpf(d);
  return inv * (2.f - d * inv);                    // one Newton step: full f32 accuracy without the IEEE divide
}
// two pivots per MFMA (K = 2): pivot J on the low half of the wave, pivot J+1 on the high half; row J+1 is
// brought up to date on the VALU while nothing else is in flight, so the serial chain per pivot pair is one MFMA.
template <int J> struct HsFactor {
  static HD void run(f32x16& acc, float (&lcol)[32], float& dinv, int col, int hi) {
    const float u0 = hs_row_bcast<J>(acc);           // u0[lane&31] = A[J][.] = L[.][J] * d_J
    const float r1 = hs_row_bcast<J + 1>(acc);
    const float inv0 = hs_rcp(fmaxf(rl(u0, J), 1e-30f));
    const float l0 = u0 * inv0;
    const float u1 = r1 - rl(l0, J + 1) * u0;        // row J+1 after eliminating pivot J
    const float inv1 = hs_rcp(fmaxf(rl(u1, J + 1), 1e-30f));
    const float l1 = u1 * inv1;
    if (col == J) dinv = inv0;
    if (col == J + 1) dinv = inv1;
    lcol[J] = l0; lcol[J + 1] = l1;
    if (J < 30) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? -l1 : -l0, hi ? u1 : u0, acc, 0, 0, 0);
    HsFactor<J + 2>::run(acc, lcol, dinv, col, hi);
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
    float Sc[6];
#pragma unroll
    for (int i = 0; i < 6; i++) Sc[i] = w.S[col][i];
    for (int c = 0; c < w.ncon; c++) {
      const int nr = w.c_nrow[c], r0 = w.c_row0[c];
      const float sg = (float)((w.c_mpos[c] >> col) & 1u) - (float)((w.c_mneg[c] >> col) & 1u);
      const float* Wn = w.c_W[c][0];
      for (int p = 0; 2 * p < nr; p++) {       // edges 2p (low half of the wave) and 2p+1 (high half)
        const float cu0 = w.r_curv[r0 + 2 * p], cu1 = (2 * p + 1 < nr) ? w.r_curv[r0 + 2 * p + 1] : 0.f;
        if (cu0 == 0.f && cu1 == 0.f) continue;
        const float* Wt = w.c_W[c][1 + p];
        const float sm = (nr > 1) ? (hi ? -w.c_mu[c][p] : w.c_mu[c][p]) : 0.f;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) v += Sc[i] * (Wn[i] + sm * Wt[i]);
        v *= sg;
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
  float* T = w.sc.T;
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

// ---- u[c][k] = (contact-frame Jacobian row k of contact c) . x, Jacobian-free:
// body spatial velocities V_b = sum_{d on the path of b} S[d] x[d], then W[c][k] . (V_b2 - V_b1)
__device__ void dev_basis_dot(const DevModel& m, const LaneK& lk, Work& w, const float* x) {
  const int tid = threadIdx.x;
  if (tid < m.nbody) {
    float V[6] = {0, 0, 0, 0, 0, 0};
    unsigned mk = lk.b_mask;
    while (mk) {
      const int d = __ffs(mk) - 1;
      mk &= mk - 1;
      const float xd = x[d];
#pragma unroll
      for (int i = 0; i < 6; i++) V[i] += w.S[d][i] * xd;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) w.bV[tid][i] = V[i];
  }
  __syncthreads();
  const int nb = w.ncon * NBASIS;
  for (int t = tid; t < nb; t += NT) {
    const int c = t / NBASIS;
    const int b1 = w.c_b1[c], b2 = w.c_b2[c];
    const float* W = w.c_W[c][t % NBASIS];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) s += W[i] * (w.bV[b2][i] - w.bV[b1][i]);
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
// (rows r < nv are always evaluated by lane r, so the friction-loss constants come from the lane's registers)
HD float dev_row_cost(const DevModel& m, const LaneK& lk, const Work& w, int r, float jar, float& force, float& curv) {
  if (r < m.nv) {
    const float f = lk.d_floss, R = lk.d_flR, D = 1.f / R;
    if (jar <= -R * f) { force = f; curv = 0.f; return -f * (0.5f * R * f + jar); }
    if (jar >= R * f) { force = -f; curv = 0.f; return -f * (0.5f * R * f - jar); }
    force = -D * jar; curv = D; return 0.5f * D * jar * jar;
  }
  const float D = (r < m.nv + w.nlim) ? w.lim_D[r - m.nv] : w.c_D[w.r_con[r]];
  if (jar < 0.f) { force = -D * jar; curv = D; return 0.5f * D * jar * jar; }
  force = 0.f; curv = 0.f; return 0.f;
}

// ---- constraint rows for the current kinematics / contacts
__device__ void dev_make_constraint(const DevModel& m, const LaneK& lk, Work& w, const float* qpos, const float* qvel) {
  const int tid = threadIdx.x;
  // joint limits (one side per joint can be active: every range is wider than twice the margin)
  {
    bool act = false; float dist = 0.f, sgn = 0.f;
    if (lk.j_limited && lk.j_type != HOIC_JNT_FREE) {
      const float q = qpos[lk.j_qadr], dl = q - lk.j_lo, du = lk.j_hi - q;
      if (dl < lk.j_margin) { act = true; dist = dl; sgn = 1.f; }
      else if (du < lk.j_margin) { act = true; dist = du; sgn = -1.f; }
    }
    const unsigned long long mask = __ballot(act);
    const int idx = __popcll(mask & ((1ull << tid) - 1ull));
    if (tid == 0) w.nlim = min(__popcll(mask), MAXLIM);
    if (act && idx < MAXLIM) {
      const int d = lk.j_dadr;
      const float imp = dev_impedance(lk.j_solimp, dist, lk.j_margin);
      const float R = fmaxf(MINVALF, (1.f - imp) * lk.j_diag / imp);
      w.lim_dof[idx] = d; w.lim_sign[idx] = sgn; w.lim_D[idx] = 1.f / R;
      w.r_aref[m.nv + idx] = -lk.j_B * (sgn * qvel[d]) - lk.j_K * imp * (dist - lk.j_margin);
    }
  }
  if (tid < m.nv) w.r_aref[tid] = -lk.d_flB * qvel[tid];
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
      w.c_b1[c] = m.pair_b1[p]; w.c_b2[c] = m.pair_b2[p];
    }
    const int incl_sum = wave_incl_scan(nrow);
    const int row0 = m.nv + w.nlim + incl_sum - nrow;
    if (c < w.ncon) {
      w.c_row0[c] = row0;
      for (int e = 0; e < nrow; e++) { w.r_con[row0 + e] = (unsigned char)c; w.r_edge[row0 + e] = (unsigned char)e; }
    }
    const int total = __builtin_amdgcn_readlane(incl_sum, NT - 1);
    if (tid == 0) w.nrow = m.nv + w.nlim + total;
  }
  __syncthreads();
  // wrench basis of every contact (lane = contact)
  if (tid < w.ncon) {
    const int c = tid, b1 = w.c_b1[c], b2 = w.c_b2[c];
    const unsigned m1 = w.k_bmask[b1], m2 = w.k_bmask[b2];
    w.c_mpos[c] = m2 & ~m1; w.c_mneg[c] = m1 & ~m2;
    const float* f = w.c_frame[c];
    for (int k = 0; k < 3; k++) {
      float pxf[3];
      cross3(w.c_pos[c], f + 3 * k, pxf);
      for (int i = 0; i < 3; i++) { w.c_W[c][k][i] = pxf[i]; w.c_W[c][k][3 + i] = f[3 * k + i]; }
    }
    for (int i = 0; i < 3; i++) { w.c_W[c][3][i] = f[i]; w.c_W[c][3][3 + i] = 0.f; }
  }
  __syncthreads();
  // reference accelerations of the contact rows
  dev_basis_dot(m, lk, w, qvel);
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
__device__ float dev_eval_rows(const DevModel& m, const LaneK& lk, Work& w, const float* x) {
  dev_basis_dot(m, lk, w, x);
  float cost = 0.f;
  for (int r = threadIdx.x; r < w.nrow; r += NT) {
    const float jar = dev_row_times(m, w, r, x) - w.r_aref[r];
    float f, cv;
    cost += dev_row_cost(m, lk, w, r, jar, f, cv);
    w.r_jar[r] = jar; w.r_force[r] = f; w.r_curv[r] = cv;
  }
  __syncthreads();
  return wave_sum(cost);
}

// out[i] = (J^T force)[i] from r_force: per-contact wrench G_c = sum_k g_k W[c][k], then S[i] . sum_c sg(i,c) G_c
__device__ void dev_jt_force(const DevModel& m, Work& w, float* out) {
  const int tid = threadIdx.x;
  if (tid < w.ncon) {
    const int c = tid, r0 = w.c_row0[c], nr = w.c_nrow[c];
    float g[NBASIS] = {0.f, 0.f, 0.f, 0.f};
    for (int e = 0; e < nr; e++) g[0] += w.r_force[r0 + e];
    if (nr > 1) for (int k = 1; 2 * k - 1 < nr; k++) g[k] = w.c_mu[c][k - 1] * (w.r_force[r0 + 2 * (k - 1)] - w.r_force[r0 + 2 * (k - 1) + 1]);
    for (int i = 0; i < 6; i++) {
      float s = 0.f;
      for (int k = 0; k < NBASIS; k++) s += g[k] * w.c_W[c][k][i];
      w.c_G[c][i] = s;
    }
  }
  __syncthreads();
  if (tid < m.nv) {
    float s = w.r_force[tid];
    for (int l = 0; l < w.nlim; l++) if (w.lim_dof[l] == tid) s += w.lim_sign[l] * w.r_force[m.nv + l];
    float G[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < w.ncon; c++) {
      const float sg = (float)((w.c_mpos[c] >> tid) & 1u) - (float)((w.c_mneg[c] >> tid) & 1u);
      if (sg != 0.f) for (int i = 0; i < 6; i++) G[i] += sg * w.c_G[c][i];
    }
    for (int i = 0; i < 6; i++) s += w.S[tid][i] * G[i];
    out[tid] = s;
  }
  __syncthreads();
}

// ---- Newton with exact line search.  In: M, fsmooth, asmooth, warm, rows.  Out: qacc, fcon.
__device__ void dev_solve(const DevModel& m, const LaneK& lk, Work& w, int maxit) {
  const int tid = threadIdx.x;
  const float scale = 1.f / (m.meaninertia * (float)max(m.nv, 1));
  // warm start choice: cost(warm) vs cost(asmooth); the row state of the LAST evaluation (warm) is reused by
  // the first iteration when warm wins (the usual case)
  float gw = 0.f;
  if (tid < m.nv) { const float Ma = dev_Mrow(w, tid, w.warm); gw = 0.5f * (Ma - w.fsmooth[tid]) * (w.warm[tid] - w.asmooth[tid]); }
  gw = wave_sum(gw);
  const float cs = dev_eval_rows(m, lk, w, w.asmooth);
  const float cw = gw + dev_eval_rows(m, lk, w, w.warm);
  bool have_eval = cw < cs;
  if (tid < NV) w.qacc[tid] = (tid < m.nv) ? (have_eval ? w.warm[tid] : w.asmooth[tid]) : 0.f;
  __syncthreads();
  int it = 0;
  bool fresh = false;     // w.tv holds J'f of the current qacc
  for (; it < maxit; it++) {
    if (tid < m.nv) w.Ma[tid] = dev_Mrow(w, tid, w.qacc);
    if (!have_eval) dev_eval_rows(m, lk, w, w.qacc);
    have_eval = false;
    dev_jt_force(m, w, w.tv);
    float g2 = 0.f;
    if (tid < NV) {
      const float g = (tid < m.nv) ? (w.Ma[tid] - w.fsmooth[tid] - w.tv[tid]) : 0.f;
      w.grad[tid] = g; w.search[tid] = -g; g2 = g * g;
    }
    g2 = wave_sum(g2);
    __syncthreads();
    if (sqrtf(g2) * scale < 1e-6f) { fresh = true; break; }
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
    dev_basis_dot(m, lk, w, w.search);
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
          dev_row_cost(m, lk, w, r, jar[k] + a * jv[k], f, cv);
          dphi -= f * jv[k]; ddphi += cv * jv[k] * jv[k];
        }
      }
      dphi = wave_sum(dphi) + gq + a * hh; ddphi = wave_sum(ddphi) + hh;
      alpha = a;
      if (fabsf(dphi) < 1e-4f * fabsf(g0) + 1e-12f) break;
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
  if (fresh) { if (tid < NV) w.fcon[tid] = w.tv[tid]; }
  else { dev_eval_rows(m, lk, w, w.qacc); dev_jt_force(m, w, w.fcon); }
  if (tid == 0) w.solver_iter = it;
  __syncthreads();
}
