// hoic_solver.h — constraint rows and the convex constraint solve, one wavefront per env.
//
// Replaces mj_makeConstraint + the Newton solver inside self.sim.step() (uhc/envs/ho_im4.py:545).
// Problem (MuJoCo's primal form, unique optimum):
//     min_a  1/2 (a - a0)' M (a - a0) + sum_r s_r(J_r a - aref_r)
// rows: dof friction loss (Huber), joint limits and pyramidal contact edges (one-sided quadratics).
// MI355X mapping: nv = 32 unknowns = half a wavefront.  Friction-loss and limit rows are one-per-dof and live
// in the registers of lane = dof (RowK / RowEval); contact rows are never materialised: each contact keeps its
// frame and p x f_k in LDS and the pyramid edges are formed on the fly.  The 32x32 Hessian is assembled,
// factorised and solved in one MFMA accumulator.
#pragma once
#include "hoic_types.h"
#include "hoic_math.h"
#include "hoic_dynamics.h"

// ---- 32x32 SPD assemble + solve on the matrix core.
// The matrix lives in ONE v_mfma_f32_32x32x2_f32 accumulator (16 VGPRs per lane; element (row, col) sits in
// lane (col + 32*((row>>2)&1)), register (row&3) + 4*(row>>3)).  Exact f32 (an fma chain), so numerics equal
// a VALU version.
//   A = M (+ diag) restricted to the leading nact x nact block, identity elsewhere
//   (+ sum over active contact rows  curv_r J_r' J_r : two rank-1 terms per MFMA, K = 2)
// Factorisation: right-looking LDL^T, two pivots per rank-2 MFMA.  Row j of the running matrix (= column j of L
// times d_j) is one value per lane, so the scaled column is also the MFMA A-operand.  The forward substitution
// rides along in the shadow of the MFMA latency; each column of L goes to the LDS scratch T as it is produced
// (one ds_write per pivot pair), and the backward substitution reads it back transposed.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

template <int LANE> HD void hs_writelane(float& dst, float uniform_val) {     // uniform_val must be wave-uniform (SGPR)
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(dst) : "s"(uniform_val), "n"(LANE));
}
// 1 / max(d, 1e-30) of a wave-uniform pivot: the clamp runs on the scalar unit (positive floats order like integers,
// negative ones are negative integers), the reciprocal is one VALU op
HD float hs_pivot_ninv(float d_uniform, float& d_clamped) {
  const int di = max(__float_as_int(d_uniform), 0x0DA24260);      // bits of 1e-30f
  d_clamped = __int_as_float(di);
  return -__builtin_amdgcn_rcpf(d_clamped);
}
// One pivot pair (J, J+1), J even.  Rows J and J+1 of the running matrix sit in two registers of the same half-wave: one
// permlane32_swap puts row J on the low lanes and row J+1 on the high lanes, a second one spreads each over both
// halves.  nl = -L[.][J] doubles as the MFMA A operand, the forward-substitution multiplier and the value stored in
// T.  No lane masks anywhere: entries of nl at lanes of pivots eliminated earlier are rounding residue and only touch entries
// of y that have already been extracted (y_J goes to lane J of yv, the pivot d_J to lane J of dv).
// prep: everything but the MFMA; Aop / Bop are its operands.
// ys: the running right-hand side the pair's own entries are read from -- the state at the start of the step (the pairs of a
// step do not touch each other's entries, so reading the snapshot takes the pairs' updates of y off each other's chains)
template <int J> HD void hs_pair_prep(const f32x16& acc, const float ys, float& y, float& yv, float& dv, int hi, float* Tcol, float& Aop, float& Bop) {
  constexpr int reg0 = (J & 3) + 4 * (J >> 3), reg1 = ((J + 1) & 3) + 4 * ((J + 1) >> 3), half = (J >> 2) & 1;
  const u32x2v p = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[reg0]), __float_as_uint(acc[reg1]), false, false);
  const unsigned q = half ? p.y : p.x;                      // low lanes: row J, high lanes: row J+1
  const u32x2v sp = __builtin_amdgcn_permlane32_swap(q, q, false, false);
  const float u0 = __uint_as_float(sp.x), r1 = __uint_as_float(sp.y);   // u0[c] = A[J][c] = L[c][J] d_J
  float d0, d1;
  const float ninv0 = hs_pivot_ninv(rl(u0, J), d0);
  const float nl0 = u0 * ninv0;
  const float u1 = fmaf(rl(nl0, J + 1), u0, r1);            // row J+1 after eliminating pivot J
  const float ninv1 = hs_pivot_ninv(rl(u1, J + 1), d1);
  const float nl1 = u1 * ninv1;
  Aop = hi ? nl1 : nl0; Bop = hi ? u1 : u0;
  Tcol[J * LD] = Aop;                                       // T[J + hi][c] = -L[c][J + hi]
  const float y0 = rl(ys, J);
  const float y1 = rl(fmaf(nl0, y0, ys), J + 1);
  y = fmaf(nl0, y0, y);
  y = fmaf(nl1, y1, y);
  hs_writelane<J>(yv, y0); hs_writelane<J + 1>(yv, y1);
  hs_writelane<J>(dv, d0); hs_writelane<J + 1>(dv, d1);
}
// Elimination order.  The joint-space inertia of the HOIC models is sparse by its tree: the five fingers (4 dofs each, dofs
// 6 + 4 f .. 9 + 4 f) couple only through the palm (dofs 0..5), the free object (26..31) couples to nothing -- and so is
// every matrix the solves see as long as no contact row joins two of those groups (diagonal shifts, contacts of the object
// with the table / floor).  Eliminating the fingers BEFORE the palm keeps that sparsity (no fill between fingers), so the
// pairs of one step below never touch each other's rows: their rows are read from ONE state of the accumulator, their VALU
// chains interleave and their rank-2 MFMAs issue back to back -- 5 dependent steps (6 + 6 + 2 + 1 + 1 MFMAs) instead of 16.
// The same order is a valid LDL^T for any SPD matrix: when contact rows couple the groups (SER) every pair reads the
// accumulator after the previous pair's MFMA.  build_model checks that the model has this structure.
//   step:        0                      1                      2        3    4
#define HS_STEPS(X) X(0, 8, 12, 16, 20, 24, 26) X(1, 6, 10, 14, 18, 22, 28)
template <int J, bool LASTPAIR> HD void hs_pair_ser(f32x16& acc, float& y, float& yv, float& dv, int hi, float* Tcol) {
  float A, B;
  const float ys = y;
  hs_pair_prep<J>(acc, ys, y, yv, dv, hi, Tcol, A, B);
  if (!LASTPAIR) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, acc, 0, 0, 0);
}
template <int J0, int J1, int J2, int J3, int J4, int J5> HD void hs_step6_par(f32x16& acc, float& y, float& yv, float& dv, int hi, float* Tcol) {
  float A0, B0, A1, B1, A2, B2, A3, B3, A4, B4, A5, B5;
  const float ys = y;
  hs_pair_prep<J0>(acc, ys, y, yv, dv, hi, Tcol, A0, B0); hs_pair_prep<J1>(acc, ys, y, yv, dv, hi, Tcol, A1, B1);
  hs_pair_prep<J2>(acc, ys, y, yv, dv, hi, Tcol, A2, B2); hs_pair_prep<J3>(acc, ys, y, yv, dv, hi, Tcol, A3, B3);
  hs_pair_prep<J4>(acc, ys, y, yv, dv, hi, Tcol, A4, B4); hs_pair_prep<J5>(acc, ys, y, yv, dv, hi, Tcol, A5, B5);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A2, B2, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A3, B3, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A4, B4, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A5, B5, acc, 0, 0, 0);
}
template <bool PAR> HD void hs_factor(f32x16& acc, float& y, float& yv, float& dv, int hi, float* Tcol) {
  if (PAR) {
    hs_step6_par<8, 12, 16, 20, 24, 26>(acc, y, yv, dv, hi, Tcol);
    hs_step6_par<6, 10, 14, 18, 22, 28>(acc, y, yv, dv, hi, Tcol);
    {   // palm pair 0 and the object's last pair: independent of each other
      float A0, B0, A1, B1;
      const float ys = y;
      hs_pair_prep<0>(acc, ys, y, yv, dv, hi, Tcol, A0, B0); hs_pair_prep<30>(acc, ys, y, yv, dv, hi, Tcol, A1, B1);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, acc, 0, 0, 0);      // (the object's last pair needs no update: nothing is left below it)
      (void)A1; (void)B1;
    }
  } else {
    hs_pair_ser<8, false>(acc, y, yv, dv, hi, Tcol); hs_pair_ser<12, false>(acc, y, yv, dv, hi, Tcol); hs_pair_ser<16, false>(acc, y, yv, dv, hi, Tcol);
    hs_pair_ser<20, false>(acc, y, yv, dv, hi, Tcol); hs_pair_ser<24, false>(acc, y, yv, dv, hi, Tcol); hs_pair_ser<26, false>(acc, y, yv, dv, hi, Tcol);
    hs_pair_ser<6, false>(acc, y, yv, dv, hi, Tcol); hs_pair_ser<10, false>(acc, y, yv, dv, hi, Tcol); hs_pair_ser<14, false>(acc, y, yv, dv, hi, Tcol);
    hs_pair_ser<18, false>(acc, y, yv, dv, hi, Tcol); hs_pair_ser<22, false>(acc, y, yv, dv, hi, Tcol); hs_pair_ser<28, false>(acc, y, yv, dv, hi, Tcol);
    hs_pair_ser<0, false>(acc, y, yv, dv, hi, Tcol); hs_pair_ser<30, false>(acc, y, yv, dv, hi, Tcol);
  }
  hs_pair_ser<2, false>(acc, y, yv, dv, hi, Tcol);
  hs_pair_ser<4, true>(acc, y, yv, dv, hi, Tcol);
}
// elimination sequence of the single pivots (the order of HS above); the backward substitution runs through it in reverse
__device__ constexpr int HS_SEQ[32] = {8, 9, 12, 13, 16, 17, 20, 21, 24, 25, 26, 27, 6, 7, 10, 11, 14, 15, 18, 19, 22, 23, 28, 29, 0, 1, 30, 31, 2, 3, 4, 5};
// backward substitution over HS_SEQ[P], P = 31 .. 0, sixteen columns of L in flight at a time (lc[i] = -L[HS_SEQ[16 BATCH + i]][col]);
// x_k goes to lane k of xv as soon as it is final.  Residue of lc at columns eliminated AFTER k only touches entries that
// are final already.
template <int P, int BATCH> struct HsBack {
  static HD void run(const float (&lc)[16], float& x, float& xv) {
    constexpr int k = HS_SEQ[P];
    const float xk = rl(x, k);
    hs_writelane<k>(xv, xk);
    if constexpr (P > 0) x = fmaf(lc[P - 16 * BATCH], xk, x);
    if constexpr (P > 16 * BATCH) HsBack<P - 1, BATCH>::run(lc, x, xv);
  }
};

// diag: per-lane diagonal increment of row/col (lane & 31); use_rows: add the active contact rows through the
// MFMA; rhs: per-lane right-hand side (lane & 31).  Returns x[lane & 31] on the low half-wave lanes AND the high ones.
template <bool ROWS>
__device__ __forceinline__ float dev_hsolve(const DevModel& m, Work& w, const MReg& M, float dg, int nact, float rhs) {
  constexpr bool use_rows = ROWS;
  const int lane = opaque(threadIdx.x), col = lane & 31, hi = lane >> 5;
  f32x16 acc;
  // Diagonal shift.  Lane (col, hi) holds the diagonal entry (col, col) in register (col & 3) + 4 (col >> 3) if its half-wave
  // owns row col, i.e. ((col >> 2) & 1) == hi: register 4 g + q has its two diagonal lanes in the 4-lane banks
  // {row g >> 1, bank 2 (g & 1)} and {row 2 + (g >> 1), bank 2 (g & 1) + 1} of the wave.  Sixteen compare-select-add triples
  // become four selects -- x_q = dg on the lanes whose diagonal register has (reg & 3) == q, 0 elsewhere -- and sixteen DPP
  // adds whose row / bank masks enable exactly those banks (the other enabled lanes add x_q = 0).
  {
    float a[16];
#pragma unroll
    for (int reg = 0; reg < 16; reg++) a[reg] = M.r[reg];
    if (!(__builtin_constant_p(dg) && dg == 0.f)) {
      const int qsel = (((col >> 2) & 1) == hi) ? (col & 3) : -1;
      float x0 = qsel == 0 ? dg : 0.f, x1 = qsel == 1 ? dg : 0.f, x2 = qsel == 2 ? dg : 0.f, x3 = qsel == 3 ? dg : 0.f;
#define HS_DIAG(r, x, rm, bm) asm("v_add_f32_dpp %0, %1, %0 quad_perm:[0,1,2,3] row_mask:" rm " bank_mask:" bm : "+v"(a[r]) : "v"(x))
      // (a DPP operand needs two wait states after the VALU instruction that wrote it: the selects above)
      asm volatile("s_nop 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
      HS_DIAG(0, x0, "0x5", "0x3"); HS_DIAG(1, x1, "0x5", "0x3"); HS_DIAG(2, x2, "0x5", "0x3"); HS_DIAG(3, x3, "0x5", "0x3");
      HS_DIAG(4, x0, "0x5", "0xc"); HS_DIAG(5, x1, "0x5", "0xc"); HS_DIAG(6, x2, "0x5", "0xc"); HS_DIAG(7, x3, "0x5", "0xc");
      HS_DIAG(8, x0, "0xa", "0x3"); HS_DIAG(9, x1, "0xa", "0x3"); HS_DIAG(10, x2, "0xa", "0x3"); HS_DIAG(11, x3, "0xa", "0x3");
      HS_DIAG(12, x0, "0xa", "0xc"); HS_DIAG(13, x1, "0xa", "0xc"); HS_DIAG(14, x2, "0xa", "0xc"); HS_DIAG(15, x3, "0xa", "0xc");
#undef HS_DIAG
    }
#pragma unroll
    for (int reg = 0; reg < 16; reg++) acc[reg] = a[reg];
  }
  if (nact < NV) {           // leading nact x nact block, identity elsewhere (the PD solve on the hand dofs)
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int r = MREG_ROW(reg, hi);
      if (r >= nact || col >= nact) acc[reg] = (r == col) ? 1.f : 0.f;
    }
  }
  bool coupled = false;      // some contact joins two of the groups (fingers, palm, object): the pairs of a step are not independent
  if (use_rows) {
    float Sc[6];
#pragma unroll
    for (int i = 0; i < 6; i++) Sc[i] = w.S[col][i];
    const unsigned objmask = 0xFC000000u;       // dofs 26..31 (checked by build_model)
    for (int c = 0; c < w.ncon; c++) {
      coupled = coupled || (((w.c_mpos[c] | w.c_mneg[c]) & ~objmask) != 0u);
      const int nr = w.c_nrow[c], r0 = w.c_row0[c];
      const float sg = (float)((w.c_mpos[c] >> col) & 1u) - (float)((w.c_mneg[c] >> col) & 1u);
      const float* fr = w.c_frame[c];
      // velocity of the contact point per unit velocity of this lane's dof: S_lin + S_ang x p; a row is f_k . that
      float wl[3];
      cross3(Sc, w.c_pos[c], wl);
      wl[0] += Sc[3]; wl[1] += Sc[4]; wl[2] += Sc[5];
      const float vn = dot3(wl, fr);
      for (int p = 0; 2 * p < nr; p++) {       // edges 2p (low half of the wave) and 2p+1 (high half)
        const float cu0 = w.cr_curv[r0 + 2 * p], cu1 = (2 * p + 1 < nr) ? w.cr_curv[r0 + 2 * p + 1] : 0.f;
        if (cu0 == 0.f && cu1 == 0.f) continue;
        float v = vn;
        if (nr > 1) {
          const float vt = (p < 2) ? dot3(wl, fr + 3 * (1 + p)) : dot3(Sc, fr);
          v += (hi ? -w.c_mu[c][p] : w.c_mu[c][p]) * vt;
        }
        v *= sg;
        const float cu = hi ? cu1 : cu0;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cu * v, v, acc, 0, 0, 0);
      }
    }
  }
  PT(15);
  float y = rhs, yv = 0.f, dv = 1.f;
  float* T = w.sc.T;
  if (use_rows && coupled) hs_factor<false>(acc, y, yv, dv, hi, T + hi * LD + col);
  else hs_factor<true>(acc, y, yv, dv, hi, T + hi * LD + col);
  PT(16);
  float x = yv * __builtin_amdgcn_rcpf(dv);       // D^-1 L^-1 rhs
  wsync();
  float xv = 0.f;
  {
    float lc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) lc[k] = T[col * LD + HS_SEQ[16 + k]];        // -L[HS_SEQ[16+k]][col]
    HsBack<31, 1>::run(lc, x, xv);
#pragma unroll
    for (int k = 0; k < 16; k++) lc[k] = T[col * LD + HS_SEQ[k]];
    HsBack<15, 0>::run(lc, x, xv);
  }
  wsync();
  PT(18);
  // lanes 0..31 of xv hold the solution; the high half-wave gets a copy
  const unsigned xb = __float_as_uint(xv);
  const u32x2v xs = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
  return __uint_as_float(xs.x);
}

// impedance d(r) from solimp [MJ-doc: getimpedance]
HD float dev_impedance(const float* s_in, float pos, float margin) {
  float s0 = fminf(fmaxf(s_in[0], 0.0001f), 0.9999f), s1 = fminf(fmaxf(s_in[1], 0.0001f), 0.9999f);
  float wdt = fmaxf(s_in[2], 0.f), mid = fminf(fmaxf(s_in[3], 0.0001f), 0.9999f), pw = fmaxf(s_in[4], 1.f);
  if (s0 == s1 || wdt <= MINVALF) return 0.5f * (s0 + s1);
  float x = fabsf(fdiv(pos - margin, wdt));
  if (x >= 1.f) return s1;
  if (x <= 0.f) return s0;
  // both branches of the power sigmoid are base^pw / bm^(pw-1) of a mirrored argument; base and bm lie in (0, 1],
  // so the hardware log2/exp2 (1 ulp) replace the ~300-instruction library powf
  const bool lo = x <= mid;
  const float base = lo ? x : 1.f - x, bm = lo ? mid : 1.f - mid;
  float r;
  if (pw == 1.f) r = base;
  else if (pw == 2.f) r = base * base * __builtin_amdgcn_rcpf(bm);
  else r = __builtin_amdgcn_exp2f(pw * __builtin_amdgcn_logf(base) - (pw - 1.f) * __builtin_amdgcn_logf(bm));
  const float y = lo ? r : 1.f - r;
  return s0 + y * (s1 - s0);
}

// rows owned by lane & 31 = dof: friction loss (always) and the joint limit of the dof's joint (sign 0: inactive)
struct RowK { float f_aref, l_sign, l_D, l_aref; };
// state of the last row evaluation: per-dof rows in registers, contact rows NCSLOT per lane (row lane + 64 k)
struct RowEval { float jar_f, force_f, curv_f, jar_l, force_l, curv_l, jar_c[NCSLOT]; };

// ---- u[c][k] = (contact-frame Jacobian row k of contact c) . x, Jacobian-free:
// body spatial velocities V_b = sum_{d on the path of b} S[d] x[d], then W[c][k] . (V_b2 - V_b1)
__device__ __forceinline__ void dev_basis_dot(const DevModel& m, Work& w, const float* x) {
  const int tid = opaque(threadIdx.x);
  if (w.ncon == 0) return;
  // only the bodies that take part in a contact (typically the object on the table: ONE body with a path of six dofs; the
  // gather over all 28 bodies' 12-dof paths was 15 % of the kernel), and the second half of the paths only when one of them
  // is a finger body (dev_make_constraint leaves both facts in w.cbod)
  const unsigned cb = w.cbod;
  if (tid < m.nbody && ((cb >> tid) & 1u)) {
    float V[6] = {0, 0, 0, 0, 0, 0};
    const unsigned bp[3] = {w.k_bpath[tid][0], w.k_bpath[tid][1], w.k_bpath[tid][2]};
    path_gather<false>(w, bp, x, nullptr, 0xFF, V, nullptr, (cb >> 31) != 0u, m.max_path > 10);
#pragma unroll
    for (int i = 0; i < 6; i++) w.bV[tid][i] = V[i];
  }
  wsync();
  const int nb = w.ncon * 4;
  for (int t = tid; t < nb; t += NT) {
    const int c = t >> 2, k = t & 3;
    const int b1 = w.c_b1[c], b2 = w.c_b2[c];
    float dV[6], vp[3];
#pragma unroll
    for (int i = 0; i < 6; i++) dV[i] = w.bV[b2][i] - w.bV[b1][i];
    cross3(dV, w.c_pos[c], vp);                 // relative velocity of the two bodies at the contact point: v + w x p
    vp[0] += dV[3]; vp[1] += dV[4]; vp[2] += dV[5];
    w.u[t] = (k < 3) ? dot3(w.c_frame[c] + 3 * k, vp) : dot3(w.c_frame[c], dV);
  }
  wsync();
}

// J_r . x for contact row r (u must hold dev_basis_dot(x))
HD float dev_crow_times(const Work& w, const float* u, int r) {
  const int ce = w.cr_ce[r], c = ce >> 3, e = ce & 7;
  const float un = u[c * 4];
  if (w.c_nrow[c] == 1) return un;
  const int k = e >> 1;
  return un + ((e & 1) ? -1.f : 1.f) * w.c_mu[c][k] * u[c * 4 + 1 + k];
}
HD float dev_crow_times(const Work& w, int r) { return dev_crow_times(w, w.u, r); }

// ---- the three Jacobian products of the solve's set-up in ONE pass: u[v][c][k] = (contact-frame Jacobian row k of contact c) .
// x_v for x = (qvel, a_smooth, warm start).  Separately (dev_basis_dot once in dev_make_constraint and twice in the warm-start
// choice) each pass paid its own path unpacking, reads of S, contact-frame reads and two hand-over points; per vector the
// arithmetic is that of dev_basis_dot, in the same order (bit-identical products).  Buffers: sc.mv (the solves' scratch, dead here).
__device__ __forceinline__ void dev_basis_dot3(const DevModel& m, Work& w, const float* x0, const float* x1, const float* x2) {
  const int tid = opaque(threadIdx.x);
  if (w.ncon == 0) return;
  const unsigned cb = w.cbod;
  if (tid < m.nbody && ((cb >> tid) & 1u)) {
    float V[3][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
    const unsigned bp[3] = {w.k_bpath[tid][0], w.k_bpath[tid][1], w.k_bpath[tid][2]};
    const float* const xs[3] = {x0, x1, x2};
    path_gather_multi<3>(w, bp, xs, V, (cb >> 31) != 0u, m.max_path > 10);
#pragma unroll
    for (int v = 0; v < 3; v++)
#pragma unroll
      for (int i = 0; i < 6; i++) w.sc.mv.bV[v][tid][i] = V[v][i];
  }
  wsync();
  const int nb = w.ncon * 4;
  for (int t = tid; t < nb; t += NT) {
    const int c = t >> 2, k = t & 3;
    const int b1 = w.c_b1[c], b2 = w.c_b2[c];
    float fr[3], fn[3], cp[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { fr[i] = w.c_frame[c][3 * (k < 3 ? k : 0) + i]; fn[i] = w.c_frame[c][i]; cp[i] = w.c_pos[c][i]; }
#pragma unroll
    for (int v = 0; v < 3; v++) {
      float dV[6], vp[3];
#pragma unroll
      for (int i = 0; i < 6; i++) dV[i] = w.sc.mv.bV[v][b2][i] - w.sc.mv.bV[v][b1][i];
      cross3(dV, cp, vp);
      vp[0] += dV[3]; vp[1] += dV[4]; vp[2] += dV[5];
      w.sc.mv.u[v][t] = (k < 3) ? dot3(fr, vp) : dot3(fn, dV);
    }
  }
  wsync();
}

// row cost pieces: return the cost, set force = -ds/djar and the curvature
HD float cost_friction(const DofK& dk, float jar, float& force, float& curv) {
  const float f = dk.floss, R = dk.flR;
  if (jar <= -R * f) { force = f; curv = 0.f; return -f * (0.5f * R * f + jar); }
  if (jar >= R * f) { force = -f; curv = 0.f; return -f * (0.5f * R * f - jar); }
  const float D = frcp(R);
  force = -D * jar; curv = D; return 0.5f * D * jar * jar;
}
HD float cost_onesided(float D, float jar, float& force, float& curv) {
  if (jar < 0.f) { force = -D * jar; curv = D; return 0.5f * D * jar * jar; }
  force = 0.f; curv = 0.f; return 0.f;
}

// ---- constraint rows for the current kinematics / contacts
__device__ __forceinline__ void dev_make_constraint(const DevModel& m, Work& w, RowK& rk, DofK& dk, const float* qpos, const float* qvel) {
  const int tid = opaque(threadIdx.x), d = tid & 31;
  // (the friction-loss constants of dev_solve are fetched with this stage's per-dof constants: one global-read latency for both)
  dk.floss = d < m.nv ? m.dof_frictionloss[d] : 0.f; dk.flR = d < m.nv ? m.dof_flR[d] : 1.f;
  // friction loss and joint limit of dof d (one side per joint can be active: every range is wider than twice
  // the margin; slide and hinge joints have exactly one dof)
  rk.f_aref = 0.f; rk.l_sign = 0.f; rk.l_D = 0.f; rk.l_aref = 0.f;
  if (d < m.nv) {
    const float qv = qvel[d];
    rk.f_aref = -m.dof_flB[d] * qv;
    if (m.dof_limited[d]) {
      const float q = qpos[m.dof_qadr[d]], margin = m.dof_margin[d];
      const float dl = q - m.dof_range[d][0], du = m.dof_range[d][1] - q;
      float dist = 0.f, sgn = 0.f;
      if (dl < margin) { dist = dl; sgn = 1.f; }
      else if (du < margin) { dist = du; sgn = -1.f; }
      if (sgn != 0.f) {
        const float si[5] = {m.dof_solimp[d][0], m.dof_solimp[d][1], m.dof_solimp[d][2], m.dof_solimp[d][3], m.dof_solimp[d][4]};
        const float imp = dev_impedance(si, dist, margin);
        const float R = fmaxf(MINVALF, fdiv((1.f - imp) * m.dof_limdiag[d], imp));
        rk.l_sign = sgn; rk.l_D = frcp(R);
        rk.l_aref = -m.dof_limB[d] * (sgn * qv) - m.dof_limK[d] * imp * (dist - margin);
      }
    }
  }
  // per-contact parameters and row layout (lane = contact)
  {
    int nrow = 0;
    const int c = tid;
    if (c < w.ncon) {
      const int p = w.c_pair[c], dim = m.pair_condim[p];
      nrow = dim == 1 ? 1 : 2 * (dim - 1);
      const float incl = m.pair_margin[p] - m.pair_gap[p];
      const float si[5] = {m.pair_solimp[p][0], m.pair_solimp[p][1], m.pair_solimp[p][2], m.pair_solimp[p][3], m.pair_solimp[p][4]};
      const float imp = dev_impedance(si, w.c_dist[c], incl);
      const float R = fmaxf(MINVALF, fdiv(1.f - imp, imp)) * m.pair_Rscale[p];
      w.c_D[c] = frcp(fmaxf(R, MINVALF));
      w.c_B[c] = m.pair_B[p];
      w.c_aref0[c] = -m.pair_K[p] * imp * (w.c_dist[c] - incl);
      for (int k = 0; k < 3; k++) w.c_mu[c][k] = m.pair_mu[p][k];
      w.c_nrow[c] = (unsigned char)nrow;
      const int b1 = m.pair_b1[p], b2 = m.pair_b2[p];
      w.c_b1[c] = (unsigned char)b1; w.c_b2[c] = (unsigned char)b2;
      w.c_mpos[c] = m.pair_mpos[p]; w.c_mneg[c] = m.pair_mneg[p];
    }
    {   // which bodies the Jacobian products have to visit (dev_basis_dot)
      unsigned bits = 0u;
      if (c < w.ncon) { const int p = w.c_pair[c]; bits = (1u << m.pair_b1[p]) | (1u << m.pair_b2[p]); }
      const unsigned cb = wave_or(bits) & 0x7FFFFFFFu;
      const bool lng = tid < m.nbody && ((cb >> tid) & 1u) && ((w.k_bpath[tid][1] >> 16) & 0xFFu) != 0xFFu;
      const unsigned cbl = cb | (__ballot(lng) != 0ull ? 0x80000000u : 0u);
      if (tid == 0) w.cbod = cbl;
    }
    const int incl_sum = wave_incl_scan(nrow);
    const int row0 = incl_sum - nrow;
    if (c < w.ncon) {
      w.c_row0[c] = (unsigned char)row0;
      for (int e = 0; e < nrow; e++) w.cr_ce[row0 + e] = (unsigned char)((c << 3) | e);     // (dev_collision cut the list so that the rows fit)
    }
    const int total = __builtin_amdgcn_readlane(incl_sum, NT - 1);
    if (tid == 0) w.nrow = total;
  }
  wsync();
  // (the reference accelerations of the contact rows need J . qvel: dev_solve forms them together with the Jacobian products of
  //  its warm-start choice, dev_rows_setup)
}

// ---- set-up of the solve: reference accelerations of the contact rows (aref_c: rows lane + 64 k, the mapping of
// RowEval::jar_c) and the row residuals jar = J x - aref at a_smooth (evs) and at the warm start (evw), the three Jacobian
// products in one pass (dev_basis_dot3)
__device__ __forceinline__ void dev_rows_setup(const DevModel& m, Work& w, const RowK& rk, float (&aref_c)[NCSLOT], const float* qvel, const float* xs,
                                               const float* xw, RowEval& evs, RowEval& evw) {
  const int tid = opaque(threadIdx.x);
  dev_basis_dot3(m, w, qvel, xs, xw);
  const bool vd = (tid & 31) < m.nv;
  const float as = vd ? xs[tid & 31] : 0.f, aw = vd ? xw[tid & 31] : 0.f;
  evs.jar_f = as - rk.f_aref; evs.jar_l = rk.l_sign * as - rk.l_aref;
  evw.jar_f = aw - rk.f_aref; evw.jar_l = rk.l_sign * aw - rk.l_aref;
  const int nrow = w.nrow;
#pragma unroll
  for (int k = 0; k < NCSLOT; k++) {
    aref_c[k] = 0.f; evs.jar_c[k] = 0.f; evw.jar_c[k] = 0.f;
    if (k * NT < nrow) {
      const int r = tid + k * NT;
      if (r < nrow) {
        const int c = w.cr_ce[r] >> 3;
        aref_c[k] = -w.c_B[c] * dev_crow_times(w, w.sc.mv.u[0], r) + w.c_aref0[c];
        evs.jar_c[k] = dev_crow_times(w, w.sc.mv.u[1], r) - aref_c[k];
        evw.jar_c[k] = dev_crow_times(w, w.sc.mv.u[2], r) - aref_c[k];
      }
    }
  }
  wsync();      // (the next writer of the solves' scratch -- the first Hessian solve -- comes after every lane has read u)
}

// ---- row state.  jar = J x - aref of every row: per-dof rows in the registers of lane & 31 = dof (both
// half-waves), contact rows NCSLOT per lane.  dev_rows_jar needs one Jacobian product (dev_basis_dot);
// dev_rows_cost turns jar into cost, forces and curvatures (contact rows: LDS cr_force / cr_curv) and is cheap.
__device__ __forceinline__ void dev_rows_jar(const DevModel& m, Work& w, const RowK& rk, const float (&aref_c)[NCSLOT], const float* x, bool with_aref, RowEval& ev) {
  const int tid = opaque(threadIdx.x);
  dev_basis_dot(m, w, x);
  const float xd = ((tid & 31) < m.nv) ? x[tid & 31] : 0.f;
  ev.jar_f = xd - (with_aref ? rk.f_aref : 0.f);
  ev.jar_l = rk.l_sign * xd - (with_aref ? rk.l_aref : 0.f);
  const int nrow = w.nrow;
#pragma unroll
  for (int k = 0; k < NCSLOT; k++) {
    ev.jar_c[k] = 0.f;
    if (k * NT < nrow) {
      const int r = tid + k * NT;
      if (r < nrow) ev.jar_c[k] = dev_crow_times(w, r) - (with_aref ? aref_c[k] : 0.f);
    }
  }
}
// D_c: curvature constant of this lane's contact rows (c_D of the row's contact), loaded once per solve
__device__ __forceinline__ float dev_rows_cost(const DevModel& m, Work& w, const DofK& dk, const RowK& rk, const float (&D_c)[NCSLOT], RowEval& ev) {
  const int tid = opaque(threadIdx.x);
  float cost = 0.f;
  ev.force_f = ev.curv_f = ev.force_l = ev.curv_l = 0.f;
  if ((tid & 31) < m.nv) {     // both half-waves keep the per-dof rows (the solve needs the curvature on all 64 lanes)
    cost = cost_friction(dk, ev.jar_f, ev.force_f, ev.curv_f);
    if (rk.l_sign != 0.f) cost += cost_onesided(rk.l_D, ev.jar_l, ev.force_l, ev.curv_l);
    if (tid >= 32) cost = 0.f;
  }
  const int nrow = w.nrow;
#pragma unroll
  for (int k = 0; k < NCSLOT; k++) {
    if (k * NT < nrow) {
      const int r = tid + k * NT;
      if (r < nrow) {
        float f, cv;
        cost += cost_onesided(D_c[k], ev.jar_c[k], f, cv);
        w.cr_force[r] = f; w.cr_curv[r] = cv;
      }
    }
  }
  wsync();
  return wave_sum(cost);
}

// (J^T force)[lane & 31] from the last dev_rows_cost: per-contact wrench G_c = sum_k g_k W[c][k], then
// S[i] . sum_c sg(i,c) G_c, plus the per-dof rows
__device__ __forceinline__ float dev_jt_force(const DevModel& m, Work& w, const RowK& rk, const RowEval& ev) {
  const int tid = opaque(threadIdx.x), d = tid & 31;
  const int ncon = w.ncon;
  if (tid < ((ncon + 3) & ~3)) {
    const int c = tid;
    float G[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < ncon) {
      const int r0 = w.c_row0[c], nr = w.c_nrow[c];
      float g[4] = {0.f, 0.f, 0.f, 0.f};
      for (int e = 0; e < nr; e++) g[0] += w.cr_force[r0 + e];
      if (nr > 1) for (int k = 1; 2 * k - 1 < nr; k++) g[k] = w.c_mu[c][k - 1] * (w.cr_force[r0 + 2 * (k - 1)] - w.cr_force[r0 + 2 * (k - 1) + 1]);
      const float* fr = w.c_frame[c];
      for (int i = 0; i < 3; i++) G[3 + i] = g[0] * fr[i] + g[1] * fr[3 + i] + g[2] * fr[6 + i];     // contact force F
      cross3(w.c_pos[c], G + 3, G);                                                                   // its moment about the origin: p x F
      for (int i = 0; i < 3; i++) G[i] += g[3] * fr[i];                                               //   + the torsional moment
    }
    for (int i = 0; i < 6; i++) w.c_G[c][i] = G[i];   // zero padding up to a multiple of 4 contacts
  }
  wsync();
  float s = ev.force_f + rk.l_sign * ev.force_l;
  if (ncon > 0) {
    float G[6] = {0, 0, 0, 0, 0, 0};
    for (int c0 = 0; c0 < ncon; c0 += 4) {     // four contacts per trip: their LDS reads share one wait
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int c = c0 + k;
        const float sg = (float)((w.c_mpos[c] >> d) & 1u) - (float)((w.c_mneg[c] >> d) & 1u);
#pragma unroll
        for (int i = 0; i < 6; i++) G[i] = fmaf(sg, w.c_G[c][i], G[i]);
      }
    }
    s += dot6(w.S[d], G);
  }
  wsync();
  return s;
}

#ifndef HOIC_IMPROVEMENT_TOL
#define HOIC_IMPROVEMENT_TOL 1e-6f
#endif
// ---- Newton with exact line search.  In: M, f_smooth / a_smooth of lane & 31 (fs, a0; a_smooth also in sc.vec.x), the warm
// start (w.qacc), rows.  Out: w.qacc, w.ftot = f_smooth + J'f (LDS).
// The row residuals jar, M qacc and qacc itself are carried along and updated by alpha * (J s, M s, s) after each
// line search (as MuJoCo's solver does), so an iteration costs one Jacobian product, not three.
__device__ __forceinline__ void dev_solve(const DevModel& m, Work& w, const MReg& M, const RowK& rk, const DofK& dk, const float* qvel, float fs, float a0, int maxit, bool shift_warm) {
  const int tid = opaque(threadIdx.x), d = tid & 31;
  const bool vd = d < m.nv;
  const float scale = frcp(m.meaninertia * (float)max(m.nv, 1));
  // Warm start.  MuJoCo starts from qacc_warmstart = the previous substep's qacc (or from a_smooth if that costs less).  The PD
  // torques move a_smooth from substep to substep by more than the constraint forces move -- eight of fifteen solves of an env
  // step fell back to a_smooth -- so the candidate here is a_smooth + (qacc - a_smooth of the previous substep): the previous
  // solution with the change of the unconstrained acceleration applied, i.e. the previous CONSTRAINT acceleration carried over.
  // The optimum is unique and the stop criteria are unchanged, so only the number of Newton iterations depends on the start;
  // it persists across launches in the state's warm-start row (zero after a reset: the start is then a_smooth).
  // (shift_warm: wave-uniform; the probe kernel passes false and starts from the qacc it was given)
  if (shift_warm) {
    if (tid < NV) w.qacc[tid] = vd ? a0 + w.acon[tid] : 0.f;
    wsync();
  }
  const float wm = w.qacc[d];
  const int nrow = w.nrow;
  float D_c[NCSLOT];
#pragma unroll
  for (int k = 0; k < NCSLOT; k++) { const int r = tid + k * NT; D_c[k] = (r < nrow) ? w.c_D[w.cr_ce[r] >> 3] : 0.f; }
  // warm start choice: cost(warm) vs cost(asmooth)
  const float Mw = vd ? dev_Mx(M, w.qacc) : 0.f;
  const float gw = wave_sum((tid < m.nv) ? 0.5f * (Mw - fs) * (wm - a0) : 0.f);
  RowEval ev, evw;
  float aref_c[NCSLOT];
  dev_rows_setup(m, w, rk, aref_c, qvel, w.sc.vec.x, w.qacc, ev, evw);
  const float cs = dev_rows_cost(m, w, dk, rk, D_c, ev);
  const float cw = gw + dev_rows_cost(m, w, dk, rk, D_c, evw);
  const bool usewarm = cw < cs;
  float cost_prev = usewarm ? cw : cs;
  float qacc = vd ? (usewarm ? wm : a0) : 0.f, Ma = vd ? (usewarm ? Mw : fs) : 0.f;   // M asmooth = fsmooth
  if (usewarm) ev = evw;
  else dev_rows_cost(m, w, dk, rk, D_c, ev);      // forces / curvatures back to the asmooth state
  PT(23);
  int it = 0;
  bool capped = true;     // the loop ran out of iterations (no stop criterion met)
  bool fresh = false;     // jtf holds J'f of the current qacc
  float jtf = 0.f;
  for (; it < maxit; it++) {
    jtf = dev_jt_force(m, w, rk, ev);
    const float g = vd ? (Ma - fs - jtf) : 0.f;
    const float g2 = wave_sum(tid < 32 ? g * g : 0.f);
    PT(11);
    if (fsqrt(g2) * scale < 1e-6f) { fresh = true; capped = false; break; }
    // Newton direction: (M + J' diag(curv) J) s = -g ; friction-loss and limit curvature sit on the diagonal
    const float sd = dev_hsolve<true>(m, w, M, ev.curv_f + ev.curv_l, m.nv, -g);
    if (tid < NV) w.sc.vec.x[tid] = vd ? sd : 0.f;      // (a_smooth is in registers by now; T is dead between two solves)
    wsync();
    // line-search quantities
    const float Ms = vd ? dev_Mx(M, w.sc.vec.x) : 0.f;
    float gq = 0.f, hh = 0.f, g0 = 0.f;
    if (tid < m.nv) { gq = (Ma - fs) * sd; hh = sd * Ms; g0 = g * sd; }
    gq = wave_sum(gq); hh = wave_sum(hh); g0 = wave_sum(g0);
    PT(12);
    RowEval jv;
    dev_rows_jar(m, w, rk, aref_c, w.sc.vec.x, false, jv);
    PT(14);
    float a = 0.f, lo = 0.f, hi = -1.f, alpha = 0.f;
    for (int ls = 0; ls < 10; ls++) {
      float dphi = 0.f, ddphi = 0.f, f, cv;
      if (tid < m.nv) {
        cost_friction(dk, ev.jar_f + a * jv.jar_f, f, cv);
        dphi -= f * jv.jar_f; ddphi += cv * jv.jar_f * jv.jar_f;
        if (rk.l_sign != 0.f) { cost_onesided(rk.l_D, ev.jar_l + a * jv.jar_l, f, cv); dphi -= f * jv.jar_l; ddphi += cv * jv.jar_l * jv.jar_l; }
      }
#pragma unroll
      for (int k = 0; k < NCSLOT; k++)
        if (k * NT < nrow) { cost_onesided(D_c[k], ev.jar_c[k] + a * jv.jar_c[k], f, cv); dphi -= f * jv.jar_c[k]; ddphi += cv * jv.jar_c[k] * jv.jar_c[k]; }
      dphi = wave_sum(dphi) + gq + a * hh; ddphi = wave_sum(ddphi) + hh;
      alpha = a;
      if (fabsf(dphi) < 1e-4f * fabsf(g0) + 1e-12f) break;
      if (dphi < 0.f) lo = a; else hi = a;
      float an = a - fdiv(dphi, ddphi);
      if (hi >= 0.f && (an <= lo || an >= hi)) an = 0.5f * (lo + hi);
      if (hi >= 0.f && hi - lo < 1e-6f * (1.f + hi)) break;
      a = an;
    }
    PT(17);
    const float dq = alpha * sd;
    qacc += vd ? dq : 0.f; Ma = fmaf(alpha, Ms, Ma);
    ev.jar_f = fmaf(alpha, jv.jar_f, ev.jar_f); ev.jar_l = fmaf(alpha, jv.jar_l, ev.jar_l);
#pragma unroll
    for (int k = 0; k < NCSLOT; k++) ev.jar_c[k] = fmaf(alpha, jv.jar_c[k], ev.jar_c[k]);
    const float crow = dev_rows_cost(m, w, dk, rk, D_c, ev);
    const float st = wave_max((tid < m.nv) ? fdiv(fabsf(dq), 1.f + fabsf(qacc)) : 0.f);
    if (st < 1e-7f) { it++; capped = false; break; }
    // MuJoCo's second criterion (engine_solver.c: improvement = scale * (oldcost - cost) < tolerance): once a Newton
    // step no longer lowers the cost, what is left of the gradient is float32 rounding and another Hessian solve
    // would only polish noise
    const float cost = crow + wave_sum((tid < m.nv) ? 0.5f * (Ma - fs) * (qacc - a0) : 0.f);
    const float improvement = scale * (cost_prev - cost);
    cost_prev = cost;
    PT(19);
    if (improvement < HOIC_IMPROVEMENT_TOL) { it++; capped = false; break; }
  }
  // forces at the final acceleration (the row state already belongs to it)
  if (!fresh) jtf = dev_jt_force(m, w, rk, ev);
  if (tid < NV) { w.ftot[tid] = vd ? fs + jtf : 0.f; w.qacc[tid] = qacc; w.acon[tid] = qacc - a0; }
  if (tid == 0) { w.solver_iter = it; w.capped = capped ? 1 : 0; }
  wsync();
}
