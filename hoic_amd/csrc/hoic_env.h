// hoic_env.h — the environment step around the dynamics: HandObjMimic4 as device code.
//
// Mirrors uhc/envs/ho_im4.py: compute_torque (:412-486) with compute_desired_accel (:393-410), gravity
// compensation (:526-536), rfc_obj (:488-501), record_contact/classify_contact (:883-889, :567-597), the
// 15-substep finite differences (:553-559), solve_rfc (:941-1083), calc_ho_diff + termination (:664-688,
// :646-662), get_full_obs_v5 (:280-356) and ho_mimic_reward_9 (uhc/envs/ho_reward.py:943-1047).
// The one-substep lag of body poses / M / bias / contacts is reproduced by keeping
// the state before the last integration (qlag, vlag) and re-running the forward pass on it.
#pragma once
#include "hoic_types.h"
#include "hoic_math.h"
#include "hoic_dynamics.h"
#include "hoic_collide.h"
#include "hoic_solver.h"

struct ExpertView {
  const DevExpert* ex;
  int off, len, start, cur_t;
  HD int frame(int delta) const {
    int ind = cur_t + delta + start;
    ind = ind < len - 1 ? ind : len - 1;
    return off + ind;
  }
};

// ---- position + velocity stages of the forward pass on state (q, v); leaves M (registers), bias, contacts, S
__device__ __forceinline__ void dev_forward_kin(const DevModel& m, const DevConfig& cfg, Work& w, MReg& M, const float* q, const float* v, int* overflow) {
  dev_kinematics(m, w, q); PT(3);
  dev_mass_matrix(m, w, M); PT(4);
  dev_bias(m, w, v); PT(5);
  dev_collision(m, w, overflow, cfg.c.mesh_single_contact); PT(6);
}

// ---- stable PD torque (ho_im4.py:412-486) using M (registers) and bias (LDS) of the previous forward pass
// act: this env's row of the action buffer (global memory; clipped on the fly, ho_im4.py:613)
// What dev_pd_torque reads from global memory for dof lane & 31 (expert frame, action, gains): fetched at the head of the substep,
// ahead of dev_record_contact, so that the two dependent read latencies (expert table pointer, then the row) pass under that
// stage instead of in front of the PD solve.  Raw values only: the arithmetic -- and with it the wait -- stays in dev_pd_torque.
struct PdFetch { float ref, act, base, scale, kp, kd; };
__device__ __forceinline__ PdFetch dev_pd_fetch(const DevModel& m, const DevConfig& cfg, const ExpertView& ev, GPTR(const float) act) {
  const int d = opaque(threadIdx.x) & 31;
  PdFetch f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (d < m.hand_nv) {
    GPTR(const float) ref = as_global(ev.ex->hand_dof) + (size_t)ev.frame(cfg.c.pd_ref_offset) * m.hand_nq;   // 0; 1 in the streaming env
    f.ref = ref[d]; f.act = act[d]; f.base = cfg.base_pose[d]; f.scale = cfg.ctrl_scale[d]; f.kp = cfg.c.jkp[d]; f.kd = cfg.c.jkd[d];
  }
  return f;
}
__device__ __forceinline__ void dev_pd_torque(const DevModel& m, const DevConfig& cfg, Work& w, const MReg& M, const PdFetch& pf) {
  const int tid = opaque(threadIdx.x), d = tid & 31, n = m.hand_nv;
  const float dt = m.timestep;
  float err = 0.f, kp = 0.f, kd = 0.f, qv = 0.f, rhs = 0.f;
  if (d < n) {
    float target;
    const float a = fminf(fmaxf(pf.act, -1.f), 1.f);
    if (d < 3) target = pf.ref + 0.1f * a;
    else if (d < 6) target = pf.ref + 0.3f * a;
    else target = (cfg.c.pd_rel ? pf.ref : pf.base) + pf.scale * a;
    qv = w.qvel[d];
    err = w.qpos[d] + qv * dt - target;
    if (d >= 3) {
      // (the reference's while loops, ho_im4.py:476-481, with a trip limit: an env whose state has run away -- test mode has no
      //  termination -- reaches |err| > 2^24 * 2 pi, where subtracting 2 pi no longer changes the float and the loop never ends:
      //  one such wavefront hangs the launch and everything behind it)
      for (int k = 0; k < 16 && err > 3.14159265358979f; k++) err -= 6.28318530717959f;
      for (int k = 0; k < 16 && err < -3.14159265358979f; k++) err += 6.28318530717959f;
    }
    kp = pf.kp; kd = pf.kd;
    rhs = -w.bias[d] - kp * err - kd * qv;
  }
  PT(20);
  // The solve runs on the whole 32 x 32 matrix, not on the leading hand block with an identity below it: the free object's
  // dofs couple to no hand dof in M (different subtrees: those entries are exact zeros, build_model checks the structure), its
  // own block is positive definite, and its right-hand side and diagonal shift are zero here -- the hand part of the solution
  // is the same, entry by entry, and the 16 masked copies of the accumulator drop out.
  // (A NON-FINITE object pose -- a diverged object: the env is flagged failed by the substep's own check -- would leak through
  //  0 * NaN of the matrix-core factorisation into the hand's torques and, through them, into this step's recorded contacts and
  //  reward inputs, which the reference's hand-block solve (ho_im4.py:455-462) keeps clean: such an env takes the masked solve.)
  const bool obj_finite = __ballot(tid >= n && tid < m.nq && !(fabsf(w.qpos[tid]) < 1e10f)) == 0ull;
  const float acc = dev_hsolve<false>(m, w, M, kd * dt, (m.nv == NV && n == NV - 6 && obj_finite) ? NV : n, rhs);
  if (tid < NV) {
    float tq = 0.f;
    if (tid < n) {
      tq = -kp * err - kd * (qv + acc * dt);
      const float lim = cfg.c.torque_lim[tid];
      tq = fminf(fmaxf(tq, -lim), lim);
    }
    w.sc.vec.ctrl[tid] = tq;        // consumed by dev_applied before the next forward pass reuses the scratch
  }
  wsync();
}

// ---- generalized applied forces: gravity compensation + residual object wrench (lagged Jacobians) + the PD torques of
// dev_pd_torque (sc.vec.ctrl)
// act: the env's action row (global memory): the residual wrench is formed from it here, every substep, instead of living in six
// registers for the whole launch (ho_im4.py:622-623)
__device__ __forceinline__ void dev_applied(const DevModel& m, const DevConfig& cfg, Work& w, GPTR(const float) act) {
  const int tid = opaque(threadIdx.x);
  float vf[3], vt[3];
  for (int i = 0; i < 3; i++) {
    vf[i] = cfg.c.residual_force ? cfg.c.residual_force_scale * fminf(fmaxf(act[m.nu + i], -1.f), 1.f) : 0.f;
    vt[i] = cfg.c.residual_force ? cfg.c.residual_torque_scale * fminf(fmaxf(act[m.nu + 3 + i], -1.f), 1.f) : 0.f;
  }
  if (tid < NV) {
    float s = 0.f;
    if (tid < m.nv) {
      const float f[3] = {0.f, 0.f, m.hand_mass * 9.8f}, z[3] = {0.f, 0.f, 0.f};
      s = dev_apply_ft_dof(m, w, tid, 3, f, z, w.gxpos[2]);                       // ho_im4.py:527-535
      if (cfg.c.residual_force) s += dev_apply_ft_dof(m, w, tid, m.obj_body, vf, vt, &w.qpos[m.hand_nq]);  // :492-500
      const int ai = m.dof_actid[tid];
      if (ai >= 0) s += w.sc.vec.ctrl[ai];                                        // actuation (motor on the dof, gear 1)
    }
    w.applied[tid] = s;
  }
  wsync();
}

// ---- record_contact (ho_im4.py:883-889): lane = hand geom, deterministic accumulation order.  The sums of the env step live
// in the hand-over record itself (rec = the env's record, zeroed at launch start): a lane touches its 13 floats only in a
// substep in which its geom has a contact with the object (8 % of the envs have one at all).
__device__ __forceinline__ void dev_record_contact(const DevModel& m, Work& w, GPTR(float) rec) {
  const int tid = opaque(threadIdx.x);
  if (tid < NHG) {
    const int g = m.hand_geom0 + tid;
    float acc[12]; float cnt = 0.f; bool any = false;
    for (int c = 0; c < w.ncon; c++) {
      const int g1 = w.c_g1[c], g2 = w.c_g2[c];
      if (g1 == g && g2 >= m.obj_geom0 && g2 <= m.obj_geom1) {
        if (!any) {
#pragma unroll
          for (int i = 0; i < 12; i++) acc[i] = rec[PB_REC + tid * 12 + i];
          cnt = rec[PB_RECCNT + tid]; any = true;
        }
        for (int i = 0; i < 3; i++) acc[i] += w.c_pos[c][i];
        for (int i = 0; i < 9; i++) acc[3 + i] += w.c_frame[c][i];
        cnt += 1.f;
      }
    }
    if (any) {
#pragma unroll
      for (int i = 0; i < 12; i++) rec[PB_REC + tid * 12 + i] = acc[i];
      rec[PB_RECCNT + tid] = cnt;
    }
  }
}

// uhc/utils/transforms.py:414 matrix_to_axis_angle
HD void dev_matrix_to_axis_angle(const float* mm, float* aa) {
  float qa[4] = {1.f + mm[0] + mm[4] + mm[8], 1.f + mm[0] - mm[4] - mm[8], 1.f - mm[0] + mm[4] - mm[8], 1.f - mm[0] - mm[4] + mm[8]};
  int best = 0;
  for (int i = 0; i < 4; i++) qa[i] = qa[i] > 0.f ? sqrtf(qa[i]) : 0.f;
  for (int i = 1; i < 4; i++) if (qa[i] > qa[best]) best = i;
  float q[4];
  if (best == 0) { q[0] = qa[0] * qa[0]; q[1] = mm[7] - mm[5]; q[2] = mm[2] - mm[6]; q[3] = mm[3] - mm[1]; }
  else if (best == 1) { q[0] = mm[7] - mm[5]; q[1] = qa[1] * qa[1]; q[2] = mm[3] + mm[1]; q[3] = mm[2] + mm[6]; }
  else if (best == 2) { q[0] = mm[2] - mm[6]; q[1] = mm[3] + mm[1]; q[2] = qa[2] * qa[2]; q[3] = mm[5] + mm[7]; }
  else { q[0] = mm[3] - mm[1]; q[1] = mm[6] + mm[2]; q[2] = mm[7] + mm[5]; q[3] = qa[3] * qa[3]; }
  const float den = 1.f / (2.f * fmaxf(qa[best], 0.1f));
  for (int i = 0; i < 4; i++) q[i] *= den;
  const float nrm = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float half = atan2f(nrm, q[0]), ang = 2.f * half;
  const float s = fabsf(ang) < 1e-6f ? 0.5f - ang * ang / 48.f : sinf(half) / ang;
  for (int i = 0; i < 3; i++) aa[i] = q[1 + i] / s;
}

// ---- classify_contact (ho_im4.py:567-597)
__device__ void dev_classify_contact(const DevModel& m, PostWork& w) {
  const int tid = threadIdx.x;
  const bool has = tid < NHG && w.rec_cnt[tid] > 0;
  const unsigned long long mask = __ballot(has);
  const int idx = __popcll(mask & ((1ull << tid) - 1ull));
  if (tid == 0) w.sc.post.n_avg = __popcll(mask);
  if (has) {
    float f[12];
    const float inv = 1.f / (float)w.rec_cnt[tid];
    for (int i = 0; i < 12; i++) f[i] = w.rec_sum[tid][i] * inv;
    float* n = f + 3; float* t1 = f + 6; float* t2 = f + 9;
    const float nn = 1.f / sqrtf(dot3(n, n));
    for (int i = 0; i < 3; i++) n[i] *= nn;
    const float ex[3] = {1.f, 0.f, 0.f}, ey[3] = {0.f, 1.f, 0.f};
    if (fabsf(n[0]) >= 1e-5f) cross3(n, ex, t1); else cross3(n, ey, t1);
    float tn = 1.f / sqrtf(dot3(t1, t1));
    for (int i = 0; i < 3; i++) t1[i] *= tn;
    cross3(n, t1, t2);
    tn = 1.f / sqrtf(dot3(t2, t2));
    for (int i = 0; i < 3; i++) t2[i] *= tn;
    for (int i = 0; i < 12; i++) w.sc.post.avg_cps[idx][i] = f[i];
    w.sc.post.avg_geom[idx] = tid + m.hand_geom0;
    w.sc.post.avg_ts[idx] = (float)w.rec_cnt[tid];
  }
  wsync();
}

// ---- solve_rfc (ho_im4.py:941-1083) in float64: Newton on the 6-D dual of the non-negative QP
//   min_l |l|^2/4 + b'l + 1/(2 eps) sum_i max(0, -(c_i + a_i'l))^2 ,  residual wrench = -l/2
// (same formulation as oracle/ho_env.c; columns a_i live in registers, 6 per lane)
// LDL^T solve of a symmetric positive definite QP_MAXP x QP_MAXP system G z = h (G packed lower row-major, destroyed;
// h in, z out).  Everything is statically indexed: the arrays stay in registers.
#define QP_MAXP 8
HD void ldl8_solve(double (&G)[QP_MAXP * (QP_MAXP + 1) / 2], double (&h)[QP_MAXP]) {
#define GP(i, j) G[(i) * ((i) + 1) / 2 + (j)]
  double dinv[QP_MAXP];
#pragma unroll
  for (int j = 0; j < QP_MAXP; j++) {
    const double d = GP(j, j);
    double r = __builtin_amdgcn_rcp(d);
    r = r * (2.0 - d * r); r = r * (2.0 - d * r);
    dinv[j] = r;
#pragma unroll
    for (int i = j + 1; i < QP_MAXP; i++) {
      const double lij = GP(i, j) * r;
#pragma unroll
      for (int k = j + 1; k <= i; k++) GP(i, k) -= lij * GP(k, j);
    }
#pragma unroll
    for (int i = j + 1; i < QP_MAXP; i++) GP(i, j) *= r;
  }
#pragma unroll
  for (int j = 0; j < QP_MAXP; j++) {
#pragma unroll
    for (int i = j + 1; i < QP_MAXP; i++) h[i] -= GP(i, j) * h[j];
  }
#pragma unroll
  for (int j = 0; j < QP_MAXP; j++) h[j] *= dinv[j];
#pragma unroll
  for (int j = QP_MAXP - 1; j >= 0; j--) {
#pragma unroll
    for (int i = j + 1; i < QP_MAXP; i++) h[j] -= GP(i, j) * h[i];
  }
#undef GP
}
HD void chol6_solve(double* H, double* x) {  // H: 21 lower-packed row-major (overwritten by its Cholesky factor), x in/out
#define HP(i, j) H[(i) * ((i) + 1) / 2 + (j)]
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double s = HP(j, j);
#pragma unroll
    for (int k = 0; k < j; k++) s -= HP(j, k) * HP(j, k);
    s = s > 1e-300 ? s : 1e-300;
    const double l = sqrt(s), il = 1.0 / l;
    HP(j, j) = l;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double t = HP(i, j);
#pragma unroll
      for (int k = 0; k < j; k++) t -= HP(i, k) * HP(j, k);
      HP(i, j) = t * il;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) { double s = x[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= HP(i, k) * x[k]; x[i] = s / HP(i, i); }
#pragma unroll
  for (int i = 5; i >= 0; i--) { double s = x[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= HP(k, i) * x[k]; x[i] = s / HP(i, i); }
#undef HP
}

// ---- the residual-force QP proper: columns a_k (6) and offsets c_k (k < ncol) in LDS at qc[i * QP_MAXCOL + k],
// right-hand side b; returns the dual optimum lambda = 2 (A x - b).  stat (optional, lane-uniform): column entries,
// small solves are not counted; [0] = active-set iterations, [1] = dual-Newton iterations of the fallback.
template <class QC> __device__ __forceinline__ void dev_nnqp(PostWork& w, QC qc, int ncol, const double (&b)[6], double (&lam)[6], int* stat) {
  const int tid = threadIdx.x;
  const double eps = 1e-7;
  const int nslot = (ncol + NT - 1) / NT;         // columns per lane actually present (typically 1-2 of at most 6)
  const double ieps = 1.0 / eps;
  // ---- primal active-set pass (Lawson-Hanson form) on
  //        min_x |A x - b|^2 + c.x + eps/2 |x|^2,   x >= 0,
  // whose dual in lambda = 2 (A x - b) is the 6-D problem the Newton iteration below solves.  At most 8 columns are
  // passive at a time (A has 6 rows; the eps term lets a few more in): their Gram matrix sits in LDS, the 8 x 8
  // LDL^T solve runs wave-uniform in registers, the steepest-column search is lane-parallel over the columns.
  // Typical cost 8 column entries / 12 small solves; the dual Newton from a cold start needed up to 37 iterations of
  // 27 float64 wave reductions plus ~10 line-search passes each, and one such env sets the kernel time.
  bool polish = false;
  {
    double xs[QP_MAXP], hb[QP_MAXP];
    int colj[QP_MAXP];
#pragma unroll
    for (int j = 0; j < QP_MAXP; j++) { xs[j] = 0.0; hb[j] = 0.0; colj[j] = 0; }
    unsigned pm = 0u, inP = 0u, ban = 0u;      // passive slots; per-lane bit jj <-> column tid + jj * NT
    double* Gm = w.sc.post.qp_G;
    float (*pa)[8] = w.sc.post.qp_a;
    // gradient tolerance relative to the size of its terms
    float amax = 0.f, cmax = 0.f;
    for (int jj = 0; jj < nslot; jj++) {
      const int col = tid + jj * NT;
      if (col < ncol) {
        for (int i = 0; i < 6; i++) amax = fmaxf(amax, fabsf(qc[i * QP_MAXCOL + col]));
        cmax = fmaxf(cmax, fabsf(qc[6 * QP_MAXCOL + col]));
      }
    }
    amax = wave_max(amax); cmax = wave_max(cmax);
    double bmax = 0.0;
    for (int i = 0; i < 6; i++) bmax = fmax(bmax, fabs(b[i]));
    const double tol = 1e-9 * (1.0 + (double)cmax + 2.0 * (double)amax * bmax);
    int it = 0;
    for (; it < 64; it++) {
      double r[6];
#pragma unroll
      for (int i = 0; i < 6; i++) r[i] = b[i];
#pragma unroll
      for (int j = 0; j < QP_MAXP; j++)
        if ((pm >> j) & 1u) {
#pragma unroll
          for (int i = 0; i < 6; i++) r[i] -= xs[j] * (double)pa[j][i];
        }
      double best = -1e300; int bestc = -1;
      for (int jj = 0; jj < nslot; jj++) {
        const int col = tid + jj * NT;
        if (col < ncol && !(((inP | ban) >> jj) & 1u)) {
          double wv = -(double)qc[6 * QP_MAXCOL + col];
#pragma unroll
          for (int i = 0; i < 6; i++) wv += 2.0 * (double)qc[i * QP_MAXCOL + col] * r[i];
          if (wv > best) { best = wv; bestc = col; }
        }
      }
      const double wmax = wave_max_d(best);
      if (!(wmax > tol)) break;
      if (pm == (1u << QP_MAXP) - 1u) { polish = true; break; }
      const int src = __ffsll((long long)__ballot(best == wmax)) - 1;
      const int k = __builtin_amdgcn_readlane(bestc, src);
      const int s = __ffs((int)~pm) - 1;
      wsync();
      if (tid < 8) pa[s][tid] = tid < 7 ? qc[tid * QP_MAXCOL + k] : 0.f;
      wsync();
      double an[6], hnew = -(double)pa[s][6];
#pragma unroll
      for (int i = 0; i < 6; i++) { an[i] = (double)pa[s][i]; hnew += 2.0 * an[i] * b[i]; }
      if (tid < QP_MAXP) {
        double g = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) g += an[i] * (double)pa[tid][i];
        g *= 2.0;
        if (tid == s) g += eps;
        const int hi_ = tid > s ? tid : s, lo_ = tid > s ? s : tid;
        Gm[hi_ * (hi_ + 1) / 2 + lo_] = g;
      }
#pragma unroll
      for (int j = 0; j < QP_MAXP; j++) if (j == s) { hb[j] = hnew; xs[j] = 0.0; colj[j] = k; }
      pm |= 1u << s;
      if (tid == (k & (NT - 1))) inP |= 1u << (k / NT);
      wsync();
      bool rejected = false, first = true;
      for (int in = 0; in < 24; in++) {
        double W[QP_MAXP * (QP_MAXP + 1) / 2], z[QP_MAXP];
#pragma unroll
        for (int i = 0; i < QP_MAXP; i++) {
          const bool ai = (pm >> i) & 1u;
          z[i] = ai ? hb[i] : 0.0;
#pragma unroll
          for (int j = 0; j <= i; j++) {
            const bool aj = (pm >> j) & 1u;
            W[i * (i + 1) / 2 + j] = (ai && aj) ? Gm[i * (i + 1) / 2 + j] : (i == j ? 1.0 : 0.0);
          }
        }
        ldl8_solve(W, z);
        double znew = 0.0;
#pragma unroll
        for (int j = 0; j < QP_MAXP; j++) if (j == s) znew = z[j];
        if (first && !(znew > 0.0)) {       // rounding: the steepest column does not want to enter after all
          pm &= ~(1u << s);
          if (tid == (k & (NT - 1))) { inP &= ~(1u << (k / NT)); ban |= 1u << (k / NT); }
          rejected = true;
          break;
        }
        first = false;
        bool allpos = true;
#pragma unroll
        for (int j = 0; j < QP_MAXP; j++) if (((pm >> j) & 1u) && !(z[j] > 0.0)) allpos = false;
        if (allpos) {
#pragma unroll
          for (int j = 0; j < QP_MAXP; j++) if ((pm >> j) & 1u) xs[j] = z[j];
          break;
        }
        double al = 1.0;
#pragma unroll
        for (int j = 0; j < QP_MAXP; j++)
          if (((pm >> j) & 1u) && !(z[j] > 0.0)) al = fmin(al, xs[j] / (xs[j] - z[j]));
        double xmax = 1.0;
#pragma unroll
        for (int j = 0; j < QP_MAXP; j++)
          if ((pm >> j) & 1u) { xs[j] += al * (z[j] - xs[j]); xmax = fmax(xmax, fabs(xs[j])); }
#pragma unroll
        for (int j = 0; j < QP_MAXP; j++)
          if (((pm >> j) & 1u) && !(z[j] > 0.0) && xs[j] <= 1e-14 * xmax) {
            pm &= ~(1u << j); xs[j] = 0.0;
            if (tid == (colj[j] & (NT - 1))) inP &= ~(1u << (colj[j] / NT));
          }
      }
      if (!rejected) ban = 0u;
    }
    if (it >= 64) polish = true;
    if (stat) stat[0] = it;
#pragma unroll
    for (int i = 0; i < 6; i++) lam[i] = -2.0 * b[i];
#pragma unroll
    for (int j = 0; j < QP_MAXP; j++)
      if ((pm >> j) & 1u) {
#pragma unroll
        for (int i = 0; i < 6; i++) lam[i] += 2.0 * xs[j] * (double)pa[j][i];
      }
    wsync();
  }
  // ---- dual Newton: only when the passive set was full or the pass ran out of iterations; starts at the
  // multipliers found above
  int n_dual = 0;
  for (int it = 0; polish && it < 60; it++) {
    n_dual++;
    double g[6], H[21];
    for (int i = 0; i < 6; i++) g[i] = 0.0;
    for (int i = 0; i < 21; i++) H[i] = 0.0;
    for (int jj = 0; jj < nslot; jj++) {
      const int col = tid + jj * NT;
      if (col < ncol) {
        double a[6];
#pragma unroll
        for (int i = 0; i < 6; i++) a[i] = (double)qc[i * QP_MAXCOL + col];
        double sk = (double)qc[6 * QP_MAXCOL + col];
#pragma unroll
        for (int i = 0; i < 6; i++) sk += a[i] * lam[i];
        if (sk < 0.0) {
          const double se = sk * ieps;
#pragma unroll
          for (int i = 0, k = 0; i < 6; i++) {
            g[i] += se * a[i];
            const double ai = a[i] * ieps;
#pragma unroll
            for (int j = 0; j <= i; j++, k++) H[k] += ai * a[j];
          }
        }
      }
    }
    for (int i = 0; i < 6; i++) g[i] = wave_sum_d(g[i]) + 0.5 * lam[i] + b[i];
    for (int i = 0; i < 21; i++) H[i] = wave_sum_d(H[i]);
    for (int i = 0, k = 0; i < 6; i++) { k += i; H[k] += 0.5; k++; }
    double gn = 0, ln = 0;
    for (int i = 0; i < 6; i++) { gn += g[i] * g[i]; ln += lam[i] * lam[i]; }
    if (sqrt(gn) < 1e-9 * (1.0 + sqrt(ln))) break;      // H >= I/2: |lambda error| <= 2 |g|, far below the float32 result
    double dir[6];
    for (int i = 0; i < 6; i++) dir[i] = -g[i];
    chol6_solve(H, dir);
    double gl = 0, dd = 0, bd = 0;
    for (int i = 0; i < 6; i++) { gl += lam[i] * dir[i]; dd += dir[i] * dir[i]; bd += b[i] * dir[i]; }
    double al = 1.0, lo = 0.0, hi = -1.0;
    for (int ls = 0; ls < 60; ls++) {
      double dphi = 0, ddphi = 0;
      for (int jj = 0; jj < nslot; jj++) {
        const int col = tid + jj * NT;
        if (col < ncol) {
          double s0 = (double)qc[6 * QP_MAXCOL + col], av = 0.0;
#pragma unroll
          for (int i = 0; i < 6; i++) { const double ai = (double)qc[i * QP_MAXCOL + col]; s0 += ai * lam[i]; av += ai * dir[i]; }
          const double sk = s0 + al * av;
          if (sk < 0.0) { dphi += sk * av * ieps; ddphi += av * av * ieps; }
        }
      }
      dphi = wave_sum_d(dphi) + 0.5 * (gl + al * dd) + bd; ddphi = wave_sum_d(ddphi) + 0.5 * dd;
      if (fabs(dphi) < 1e-13 * (1.0 + fabs(bd) + fabs(gl))) break;
      if (dphi < 0) lo = al; else hi = al;
      double an = al - dphi / ddphi;
      if (hi >= 0 && (an <= lo || an >= hi)) an = 0.5 * (lo + hi);
      if (hi < 0 && an <= lo) an = 2 * al + 1e-12;
      if (hi >= 0 && hi - lo < 1e-15 * (1 + hi)) break;
      // the 1-D function is piecewise quadratic: once the active set stops changing the Newton step is exact and
      // repeats itself (the derivative test above sits below its own rounding noise, which is scaled by 1/eps)
      const bool same = fabs(an - al) <= 1e-14 * (1.0 + fabs(al));
      al = an;
      if (same) break;
    }
    double st = 0;
    for (int i = 0; i < 6; i++) { lam[i] += al * dir[i]; st += al * al * dir[i] * dir[i]; }
    if (sqrt(st) < 1e-15 * (1.0 + sqrt(ln))) break;
  }
  if (stat) stat[1] = n_dual;
}

// qc: the env's QP_COL_FLOATS floats of DevState::qpcol (global memory, L2-resident while in use: the columns exist for the 8 % of
// the envs that have a hand-object contact, and keeping 10.6 KB of LDS for them in every workgroup cost the post-step kernel its
// place beside the substep workgroups)
__device__ float dev_solve_rfc(const DevModel& m, const DevConfig& cfg, PostWork& w, const float* vf, const float* vt, double* warm_lam, GPTR(float) qc) {
  const int tid = threadIdx.x;
  const double w_t = 1e4, swt = 100.0, mu = 0.75, dx = 0.0025;
  if (!cfg.c.explain_force)
    return sqrtf(dot3(vf, vf)) + (float)w_t * sqrtf(dot3(vt, vt));
  const int nq = m.nq, ob = m.obj_body, lastg = m.ngeom - 1;
  double Rm[9];
  {
    float Rf[9];
    quat_matrix_ref(&w.qpos[nq - 4], Rf);
    for (int i = 0; i < 9; i++) Rm[i] = Rf[i];
  }
  double F[3], tau[3], I[9], ow[3], oa[3], ooa[3];
  for (int i = 0; i < 3; i++) { ow[i] = w.sc.post.gangvel[lastg][i]; oa[i] = w.sc.post.obj_avg_acc[i]; ooa[i] = w.sc.post.obj_avg_acc[3 + i]; }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += Rm[3 * i + k] * (double)m.body_inertia[ob][k] * Rm[3 * j + k];
      I[3 * i + j] = s;
    }
  const double mass = m.body_mass[ob];
  F[0] = mass * oa[0]; F[1] = mass * oa[1]; F[2] = mass * (oa[2] + 9.8);
  double Iw[3], Ioa[3];
  for (int i = 0; i < 3; i++) {
    Iw[i] = I[3 * i] * ow[0] + I[3 * i + 1] * ow[1] + I[3 * i + 2] * ow[2];
    Ioa[i] = I[3 * i] * ooa[0] + I[3 * i + 1] * ooa[1] + I[3 * i + 2] * ooa[2];
  }
  tau[0] = Ioa[0] + ow[1] * Iw[2] - ow[2] * Iw[1];
  tau[1] = Ioa[1] + ow[2] * Iw[0] - ow[0] * Iw[2];
  tau[2] = Ioa[2] + ow[0] * Iw[1] - ow[1] * Iw[0];
  if (w.sc.post.n_avg == 0) { if (tid == 6) warm_lam[6] = 0.0; }
  if (w.sc.post.n_avg == 0)
    return (float)(sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2]) + w_t * sqrt(tau[0] * tau[0] + tau[1] * tau[1] + tau[2] * tau[2]));
  const int npt = cfg.c.surface_contact ? 5 : 1, ncol = w.sc.post.n_avg * npt * 4;
  const int nslot = (ncol + NT - 1) / NT;         // columns per lane actually present (typically 1-2 of at most 6)
  const double inv = 1.0 / sqrt(1.0 + mu * mu);
  // columns a_i (6) and offsets c_i live in LDS as float32 (their inputs are float32 quantities; all arithmetic on
  // them is float64): the QP then needs ~130 registers instead of ~460, and its loops run over the columns that
  // exist instead of six unrolled slots per lane
  const double obj_p[3] = {w.qpos[nq - 7], w.qpos[nq - 6], w.qpos[nq - 5]};
  const double obj_v[3] = {w.sc.post.gvel[lastg][0], w.sc.post.gvel[lastg][1], w.sc.post.gvel[lastg][2]};
  for (int jj = 0; jj < nslot; jj++) {
    const int col = tid + jj * NT;
    if (col >= ncol) continue;
    const int pt = col >> 2, e = col & 3, ci = pt / npt, j = pt % npt;
    const float* cp = w.sc.post.avg_cps[ci];
    double pos[3], fn[3], t1[3], t2[3];
    for (int i = 0; i < 3; i++) { pos[i] = cp[i]; fn[i] = cp[3 + i]; t1[i] = cp[6 + i]; t2[i] = cp[9 + i]; }
    const int g1 = w.sc.post.avg_geom[ci];
    const double* dl = (j == 1 || j == 2) ? t1 : t2;
    const double sg = (j == 0) ? 0.0 : ((j & 1) ? dx : -dx);
    double p[3], crh[3], cro[3], rel[3], relt[3];
    for (int k = 0; k < 3; k++) { p[k] = pos[k] + sg * dl[k]; crh[k] = p[k] - (double)w.gxpos[g1][k]; cro[k] = p[k] - obj_p[k]; }
    const double gw[3] = {w.sc.post.gangvel[g1][0], w.sc.post.gangvel[g1][1], w.sc.post.gangvel[g1][2]};
    const double cvh[3] = {w.sc.post.gvel[g1][0] + gw[1] * crh[2] - gw[2] * crh[1], w.sc.post.gvel[g1][1] + gw[2] * crh[0] - gw[0] * crh[2],
                           w.sc.post.gvel[g1][2] + gw[0] * crh[1] - gw[1] * crh[0]};
    const double cvo[3] = {obj_v[0] + ow[1] * cro[2] - ow[2] * cro[1], obj_v[1] + ow[2] * cro[0] - ow[0] * cro[2],
                           obj_v[2] + ow[0] * cro[1] - ow[1] * cro[0]};
    for (int k = 0; k < 3; k++) rel[k] = cvo[k] - cvh[k];
    const double nn = fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2];
    const double vn = fn[0] * rel[0] + fn[1] * rel[1] + fn[2] * rel[2];
    for (int k = 0; k < 3; k++) relt[k] = rel[k] - vn * fn[k];
    const double nvn = fabs(vn) * sqrt(nn), nvt = sqrt(relt[0] * relt[0] + relt[1] * relt[1] + relt[2] * relt[2]);
    const double ts = (double)w.sc.post.avg_ts[pt / 5] / (double)cfg.c.sim_step;   // cp_ts[i // 5] quirk (:1012)
    const double d1 = relt[0] * t1[0] + relt[1] * t1[1] + relt[2] * t1[2], d2 = relt[0] * t2[0] + relt[1] * t2[1] + relt[2] * t2[2];
    int am = 0;                       // argmax of {-d1, d1, -d2, d2}, first maximum (static indexing: no private array)
    double bestd = -d1;
    if (d1 > bestd) { bestd = d1; am = 1; }
    if (-d2 > bestd) { bestd = -d2; am = 2; }
    if (d2 > bestd) { bestd = d2; am = 3; }
    const double* tt = e < 2 ? t1 : t2;
    const double sgn = (e & 1) ? -1.0 : 1.0;
    double xv[3];
    for (int k = 0; k < 3; k++) xv[k] = (fn[k] + sgn * mu * tt[k]) * inv * ts;
    qc[0 * QP_MAXCOL + col] = (float)xv[0]; qc[1 * QP_MAXCOL + col] = (float)xv[1]; qc[2 * QP_MAXCOL + col] = (float)xv[2];
    qc[3 * QP_MAXCOL + col] = (float)(swt * (cro[1] * xv[2] - cro[2] * xv[1]));
    qc[4 * QP_MAXCOL + col] = (float)(swt * (cro[2] * xv[0] - cro[0] * xv[2]));
    qc[5 * QP_MAXCOL + col] = (float)(swt * (cro[0] * xv[1] - cro[1] * xv[0]));
    qc[6 * QP_MAXCOL + col] = (float)(((vn * nn <= 0.0) ? nvn : 0.0) + (e == am ? 0.0 : nvt));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // the columns travel between lanes through global memory
  wsync();
  const double b[6] = {F[0], F[1], F[2], swt * tau[0], swt * tau[1], swt * tau[2]};
  double lam[6];
#ifdef HOIC_TRACE_DISPATCH
  int qstat[2];
  dev_nnqp(w, (GPTR(const float))qc, ncol, b, lam, qstat);
  if (tid == 0) { g_trace_qp[blockIdx.x * 4] = qstat[0]; g_trace_qp[blockIdx.x * 4 + 1] = qstat[1]; g_trace_qp[blockIdx.x * 4 + 2] = ncol; g_trace_qp[blockIdx.x * 4 + 3] = qstat[1] > 0; }
#else
  dev_nnqp(w, (GPTR(const float))qc, ncol, b, lam, nullptr);
#endif
  if (tid < 6) warm_lam[tid] = lam[tid];
  if (tid == 6) warm_lam[6] = 1.0;
  const double rf = 0.5 * sqrt(lam[0] * lam[0] + lam[1] * lam[1] + lam[2] * lam[2]);
  const double rt = 0.5 * sqrt(lam[3] * lam[3] + lam[4] * lam[4] + lam[5] * lam[5]);
  return (float)(rf + rt);   // |rest_force| + sqrt(w_t) |rest_torque| (:1083)
}

// ---- termination diffs (calc_ho_diff, ho_im4.py:664-688); out: pos, rot, jpos, obj, obj_rot(=0)
template <class W> __device__ __forceinline__ void dev_ho_diff(const DevModel& m, const W& w, const ExpertView& ev, float* out) {
  const int tid = threadIdx.x, hb0 = m.hand_body0, fr = ev.frame(0);
  GPTR(const float) ep = as_global(ev.ex->body_pos) + (size_t)fr * NHB * 3; GPTR(const float) eq = as_global(ev.ex->body_quat) + (size_t)fr * NHB * 4;
  float s = 0.f;
  if (tid < NHB) { float dv[3]; for (int i = 0; i < 3; i++) dv[i] = w.xpos[hb0 + tid][i] - ep[3 * tid + i]; s = sqrtf(dot3(dv, dv)); }
  const float root = rl(s, 0);
  out[0] = root;
  out[2] = wave_sum(s) / (float)NHB;
  float qi[4], qd[4], e4[4] = {eq[0], eq[1], eq[2], eq[3]};
  quat_inv(w.xquat[hb0], qi); mulquat(e4, qi, qd);
  out[1] = 2.f * asinf(fminf(fmaxf(sqrtf(qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]), 0.f), 1.f));
  GPTR(const float) eo = as_global(ev.ex->obj_pose) + (size_t)fr * 7;
  float dv[3];
  for (int i = 0; i < 3; i++) dv[i] = w.qpos[m.hand_nq + i] - eo[i];
  out[3] = sqrtf(dot3(dv, dv));
  float eo4[4] = {eo[3], eo[4], eo[5], eo[6]};
  quat_inv(eo4, qi); mulquat(eo4, qi, qd);   // :685 uses the expert quaternion twice
  out[4] = 2.f * asinf(fminf(fmaxf(sqrtf(qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]), 0.f), 1.f));
}

// ---- ho_mimic_reward_9 (uhc/envs/ho_reward.py:943-1047); out[0] reward, out[1..9] info
__device__ void dev_reward(const DevModel& m, const DevConfig& cfg, const PostWork& w, const ExpertView& ev, float rfc_score, float* out) {
  const int tid = threadIdx.x, nh = m.hand_nq, hb0 = m.hand_body0, fr = ev.frame(0);
  const float* wk = cfg.rp.wk;
  GPTR(const float) eq = as_global(ev.ex->hand_dof) + (size_t)fr * nh; GPTR(const float) evel = as_global(ev.ex->hand_dof_vel) + (size_t)fr * nh;
  GPTR(const float) ebq = as_global(ev.ex->body_quat) + (size_t)fr * NHB * 4; GPTR(const float) ebp = as_global(ev.ex->body_pos) + (size_t)fr * NHB * 3;
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  if (tid >= 6 && tid < nh) a = fabsf(w.qpos[tid] - eq[tid]);
  if (tid < m.hand_nv) c = fabsf(w.qvel[tid] - evel[tid]);
  if (tid < NHB) {
    float qi[4], qd[4], e4[4] = {ebq[4 * tid], ebq[4 * tid + 1], ebq[4 * tid + 2], ebq[4 * tid + 3]};
    quat_inv(e4, qi); mulquat(w.xquat[hb0 + tid], qi, qd);
    const float w0 = fabsf(qd[0]) - 1.f;
    b = sqrtf(w0 * w0 + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]);
    float dv[3];
    for (int i = 0; i < 3; i++) dv[i] = w.xpos[hb0 + tid][i] - ebp[3 * tid + i];
    d = sqrtf(dot3(dv, dv));
  }
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); d = wave_sum(d);
  const float pose_r = expf(-wk[8] * a / (float)(nh - 6)), wpose_r = expf(-wk[9] * b / (float)NHB);
  const float vel_r = expf(-wk[10] * c / (float)m.hand_nv), jpos_r = expf(-wk[11] * d / (float)NHB);
  GPTR(const float) eo = as_global(ev.ex->obj_pose) + (size_t)fr * 7;
  float dv[3];
  for (int i = 0; i < 3; i++) dv[i] = w.qpos[nh + i] - eo[i];
  const float opos_r = expf(-wk[12] * sqrtf(dot3(dv, dv)));
  float qi[4], qd[4], eo4[4] = {eo[3], eo[4], eo[5], eo[6]};
  quat_inv(eo4, qi); mulquat(&w.qpos[nh + 3], qi, qd);
  const float w0 = fabsf(qd[0]) - 1.f;
  const float orot_r = expf(-wk[13] * sqrtf(w0 * w0 + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]));
  float s = 0.f;
  for (int i = 0; i < 3; i++) s += fabsf(w.qvel[m.hand_nv + i] - as_global(ev.ex->obj_vel)[(size_t)fr * 3 + i]) +
                                   fabsf(w.qvel[m.hand_nv + 3 + i] - as_global(ev.ex->obj_angvel)[(size_t)fr * 3 + i]);
  const float ovel_r = expf(-wk[14] * s / 6.f);
  const float orfc_r = cfg.c.residual_force ? expf(-wk[15] * rfc_score) : 1.f;
  const float hand = (wk[0] * pose_r + wk[1] * wpose_r + wk[3] * jpos_r + wk[2] * vel_r) / (wk[0] + wk[1] + wk[3] + wk[2]);
  const float obj = (wk[4] * opos_r + wk[5] * orot_r + wk[6] * ovel_r + wk[7] * orfc_r) / (wk[4] + wk[5] + wk[6] + wk[7]);
  out[0] = hand * obj; out[1] = pose_r; out[2] = wpose_r; out[3] = jpos_r; out[4] = vel_r; out[5] = opos_r;
  out[6] = orot_r; out[7] = ovel_r; out[8] = orfc_r; out[9] = 1.f;
}

// ---- get_full_obs_v5(5) (ho_im4.py:280-356): 617 floats written straight to HBM, coalesced per segment
template <class W> __device__ __forceinline__ void dev_write_obs(const DevModel& m, const W& w, const ExpertView& ev, float* __restrict__ obs) {
  const int tid = threadIdx.x, nh = m.hand_nq, hb0 = m.hand_body0;
  const DevExpert& x = *ev.ex;
  float R[9], Rqi[4];
  quat_matrix_ref(w.xquat[hb0], R);
  quat_inv(w.xquat[hb0], Rqi);
  const float* P = w.xpos[hb0];
  // (component selection by compares, not by indexing R with a lane-dependent index: that would put R into scratch memory)
#define RT(comp, v) (sel3(R[0], R[1], R[2], (comp)) * (v)[0] + sel3(R[3], R[4], R[5], (comp)) * (v)[1] + sel3(R[6], R[7], R[8], (comp)) * (v)[2])
  if (tid < 6) { const int rw = tid >> 1, cl = tid & 1; obs[tid] = cl ? sel3(R[1], R[4], R[7], rw) : sel3(R[0], R[3], R[6], rw); }
  if (tid < 20) { obs[6 + tid] = w.qpos[6 + tid]; obs[126 + tid] = w.qvel[6 + tid]; }
  for (int t = tid; t < 100; t += NT) {
    const int k = t / 20, i = t % 20;
    obs[26 + t] = x.hand_dof[(size_t)ev.frame(k + 1) * nh + 6 + i] - w.qpos[6 + i];
  }
  if (tid < 6) { const float* v = &w.qvel[tid < 3 ? 0 : 3]; obs[146 + tid] = RT(tid % 3, v); }
  if (tid < 5) {
    const int fr = ev.frame(tid + 1);
    const float* tp = x.body_pos + (size_t)fr * NHB * 3; const float* tq = x.body_quat + (size_t)fr * NHB * 4;
    float t3[3] = {tp[0] - P[0], tp[1] - P[1], tp[2] - P[2]}, q[4], Mm[9], tq4[4] = {tq[0], tq[1], tq[2], tq[3]};
    float* o = obs + 152 + 9 * tid;
    for (int c = 0; c < 3; c++) o[c] = RT(c, t3);
    mulquat(Rqi, tq4, q); quat_matrix_ref(q, Mm);
    o[3] = Mm[0]; o[4] = Mm[1]; o[5] = Mm[3]; o[6] = Mm[4]; o[7] = Mm[6]; o[8] = Mm[7];
  }
  if (tid < 60) {  // component-major (3,20) block: transform_vec_batch quirk (math_utils.py:117-130)
    const int comp = tid / 20, b = tid % 20 + 1;
    float v[3] = {w.xpos[hb0 + b][0] - w.qpos[0], w.xpos[hb0 + b][1] - w.qpos[1], w.xpos[hb0 + b][2] - w.qpos[2]};
    obs[197 + tid] = RT(comp, v);
  }
  for (int t = tid; t < 300; t += NT) {
    const int k = t / 60, r = t % 60, comp = r / 20, b = r % 20 + 1;
    const float* tp = x.body_pos + (size_t)ev.frame(k + 1) * NHB * 3 + 3 * b;
    float v[3] = {tp[0] - w.xpos[hb0 + b][0], tp[1] - w.xpos[hb0 + b][1], tp[2] - w.xpos[hb0 + b][2]};
    obs[257 + t] = RT(comp, v);
  }
  const float* op = &w.qpos[nh]; const float* oq = &w.qpos[nh + 3];
  if (tid == 0) {
    float t3[3] = {op[0] - P[0], op[1] - P[1], op[2] - P[2]}, q[4], Mm[9];
    float* o = obs + 557;
    for (int c = 0; c < 3; c++) o[c] = RT(c, t3);
    mulquat(Rqi, oq, q); quat_matrix_ref(q, Mm);
    o[3] = Mm[0]; o[4] = Mm[1]; o[5] = Mm[3]; o[6] = Mm[4]; o[7] = Mm[6]; o[8] = Mm[7];
  }
  if (tid < 6) { const float* v = &w.qvel[m.hand_nv + (tid < 3 ? 0 : 3)]; obs[566 + tid] = RT(tid % 3, v); }
  if (tid < 5) {
    const float* tp = x.obj_pose + (size_t)ev.frame(tid + 1) * 7;
    float t3[3] = {tp[0] - op[0], tp[1] - op[1], tp[2] - op[2]}, oqi[4], q2[4], q[4], Mm[9], tq4[4] = {tp[3], tp[4], tp[5], tp[6]};
    float* o = obs + 572 + 9 * tid;
    for (int c = 0; c < 3; c++) o[c] = RT(c, t3);
    quat_inv(oq, oqi); mulquat(tq4, oqi, q2); mulquat(Rqi, q2, q); quat_matrix_ref(q, Mm);
    o[3] = Mm[0]; o[4] = Mm[1]; o[5] = Mm[3]; o[6] = Mm[4]; o[7] = Mm[6]; o[8] = Mm[7];
  }
#undef RT
}

// ---- reset_model (ho_im4.py:690-716): state <- expert frame `start` of sequence `seq`; the warm start is cleared.  The caller
// stores the state as the lagged state as well (store_state with lag_too)
template <class W> __device__ __forceinline__ void dev_reset_state(const DevModel& m, W& w, const DevExpert& x, int seq, int start) {
  const int tid = threadIdx.x;
  const int len = x.seq_len[seq], off = x.seq_off[seq];
  const int fr = off + (start < len - 1 ? start : len - 1), nh = m.hand_nq;
  if (tid < nh) { w.qpos[tid] = x.hand_dof[(size_t)fr * nh + tid]; w.qvel[tid] = x.hand_dof_vel[(size_t)fr * nh + tid]; }
  if (tid < 7) w.qpos[nh + tid] = x.obj_pose[(size_t)fr * 7 + tid];
  if (tid < 3) { w.qvel[m.hand_nv + tid] = x.obj_vel[(size_t)fr * 3 + tid]; w.qvel[m.hand_nv + 3 + tid] = x.obj_angvel[(size_t)fr * 3 + tid]; }
  if (tid >= m.nq && tid < NQP) w.qpos[tid] = 0.f;
  if (tid < NV) w.qacc[tid] = 0.f;
  wsync();
}
