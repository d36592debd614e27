/* ho_env.c — CPU oracle, environment glue (TEST INFRASTRUCTURE).
 *
 * Scalar float64 restatement of HandObjMimic4 (uhc/envs/ho_im4.py) and ho_mimic_reward_9
 * (uhc/envs/ho_reward.py:943-1047) around the physics in ho_sim.c.  Every function cites the
 * reference lines it follows; these parts ARE pinned by golden vectors produced by importing the
 * reference's own Python (tests/golden/gen_golden.py -> tests/golden/*.npz).
 */
#include "ho_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define NHB HOIC_NHANDBODY
#define NHG 19 /* hand collision geoms recorded by record_contact */
#define MAXQP (NHG * 20)

typedef struct ho_cfg {
  double jkp[NU], jkd[NU], torque_lim[NU];
  double pos_diff_thresh, rot_diff_thresh, jpos_diff_thresh, obj_pos_diff_thresh, obj_rot_diff_thresh;
  double residual_force_scale, residual_torque_scale;
  int sim_step, w_size, residual_force, explain_force, surface_contact, mode_train, pd_rel;
  int pd_ref_offset;   /* 1: streaming env (uhc/envs/ho_im_test.py: the frame is inserted before step, so delta_t = 0 is frame t+1) */
} ho_cfg;

typedef struct ho_expert {
  int T;
  double *hand_dof, *hand_dof_vel, *obj_pose, *obj_vel, *obj_angvel, *body_pos, *body_quat;
} ho_expert;

typedef struct ho_env {
  ho_model m;
  ho_data d;
  ho_cfg cfg;
  ho_expert e;
  int cur_t, start_ind;
  double base_pose[NU], ctrl_scale[NU];
  double obj_vf[3], obj_vt[3], rfc_score, rest_force[3], rest_torque[3];
  double contact_sum[NG][12]; int contact_count[NG];
  double obj_avg_acc[6], geom_avg_vel[NG][3], geom_avg_ang_vel[NG][3];
  int n_avg; double avg_cps[NG][12]; int avg_cp_geom[NG]; double cp_ts[NG];
  int qp_iter;
} ho_env;

/* ------------------------------------------------------------------ helpers (uhc/utils/transformation.py) */
static void quat_inv(const double q[4], double o[4]) { /* transformation.py:1509 quaternion_inverse */
  double n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  o[0] = q[0] / n; o[1] = -q[1] / n; o[2] = -q[2] / n; o[3] = -q[3] / n;
}
static void quat_matrix(const double qin[4], double R[9]) { /* transformation.py:1344 quaternion_matrix */
  double n = qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3];
  if (n < 8.881784197001252e-16) { memset(R, 0, 72); R[0] = R[4] = R[8] = 1; return; }
  double s = sqrt(2.0 / n), q[4] = {qin[0] * s, qin[1] * s, qin[2] * s, qin[3] * s};
  R[0] = 1 - q[2] * q[2] - q[3] * q[3]; R[1] = q[1] * q[2] - q[3] * q[0]; R[2] = q[1] * q[3] + q[2] * q[0];
  R[3] = q[1] * q[2] + q[3] * q[0]; R[4] = 1 - q[1] * q[1] - q[3] * q[3]; R[5] = q[2] * q[3] - q[1] * q[0];
  R[6] = q[1] * q[3] - q[2] * q[0]; R[7] = q[2] * q[3] + q[1] * q[0]; R[8] = 1 - q[1] * q[1] - q[2] * q[2];
}
static void rot_t_vec(const double R[9], const double v[3], double o[3]) { /* math_utils.py:103 transform_vec */
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
         z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void rot6(const double R[9], double* o) { /* R[:, 0:2].flatten() row-major, ho_im4.py:289 */
  o[0] = R[0]; o[1] = R[1]; o[2] = R[3]; o[3] = R[4]; o[4] = R[6]; o[5] = R[7];
}
/* uhc/utils/transforms.py:414 matrix_to_axis_angle = quaternion_to_axis_angle(matrix_to_quaternion) */
static void matrix_to_axis_angle(const double m[9], double aa[3]) {
  double qa[4] = {1.0 + m[0] + m[4] + m[8], 1.0 + m[0] - m[4] - m[8], 1.0 - m[0] + m[4] - m[8], 1.0 - m[0] - m[4] + m[8]};
  int best = 0;
  for (int i = 0; i < 4; i++) qa[i] = qa[i] > 0 ? sqrt(qa[i]) : 0;
  for (int i = 1; i < 4; i++) if (qa[i] > qa[best]) best = i;
  double cand[4][4] = {{qa[0] * qa[0], m[7] - m[5], m[2] - m[6], m[3] - m[1]},
                       {m[7] - m[5], qa[1] * qa[1], m[3] + m[1], m[2] + m[6]},
                       {m[2] - m[6], m[3] + m[1], qa[2] * qa[2], m[5] + m[7]},
                       {m[3] - m[1], m[6] + m[2], m[7] + m[5], qa[3] * qa[3]}};
  double den = 2.0 * fmax(qa[best], 0.1), q[4];
  for (int i = 0; i < 4; i++) q[i] = cand[best][i] / den;
  double nrm = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double half = atan2(nrm, q[0]), ang = 2 * half, s;
  if (fabs(ang) < 1e-6) s = 0.5 - ang * ang / 48; else s = sin(half) / ang;
  for (int i = 0; i < 3; i++) aa[i] = q[1 + i] / s;
}

/* ------------------------------------------------------------------ expert access (ho_im4.py:739-782) */
static int eidx(const ho_env* e, int delta) {
  int ind = e->cur_t + delta + e->start_ind;
  return ind < e->e.T - 1 ? ind : e->e.T - 1;
}

/* ------------------------------------------------------------------ create / configure */
ho_env* hoo_env_create(const void* blob, size_t nbytes) {
  ho_env* e = (ho_env*)calloc(1, sizeof(ho_env));
  if (!e) return NULL;
  if (ho_model_load(&e->m, blob, nbytes) != 0) { free(e); return NULL; }
  ho_data_reset(&e->m, &e->d);
  ho_cfg* c = &e->cfg;
  c->sim_step = 15; c->w_size = 5; c->residual_force = 1; c->explain_force = 1; c->surface_contact = 1;
  c->mode_train = 1; c->pd_rel = 1;
  c->pos_diff_thresh = 0.1; c->rot_diff_thresh = 1.0; c->jpos_diff_thresh = 0.1;
  c->obj_pos_diff_thresh = 0.1; c->obj_rot_diff_thresh = 1.0;
  c->residual_force_scale = 2.5; c->residual_torque_scale = 0.125;
  e->m.pd_wrap_cap = 16;
  /* ho_im4.py:103-107 */
  for (int j = 0; j < e->m.hand_nq; j++) {
    double lo = e->m.jnt_range[j][0], hi = e->m.jnt_range[j][1];
    e->base_pose[j] = 0.5 * (hi + lo);
    e->ctrl_scale[j] = hi - e->base_pose[j];
    if (j >= 6) e->ctrl_scale[j] *= 1.2;
  }
  return e;
}
static void free_expert(ho_expert* x) {
  free(x->hand_dof); free(x->hand_dof_vel); free(x->obj_pose); free(x->obj_vel); free(x->obj_angvel);
  free(x->body_pos); free(x->body_quat);
  memset(x, 0, sizeof(*x));
}
void hoo_env_set_pd_ref_offset(ho_env* e, int off) { e->cfg.pd_ref_offset = off; }
void hoo_env_set_mesh_single_contact(ho_env* e, int on) { e->m.mesh_single_contact = on ? 1 : 0; }
void hoo_env_set_obb_reject(ho_env* e, int on) { e->m.no_obb_reject = on ? 0 : 1; }     /* default on (ho_sim.c ho_collision) */
/* Reference-faithful mode: the two places where this oracle was changed in lock-step with the HIP kernel are switched back to
   what the reference does -- no oriented-box rejection in the collision driver (MuJoCo's broad phase is bounding spheres /
   AABBs; the reject only drops shallow hull contacts of separated geoms) and the unbounded angle wrap of compute_torque.
   tests/test_gpu_parity.py runs whole episodes of the HIP kernel against this mode. */
void hoo_env_set_reference_faithful(ho_env* e, int on) { e->m.no_obb_reject = on ? 1 : 0; e->m.pd_wrap_cap = on ? 0 : 16; }
/* Control arm for the whole-episode parity tests: the state is rounded to float32 after every substep (ho_sim.c euler). */
void hoo_env_set_state_float32(ho_env* e, int on) { e->m.state_float32 = on ? 1 : 0; }
void hoo_env_set_solver_stop(ho_env* e, double tol, int maxit) { e->m.solver_tol = tol; e->m.solver_maxit = maxit; }
void hoo_env_destroy(ho_env* e) { if (e) { free_expert(&e->e); free(e); } }

void hoo_env_set_cfg(ho_env* e, const double* jkp, const double* jkd, const double* torque_lim,
                     const double* thresh5, double rf_scale, double rt_scale, const int* flags7) {
  memcpy(e->cfg.jkp, jkp, sizeof(double) * e->m.nu); memcpy(e->cfg.jkd, jkd, sizeof(double) * e->m.nu);
  memcpy(e->cfg.torque_lim, torque_lim, sizeof(double) * e->m.nu);
  e->cfg.pos_diff_thresh = thresh5[0]; e->cfg.rot_diff_thresh = thresh5[1]; e->cfg.jpos_diff_thresh = thresh5[2];
  e->cfg.obj_pos_diff_thresh = thresh5[3]; e->cfg.obj_rot_diff_thresh = thresh5[4];
  e->cfg.residual_force_scale = rf_scale; e->cfg.residual_torque_scale = rt_scale;
  e->cfg.sim_step = flags7[0]; e->cfg.w_size = flags7[1]; e->cfg.residual_force = flags7[2];
  e->cfg.explain_force = flags7[3]; e->cfg.surface_contact = flags7[4]; e->cfg.mode_train = flags7[5];
  e->cfg.pd_rel = flags7[6];
}
static double* dup(const double* s, size_t n) { double* p = (double*)malloc(n * 8); memcpy(p, s, n * 8); return p; }
/* set_expert (ho_im4.py:135-137); arrays as produced by DatasetSingleDepth.load_seq (dataset_singledepth.py:239) */
void hoo_env_set_expert(ho_env* e, int T, const double* hand_dof, const double* hand_dof_vel, const double* obj_pose,
                        const double* obj_vel, const double* obj_angvel, const double* body_pos, const double* body_quat) {
  free_expert(&e->e);
  int nh = e->m.hand_nq;
  e->e.T = T;
  e->e.hand_dof = dup(hand_dof, (size_t)T * nh); e->e.hand_dof_vel = dup(hand_dof_vel, (size_t)T * nh);
  e->e.obj_pose = dup(obj_pose, (size_t)T * 7); e->e.obj_vel = dup(obj_vel, (size_t)T * 3);
  e->e.obj_angvel = dup(obj_angvel, (size_t)T * 3);
  e->e.body_pos = dup(body_pos, (size_t)T * NHB * 3); e->e.body_quat = dup(body_quat, (size_t)T * NHB * 4);
}

/* ------------------------------------------------------------------ observation (ho_im4.py:280-356, get_full_obs_v5) */
void hoo_env_get_obs(const ho_env* e, double* obs) {
  const ho_model* m = &e->m; const ho_data* d = &e->d; const ho_expert* x = &e->e;
  int nh = m->hand_nq, w = e->cfg.w_size, hb0 = m->hand_body0, o = 0;
  const double* qpos = d->qpos; const double* qvel = d->qvel;
  const double* P = d->xpos[hb0]; const double* Rq = d->xquat[hb0];   /* get_hand_root_pose :819 (lagged body pose) */
  double R[9], Rqi[4], t[3], q[4], q2[4], M[9];
  quat_matrix(Rq, R); quat_inv(Rq, Rqi);
  rot6(R, obs + o); o += 6;
  for (int i = 6; i < nh; i++) obs[o++] = qpos[i];
  for (int k = 1; k <= w; k++) { const double* tq = x->hand_dof + (size_t)eidx(e, k) * nh; for (int i = 6; i < nh; i++) obs[o++] = tq[i] - qpos[i]; }
  for (int i = 6; i < m->hand_nv; i++) obs[o++] = qvel[i];
  rot_t_vec(R, qvel, obs + o); o += 3;
  rot_t_vec(R, qvel + 3, obs + o); o += 3;
  for (int k = 1; k <= w; k++) {
    int ix = eidx(e, k);
    const double* tp = x->body_pos + (size_t)ix * NHB * 3; const double* tq = x->body_quat + (size_t)ix * NHB * 4;
    for (int i = 0; i < 3; i++) t[i] = tp[i] - P[i];
    rot_t_vec(R, t, obs + o); o += 3;
    ho_mulquat(Rqi, tq, q); quat_matrix(q, M); rot6(M, obs + o); o += 6;
  }
  /* transform_vec_batch (math_utils.py:117-130) returns rot.T.dot(v[:, :, None]).squeeze() of shape (3, 20):
     the flattened block is COMPONENT-major: x of the 20 bodies, then y, then z. */
  const int nb1 = NHB - 1;
  for (int b = 1; b < NHB; b++) {
    double r[3];
    for (int i = 0; i < 3; i++) t[i] = d->xpos[hb0 + b][i] - qpos[i];   /* minus qpos[:3], not P (:318) */
    rot_t_vec(R, t, r);
    for (int i = 0; i < 3; i++) obs[o + i * nb1 + (b - 1)] = r[i];
  }
  o += 3 * nb1;
  for (int k = 1; k <= w; k++) {
    const double* tp = x->body_pos + (size_t)eidx(e, k) * NHB * 3;
    for (int b = 1; b < NHB; b++) {
      double r[3];
      for (int i = 0; i < 3; i++) t[i] = tp[3 * b + i] - d->xpos[hb0 + b][i];
      rot_t_vec(R, t, r);
      for (int i = 0; i < 3; i++) obs[o + i * nb1 + (b - 1)] = r[i];
    }
    o += 3 * nb1;
  }
  const double* op = qpos + nh; const double* oq = qpos + nh + 3;
  for (int i = 0; i < 3; i++) t[i] = op[i] - P[i];
  rot_t_vec(R, t, obs + o); o += 3;
  ho_mulquat(Rqi, oq, q); quat_matrix(q, M); rot6(M, obs + o); o += 6;
  rot_t_vec(R, qvel + m->hand_nv, obs + o); o += 3;
  rot_t_vec(R, qvel + m->hand_nv + 3, obs + o); o += 3;
  double oqi[4];
  quat_inv(oq, oqi);
  for (int k = 1; k <= w; k++) {
    const double* tp = x->obj_pose + (size_t)eidx(e, k) * 7;
    for (int i = 0; i < 3; i++) t[i] = tp[i] - op[i];
    rot_t_vec(R, t, obs + o); o += 3;
    ho_mulquat(tp + 3, oqi, q2); ho_mulquat(Rqi, q2, q); quat_matrix(q, M); rot6(M, obs + o); o += 6;
  }
}
int hoo_env_obs_dim(const ho_env* e) { int w = e->cfg.w_size, nf = e->m.hand_nq - 6; return 6 + nf + w * nf + nf + 6 + w * 9 + 60 + w * 60 + 9 + 6 + w * 9; }

/* ------------------------------------------------------------------ PD control (ho_im4.py:412-486, 393-410) */
void hoo_env_compute_torque(const ho_env* e, const double* ctrl, double* torque) {
  const ho_model* m = &e->m; const ho_data* d = &e->d;
  int n = m->hand_nv;
  double dt = m->timestep, target[NU], err[NU], rhs[NU], A[NU * NU];
  const double* ref = e->e.hand_dof + (size_t)eidx(e, e->cfg.pd_ref_offset) * m->hand_nq;
  for (int i = 0; i < 3; i++) target[i] = ref[i] + 0.1 * ctrl[i];
  for (int i = 3; i < 6; i++) target[i] = ref[i] + 0.3 * ctrl[i];
  for (int i = 6; i < n; i++) target[i] = (e->cfg.pd_rel ? ref[i] : e->base_pose[i]) + e->ctrl_scale[i] * ctrl[i];
  for (int i = 0; i < n; i++) err[i] = d->qpos[i] + d->qvel[i] * dt - target[i];
  /* the reference's while loops (ho_im4.py:476-481) with the trip limit of the HIP kernel (hoic_env.h dev_pd_torque): a
     runaway test-mode state (|err| > 33 pi) would otherwise spin for ages here and hang the GPU launch there; both sides
     give the same result for every |err| */
  const int cap = m->pd_wrap_cap > 0 ? m->pd_wrap_cap : 0x7fffffff;      /* reference-faithful mode: the plain while loops */
  for (int i = 3; i < n; i++) {
    for (int k = 0; k < cap && err[i] > M_PI; k++) err[i] -= 2 * M_PI;
    for (int k = 0; k < cap && err[i] < -M_PI; k++) err[i] += 2 * M_PI;
  }
  /* compute_desired_accel: chol(M[:26,:26] + Kd dt)^-1 (-C - Kp e - Kd qd), M and C from the LAST forward pass */
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) A[i * NU + j] = d->qM[i * NV + j];
    A[i * NU + i] += e->cfg.jkd[i] * dt;
    rhs[i] = -d->qfrc_bias[i] - e->cfg.jkp[i] * err[i] - e->cfg.jkd[i] * d->qvel[i];
  }
  ho_cholesky(A, n, NU);
  ho_cholsolve(A, n, NU, rhs);
  for (int i = 0; i < n; i++) torque[i] = -e->cfg.jkp[i] * err[i] - e->cfg.jkd[i] * (d->qvel[i] + rhs[i] * dt);
}

/* ------------------------------------------------------------------ contact bookkeeping (ho_im4.py:883-889, 567-597) */
static void record_contact(ho_env* e) {
  const ho_model* m = &e->m; const ho_data* d = &e->d;
  for (int c = 0; c < d->ncon; c++) {
    int g1 = d->contact[c].geom1, g2 = d->contact[c].geom2;
    if (g1 >= m->hand_geom0 && g1 <= m->hand_geom1 && g2 >= m->obj_geom0 && g2 <= m->obj_geom1) {
      int k = g1 - m->hand_geom0;
      for (int i = 0; i < 3; i++) e->contact_sum[k][i] += d->contact[c].pos[i];
      for (int i = 0; i < 9; i++) e->contact_sum[k][3 + i] += d->contact[c].frame[i];
      e->contact_count[k]++;
    }
  }
}
void hoo_env_classify_contact(ho_env* e) {
  const ho_model* m = &e->m;
  int nhg = m->hand_geom1 - m->hand_geom0 + 1;
  e->n_avg = 0;
  for (int k = 0; k < nhg; k++) {
    if (e->contact_count[k] == 0) continue;
    double* f = e->avg_cps[e->n_avg];
    for (int i = 0; i < 12; i++) f[i] = e->contact_sum[k][i] / e->contact_count[k];
    double* n = f + 3; double* t1 = f + 6; double* t2 = f + 9;
    double nn = sqrt(ho_dot3(n, n));
    for (int i = 0; i < 3; i++) n[i] /= nn;
    double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0};
    if (fabs(n[0]) >= 1e-5) ho_cross(n, ex, t1); else ho_cross(n, ey, t1);
    double tn = sqrt(ho_dot3(t1, t1));
    for (int i = 0; i < 3; i++) t1[i] /= tn;
    ho_cross(n, t1, t2);
    tn = sqrt(ho_dot3(t2, t2));
    for (int i = 0; i < 3; i++) t2[i] /= tn;
    e->avg_cp_geom[e->n_avg] = k + m->hand_geom0;
    e->cp_ts[e->n_avg] = e->contact_count[k];
    e->n_avg++;
  }
}

/* ------------------------------------------------------------------ residual-force explanation QP (ho_im4.py:941-1083)
 * min_x 1/2 x'Qx + p'x, x >= 0, Q = 2(Jf'Jf + w_t Jt'Jt) + 1e-7 I  (:1063-1068, solved by daqp in the reference;
 * qpsolvers/daqp are third-party and absent).  Q = 2 A'A + eps I with A = [Jf; sqrt(w_t) Jt] (6 x n), so the
 * strictly convex problem is solved through its 6-dimensional dual
 *   min_l |l|^2/4 + b'l + 1/(2 eps) sum_i max(0, -(c_i + a_i'l))^2,   residual b - A x = -l/2,
 * by Newton with an exact line search (finite termination; unique optimum = the QP's optimum). */
static void solve_nnqp_dual(int n, const double (*a)[6], const double* c, const double b[6], double eps,
                            double lam[6], int* iters) {
  double s[MAXQP], av[MAXQP];
  for (int i = 0; i < 6; i++) lam[i] = -2 * b[i];
  int it;
  for (it = 0; it < 200; it++) {
    double g[6], H[36], dir[6];
    for (int i = 0; i < 6; i++) g[i] = 0.5 * lam[i] + b[i];
    memset(H, 0, sizeof(H));
    for (int i = 0; i < 6; i++) H[i * 6 + i] = 0.5;
    for (int k = 0; k < n; k++) {
      double sk = c[k];
      for (int i = 0; i < 6; i++) sk += a[k][i] * lam[i];
      s[k] = sk;
      if (sk < 0) {
        for (int i = 0; i < 6; i++) { g[i] += sk / eps * a[k][i]; for (int j = 0; j <= i; j++) H[i * 6 + j] += a[k][i] * a[k][j] / eps; }
      }
    }
    double gn = 0, ln = 0;
    for (int i = 0; i < 6; i++) { gn += g[i] * g[i]; ln += lam[i] * lam[i]; }
    if (sqrt(gn) < 1e-13 * (1 + sqrt(ln))) break;
    ho_cholesky(H, 6, 6);
    for (int i = 0; i < 6; i++) dir[i] = -g[i];
    ho_cholsolve(H, 6, 6, dir);
    double gl = 0, dd = 0, bd = 0;
    for (int i = 0; i < 6; i++) { gl += lam[i] * dir[i]; dd += dir[i] * dir[i]; bd += b[i] * dir[i]; }
    for (int k = 0; k < n; k++) { double v = 0; for (int i = 0; i < 6; i++) v += a[k][i] * dir[i]; av[k] = v; }
    double al = 1, lo = 0, hi = -1;
    for (int ls = 0; ls < 200; ls++) {
      double dphi = 0.5 * (gl + al * dd) + bd, ddphi = 0.5 * dd;
      for (int k = 0; k < n; k++) { double sk = s[k] + al * av[k]; if (sk < 0) { dphi += sk * av[k] / eps; ddphi += av[k] * av[k] / eps; } }
      if (fabs(dphi) < 1e-14 * (1 + fabs(bd) + fabs(gl))) break;
      if (dphi < 0) lo = al; else hi = al;
      double an = al - dphi / ddphi;
      if (hi >= 0 && (an <= lo || an >= hi)) an = 0.5 * (lo + hi);
      if (hi < 0 && an <= lo) an = 2 * al + 1e-12;
      if (hi >= 0 && hi - lo < 1e-16 * (1 + hi)) break;
      al = an;
    }
    double st = 0;
    for (int i = 0; i < 6; i++) { lam[i] += al * dir[i]; st += al * al * dir[i] * dir[i]; }
    if (sqrt(st) < 1e-16 * (1 + sqrt(ln))) { it++; break; }
  }
  *iters = it;
}

void hoo_env_solve_rfc(ho_env* e) {
  const ho_model* m = &e->m; const ho_data* d = &e->d;
  const double dx = 0.0025, mu = 0.75, w_t = 1e4;
  int nq = m->nq, nv = m->nv, ob = m->obj_body, lastg = m->ngeom - 1;
  if (!e->cfg.explain_force) { /* :951-952 */
    memcpy(e->rest_force, e->obj_vf, 24); memcpy(e->rest_torque, e->obj_vt, 24);
    e->rfc_score = sqrt(ho_dot3(e->obj_vf, e->obj_vf)) + w_t * sqrt(ho_dot3(e->obj_vt, e->obj_vt));
    return;
  }
  const double* obj_p = d->qpos + nq - 7;
  double Rm[9], I[9], F[3], tau[3], Iw[3], Ioa[3], wxIw[3];
  quat_matrix(d->qpos + nq - 4, Rm);  /* t3d quat2mat (:963) */
  const double* obj_v = e->geom_avg_vel[lastg]; const double* obj_w = e->geom_avg_ang_vel[lastg];
  const double* obj_a = e->obj_avg_acc; const double* obj_oa = e->obj_avg_acc + 3;
  double mass = m->body_mass[ob];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += Rm[3 * i + k] * m->body_inertia[ob][k] * Rm[3 * j + k];
      I[3 * i + j] = s;
    }
  F[0] = mass * obj_a[0]; F[1] = mass * obj_a[1]; F[2] = mass * (obj_a[2] + 9.8);   /* :975 */
  for (int i = 0; i < 3; i++) { Iw[i] = I[3 * i] * obj_w[0] + I[3 * i + 1] * obj_w[1] + I[3 * i + 2] * obj_w[2];
                                Ioa[i] = I[3 * i] * obj_oa[0] + I[3 * i + 1] * obj_oa[1] + I[3 * i + 2] * obj_oa[2]; }
  ho_cross(obj_w, Iw, wxIw);
  for (int i = 0; i < 3; i++) tau[i] = Ioa[i] + wxIw[i];
  (void)nv;
  if (e->n_avg == 0) { /* :980-981 (note: w_t, not sqrt(w_t)) */
    memcpy(e->rest_force, F, 24); memcpy(e->rest_torque, tau, 24);
    e->rfc_score = sqrt(ho_dot3(F, F)) + w_t * sqrt(ho_dot3(tau, tau));
    return;
  }
  int npt = e->cfg.surface_contact ? 5 : 1, n = 0;
  static double a[MAXQP][6]; static double c[MAXQP];
  double inv = 1.0 / sqrt(1 + mu * mu), swt = sqrt(w_t);
  for (int i = 0; i < e->n_avg; i++) {
    const double* pos = e->avg_cps[i]; const double* fn = e->avg_cps[i] + 3;
    const double* t1 = e->avg_cps[i] + 6; const double* t2 = e->avg_cps[i] + 9;
    int g1 = e->avg_cp_geom[i];
    for (int j = 0; j < npt; j++) {
      double p[3], crh[3], cro[3], cvh[3], cvo[3], tmp[3], rel[3], relt[3];
      const double* dl = (j == 1 || j == 2) ? t1 : t2; double sg = (j == 0) ? 0 : ((j & 1) ? dx : -dx);
      for (int k = 0; k < 3; k++) { p[k] = pos[k] + sg * dl[k]; crh[k] = p[k] - d->geom_xpos[g1][k]; cro[k] = p[k] - obj_p[k]; }
      ho_cross(e->geom_avg_ang_vel[g1], crh, tmp);
      for (int k = 0; k < 3; k++) cvh[k] = e->geom_avg_vel[g1][k] + tmp[k];
      ho_cross(obj_w, cro, tmp);
      for (int k = 0; k < 3; k++) { cvo[k] = obj_v[k] + tmp[k]; rel[k] = cvo[k] - cvh[k]; }
      double vn = ho_dot3(fn, rel);
      for (int k = 0; k < 3; k++) relt[k] = rel[k] - vn * fn[k];
      double nvn = fabs(vn) * sqrt(ho_dot3(fn, fn)), nvt = sqrt(ho_dot3(relt, relt));
      int idx = n / 4;                                   /* expanded point index i in the reference loop :1009 */
      double ts = e->cp_ts[idx / 5] / e->cfg.sim_step;   /* :1012-1013 (cp_ts[i // 5] quirk) */
      double dirs[4] = {-ho_dot3(relt, t1), ho_dot3(relt, t1), -ho_dot3(relt, t2), ho_dot3(relt, t2)};
      int am = 0;
      for (int k = 1; k < 4; k++) if (dirs[k] > dirs[am]) am = k;
      double vn_dir = vn * ho_dot3(fn, fn);              /* dot(rel_vn, frame[0]) :1028 */
      for (int col = 0; col < 4; col++) {
        const double* tt = col < 2 ? t1 : t2; double sgn = (col & 1) ? -1 : 1, xv[3], rx[3];
        for (int k = 0; k < 3; k++) xv[k] = (fn[k] + sgn * mu * tt[k]) * inv * ts;
        ho_cross(cro, xv, rx);
        for (int k = 0; k < 3; k++) { a[n][k] = xv[k]; a[n][3 + k] = swt * rx[k]; }
        c[n] = (vn_dir <= 0 ? nvn : 0) + (col == am ? 0 : nvt);
        n++;
      }
    }
  }
  double b[6] = {F[0], F[1], F[2], swt * tau[0], swt * tau[1], swt * tau[2]}, lam[6];
  solve_nnqp_dual(n, a, c, b, 1e-7, lam, &e->qp_iter);
  for (int i = 0; i < 3; i++) { e->rest_force[i] = -0.5 * lam[i]; e->rest_torque[i] = -0.5 * lam[3 + i] / swt; }
  e->rfc_score = sqrt(ho_dot3(e->rest_force, e->rest_force)) + swt * sqrt(ho_dot3(e->rest_torque, e->rest_torque)); /* :1083 */
}

/* ------------------------------------------------------------------ do_simulation (ho_im4.py:503-565) */
static void do_simulation(ho_env* e, const double* action) {
  ho_model* m = &e->m; ho_data* d = &e->d;
  int nv = m->nv, n = e->cfg.sim_step;
  double old_obj_vel[6], old_xpos[NG][3], old_xmat[NG][9];
  memcpy(old_obj_vel, d->qvel + nv - 6, sizeof(old_obj_vel));
  memcpy(old_xpos, d->geom_xpos, sizeof(old_xpos)); memcpy(old_xmat, d->geom_xmat, sizeof(old_xmat));
  memset(e->contact_sum, 0, sizeof(e->contact_sum)); memset(e->contact_count, 0, sizeof(e->contact_count));
  for (int i = 0; i < n; i++) {
    double torque[NU];
    hoo_env_compute_torque(e, action, torque);                                    /* :518 */
    for (int k = 0; k < m->nu; k++) d->ctrl[k] = fmin(fmax(torque[k], -e->cfg.torque_lim[k]), e->cfg.torque_lim[k]);
    /* gravity compensation at geom_xpos[2] on body 3 (:526-536), kinematics of the last forward pass */
    double f[3] = {0, 0, m->hand_mass * 9.8}, z[3] = {0, 0, 0};
    memset(d->qfrc_applied, 0, sizeof(d->qfrc_applied));
    ho_apply_ft(m, d, f, z, d->geom_xpos[2], 3, d->qfrc_applied);
    if (e->cfg.residual_force)                                                    /* rfc_obj :488-501 */
      ho_apply_ft(m, d, e->obj_vf, e->obj_vt, d->qpos + m->hand_nq, m->obj_body, d->qfrc_applied);
    record_contact(e);                                                            /* :543 */
    ho_step(m, d);                                                                /* :545 */
  }
  double delta_t = n * m->timestep;
  for (int i = 0; i < 6; i++) e->obj_avg_acc[i] = (d->qvel[nv - 6 + i] - old_obj_vel[i]) / delta_t;
  for (int g = 0; g < m->ngeom; g++) {
    for (int i = 0; i < 3; i++) e->geom_avg_vel[g][i] = (d->geom_xpos[g][i] - old_xpos[g][i]) / delta_t;
    double Rd[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += d->geom_xmat[g][3 * i + k] * old_xmat[g][3 * j + k];
        Rd[3 * i + j] = s;
      }
    double aa[3];
    matrix_to_axis_angle(Rd, aa);
    for (int i = 0; i < 3; i++) e->geom_avg_ang_vel[g][i] = aa[i] / delta_t;
  }
  hoo_env_classify_contact(e);
}

/* ------------------------------------------------------------------ termination diffs (ho_im4.py:664-688) */
void hoo_env_calc_ho_diff(const ho_env* e, double out[5]) {
  const ho_model* m = &e->m; const ho_data* d = &e->d;
  int ix = eidx(e, 0), hb0 = m->hand_body0;
  const double* ep = e->e.body_pos + (size_t)ix * NHB * 3; const double* eq = e->e.body_quat + (size_t)ix * NHB * 4;
  double dv[3], s = 0, qi[4], qd[4];
  for (int i = 0; i < 3; i++) dv[i] = d->xpos[hb0][i] - ep[i];
  out[0] = sqrt(ho_dot3(dv, dv));
  for (int b = 0; b < NHB; b++) {
    for (int i = 0; i < 3; i++) dv[i] = d->xpos[hb0 + b][i] - ep[3 * b + i];
    s += sqrt(ho_dot3(dv, dv));
  }
  out[2] = s / NHB;
  quat_inv(d->xquat[hb0], qi); ho_mulquat(eq, qi, qd);
  out[1] = 2.0 * asin(fmin(fmax(sqrt(qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]), 0), 1));
  const double* eo = e->e.obj_pose + (size_t)ix * 7;
  for (int i = 0; i < 3; i++) dv[i] = d->qpos[m->hand_nq + i] - eo[i];
  out[3] = sqrt(ho_dot3(dv, dv));
  quat_inv(eo + 3, qi); ho_mulquat(eo + 3, qi, qd);     /* :685 multiplies the expert quat by its own inverse */
  out[4] = 2.0 * asin(fmin(fmax(sqrt(qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]), 0), 1));
}

/* ------------------------------------------------------------------ reset / step (mujoco_env.py:95-114, ho_im4.py:690-716, 611-662) */
void hoo_env_reset(ho_env* e, int start_ind, double* obs) {
  ho_model* m = &e->m; ho_data* d = &e->d;
  ho_data_reset(m, d);
  e->cur_t = 0; e->start_ind = start_ind;
  int ix = start_ind < e->e.T - 1 ? start_ind : e->e.T - 1, nh = m->hand_nq;
  memcpy(d->qpos, e->e.hand_dof + (size_t)ix * nh, 8 * nh);
  memcpy(d->qpos + nh, e->e.obj_pose + (size_t)ix * 7, 8 * 7);
  memcpy(d->qvel, e->e.hand_dof_vel + (size_t)ix * nh, 8 * nh);
  memcpy(d->qvel + m->hand_nv, e->e.obj_vel + (size_t)ix * 3, 24);
  memcpy(d->qvel + m->hand_nv + 3, e->e.obj_angvel + (size_t)ix * 3, 24);
  ho_forward(m, d);
  e->rfc_score = 0;
  if (obs) hoo_env_get_obs(e, obs);
}

/* info: fail, end, done, percent, rfc_score */
void hoo_env_step(ho_env* e, const double* action_in, double* obs, double* info) {
  const ho_model* m = &e->m;
  double a[HOIC_ACT_DIM];
  for (int i = 0; i < HOIC_ACT_DIM; i++) a[i] = fmin(fmax(action_in[i], -1), 1);
  e->rfc_score = 0;
  for (int i = 0; i < 3; i++) {
    e->obj_vf[i] = e->cfg.residual_force ? e->cfg.residual_force_scale * a[m->nu + i] : 0;
    e->obj_vt[i] = e->cfg.residual_force ? e->cfg.residual_torque_scale * a[m->nu + 3 + i] : 0;
  }
  int fail = 0;
  do_simulation(e, a);
  if (e->d.warning) fail = 1;                      /* MuJoCo exception -> fail (:635-637) */
  else if (e->cfg.residual_force) hoo_env_solve_rfc(e);
  if (!isfinite(e->rfc_score)) { fail = 1; e->rfc_score = 0; }
  e->cur_t += 1;
  double df[5];
  hoo_env_calc_ho_diff(e, df);
  int body_fail = df[0] > e->cfg.pos_diff_thresh || df[1] > e->cfg.rot_diff_thresh || df[2] > e->cfg.jpos_diff_thresh ||
                  df[3] > e->cfg.obj_pos_diff_thresh || df[4] > e->cfg.obj_rot_diff_thresh;
  if (e->cfg.mode_train) fail = fail || body_fail;
  int expert_len = e->e.T - e->start_ind;  /* load_seq slices [start:], reset_model sets start_ind = 0 there */
  int end = e->cur_t >= expert_len - e->cfg.w_size - 1;
  info[0] = fail; info[1] = end; info[2] = fail || end; info[3] = (double)e->cur_t / (expert_len - 1); info[4] = e->rfc_score;
  if (obs) hoo_env_get_obs(e, obs);
}

/* ------------------------------------------------------------------ reward (ho_reward.py:943-1047)
 * wk = [w_p,w_wp,w_v,w_j,w_op,w_or,w_ov,w_orfc, k_p,k_wp,k_v,k_j,k_op,k_or,k_ov,k_orfc]; out = reward + 9 info */
void hoo_env_reward(const ho_env* e, const double* wk, double* out) {
  const ho_model* m = &e->m; const ho_data* d = &e->d;
  int ix = eidx(e, 0), nh = m->hand_nq, hb0 = m->hand_body0;
  const double* eq = e->e.hand_dof + (size_t)ix * nh; const double* ev = e->e.hand_dof_vel + (size_t)ix * nh;
  const double* ebq = e->e.body_quat + (size_t)ix * NHB * 4; const double* ebp = e->e.body_pos + (size_t)ix * NHB * 3;
  double s = 0, qi[4], qd[4], dv[3];
  for (int i = 6; i < nh; i++) s += fabs(d->qpos[i] - eq[i]);
  double pose_r = exp(-wk[8] * s / (nh - 6));
  s = 0;
  for (int b = 0; b < NHB; b++) {   /* multi_quat_diff + multi_quat_norm_v2 (math_utils.py:210-236) */
    quat_inv(ebq + 4 * b, qi); ho_mulquat(d->xquat[hb0 + b], qi, qd);
    double w0 = fabs(qd[0]) - 1.0;
    s += sqrt(w0 * w0 + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]);
  }
  double wpose_r = exp(-wk[9] * s / NHB);
  s = 0;
  for (int i = 0; i < m->hand_nv; i++) s += fabs(d->qvel[i] - ev[i]);
  double vel_r = exp(-wk[10] * s / m->hand_nv);
  s = 0;
  for (int b = 0; b < NHB; b++) { for (int i = 0; i < 3; i++) dv[i] = d->xpos[hb0 + b][i] - ebp[3 * b + i]; s += sqrt(ho_dot3(dv, dv)); }
  double jpos_r = exp(-wk[11] * s / NHB);
  const double* eo = e->e.obj_pose + (size_t)ix * 7;
  for (int i = 0; i < 3; i++) dv[i] = d->qpos[nh + i] - eo[i];
  double opos_r = exp(-wk[12] * sqrt(ho_dot3(dv, dv)));
  quat_inv(eo + 3, qi); ho_mulquat(d->qpos + nh + 3, qi, qd);
  double w0 = fabs(qd[0]) - 1.0;
  double orot_r = exp(-wk[13] * fabs(sqrt(w0 * w0 + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3])));
  s = 0;
  for (int i = 0; i < 3; i++) s += fabs(d->qvel[m->hand_nv + i] - e->e.obj_vel[(size_t)ix * 3 + i]) +
                                   fabs(d->qvel[m->hand_nv + 3 + i] - e->e.obj_angvel[(size_t)ix * 3 + i]);
  double ovel_r = exp(-wk[14] * s / 6);
  double orfc_r = e->cfg.residual_force ? exp(-wk[15] * e->rfc_score) : 1.0;
  double hand = (wk[0] * pose_r + wk[1] * wpose_r + wk[3] * jpos_r + wk[2] * vel_r) / (wk[0] + wk[1] + wk[3] + wk[2]);
  double obj = (wk[4] * opos_r + wk[5] * orot_r + wk[6] * ovel_r + wk[7] * orfc_r) / (wk[4] + wk[5] + wk[6] + wk[7]);
  out[0] = hand * obj;
  out[1] = pose_r; out[2] = wpose_r; out[3] = jpos_r; out[4] = vel_r; out[5] = opos_r; out[6] = orot_r;
  out[7] = ovel_r; out[8] = orfc_r; out[9] = 1.0;
}

/* ------------------------------------------------------------------ generic field access for tests */
typedef struct { const char* name; size_t off; size_t n; int is_int; } field;
#define FD(name) {#name, offsetof(ho_env, d.name), sizeof(((ho_env*)0)->d.name) / 8, 0}
#define FE(name) {#name, offsetof(ho_env, name), sizeof(((ho_env*)0)->name) / 8, 0}
#define FI(name, path) {#name, offsetof(ho_env, path), sizeof(((ho_env*)0)->path) / 4, 1}
#define FM(name) {#name, offsetof(ho_env, m.name), sizeof(((ho_env*)0)->m.name) / 8, 0}
static const field FIELDS[] = {
  FM(body_mass), FM(body_inertia), FM(jnt_range), FM(dof_damping), FM(dof_frictionloss), FM(gravity),
  FD(qpos), FD(qvel), FD(qacc), FD(qacc_warmstart), FD(ctrl), FD(qfrc_applied), FD(xpos), FD(xquat), FD(xmat),
  FD(xipos), FD(ximat), FD(geom_xpos), FD(geom_xmat), FD(qM), FD(qfrc_bias), FD(qfrc_passive), FD(qfrc_smooth),
  FD(qacc_smooth), FD(qfrc_constraint), FD(efc_J), FD(efc_pos), FD(efc_R), FD(efc_D), FD(efc_aref), FD(efc_force),
  FD(efc_KBIP), FD(efc_diagApprox), FD(cvel), FD(S), FD(xanchor), FD(xaxis), FD(solver_gradnorm),
  FE(obj_vf), FE(obj_vt), FE(rfc_score), FE(rest_force), FE(rest_torque), FE(contact_sum), FE(obj_avg_acc),
  FE(geom_avg_vel), FE(geom_avg_ang_vel), FE(avg_cps), FE(cp_ts), FE(base_pose), FE(ctrl_scale),
  FI(ncon, d.ncon), FI(nefc, d.nefc), FI(nf, d.nf), FI(nl, d.nl), FI(solver_iter, d.solver_iter), FI(warning, d.warning),
  FI(cur_t, cur_t), FI(start_ind, start_ind), FI(contact_count, contact_count), FI(n_avg, n_avg),
  FI(avg_cp_geom, avg_cp_geom), FI(efc_type, d.efc_type), FI(efc_id, d.efc_id), FI(qp_iter, qp_iter),
};
static const field* find_field(const char* name) {
  for (size_t i = 0; i < sizeof(FIELDS) / sizeof(FIELDS[0]); i++) if (strcmp(FIELDS[i].name, name) == 0) return &FIELDS[i];
  return NULL;
}
int hoo_get(const ho_env* e, const char* name, void* out, int maxn) {
  const field* f = find_field(name);
  if (!f) return -1;
  int n = (int)f->n < maxn ? (int)f->n : maxn;
  memcpy(out, (const char*)e + f->off, (size_t)n * (f->is_int ? 4 : 8));
  return n;
}
int hoo_set(ho_env* e, const char* name, const void* in, int n) {
  const field* f = find_field(name);
  if (!f || n > (int)f->n) return -1;
  memcpy((char*)e + f->off, in, (size_t)n * (f->is_int ? 4 : 8));
  return n;
}
/* contacts as rows: dist, pos[3], frame[9], geom1, geom2, dim */
int hoo_get_contacts(const ho_env* e, double* out, int maxcon) {
  int n = e->d.ncon < maxcon ? e->d.ncon : maxcon;
  for (int c = 0; c < n; c++) {
    const ho_contact* k = &e->d.contact[c];
    double* r = out + 16 * c;
    r[0] = k->dist; memcpy(r + 1, k->pos, 24); memcpy(r + 4, k->frame, 72); r[13] = k->geom1; r[14] = k->geom2; r[15] = k->dim;
  }
  return n;
}
/* mj_contactForce (mujoco_py functions.mj_contactForce, call site uhc/envs/ho_im4.py:875): the 6-vector of contact c in the
   contact frame decoded from the pyramid's edge forces [MJ-doc: mju_decodePyramid]: force[0] = sum of the edge forces (normal),
   force[1 + i] = mu_i (f_2i - f_2i+1) -- two tangential components, for condim 4 the torsional moment; frictionless: the one row */
int hoo_contact_force(const ho_env* e, int c, double* out6) {
  if (c < 0 || c >= e->d.ncon) return -1;
  const ho_contact* k = &e->d.contact[c];
  const double* f = e->d.efc_force + k->efc_address;
  for (int i = 0; i < 6; i++) out6[i] = 0;
  if (k->dim == 1) { out6[0] = f[0]; return 0; }
  for (int i = 0; i < 2 * (k->dim - 1); i++) out6[0] += f[i];
  for (int i = 0; i < k->dim - 1; i++) out6[1 + i] = k->friction[i] * (f[2 * i] - f[2 * i + 1]);
  return 0;
}
void hoo_forward(ho_env* e) { ho_forward(&e->m, &e->d); }
int hoo_solve_dual_pgs(ho_env* e, int max_sweeps, double tol, double* qacc_out, double* force_out) {
  return ho_solve_dual_pgs(&e->m, &e->d, max_sweeps, tol, qacc_out, force_out);
}
void hoo_step(ho_env* e) { ho_step(&e->m, &e->d); }
void hoo_fwd_position(ho_env* e) { ho_fwd_position(&e->m, &e->d); }
void hoo_do_simulation(ho_env* e, const double* action) { do_simulation(e, action); }
void hoo_record_contact(ho_env* e) { record_contact(e); }
int hoo_sizeof_env(void) { return (int)sizeof(ho_env); }
/* standalone NNQP entry (tests): A is n x 6 row-major */
void hoo_nnqp_dual(int n, const double* a, const double* c, const double* b, double eps, double* lam, int* iters) {
  solve_nnqp_dual(n, (const double(*)[6])a, c, b, eps, lam, iters);
}
