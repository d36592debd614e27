/* ho_sim.c — CPU oracle, physics part (TEST INFRASTRUCTURE).
 *
 * Restates what `self.sim.step()` / `self.sim.forward()` compute for the HOIC model
 * (call sites uhc/envs/ho_im4.py:545, uhc/khrylib/rl/envs/common/mujoco_env.py:114).
 * The arithmetic lives in MuJoCo 2.1.0 (mujoco_py==2.1.2.14, requirements.txt:16), which is not in
 * /root/reference; the stages below follow MuJoCo's published pipeline [MJ-doc]:
 *   kinematics -> composite inertia (CRBA) -> factor -> collision -> constraint rows -> bias (RNE)
 *   -> passive -> actuation -> unconstrained acceleration -> convex constraint solve (Newton)
 *   -> semi-implicit Euler with implicit joint damping.
 * PARITY TO MuJoCo UNPINNED (no MuJoCo here, no reference test at this boundary); pinned only by
 * analytic checks in tests/test_oracle_physics.py.
 */
#include "ho_oracle.h"
#include <math.h>
#include <string.h>
#include <stdio.h>

/* ------------------------------------------------------------------ small math */
double ho_dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
void ho_cross(const double a[3], const double b[3], double o[3]) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
double ho_normalize3(double a[3]) {
  double n = sqrt(ho_dot3(a, a));
  if (n < HO_MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
void ho_quat2mat(const double q[4], double R[9]) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}
void ho_mulquat(const double a[4], const double b[4], double o[4]) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static void normquat(double q[4]) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < HO_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}
static void mat_vec(const double R[9], const double v[3], double o[3]) {
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
/* tangents from a normal stored in frame[0:3]  [MJ-doc: mju_makeFrame] */
void ho_make_frame(double f[9]) {
  double* x = f; double* y = f + 3; double* z = f + 6;
  ho_normalize3(x);
  if (fabs(x[1]) < 0.5) { y[0] = 0; y[1] = 1; y[2] = 0; } else { y[0] = 0; y[1] = 0; y[2] = 1; }
  double dp = ho_dot3(x, y);
  for (int i = 0; i < 3; i++) y[i] -= dp * x[i];
  ho_normalize3(y);
  ho_cross(x, y, z);
}
/* dense Cholesky A = L L^T, lower triangle in place; returns rank deficiency count */
int ho_cholesky(double* A, int n, int lda) {
  int bad = 0;
  for (int j = 0; j < n; j++) {
    double s = A[j * lda + j];
    for (int k = 0; k < j; k++) s -= A[j * lda + k] * A[j * lda + k];
    if (s < HO_MINVAL) { s = HO_MINVAL; bad++; }
    double l = sqrt(s);
    A[j * lda + j] = l;
    for (int i = j + 1; i < n; i++) {
      double t = A[i * lda + j];
      for (int k = 0; k < j; k++) t -= A[i * lda + k] * A[j * lda + k];
      A[i * lda + j] = t / l;
    }
  }
  return bad;
}
void ho_cholsolve(const double* L, int n, int lda, double* x) {
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[i * lda + k] * x[k];
    x[i] = s / L[i * lda + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k * lda + i] * x[k];
    x[i] = s / L[i * lda + i];
  }
}

/* ------------------------------------------------------------------ blob loader */
static const hoic_blob_entry* find_entry(const void* blob, const char* name) {
  const hoic_blob_header* h = (const hoic_blob_header*)blob;
  const hoic_blob_entry* e = (const hoic_blob_entry*)((const char*)blob + sizeof(hoic_blob_header));
  for (int i = 0; i < h->nentries; i++)
    if (strncmp(e[i].name, name, 32) == 0) return &e[i];
  return NULL;
}
static int load_f64(const void* blob, const char* name, double* dst, size_t maxn) {
  const hoic_blob_entry* e = find_entry(blob, name);
  if (!e || e->dtype != 0) { fprintf(stderr, "ho_model_load: missing f64 '%s'\n", name); return -1; }
  size_t n = (size_t)e->nbytes / 8;
  if (n > maxn) { fprintf(stderr, "ho_model_load: '%s' too large (%zu > %zu)\n", name, n, maxn); return -1; }
  memcpy(dst, (const char*)blob + e->offset, n * 8);
  return (int)n;
}
static int load_i32(const void* blob, const char* name, int* dst, size_t maxn) {
  const hoic_blob_entry* e = find_entry(blob, name);
  if (!e || e->dtype != 1) { fprintf(stderr, "ho_model_load: missing i32 '%s'\n", name); return -1; }
  size_t n = (size_t)e->nbytes / 4;
  if (n > maxn) { fprintf(stderr, "ho_model_load: '%s' too large (%zu > %zu)\n", name, n, maxn); return -1; }
  memcpy(dst, (const char*)blob + e->offset, n * 4);
  return (int)n;
}
#define LF(name, field) if (load_f64(blob, name, (double*)(field), sizeof(field) / 8) < 0) return -1
#define LI(name, field) if (load_i32(blob, name, (int*)(field), sizeof(field) / 4) < 0) return -1
#define LI1(name, field) { int t_; if (load_i32(blob, name, &t_, 1) < 0) return -1; (field) = t_; }
#define LF1(name, field) { double t_; if (load_f64(blob, name, &t_, 1) < 0) return -1; (field) = t_; }

int ho_model_load(ho_model* m, const void* blob, size_t nbytes) {
  if (nbytes < sizeof(hoic_blob_header) || memcmp(blob, HOIC_BLOB_MAGIC, 8) != 0) return -1;
  memset(m, 0, sizeof(*m));
  LI1("nbody", m->nbody); LI1("njnt", m->njnt); LI1("nq", m->nq); LI1("nv", m->nv); LI1("nu", m->nu);
  LI1("ngeom", m->ngeom); LI1("npair", m->npair); LI1("nmesh", m->nmesh); LI1("iterations", m->iterations);
  if (m->nbody > NB || m->njnt > NJ || m->nq > NQ || m->nv > NV || m->nu > NU || m->ngeom > NG || m->npair > NP)
    return -1;
  LF1("timestep", m->timestep); LF("gravity", m->gravity); LF1("tolerance", m->tolerance);
  LF1("impratio", m->impratio); LF1("meaninertia", m->meaninertia); LF1("hand_mass", m->hand_mass);
  LF("qpos0", m->qpos0);
  LI("body_parent", m->body_parent); LI("body_jntadr", m->body_jntadr); LI("body_jntnum", m->body_jntnum);
  LI("body_dofadr", m->body_dofadr); LI("body_dofnum", m->body_dofnum); LI("body_lastdof", m->body_lastdof);
  LI("body_weldid", m->body_weldid);
  LF("body_pos", m->body_pos); LF("body_quat", m->body_quat); LF("body_ipos", m->body_ipos);
  LF("body_iquat", m->body_iquat); LF("body_mass", m->body_mass); LF("body_inertia", m->body_inertia);
  LF("body_invweight0", m->body_invweight0);
  LI("jnt_type", m->jnt_type); LI("jnt_bodyid", m->jnt_bodyid); LI("jnt_qposadr", m->jnt_qposadr);
  LI("jnt_dofadr", m->jnt_dofadr); LI("jnt_limited", m->jnt_limited);
  LF("jnt_pos", m->jnt_pos); LF("jnt_axis", m->jnt_axis); LF("jnt_range", m->jnt_range);
  LF("jnt_margin", m->jnt_margin); LF("jnt_solref", m->jnt_solref); LF("jnt_solimp", m->jnt_solimp);
  LI("dof_bodyid", m->dof_bodyid); LI("dof_jntid", m->dof_jntid); LI("dof_parentid", m->dof_parentid);
  LF("dof_armature", m->dof_armature); LF("dof_damping", m->dof_damping);
  LF("dof_frictionloss", m->dof_frictionloss); LF("dof_invweight0", m->dof_invweight0);
  LF("dof_solref", m->dof_solref); LF("dof_solimp", m->dof_solimp);
  LI("geom_type", m->geom_type); LI("geom_bodyid", m->geom_bodyid); LI("geom_meshid", m->geom_meshid);
  LF("geom_size", m->geom_size); LF("geom_pos", m->geom_pos); LF("geom_quat", m->geom_quat);
  LF("geom_rbound", m->geom_rbound);
  LI("act_dofid", m->act_dofid);
  LI("pair_geom1", m->pair_geom1); LI("pair_geom2", m->pair_geom2); LI("pair_condim", m->pair_condim);
  LF("pair_friction", m->pair_friction); LF("pair_solref", m->pair_solref); LF("pair_solimp", m->pair_solimp);
  LF("pair_margin", m->pair_margin); LF("pair_gap", m->pair_gap);
  LI("mesh_vertadr", m->mesh_vertadr); LI("mesh_vertnum", m->mesh_vertnum); LF("mesh_vert", m->mesh_vert);
  LI("mesh_planeadr", m->mesh_planeadr); LI("mesh_planenum", m->mesh_planenum); LF("mesh_plane", m->mesh_plane);
  for (int i = 0; i < m->nmesh && i < HOIC_MAX_MESH; i++) {
    for (int k = 0; k < 3; k++) { m->mesh_aabb[i][k] = 1e300; m->mesh_aabb[i][3 + k] = -1e300; }
    for (int v = m->mesh_vertadr[i]; v < m->mesh_vertadr[i] + m->mesh_vertnum[i]; v++)
      for (int k = 0; k < 3; k++) {
        if (m->mesh_vert[v][k] < m->mesh_aabb[i][k]) m->mesh_aabb[i][k] = m->mesh_vert[v][k];
        if (m->mesh_vert[v][k] > m->mesh_aabb[i][3 + k]) m->mesh_aabb[i][3 + k] = m->mesh_vert[v][k];
      }
  }
  LI1("hand_body0", m->hand_body0); LI1("hand_nbody", m->hand_nbody); LI1("obj_body", m->obj_body);
  LI1("hand_geom0", m->hand_geom0); LI1("hand_geom1", m->hand_geom1);
  LI1("obj_geom0", m->obj_geom0); LI1("obj_geom1", m->obj_geom1);
  LI1("hand_nq", m->hand_nq); LI1("hand_nv", m->hand_nv);
  return 0;
}

/* ------------------------------------------------------------------ reset */
void ho_data_reset(const ho_model* m, ho_data* d) {
  memset(d, 0, sizeof(*d));
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
}

/* ------------------------------------------------------------------ kinematics [MJ-doc: mj_kinematics] */
static void kinematics(const ho_model* m, ho_data* d) {
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  d->xpos[0][0] = d->xpos[0][1] = d->xpos[0][2] = 0;
  ho_quat2mat(d->xquat[0], d->xmat[0]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b], ja = m->body_jntadr[b], jn = m->body_jntnum[b];
    double pos[3], quat[4], R[9], t[3];
    if (jn == 1 && m->jnt_type[ja] == HOIC_JNT_FREE) {
      int qa = m->jnt_qposadr[ja];
      for (int i = 0; i < 3; i++) pos[i] = d->qpos[qa + i];
      for (int i = 0; i < 4; i++) quat[i] = d->qpos[qa + 3 + i];
      normquat(quat);
      ho_quat2mat(quat, R);
      for (int i = 0; i < 3; i++) { d->xanchor[ja][i] = pos[i]; d->xaxis[ja][i] = R[3 * i + 2]; }
    } else {
      mat_vec(d->xmat[p], m->body_pos[b], t);
      for (int i = 0; i < 3; i++) pos[i] = d->xpos[p][i] + t[i];
      ho_mulquat(d->xquat[p], m->body_quat[b], quat);
      for (int j = ja; j < ja + jn; j++) {
        ho_quat2mat(quat, R);
        mat_vec(R, m->jnt_pos[j], t);
        for (int i = 0; i < 3; i++) d->xanchor[j][i] = pos[i] + t[i];
        mat_vec(R, m->jnt_axis[j], d->xaxis[j]);
        double q = d->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]];
        if (m->jnt_type[j] == HOIC_JNT_SLIDE) {
          for (int i = 0; i < 3; i++) pos[i] += d->xaxis[j][i] * q;
        } else { /* hinge: rotate about the joint axis through the anchor */
          double s = sin(0.5 * q), ql[4] = {cos(0.5 * q), s * m->jnt_axis[j][0], s * m->jnt_axis[j][1], s * m->jnt_axis[j][2]};
          double qn[4];
          ho_mulquat(quat, ql, qn);
          memcpy(quat, qn, sizeof(qn));
          ho_quat2mat(quat, R);
          mat_vec(R, m->jnt_pos[j], t);
          for (int i = 0; i < 3; i++) pos[i] = d->xanchor[j][i] - t[i];
        }
      }
      normquat(quat);
    }
    memcpy(d->xpos[b], pos, sizeof(pos));
    memcpy(d->xquat[b], quat, sizeof(quat));
    ho_quat2mat(quat, d->xmat[b]);
    mat_vec(d->xmat[b], m->body_ipos[b], t);
    for (int i = 0; i < 3; i++) d->xipos[b][i] = pos[i] + t[i];
    double qi[4];
    ho_mulquat(quat, m->body_iquat[b], qi);
    ho_quat2mat(qi, d->ximat[b]);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double t[3], q[4];
    mat_vec(d->xmat[b], m->geom_pos[g], t);
    for (int i = 0; i < 3; i++) d->geom_xpos[g][i] = d->xpos[b][i] + t[i];
    ho_mulquat(d->xquat[b], m->geom_quat[g], q);
    ho_quat2mat(q, d->geom_xmat[g]);
  }
}

/* spatial motion axes about the world origin */
static void motion_axes(const ho_model* m, ho_data* d) {
  memset(d->S, 0, sizeof(d->S));
  for (int j = 0; j < m->njnt; j++) {
    int a = m->jnt_dofadr[j];
    if (m->jnt_type[j] == HOIC_JNT_SLIDE) {
      for (int i = 0; i < 3; i++) d->S[a][3 + i] = d->xaxis[j][i];
    } else if (m->jnt_type[j] == HOIC_JNT_HINGE) {
      for (int i = 0; i < 3; i++) d->S[a][i] = d->xaxis[j][i];
      ho_cross(d->xanchor[j], d->xaxis[j], d->S[a] + 3);
    } else { /* free: 3 world translations, then 3 rotations about the body's own axes */
      int b = m->jnt_bodyid[j];
      for (int k = 0; k < 3; k++) {
        d->S[a + k][3 + k] = 1;
        double ax[3] = {d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k]};
        for (int i = 0; i < 3; i++) d->S[a + 3 + k][i] = ax[i];
        ho_cross(d->xpos[b], ax, d->S[a + 3 + k] + 3);
      }
    }
  }
}

/* composite inertia: 10 numbers (m, h = m c, Io sym xx,yy,zz,xy,xz,yz) about the origin */
static void inert_mul(const double I[10], const double v[6], double f[6]) {
  const double* w = v; const double* vo = v + 3;
  double mass = I[0]; const double* h = I + 1;
  double wxh[3], hxv[3];
  ho_cross(w, h, wxh); ho_cross(h, vo, hxv);
  f[0] = I[4] * w[0] + I[7] * w[1] + I[8] * w[2] + hxv[0];
  f[1] = I[7] * w[0] + I[5] * w[1] + I[9] * w[2] + hxv[1];
  f[2] = I[8] * w[0] + I[9] * w[1] + I[6] * w[2] + hxv[2];
  f[3] = mass * vo[0] + wxh[0]; f[4] = mass * vo[1] + wxh[1]; f[5] = mass * vo[2] + wxh[2];
}
static void body_inertia_origin(const ho_model* m, const ho_data* d, int b, double I[10]) {
  double mass = m->body_mass[b]; const double* c = d->xipos[b]; const double* R = d->ximat[b];
  const double* pm = m->body_inertia[b];
  double Ic[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      Ic[3 * i + j] = R[3 * i] * pm[0] * R[3 * j] + R[3 * i + 1] * pm[1] * R[3 * j + 1] + R[3 * i + 2] * pm[2] * R[3 * j + 2];
  double cc = ho_dot3(c, c);
  I[0] = mass; I[1] = mass * c[0]; I[2] = mass * c[1]; I[3] = mass * c[2];
  I[4] = Ic[0] + mass * (cc - c[0] * c[0]); I[5] = Ic[4] + mass * (cc - c[1] * c[1]); I[6] = Ic[8] + mass * (cc - c[2] * c[2]);
  I[7] = Ic[1] - mass * c[0] * c[1]; I[8] = Ic[2] - mass * c[0] * c[2]; I[9] = Ic[5] - mass * c[1] * c[2];
}

/* [MJ-doc: mj_crb + armature] */
static void crb(const ho_model* m, ho_data* d) {
  int nv = m->nv;
  for (int b = 0; b < m->nbody; b++) body_inertia_origin(m, d, b, d->Ic[b]);
  for (int b = m->nbody - 1; b > 0; b--)
    for (int k = 0; k < 10; k++) d->Ic[m->body_parent[b]][k] += d->Ic[b][k];
  memset(d->qM, 0, sizeof(double) * NV * NV);
  for (int i = 0; i < nv; i++) {
    double f[6];
    inert_mul(d->Ic[m->dof_bodyid[i]], d->S[i], f);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double v = 0;
      for (int k = 0; k < 6; k++) v += d->S[j][k] * f[k];
      d->qM[i * NV + j] = d->qM[j * NV + i] = v;
    }
    d->qM[i * NV + i] += m->dof_armature[i];
  }
  memcpy(d->qL, d->qM, sizeof(double) * NV * NV);
  ho_cholesky(d->qL, nv, NV);
}

void ho_fwd_position(const ho_model* m, ho_data* d) {
  kinematics(m, d);
  motion_axes(m, d);
  crb(m, d);
  ho_collision(m, d);
  ho_make_constraint(m, d);
}

/* spatial cross products, vectors are [angular; linear] */
static void cross_motion(const double v[6], const double s[6], double o[6]) {
  double a[3], b[3], c[3];
  ho_cross(v, s, a); ho_cross(v, s + 3, b); ho_cross(v + 3, s, c);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static void cross_force(const double v[6], const double f[6], double o[6]) {
  double a[3], b[3], c[3];
  ho_cross(v, f, a); ho_cross(v + 3, f + 3, b); ho_cross(v, f + 3, c);
  o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}

/* bias = Coriolis + centrifugal + gravity by recursive Newton-Euler with qacc = 0 [MJ-doc: mj_rne] */
static void rne_bias(const ho_model* m, ho_data* d) {
  double cacc[NB][6], cfrc[NB][6];
  memset(cacc, 0, sizeof(cacc)); memset(cfrc, 0, sizeof(cfrc)); memset(d->cvel, 0, sizeof(d->cvel));
  for (int i = 0; i < 3; i++) cacc[0][3 + i] = -m->gravity[i];
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b];
    double v[6], a[6];
    memcpy(v, d->cvel[p], sizeof(v)); memcpy(a, cacc[p], sizeof(a));
    int da = m->body_dofadr[b];
    for (int k = 0; k < m->body_dofnum[b]; k++) {
      int dd = da + k; double sd[6];
      cross_motion(v, d->S[dd], sd);           /* dS/dt seen with the velocity accumulated so far */
      for (int i = 0; i < 6; i++) { a[i] += sd[i] * d->qvel[dd]; }
      for (int i = 0; i < 6; i++) { v[i] += d->S[dd][i] * d->qvel[dd]; }
    }
    memcpy(d->cvel[b], v, sizeof(v)); memcpy(cacc[b], a, sizeof(a));
    double I[10], Iv[6], Ia[6], vxIv[6];
    body_inertia_origin(m, d, b, I);
    inert_mul(I, v, Iv); inert_mul(I, a, Ia); cross_force(v, Iv, vxIv);
    for (int i = 0; i < 6; i++) cfrc[b][i] = Ia[i] + vxIv[i];
  }
  for (int b = m->nbody - 1; b > 0; b--)
    for (int i = 0; i < 6; i++) cfrc[m->body_parent[b]][i] += cfrc[b][i];
  for (int i = 0; i < m->nv; i++) {
    double s = 0;
    for (int k = 0; k < 6; k++) s += d->S[i][k] * cfrc[m->dof_bodyid[i]][k];
    d->qfrc_bias[i] = s;
  }
}

void ho_fwd_velocity(const ho_model* m, ho_data* d) {
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
  rne_bias(m, d);
}

/* [MJ-doc: mj_jac] translational / rotational Jacobian of a world point attached to a body; 3 x nv row-major */
void ho_jac(const ho_model* m, const ho_data* d, double* jacp, double* jacr, const double point[3], int body) {
  int nv = m->nv;
  if (jacp) memset(jacp, 0, sizeof(double) * 3 * nv);
  if (jacr) memset(jacr, 0, sizeof(double) * 3 * nv);
  for (int i = m->body_lastdof[body]; i >= 0; i = m->dof_parentid[i]) {
    double wxp[3];
    ho_cross(d->S[i], point, wxp);
    for (int k = 0; k < 3; k++) {
      if (jacp) jacp[k * nv + i] = wxp[k] + d->S[i][3 + k];
      if (jacr) jacr[k * nv + i] = d->S[i][k];
    }
  }
}

/* [MJ-doc: mj_applyFT] qfrc += Jp^T f + Jr^T t with the kinematics currently stored in d */
void ho_apply_ft(const ho_model* m, const ho_data* d, const double f[3], const double t[3],
                 const double point[3], int body, double* qfrc) {
  double jp[3 * NV], jr[3 * NV];
  int nv = m->nv;
  ho_jac(m, d, jp, jr, point, body);
  for (int i = 0; i < nv; i++)
    for (int k = 0; k < 3; k++) qfrc[i] += jp[k * nv + i] * f[k] + jr[k * nv + i] * t[k];
}

/* Separating-axis test of two oriented boxes, all 15 axes: 1 = along one of them the boxes are more than `gap` apart.
 * Box A: centre pa + Ra ca, axes = columns of Ra, half sizes a; box B: centre pb, axes = columns of Rb, half sizes b. */
static int obb_separated(const double pa[3], const double Ra[9], const double ca[3], const double a[3],
                         const double pb[3], const double Rb[9], const double b[3], double gap) {
  double R[3][3], Q[3][3], t[3], d[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  for (int i = 0; i < 3; i++) {
    t[i] = d[0] * Ra[i] + d[1] * Ra[3 + i] + d[2] * Ra[6 + i] - ca[i];
    for (int j = 0; j < 3; j++) { R[i][j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j]; Q[i][j] = fabs(R[i][j]) + 1e-6; }
  }
  const double g = gap + 1e-6;
  for (int i = 0; i < 3; i++) {
    if (fabs(t[i]) > (a[i] + b[0] * Q[i][0] + b[1] * Q[i][1] + b[2] * Q[i][2] + g) * 1.00001) return 1;
    if (fabs(t[0] * R[0][i] + t[1] * R[1][i] + t[2] * R[2][i]) > (b[i] + a[0] * Q[0][i] + a[1] * Q[1][i] + a[2] * Q[2][i] + g) * 1.00001) return 1;
  }
  for (int i = 0; i < 3; i++) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    for (int j = 0; j < 3; j++) {
      int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      if (fabs(t[i2] * R[i1][j] - t[i1] * R[i2][j]) > (a[i1] * Q[i2][j] + a[i2] * Q[i1][j] + b[j1] * Q[i][j2] + b[j2] * Q[i][j1] + g) * 1.00001) return 1;
    }
  }
  return 0;
}

/* test hook (tests/test_oracle_physics.py checks the test against sampled box-box distances) */
int hoo_obb_separated(const double* pa, const double* Ra, const double* ca, const double* a, const double* pb, const double* Rb,
                      const double* b, double gap) { return obb_separated(pa, Ra, ca, a, pb, Rb, b, gap); }

/* ------------------------------------------------------------------ collision driver */
void ho_collision(const ho_model* m, ho_data* d) {
  d->ncon = 0;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    double margin = m->pair_margin[p];
    /* bounding-sphere rejection (planes are unbounded) [MJ-doc: mj_collideGeoms] */
    if (m->geom_type[g1] != HOIC_GEOM_PLANE && m->geom_type[g2] != HOIC_GEOM_PLANE) {
      double dv[3];
      for (int i = 0; i < 3; i++) dv[i] = d->geom_xpos[g1][i] - d->geom_xpos[g2][i];
      double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
      if (ho_dot3(dv, dv) > bound * bound) continue;
      /* bounding-box rejection: both geoms as oriented boxes (a capsule inside r x r x (l + r) along its axis, a hull
       * inside its bounding box in the mesh frame), 15-axis separating-axis test with the pair's margin.  A pair it drops
       * is farther apart than the margin, so the exact routines (box, capsule) return nothing for it anyway; the hull
       * routines' max-over-face-planes distance under-estimates next to sharp hull vertices (ho_collide.c) and can report
       * a shallow contact for such a pair -- with this test it does not, which is what a geometric collider (the
       * reference's libccd) gives.  Same test, same constants, in the kernel (hoic_collide.h obb_separated). */
      int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
      if (!m->no_obb_reject && (t1 == HOIC_GEOM_CAPSULE || t1 == HOIC_GEOM_BOX) &&
          (t2 == HOIC_GEOM_CAPSULE || t2 == HOIC_GEOM_BOX || t2 == HOIC_GEOM_MESH)) {
        const double *z1 = m->geom_size[g1], *z2 = m->geom_size[g2];
        double h1[3] = {z1[0], t1 == HOIC_GEOM_CAPSULE ? z1[0] : z1[1], t1 == HOIC_GEOM_CAPSULE ? z1[1] + z1[0] : z1[2]};
        double h2[3] = {z2[0], t2 == HOIC_GEOM_CAPSULE ? z2[0] : z2[1], t2 == HOIC_GEOM_CAPSULE ? z2[1] + z2[0] : z2[2]};
        double c2[3] = {0, 0, 0};
        if (t2 == HOIC_GEOM_MESH) {
          const double* bb = m->mesh_aabb[m->geom_meshid[g2]];
          for (int i = 0; i < 3; i++) { c2[i] = 0.5 * (bb[i] + bb[3 + i]); h2[i] = 0.5 * (bb[3 + i] - bb[i]); }
        }
        if (obb_separated(d->geom_xpos[g2], d->geom_xmat[g2], c2, h2, d->geom_xpos[g1], d->geom_xmat[g1], h1, margin)) continue;
      }
    }
    ho_contact tmp[8];
    int n = ho_collide_pair(m, d, p, tmp, 8);
    if (m->mesh_single_contact && n > 1 && m->geom_type[g2] == HOIC_GEOM_MESH) {   /* the deepest point only (first on ties) */
      int b = 0;
      for (int k = 1; k < n; k++) if (tmp[k].dist < tmp[b].dist) b = k;
      tmp[0] = tmp[b]; n = 1;
    }
    for (int k = 0; k < n && d->ncon < HO_MAXCON; k++) {
      ho_contact* c = &d->contact[d->ncon];
      *c = tmp[k];
      if (c->dist >= margin) continue;
      ho_make_frame(c->frame);
      c->geom1 = g1; c->geom2 = g2; c->dim = m->pair_condim[p];
      c->includemargin = margin - m->pair_gap[p];
      memcpy(c->friction, m->pair_friction[p], sizeof(c->friction));
      memcpy(c->solref, m->pair_solref[p], sizeof(c->solref));
      memcpy(c->solimp, m->pair_solimp[p], sizeof(c->solimp));
      d->ncon++;
    }
  }
}

/* ------------------------------------------------------------------ constraints [MJ-doc: mj_makeConstraint/mj_makeImpedance] */
static void get_impedance(const double* solimp_in, double pos, double margin, double* imp, double* impP) {
  double s[5];
  memcpy(s, solimp_in, sizeof(s));
  /* [MJ-doc: getsolparam clamps] */
  for (int i = 0; i < 2; i++) { if (s[i] < 0.0001) s[i] = 0.0001; if (s[i] > 0.9999) s[i] = 0.9999; }
  if (s[2] < 0) s[2] = 0;
  if (s[3] < 0.0001) s[3] = 0.0001; if (s[3] > 0.9999) s[3] = 0.9999;
  if (s[4] < 1) s[4] = 1;
  if (s[0] == s[1] || s[2] <= HO_MINVAL) { *imp = 0.5 * (s[0] + s[1]); *impP = 0; return; }
  double x = (pos - margin) / s[2], sgn = 1;
  if (x < 0) { x = -x; sgn = -1; }
  if (x >= 1 || x <= 0) { *imp = (x >= 1 ? s[1] : s[0]); *impP = 0; return; }
  double y, yP;
  if (s[4] == 1) { y = x; yP = 1; }
  else if (x <= s[3]) { double a = 1 / pow(s[3], s[4] - 1); y = a * pow(x, s[4]); yP = s[4] * a * pow(x, s[4] - 1); }
  else { double b = 1 / pow(1 - s[3], s[4] - 1); y = 1 - b * pow(1 - x, s[4]); yP = s[4] * b * pow(1 - x, s[4] - 1); }
  *imp = s[0] + y * (s[1] - s[0]);
  *impP = yP * sgn * (s[1] - s[0]) / s[2];
}

static void set_kbip(const ho_model* m, ho_data* d, int row, const double* solref_in, const double* solimp,
                     double imp, double impP, int is_friction) {
  double ref[2] = {solref_in[0], solref_in[1]};
  double dmax = solimp[1];
  if (dmax < 0.0001) dmax = 0.0001; if (dmax > 0.9999) dmax = 0.9999;
  if ((ref[0] > 0) != (ref[1] > 0)) { ref[0] = 0.02; ref[1] = 1; }  /* mixed format -> default */
  if (ref[0] > 0 && ref[0] < 2 * m->timestep) ref[0] = 2 * m->timestep; /* refsafe */
  double K, B;
  if (is_friction) K = 0;
  else if (ref[0] > 0) K = 1 / fmax(HO_MINVAL, dmax * dmax * ref[0] * ref[0] * ref[1] * ref[1]);
  else K = -ref[0] / fmax(HO_MINVAL, dmax * dmax);
  if (ref[1] > 0) B = 2 / fmax(HO_MINVAL, dmax * ref[0]);
  else B = -ref[1] / fmax(HO_MINVAL, dmax);
  d->efc_KBIP[row][0] = K; d->efc_KBIP[row][1] = B; d->efc_KBIP[row][2] = imp; d->efc_KBIP[row][3] = impP;
}

void ho_make_constraint(const ho_model* m, ho_data* d) {
  int nv = m->nv, n = 0;
  /* 1. dof friction loss rows */
  for (int i = 0; i < nv; i++) {
    if (m->dof_frictionloss[i] <= 0) continue;
    memset(d->efc_J[n], 0, sizeof(double) * NV);
    d->efc_J[n][i] = 1;
    d->efc_type[n] = HO_EFC_FRICTION; d->efc_id[n] = i;
    d->efc_pos[n] = 0; d->efc_margin[n] = 0; d->efc_frictionloss[n] = m->dof_frictionloss[i];
    d->efc_diagApprox[n] = m->dof_invweight0[i];
    double imp, impP;
    get_impedance(m->dof_solimp[i], 0, 0, &imp, &impP);
    d->efc_R[n] = fmax(HO_MINVAL, (1 - imp) * d->efc_diagApprox[n] / imp);
    set_kbip(m, d, n, m->dof_solref[i], m->dof_solimp[i], imp, impP, 1);
    n++;
  }
  d->nf = n;
  /* 2. joint limit rows (slide / hinge) */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || m->jnt_type[j] == HOIC_JNT_FREE) continue;
    double q = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[j][side < 0 ? 0 : 1] - q);
      if (dist >= margin) continue;
      memset(d->efc_J[n], 0, sizeof(double) * NV);
      d->efc_J[n][m->jnt_dofadr[j]] = -side;
      d->efc_type[n] = HO_EFC_LIMIT; d->efc_id[n] = j;
      d->efc_pos[n] = dist; d->efc_margin[n] = margin; d->efc_frictionloss[n] = 0;
      d->efc_diagApprox[n] = m->dof_invweight0[m->jnt_dofadr[j]];
      double imp, impP;
      get_impedance(m->jnt_solimp[j], dist, margin, &imp, &impP);
      d->efc_R[n] = fmax(HO_MINVAL, (1 - imp) * d->efc_diagApprox[n] / imp);
      set_kbip(m, d, n, m->jnt_solref[j], m->jnt_solimp[j], imp, impP, 0);
      n++;
    }
  }
  d->nl = n - d->nf;
  /* 3. contacts: frictionless (condim 1) or pyramidal */
  for (int c = 0; c < d->ncon; c++) {
    ho_contact* con = &d->contact[c];
    con->efc_address = n;
    int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2];
    double jp1[3 * NV], jr1[3 * NV], jp2[3 * NV], jr2[3 * NV], jd[6][NV];
    ho_jac(m, d, jp1, jr1, con->pos, b1);
    ho_jac(m, d, jp2, jr2, con->pos, b2);
    /* relative Jacobian in the contact frame: rows 0..2 translational (n,t1,t2), 3..5 rotational */
    for (int r = 0; r < 3; r++)
      for (int i = 0; i < nv; i++) {
        double sp = 0, sr = 0;
        for (int k = 0; k < 3; k++) {
          sp += con->frame[3 * r + k] * (jp2[k * nv + i] - jp1[k * nv + i]);
          sr += con->frame[3 * r + k] * (jr2[k * nv + i] - jr1[k * nv + i]);
        }
        jd[r][i] = sp; jd[3 + r][i] = sr;
      }
    double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
    double rot = m->body_invweight0[b1][1] + m->body_invweight0[b2][1];
    double imp, impP;
    get_impedance(con->solimp, con->dist, con->includemargin, &imp, &impP);
    int nrow = con->dim == 1 ? 1 : 2 * (con->dim - 1);
    if (n + nrow > HO_MAXEFC) break;
    for (int r = 0; r < nrow; r++) {
      int row = n + r;
      memset(d->efc_J[row], 0, sizeof(double) * NV);
      if (con->dim == 1) {
        for (int i = 0; i < nv; i++) d->efc_J[row][i] = jd[0][i];
        d->efc_diagApprox[row] = tran;
      } else {
        int k = 1 + r / 2; double sgn = (r & 1) ? -1 : 1, fri = con->friction[r / 2];
        for (int i = 0; i < nv; i++) d->efc_J[row][i] = jd[0][i] + sgn * fri * jd[k][i];
        d->efc_diagApprox[row] = tran + fri * fri * (r < 4 ? tran : rot);
      }
      d->efc_type[row] = HO_EFC_CONTACT; d->efc_id[row] = c;
      d->efc_pos[row] = con->dist; d->efc_margin[row] = con->includemargin; d->efc_frictionloss[row] = 0;
      d->efc_R[row] = fmax(HO_MINVAL, (1 - imp) * d->efc_diagApprox[row] / imp);
      set_kbip(m, d, row, con->solref, con->solimp, imp, impP, 0);
    }
    if (con->dim > 1) { /* pyramidal: one regulariser for all edges, Rpy = 2 mu^2 R[0], mu = friction[0]/sqrt(impratio) */
      double mu = con->friction[0] * sqrt(1.0 / fmax(HO_MINVAL, m->impratio));
      double Rpy = 2 * mu * mu * d->efc_R[n];
      for (int r = 0; r < nrow; r++) d->efc_R[n + r] = Rpy;
    }
    n += nrow;
  }
  d->nefc = n;
  for (int i = 0; i < n; i++) d->efc_D[i] = 1.0 / d->efc_R[i];
}

/* reference acceleration [MJ-doc: mj_referenceConstraint] */
static void reference_constraint(const ho_model* m, ho_data* d) {
  for (int i = 0; i < d->nefc; i++) {
    double vel = 0;
    for (int k = 0; k < m->nv; k++) vel += d->efc_J[i][k] * d->qvel[k];
    d->efc_aref[i] = -d->efc_KBIP[i][1] * vel - d->efc_KBIP[i][0] * d->efc_KBIP[i][2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
}

/* per-row cost s_i(jar), its negative derivative (force) and curvature */
static inline double row_cost(const ho_data* d, int i, double jar, double* force, double* curv) {
  double D = d->efc_D[i];
  if (d->efc_type[i] == HO_EFC_FRICTION) {
    double f = d->efc_frictionloss[i], R = d->efc_R[i];
    if (jar <= -R * f) { *force = f; *curv = 0; return -f * (0.5 * R * f + jar); }
    if (jar >= R * f) { *force = -f; *curv = 0; return -f * (0.5 * R * f - jar); }
    *force = -D * jar; *curv = D; return 0.5 * D * jar * jar;
  }
  if (jar < 0) { *force = -D * jar; *curv = D; return 0.5 * D * jar * jar; }
  *force = 0; *curv = 0; return 0;
}

static double total_cost(const ho_model* m, ho_data* d, const double* qacc, double* jar_out) {
  int nv = m->nv;
  double cost = 0;
  for (int i = 0; i < nv; i++) {
    double Ma = 0;
    for (int k = 0; k < nv; k++) Ma += d->qM[i * NV + k] * qacc[k];
    cost += 0.5 * (Ma - d->qfrc_smooth[i]) * (qacc[i] - d->qacc_smooth[i]);
  }
  for (int i = 0; i < d->nefc; i++) {
    double jar = -d->efc_aref[i], f, c;
    for (int k = 0; k < nv; k++) jar += d->efc_J[i][k] * qacc[k];
    if (jar_out) jar_out[i] = jar;
    cost += row_cost(d, i, jar, &f, &c);
  }
  return cost;
}

/* Newton on the primal convex problem  min_a 1/2 (a-a0)' M (a-a0) + sum_i s_i(J_i a - aref_i)
 * [MJ-doc: Computation chapter, "Newton solver"]; run to tight convergence (unique optimum). */
void ho_solve(const ho_model* m, ho_data* d) {
  int nv = m->nv, ne = d->nefc;
  const int MAXIT = m->solver_maxit > 0 ? m->solver_maxit : 100;
  const double gtol = m->solver_tol > 0 ? m->solver_tol : 1e-14;
  double qacc[NV], jar[HO_MAXEFC], jv[HO_MAXEFC], grad[NV], search[NV], Mv[NV], H[NV * NV];
  memset(d->qfrc_constraint, 0, sizeof(d->qfrc_constraint));
  d->solver_iter = 0; d->solver_gradnorm = 0;
  if (ne == 0) { memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv); return; }
  /* warm start: previous qacc if it is cheaper than the unconstrained acceleration */
  double cw = total_cost(m, d, d->qacc_warmstart, NULL), cs = total_cost(m, d, d->qacc_smooth, NULL);
  memcpy(qacc, cw < cs ? d->qacc_warmstart : d->qacc_smooth, sizeof(double) * nv);
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  for (int it = 0; it < MAXIT; it++) {
    /* gradient and Hessian at qacc */
    for (int i = 0; i < nv; i++) {
      double Ma = 0;
      for (int k = 0; k < nv; k++) Ma += d->qM[i * NV + k] * qacc[k];
      grad[i] = Ma - d->qfrc_smooth[i];
    }
    memcpy(H, d->qM, sizeof(double) * NV * NV);
    for (int i = 0; i < ne; i++) {
      double j = -d->efc_aref[i], f, c;
      for (int k = 0; k < nv; k++) j += d->efc_J[i][k] * qacc[k];
      jar[i] = j;
      row_cost(d, i, j, &f, &c);
      d->efc_force[i] = f;
      if (f != 0) for (int k = 0; k < nv; k++) grad[k] -= d->efc_J[i][k] * f;
      if (c != 0)
        for (int a = 0; a < nv; a++) {
          double ja = d->efc_J[i][a];
          if (ja == 0) continue;
          for (int b = 0; b <= a; b++) H[a * NV + b] += c * ja * d->efc_J[i][b];
        }
    }
    double gn = 0;
    for (int i = 0; i < nv; i++) gn += grad[i] * grad[i];
    gn = sqrt(gn) * scale;
    d->solver_gradnorm = gn; d->solver_iter = it;
    if (gn < gtol) break;
    ho_cholesky(H, nv, NV);
    for (int i = 0; i < nv; i++) search[i] = -grad[i];
    ho_cholsolve(H, nv, NV, search);
    /* exact line search on the piecewise-quadratic phi(alpha) */
    double g0 = 0, h0 = 0;
    for (int i = 0; i < nv; i++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->qM[i * NV + k] * search[k];
      Mv[i] = s; g0 += grad[i] * search[i]; h0 += search[i] * s;
    }
    /* the Gauss part of dphi/dalpha at alpha: (M(a-a0)).s + alpha s'Ms; grad[] above already holds the
       constraint part at alpha = 0, so rebuild it separately */
    double gq0 = 0;
    for (int i = 0; i < nv; i++) {
      double Ma = 0;
      for (int k = 0; k < nv; k++) Ma += d->qM[i * NV + k] * qacc[k];
      gq0 += (Ma - d->qfrc_smooth[i]) * search[i];
    }
    for (int i = 0; i < ne; i++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->efc_J[i][k] * search[k];
      jv[i] = s;
    }
    double lo = 0, hi = -1, alpha = 0, dlo = g0;
    (void)dlo;
    double a = 0;
    for (int ls = 0; ls < 100; ls++) {
      double dphi = gq0 + a * h0, ddphi = h0;
      for (int i = 0; i < ne; i++) {
        double f, c;
        row_cost(d, i, jar[i] + a * jv[i], &f, &c);
        dphi -= f * jv[i]; ddphi += c * jv[i] * jv[i];
      }
      alpha = a;
      if (fabs(dphi) < 1e-15 * (1 + fabs(g0))) break;
      if (dphi < 0) lo = a; else hi = a;
      double an = a - dphi / ddphi;
      if (hi >= 0 && (an <= lo || an >= hi)) an = 0.5 * (lo + hi);
      if (hi >= 0 && hi - lo < 1e-16 * (1 + hi)) break;
      a = an;
    }
    double step2 = 0;
    for (int i = 0; i < nv; i++) { qacc[i] += alpha * search[i]; step2 += alpha * alpha * search[i] * search[i]; }
    if (sqrt(step2) * scale < 1e-16) { d->solver_iter = it + 1; break; }
  }
  /* final forces */
  for (int i = 0; i < ne; i++) {
    double j = -d->efc_aref[i], f, c;
    for (int k = 0; k < nv; k++) j += d->efc_J[i][k] * qacc[k];
    row_cost(d, i, j, &f, &c);
    d->efc_force[i] = f;
    for (int k = 0; k < nv; k++) d->qfrc_constraint[k] += d->efc_J[i][k] * f;
  }
  memcpy(d->qacc, qacc, sizeof(double) * nv);
}

/* ------------------------------------------------------------------ independent second solver (dual PGS)
 * The same constraint problem in its DUAL form, which is what a projected Gauss-Seidel kernel iterates and what the
 * north-star text names [MJ-doc: Computation chapter, "PGS solver"]:
 *     find f in the per-row box  ([-eta, eta] friction loss, [0, inf) limits and pyramidal contact edges)  with
 *     (A + diag R) f + b  complementary,   A = J M^-1 J',  b = J a0 - aref,   then  qacc = a0 + M^-1 J' f.
 * Shares nothing with ho_solve but the rows: no cost function, no Hessian, no line search.  The primal problem is
 * strictly convex, so both must reach the same qacc; tests/test_oracle_physics.py asserts it — protection against a
 * common-mode error in the Newton solver that every other parity test would inherit.
 * Call after ho_forward (rows, aref, qacc_smooth, qL are taken from d).  Returns the number of sweeps used. */
int ho_solve_dual_pgs(const ho_model* m, const ho_data* d, int max_sweeps, double tol, double* qacc_out, double* force_out) {
  int nv = m->nv, ne = d->nefc;
  static double A[HO_MAXEFC][HO_MAXEFC], MinvJt[HO_MAXEFC][NV];
  double b[HO_MAXEFC], f[HO_MAXEFC];
  memcpy(qacc_out, d->qacc_smooth, sizeof(double) * nv);
  if (ne == 0) return 0;
  for (int i = 0; i < ne; i++) {
    memcpy(MinvJt[i], d->efc_J[i], sizeof(double) * NV);
    ho_cholsolve(d->qL, nv, NV, MinvJt[i]);
    double s = -d->efc_aref[i];
    for (int k = 0; k < nv; k++) s += d->efc_J[i][k] * d->qacc_smooth[k];
    b[i] = s; f[i] = 0;
  }
  for (int i = 0; i < ne; i++)
    for (int j = 0; j <= i; j++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->efc_J[i][k] * MinvJt[j][k];
      A[i][j] = A[j][i] = s;
    }
  int sweep = 0;
  for (; sweep < max_sweeps; sweep++) {
    double change = 0, mag = 0;
    for (int i = 0; i < ne; i++) {
      double r = b[i] + d->efc_R[i] * f[i];
      for (int j = 0; j < ne; j++) r += A[i][j] * f[j];
      double fn = f[i] - r / (A[i][i] + d->efc_R[i]);
      if (d->efc_type[i] == HO_EFC_FRICTION) {
        double eta = d->efc_frictionloss[i];
        fn = fn < -eta ? -eta : (fn > eta ? eta : fn);
      } else if (fn < 0) fn = 0;
      change = fmax(change, fabs(fn - f[i])); mag = fmax(mag, fabs(fn));
      f[i] = fn;
    }
    if (change <= tol * (1e-300 + mag)) { sweep++; break; }
  }
  for (int i = 0; i < ne; i++) {
    if (force_out) force_out[i] = f[i];
    for (int k = 0; k < nv; k++) qacc_out[k] += MinvJt[i][k] * f[i];
  }
  return sweep;
}

/* ------------------------------------------------------------------ forward / step */
static int finite_vec(const double* v, int n) {
  for (int i = 0; i < n; i++) if (!isfinite(v[i]) || fabs(v[i]) > 1e10) return 0;
  return 1;
}

void ho_forward(const ho_model* m, ho_data* d) {
  int nv = m->nv;
  ho_fwd_position(m, d);
  ho_fwd_velocity(m, d);
  reference_constraint(m, d);
  memset(d->qfrc_actuator, 0, sizeof(d->qfrc_actuator));
  for (int u = 0; u < m->nu; u++) d->qfrc_actuator[m->act_dofid[u]] += d->ctrl[u]; /* motor, gear 1 */
  for (int i = 0; i < nv; i++) {
    d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_applied[i] + d->qfrc_actuator[i];
    d->qacc_smooth[i] = d->qfrc_smooth[i];
  }
  ho_cholsolve(d->qL, nv, NV, d->qacc_smooth);
  ho_solve(m, d);
}

/* [MJ-doc: mj_Euler] velocity update implicit in joint damping, then position update */
static void euler(const ho_model* m, ho_data* d) {
  int nv = m->nv;
  double h = m->timestep, qacc[NV];
  int damped = 0;
  for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) damped = 1;
  if (!damped) memcpy(qacc, d->qacc, sizeof(double) * nv);
  else {
    double A[NV * NV];
    memcpy(A, d->qM, sizeof(A));
    for (int i = 0; i < nv; i++) { A[i * NV + i] += h * m->dof_damping[i]; qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i]; }
    ho_cholesky(A, nv, NV);
    ho_cholsolve(A, nv, NV, qacc);
  }
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == HOIC_JNT_FREE) {
      for (int i = 0; i < 3; i++) d->qpos[qa + i] += h * d->qvel[da + i];
      double w[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]};  /* body frame */
      double ang = ho_normalize3(w) * h;
      if (ang != 0) {
        double s = sin(0.5 * ang), dq[4] = {cos(0.5 * ang), s * w[0], s * w[1], s * w[2]}, qn[4];
        ho_mulquat(d->qpos + qa + 3, dq, qn);
        normquat(qn);
        memcpy(d->qpos + qa + 3, qn, sizeof(qn));
      }
    } else d->qpos[qa] += h * d->qvel[da];
  }
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
  if (m->state_float32) {      /* control arm: the state a float32 simulator would carry to the next substep */
    for (int i = 0; i < m->nq; i++) d->qpos[i] = (double)(float)d->qpos[i];
    for (int i = 0; i < nv; i++) { d->qvel[i] = (double)(float)d->qvel[i]; d->qacc_warmstart[i] = (double)(float)d->qacc_warmstart[i]; }
  }
}

void ho_step(const ho_model* m, ho_data* d) {
  if (!finite_vec(d->qpos, m->nq) || !finite_vec(d->qvel, m->nv)) { d->warning = 1; return; }
  ho_forward(m, d);
  if (!finite_vec(d->qacc, m->nv)) { d->warning = 1; return; }
  euler(m, d);
}
