// hoic_dynamics.h — wave-cooperative kinematics, composite-inertia mass matrix and bias forces.
//
// Replaces the position/velocity stages of mj_forward that the reference runs through
// self.sim.step() / self.sim.forward() (uhc/envs/ho_im4.py:545, mujoco_env.py:114): kinematics,
// mass matrix (data.qM, read back by ho_im4.py:398) and qfrc_bias (ho_im4.py:401).
// Design: lanes = bodies walked level by level down the kinematic tree; subtree sums use the
// depth-first body order (a subtree is an index range), so no atomics and no recursion.
// Spatial vectors are [angular; linear-at-world-origin].
#pragma once
#include "hoic_types.h"
#include "hoic_math.h"

// ---- kinematics: body frames, joint anchors/axes, geoms, motion axes S, body inertias about the origin
__device__ void dev_kinematics(const DevModel& m, Work& w, const float* q) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    w.xpos[0][0] = w.xpos[0][1] = w.xpos[0][2] = 0.f;
    w.xquat[0][0] = 1.f; w.xquat[0][1] = w.xquat[0][2] = w.xquat[0][3] = 0.f;
    for (int i = 0; i < 9; i++) w.xmat[0][i] = (i % 4 == 0) ? 1.f : 0.f;
  }
  __syncthreads();
  for (int lev = 1; lev <= m.nlevel; lev++) {
    if (tid < m.nbody && m.body_depth[tid] == lev) {
      const int b = tid, p = m.body_parent[b], ja = m.body_jntadr[b], jn = m.body_jntnum[b];
      float pos[3], quat[4], R[9], t[3];
      if (jn == 1 && m.jnt_type[ja] == HOIC_JNT_FREE) {
        const int qa = m.jnt_qposadr[ja];
        for (int i = 0; i < 3; i++) pos[i] = q[qa + i];
        for (int i = 0; i < 4; i++) quat[i] = q[qa + 3 + i];
        normquat(quat);
        quat2mat(quat, R);
        for (int i = 0; i < 3; i++) { w.xanchor[ja][i] = pos[i]; w.xaxis[ja][i] = R[3 * i + 2]; }
      } else {
        matvec(w.xmat[p], m.body_pos[b], t);
        for (int i = 0; i < 3; i++) pos[i] = w.xpos[p][i] + t[i];
        mulquat(w.xquat[p], m.body_quat[b], quat);
        for (int j = ja; j < ja + jn; j++) {
          quat2mat(quat, R);
          matvec(R, m.jnt_pos[j], t);
          float anchor[3], axis[3];
          for (int i = 0; i < 3; i++) anchor[i] = pos[i] + t[i];
          matvec(R, m.jnt_axis[j], axis);
          for (int i = 0; i < 3; i++) { w.xanchor[j][i] = anchor[i]; w.xaxis[j][i] = axis[i]; }
          const float qq = q[m.jnt_qposadr[j]] - m.qpos0[m.jnt_qposadr[j]];
          if (m.jnt_type[j] == HOIC_JNT_SLIDE) {
            for (int i = 0; i < 3; i++) pos[i] += axis[i] * qq;
          } else {
            float s, c;
            sincosf(0.5f * qq, &s, &c);
            float ql[4] = {c, s * m.jnt_axis[j][0], s * m.jnt_axis[j][1], s * m.jnt_axis[j][2]}, qn[4];
            mulquat(quat, ql, qn);
            for (int i = 0; i < 4; i++) quat[i] = qn[i];
            quat2mat(quat, R);
            matvec(R, m.jnt_pos[j], t);
            for (int i = 0; i < 3; i++) pos[i] = anchor[i] - t[i];
          }
        }
        normquat(quat);
        quat2mat(quat, R);
      }
      for (int i = 0; i < 3; i++) w.xpos[b][i] = pos[i];
      for (int i = 0; i < 4; i++) w.xquat[b][i] = quat[i];
      for (int i = 0; i < 9; i++) w.xmat[b][i] = R[i];
    }
    __syncthreads();
  }
  // per body: inertial frame and spatial inertia about the world origin (m, h = m c, Io: xx yy zz xy xz yz)
  if (tid < m.nbody) {
    const int b = tid;
    float c[3], t[3], qi[4], Ri[9];
    matvec(w.xmat[b], m.body_ipos[b], t);
    for (int i = 0; i < 3; i++) { c[i] = w.xpos[b][i] + t[i]; w.xipos[b][i] = c[i]; }
    mulquat(w.xquat[b], m.body_iquat[b], qi);
    quat2mat(qi, Ri);
    const float mass = m.body_mass[b], p0 = m.body_inertia[b][0], p1 = m.body_inertia[b][1], p2 = m.body_inertia[b][2];
    float Ic[6];  // xx yy zz xy xz yz about the centre of mass
    Ic[0] = Ri[0] * p0 * Ri[0] + Ri[1] * p1 * Ri[1] + Ri[2] * p2 * Ri[2];
    Ic[1] = Ri[3] * p0 * Ri[3] + Ri[4] * p1 * Ri[4] + Ri[5] * p2 * Ri[5];
    Ic[2] = Ri[6] * p0 * Ri[6] + Ri[7] * p1 * Ri[7] + Ri[8] * p2 * Ri[8];
    Ic[3] = Ri[0] * p0 * Ri[3] + Ri[1] * p1 * Ri[4] + Ri[2] * p2 * Ri[5];
    Ic[4] = Ri[0] * p0 * Ri[6] + Ri[1] * p1 * Ri[7] + Ri[2] * p2 * Ri[8];
    Ic[5] = Ri[3] * p0 * Ri[6] + Ri[4] * p1 * Ri[7] + Ri[5] * p2 * Ri[8];
    const float cc = dot3(c, c);
    float* I = w.I10[b];
    I[0] = mass; I[1] = mass * c[0]; I[2] = mass * c[1]; I[3] = mass * c[2];
    I[4] = Ic[0] + mass * (cc - c[0] * c[0]); I[5] = Ic[1] + mass * (cc - c[1] * c[1]); I[6] = Ic[2] + mass * (cc - c[2] * c[2]);
    I[7] = Ic[3] - mass * c[0] * c[1]; I[8] = Ic[4] - mass * c[0] * c[2]; I[9] = Ic[5] - mass * c[1] * c[2];
  }
  // per geom (second half-wave so it overlaps the body work)
  {
    const int g = tid - 32;
    if (g >= 0 && g < m.ngeom) {
      const int b = m.geom_bodyid[g];
      float t[3], qg[4];
      matvec(w.xmat[b], m.geom_pos[g], t);
      for (int i = 0; i < 3; i++) w.gxpos[g][i] = w.xpos[b][i] + t[i];
      mulquat(w.xquat[b], m.geom_quat[g], qg);
      quat2mat(qg, w.gxmat[g]);
    }
  }
  // per dof: motion axis about the origin
  if (tid < m.nv) {
    const int d = tid, j = m.dof_jntid[d], ty = m.jnt_type[j];
    float* S = w.S[d];
    if (ty == HOIC_JNT_SLIDE) {
      S[0] = S[1] = S[2] = 0.f;
      for (int i = 0; i < 3; i++) S[3 + i] = w.xaxis[j][i];
    } else if (ty == HOIC_JNT_HINGE) {
      for (int i = 0; i < 3; i++) S[i] = w.xaxis[j][i];
      cross3(w.xanchor[j], w.xaxis[j], S + 3);
    } else {
      const int k = d - m.jnt_dofadr[j], b = m.jnt_bodyid[j];
      if (k < 3) { for (int i = 0; i < 6; i++) S[i] = 0.f; S[3 + k] = 1.f; }
      else {
        float ax[3] = {w.xmat[b][k - 3], w.xmat[b][3 + k - 3], w.xmat[b][6 + k - 3]};
        for (int i = 0; i < 3; i++) S[i] = ax[i];
        cross3(w.xpos[b], ax, S + 3);
      }
    }
  }
  __syncthreads();
}

HD void inert_mul(const float* I, const float* v, float* f) {
  const float* wv = v; const float* vo = v + 3; const float* h = I + 1;
  float wxh[3], hxv[3];
  cross3(wv, h, wxh); cross3(h, vo, hxv);
  f[0] = I[4] * wv[0] + I[7] * wv[1] + I[8] * wv[2] + hxv[0];
  f[1] = I[7] * wv[0] + I[5] * wv[1] + I[9] * wv[2] + hxv[1];
  f[2] = I[8] * wv[0] + I[9] * wv[1] + I[6] * wv[2] + hxv[2];
  f[3] = I[0] * vo[0] + wxh[0]; f[4] = I[0] * vo[1] + wxh[1]; f[5] = I[0] * vo[2] + wxh[2];
}

// ---- joint-space inertia: composite rigid body sums over index ranges, then the tree-sparse entries of M
__device__ void dev_mass_matrix(const DevModel& m, Work& w) {
  const int tid = threadIdx.x;
  if (tid < m.nbody) {
    float acc[10];
    for (int k = 0; k < 10; k++) acc[k] = 0.f;
    const int e = tid + m.body_subtree[tid];
    for (int b = tid; b < e; b++)
      for (int k = 0; k < 10; k++) acc[k] += w.I10[b][k];
    for (int k = 0; k < 10; k++) w.Ic[tid][k] = acc[k];
  }
  __syncthreads();
  if (tid < m.nv) inert_mul(w.Ic[m.dof_bodyid[tid]], w.S[tid], w.fS[tid]);
  __syncthreads();
  for (int k = tid; k < m.nM; k += NT) {
    const int i = m.mi[k], j = m.mj[k];
    float v = 0.f;
    for (int c = 0; c < 6; c++) v += w.S[j][c] * w.fS[i][c];
    if (i == j) v += m.dof_armature[i];
    w.M[i * LD + j] = v;
    w.M[j * LD + i] = v;
  }
  __syncthreads();
}

HD void cross_motion(const float* v, const float* s, float* o) {
  float a[3], b[3], c[3];
  cross3(v, s, a); cross3(v, s + 3, b); cross3(v + 3, s, c);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
HD void cross_force(const float* v, const float* f, float* o) {
  float a[3], b[3], c[3];
  cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
  o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}

// ---- bias forces (Coriolis, centrifugal, gravity): recursive Newton-Euler with zero joint acceleration
__device__ void dev_bias(const DevModel& m, Work& w, const float* qvel) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    for (int i = 0; i < 6; i++) { w.cvel[0][i] = 0.f; w.cacc[0][i] = 0.f; w.cfrc[0][i] = 0.f; }
    for (int i = 0; i < 3; i++) w.cacc[0][3 + i] = -m.gravity[i];
  }
  __syncthreads();
  for (int lev = 1; lev <= m.nlevel; lev++) {
    if (tid < m.nbody && m.body_depth[tid] == lev) {
      const int b = tid, p = m.body_parent[b];
      float v[6], a[6];
      for (int i = 0; i < 6; i++) { v[i] = w.cvel[p][i]; a[i] = w.cacc[p][i]; }
      const int da = m.body_dofadr[b];
      for (int k = 0; k < m.body_dofnum[b]; k++) {
        const int dd = da + k;
        float sd[6];
        const float qd = qvel[dd];
        cross_motion(v, w.S[dd], sd);
        for (int i = 0; i < 6; i++) { a[i] += sd[i] * qd; v[i] += w.S[dd][i] * qd; }
      }
      float Iv[6], Ia[6], x[6];
      inert_mul(w.I10[b], v, Iv); inert_mul(w.I10[b], a, Ia); cross_force(v, Iv, x);
      for (int i = 0; i < 6; i++) { w.cvel[b][i] = v[i]; w.cacc[b][i] = a[i]; w.cfrc[b][i] = Ia[i] + x[i]; }
    }
    __syncthreads();
  }
  // subtree force sums (range sums again), then project on the motion axes
  float sub[6] = {0, 0, 0, 0, 0, 0};
  if (tid < m.nbody) {
    const int e = tid + m.body_subtree[tid];
    for (int b = tid; b < e; b++)
      for (int k = 0; k < 6; k++) sub[k] += w.cfrc[b][k];
  }
  __syncthreads();
  if (tid < m.nbody) for (int k = 0; k < 6; k++) w.cacc[tid][k] = sub[k];  // reuse cacc as subtree force
  __syncthreads();
  if (tid < m.nv) {
    const int b = m.dof_bodyid[tid];
    float s = 0.f;
    for (int k = 0; k < 6; k++) s += w.S[tid][k] * w.cacc[b][k];
    w.bias[tid] = s;
    w.passive[tid] = -m.dof_damping[tid] * qvel[tid];
  }
  __syncthreads();
}

// J^T (f at point, torque) of a body into qfrc, with the kinematics currently in the workspace (mj_applyFT,
// call sites uhc/envs/ho_im4.py:492-500,527-535)
HD float dev_apply_ft_dof(const DevModel& m, const Work& w, int dof, int body, const float* f, const float* tq,
                          const float* point) {
  if (!((m.body_dofmask[body] >> dof) & 1u)) return 0.f;
  float wxp[3];
  cross3(w.S[dof], point, wxp);
  float r = 0.f;
  for (int k = 0; k < 3; k++) r += (wxp[k] + w.S[dof][3 + k]) * f[k] + w.S[dof][k] * tq[k];
  return r;
}
