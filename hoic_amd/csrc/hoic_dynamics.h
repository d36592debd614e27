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

// ---- one-time load of the lane-resident constants and of the joint tables kept in LDS
__device__ void dev_load_constants(const DevModel& m, Work& w, LaneK& lk) {
  const int tid = threadIdx.x;
  const int b = tid < m.nbody ? tid : 0;
  lk.b_parent = m.body_parent[b]; lk.b_depth = tid < m.nbody ? m.body_depth[b] : -1; lk.b_jntadr = m.body_jntadr[b];
  lk.b_jntnum = m.body_jntnum[b]; lk.b_dofadr = m.body_dofadr[b]; lk.b_dofnum = m.body_dofnum[b];
  lk.b_subtree = m.body_subtree[b]; lk.b_mask = m.body_dofmask[b];
  for (int i = 0; i < 3; i++) { lk.b_pos[i] = m.body_pos[b][i]; lk.b_ipos[i] = m.body_ipos[b][i]; lk.b_inertia[i] = m.body_inertia[b][i]; }
  for (int i = 0; i < 4; i++) { lk.b_quat[i] = m.body_quat[b][i]; lk.b_iquat[i] = m.body_iquat[b][i]; }
  lk.b_mass = m.body_mass[b];
  const int g = (tid >= 32 && tid - 32 < m.ngeom) ? tid - 32 : 0;
  lk.g_body = m.geom_bodyid[g];
  for (int i = 0; i < 3; i++) lk.g_pos[i] = m.geom_pos[g][i];
  for (int i = 0; i < 4; i++) lk.g_quat[i] = m.geom_quat[g][i];
  const int d = tid < m.nv ? tid : 0;
  lk.d_body = m.dof_bodyid[d]; lk.d_jnt = m.dof_jntid[d]; lk.d_jtype = m.jnt_type[lk.d_jnt]; lk.d_k = d - m.jnt_dofadr[lk.d_jnt];
  lk.d_arm = m.dof_armature[d]; lk.d_damp = m.dof_damping[d]; lk.d_floss = m.dof_frictionloss[d]; lk.d_flR = m.dof_flR[d]; lk.d_flB = m.dof_flB[d];
  lk.d_act = -1;
  for (int u = 0; u < m.nu; u++) if (m.act_dofid[u] == d) lk.d_act = u;
  const int j = tid < m.njnt ? tid : 0;
  lk.j_type = m.jnt_type[j]; lk.j_qadr = m.jnt_qposadr[j]; lk.j_dadr = m.jnt_dofadr[j];
  lk.j_limited = (tid < m.njnt) ? m.jnt_limited[j] : 0;
  lk.j_lo = m.jnt_range[j][0]; lk.j_hi = m.jnt_range[j][1]; lk.j_margin = m.jnt_margin[j]; lk.j_K = m.jnt_K[j]; lk.j_B = m.jnt_B[j];
  lk.j_diag = m.jnt_diag[j];
  for (int i = 0; i < 5; i++) lk.j_solimp[i] = m.jnt_solimp[j][i];
  for (int ps = 0; ps < 2; ps++) {
    const int p = ps * NT + tid, pp = p < m.npair ? p : 0;
    const int g1 = m.pair_geom1[pp], g2 = m.pair_geom2[pp];
    lk.p_g1[ps] = g1; lk.p_g2[ps] = g2; lk.p_t1[ps] = p < m.npair ? m.geom_type[g1] : -1; lk.p_t2[ps] = m.geom_type[g2];
    for (int i = 0; i < 3; i++) { lk.p_s1[ps][i] = m.geom_size[g1][i]; lk.p_s2[ps][i] = m.geom_size[g2][i]; }
    lk.p_mesh[ps] = m.geom_meshid[g2];
    lk.p_margin[ps] = m.pair_margin[pp];
    lk.p_bound[ps] = m.geom_rbound[g1] + m.geom_rbound[g2] + m.pair_margin[pp];
  }
  for (int r = 0; r < 4; r++) {
    const int e = tid + r * NT;
    if (e < m.nM) { lk.m_i[r] = m.mi[e]; lk.m_j[r] = m.mj[e]; lk.m_arm[r] = (lk.m_i[r] == lk.m_j[r]) ? m.dof_armature[lk.m_i[r]] : 0.f; }
    else { lk.m_i[r] = -1; lk.m_j[r] = 0; lk.m_arm[r] = 0.f; }
  }
  if (tid < m.njnt) {
    for (int i = 0; i < 3; i++) { w.k_jaxis[tid][i] = m.jnt_axis[tid][i]; w.k_jpos[tid][i] = m.jnt_pos[tid][i]; }
    w.k_jq0[tid] = m.qpos0[m.jnt_qposadr[tid]];
    w.k_jtype[tid] = (unsigned char)m.jnt_type[tid]; w.k_jqadr[tid] = (unsigned char)m.jnt_qposadr[tid];
  }
  if (tid < NB) w.k_bmask[tid] = tid < m.nbody ? m.body_dofmask[tid] : 0u;
  __syncthreads();
}

// ---- kinematics: body frames, joint anchors/axes, geoms, motion axes S, body inertias about the origin
__device__ void dev_kinematics(const DevModel& m, const LaneK& lk, Work& w, const float* q) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    w.xpos[0][0] = w.xpos[0][1] = w.xpos[0][2] = 0.f;
    w.xquat[0][0] = 1.f; w.xquat[0][1] = w.xquat[0][2] = w.xquat[0][3] = 0.f;
    for (int i = 0; i < 9; i++) w.xmat[0][i] = (i % 4 == 0) ? 1.f : 0.f;
  }
  __syncthreads();
  for (int lev = 1; lev <= m.nlevel; lev++) {
    if (lk.b_depth == lev) {
      const int b = tid, p = lk.b_parent, ja = lk.b_jntadr, jn = lk.b_jntnum;
      float pos[3], quat[4], R[9], t[3];
      if (jn == 1 && w.k_jtype[ja] == HOIC_JNT_FREE) {
        const int qa = w.k_jqadr[ja];
        for (int i = 0; i < 3; i++) pos[i] = q[qa + i];
        for (int i = 0; i < 4; i++) quat[i] = q[qa + 3 + i];
        normquat(quat);
        quat2mat(quat, R);
        for (int i = 0; i < 3; i++) { w.xanchor[ja][i] = pos[i]; w.xaxis[ja][i] = R[3 * i + 2]; }
      } else {
        matvec(w.xmat[p], lk.b_pos, t);
        for (int i = 0; i < 3; i++) pos[i] = w.xpos[p][i] + t[i];
        mulquat(w.xquat[p], lk.b_quat, quat);
        quat2mat(quat, R);
        for (int j = ja; j < ja + jn; j++) {
          const float jax[3] = {w.k_jaxis[j][0], w.k_jaxis[j][1], w.k_jaxis[j][2]};
          const float jps[3] = {w.k_jpos[j][0], w.k_jpos[j][1], w.k_jpos[j][2]};
          matvec(R, jps, t);
          float anchor[3], axis[3];
          for (int i = 0; i < 3; i++) anchor[i] = pos[i] + t[i];
          matvec(R, jax, axis);
          for (int i = 0; i < 3; i++) { w.xanchor[j][i] = anchor[i]; w.xaxis[j][i] = axis[i]; }
          const float qq = q[w.k_jqadr[j]] - w.k_jq0[j];
          if (w.k_jtype[j] == HOIC_JNT_SLIDE) {
            for (int i = 0; i < 3; i++) pos[i] += axis[i] * qq;
          } else {
            float sn, cs;
            sincosf(0.5f * qq, &sn, &cs);
            float ql[4] = {cs, sn * jax[0], sn * jax[1], sn * jax[2]}, qn[4];
            mulquat(quat, ql, qn);
            for (int i = 0; i < 4; i++) quat[i] = qn[i];
            quat2mat(quat, R);
            matvec(R, jps, t);
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
  PT(21);
  // per body: inertial frame and spatial inertia about the world origin (m, h = m c, Io: xx yy zz xy xz yz)
  if (tid < m.nbody) {
    const int b = tid;
    float c[3], t[3], qi[4], Ri[9];
    matvec(w.xmat[b], lk.b_ipos, t);
    for (int i = 0; i < 3; i++) { c[i] = w.xpos[b][i] + t[i]; w.xipos[b][i] = c[i]; }
    mulquat(w.xquat[b], lk.b_iquat, qi);
    quat2mat(qi, Ri);
    const float mass = lk.b_mass, p0 = lk.b_inertia[0], p1 = lk.b_inertia[1], p2 = lk.b_inertia[2];
    float Ic[6];  // xx yy zz xy xz yz about the centre of mass
    Ic[0] = Ri[0] * p0 * Ri[0] + Ri[1] * p1 * Ri[1] + Ri[2] * p2 * Ri[2];
    Ic[1] = Ri[3] * p0 * Ri[3] + Ri[4] * p1 * Ri[4] + Ri[5] * p2 * Ri[5];
    Ic[2] = Ri[6] * p0 * Ri[6] + Ri[7] * p1 * Ri[7] + Ri[8] * p2 * Ri[8];
    Ic[3] = Ri[0] * p0 * Ri[3] + Ri[1] * p1 * Ri[4] + Ri[2] * p2 * Ri[5];
    Ic[4] = Ri[0] * p0 * Ri[6] + Ri[1] * p1 * Ri[7] + Ri[2] * p2 * Ri[8];
    Ic[5] = Ri[3] * p0 * Ri[6] + Ri[4] * p1 * Ri[7] + Ri[5] * p2 * Ri[8];
    const float cc = dot3(c, c);
    float* I = w.sc.dyn.I10[b];
    I[0] = mass; I[1] = mass * c[0]; I[2] = mass * c[1]; I[3] = mass * c[2];
    I[4] = Ic[0] + mass * (cc - c[0] * c[0]); I[5] = Ic[1] + mass * (cc - c[1] * c[1]); I[6] = Ic[2] + mass * (cc - c[2] * c[2]);
    I[7] = Ic[3] - mass * c[0] * c[1]; I[8] = Ic[4] - mass * c[0] * c[2]; I[9] = Ic[5] - mass * c[1] * c[2];
  }
  PT(22);
  // per geom (second half-wave so it overlaps the body work)
  {
    const int g = tid - 32;
    if (g >= 0 && g < m.ngeom) {
      const int b = lk.g_body;
      float t[3], qg[4];
      matvec(w.xmat[b], lk.g_pos, t);
      for (int i = 0; i < 3; i++) w.gxpos[g][i] = w.xpos[b][i] + t[i];
      mulquat(w.xquat[b], lk.g_quat, qg);
      quat2mat(qg, w.gxmat[g]);
    }
  }
  // per dof: motion axis about the origin
  if (tid < m.nv) {
    const int d = tid, j = lk.d_jnt, ty = lk.d_jtype;
    float* S = w.S[d];
    if (ty == HOIC_JNT_SLIDE) {
      S[0] = S[1] = S[2] = 0.f;
      for (int i = 0; i < 3; i++) S[3 + i] = w.xaxis[j][i];
    } else if (ty == HOIC_JNT_HINGE) {
      for (int i = 0; i < 3; i++) S[i] = w.xaxis[j][i];
      cross3(w.xanchor[j], w.xaxis[j], S + 3);
    } else {
      const int kk = lk.d_k, b = lk.d_body;
      if (kk < 3) { for (int i = 0; i < 6; i++) S[i] = 0.f; S[3 + kk] = 1.f; }
      else {
        float ax[3] = {w.xmat[b][kk - 3], w.xmat[b][3 + kk - 3], w.xmat[b][6 + kk - 3]};
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
__device__ void dev_mass_matrix(const DevModel& m, const LaneK& lk, Work& w) {
  const int tid = threadIdx.x;
  if (tid < m.nbody) {
    float acc[10];
    for (int i = 0; i < 10; i++) acc[i] = 0.f;
    const int e = tid + lk.b_subtree;
    for (int b = tid; b < e; b++)
      for (int i = 0; i < 10; i++) acc[i] += w.sc.dyn.I10[b][i];
    for (int i = 0; i < 10; i++) w.sc.dyn.Ic[tid][i] = acc[i];
  }
  __syncthreads();
  if (tid < m.nv) inert_mul(w.sc.dyn.Ic[lk.d_body], w.S[tid], w.sc.dyn.fS[tid]);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = lk.m_i[r], j = lk.m_j[r];
    if (i >= 0) {
      float v = lk.m_arm[r];
      for (int c = 0; c < 6; c++) v += w.S[j][c] * w.sc.dyn.fS[i][c];
      w.M[i * LD + j] = v;
      w.M[j * LD + i] = v;
    }
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
__device__ void dev_bias(const DevModel& m, const LaneK& lk, Work& w, const float* qvel) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    for (int i = 0; i < 6; i++) { w.sc.dyn.cvel[0][i] = 0.f; w.sc.dyn.cacc[0][i] = 0.f; w.sc.dyn.cfrc[0][i] = 0.f; }
    for (int i = 0; i < 3; i++) w.sc.dyn.cacc[0][3 + i] = -m.gravity[i];
  }
  __syncthreads();
  for (int lev = 1; lev <= m.nlevel; lev++) {
    if (lk.b_depth == lev) {
      const int b = tid, p = lk.b_parent;
      float v[6], a[6];
      for (int i = 0; i < 6; i++) { v[i] = w.sc.dyn.cvel[p][i]; a[i] = w.sc.dyn.cacc[p][i]; }
      const int da = lk.b_dofadr;
      for (int kk = 0; kk < lk.b_dofnum; kk++) {
        const int dd = da + kk;
        float sd[6];
        const float qd = qvel[dd];
        cross_motion(v, w.S[dd], sd);
        for (int i = 0; i < 6; i++) { a[i] += sd[i] * qd; v[i] += w.S[dd][i] * qd; }
      }
      float Iv[6], Ia[6], x[6];
      inert_mul(w.sc.dyn.I10[b], v, Iv); inert_mul(w.sc.dyn.I10[b], a, Ia); cross_force(v, Iv, x);
      for (int i = 0; i < 6; i++) { w.sc.dyn.cvel[b][i] = v[i]; w.sc.dyn.cacc[b][i] = a[i]; w.sc.dyn.cfrc[b][i] = Ia[i] + x[i]; }
    }
    __syncthreads();
  }
  // subtree force sums (range sums again), then project on the motion axes
  float sub[6] = {0, 0, 0, 0, 0, 0};
  if (tid < m.nbody) {
    const int e = tid + lk.b_subtree;
    for (int b = tid; b < e; b++)
      for (int i = 0; i < 6; i++) sub[i] += w.sc.dyn.cfrc[b][i];
  }
  __syncthreads();
  if (tid < m.nbody) for (int i = 0; i < 6; i++) w.sc.dyn.cacc[tid][i] = sub[i];  // reuse cacc as subtree force
  __syncthreads();
  if (tid < m.nv) {
    const int b = lk.d_body;
    float s = 0.f;
    for (int i = 0; i < 6; i++) s += w.S[tid][i] * w.sc.dyn.cacc[b][i];
    w.bias[tid] = s;
    w.passive[tid] = -lk.d_damp * qvel[tid];
  }
  __syncthreads();
}

// J^T (f at point, torque) of a body into qfrc, with the kinematics currently in the workspace (mj_applyFT,
// call sites uhc/envs/ho_im4.py:492-500,527-535)
HD float dev_apply_ft_dof(const Work& w, int dof, int body, const float* f, const float* tq, const float* point) {
  if (!((w.k_bmask[body] >> dof) & 1u)) return 0.f;
  float wxp[3];
  cross3(w.S[dof], point, wxp);
  float r = 0.f;
  for (int k = 0; k < 3; k++) r += (wxp[k] + w.S[dof][3 + k]) * f[k] + w.S[dof][k] * tq[k];
  return r;
}
