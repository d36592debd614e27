// hoic_dynamics.h — wave-cooperative kinematics, composite-inertia mass matrix and bias forces.
//
// Replaces the position/velocity stages of mj_forward that the reference runs through
// self.sim.step() / self.sim.forward() (uhc/envs/ho_im4.py:545, mujoco_env.py:114): kinematics,
// mass matrix (data.qM, read back by ho_im4.py:398) and qfrc_bias (ho_im4.py:401).
//
// Design (no level-by-level tree walks: every stage is a fixed number of wave-wide steps):
//   * kinematics: every body lane builds its transform relative to its parent, three pointer-jumping rounds
//     compose them into world poses;
//   * velocities / velocity-product accelerations are masked sums over the dofs on a body's path (motion axes
//     are expressed about the world origin, so spatial vectors of different bodies simply add);
//   * subtree sums use the depth-first body order (a subtree is an index range): no atomics, no recursion;
//   * the joint-space inertia matrix is produced one row per lane, in registers (MReg).
// Spatial vectors are [angular; linear-at-world-origin].
#pragma once
#include "hoic_types.h"
#include "hoic_math.h"

// the packed dof paths of the bodies -> LDS (once per launch): read several times per Newton iteration; every other per-dof
// constant is re-read from the L1/L2-resident DevModel where it is used (one load level, issued ahead of its use)
__device__ __forceinline__ void dev_load_constants(const DevModel& m, Work& w) {
  const int t = threadIdx.x;
  if (t < NB) for (int i = 0; i < 3; i++) w.k_bpath[t][i] = t < m.nbody ? m.body_path[t][i] : 0xFFFFFFFFu;
  if (t < NV) w.k_damp[t] = t < m.nv ? m.dof_damping[t] : 0.f;
  wsync();
}

// sum_{d on the packed path} S[d] * x[d]  (+ optional extra[d]) as straight-line code: all LDS reads are issued
// before the first FMA needs them (one latency instead of one per dof); `below` keeps only dofs < below
template <bool EXTRA> HD void path_gather(const Work& w, const unsigned (&path)[3], const float* x, const float (*extra)[6],
                                          int below, float* V, float* A, bool second = true, bool tail = true) {
  // (the packed path is loop-invariant over the substeps: hidden from the optimiser, or the 12 unpacked indices and
  // their scaled copies are hoisted out of the substep loop and spilled)
  // second (wave-uniform): also entries 6..11 -- a caller that knows no lane's path is longer than six dofs (the free object,
  // the palm) skips that batch; tail (wave-uniform): entries 10, 11 -- no body of the HOIC hand has more than ten dofs on its
  // path (palm 6 + finger 4: DevModel::max_path), so they would be two masked-off gathers in every call
  const unsigned pk[3] = {path[0], path[1], path[2]};
  // The tables' LDS addresses as (wave-uniform) base registers: an entry's address is then ONE multiply-add, d * 24 + base, and
  // its three two-word reads carry their offsets as immediates.  Left to itself the compiler forms d * 24 and adds the table's
  // 11 KB offset separately for every read (three adds per table and entry: the offset does not fit the reads' 8-bit fields).
  LPTR(const float) Sb = (LPTR(const float))&w.S[0][0];
  LPTR(const float) xb = (LPTR(const float))x;
  LPTR(const float) eb = (LPTR(const float))(EXTRA ? &extra[0][0] : &w.S[0][0]);
  asm volatile("" : "+s"(Sb), "+s"(xb), "+s"(eb));
#pragma unroll
  for (int h = 0; h < 2; h++) {
    if (h == 1) { __builtin_amdgcn_sched_barrier(0); if (!second) break; }   // two batches of six gathers in flight, not twelve (register peak)
#pragma unroll
    for (int i = 6 * h; i < 6 * h + 6; i++) {
      if (i == 10 && !tail) break;
      const unsigned e = (pk[i >> 2] >> (8 * (i & 3))) & 0xFFu;
      const bool on = e < (unsigned)below;
      const int d = on ? (int)e : 0;
      const float msk = on ? 1.f : 0.f;
      const float xd = xb[d] * msk;              // unconditional loads (index 0 when off): no exec-masked branches
      LPTR(const float) Sd = Sb + 6 * d;
#pragma unroll
      for (int k = 0; k < 6; k++) V[k] = fmaf(Sd[k], xd, V[k]);
      if (EXTRA) {
        LPTR(const float) ed = eb + 6 * d;
#pragma unroll
        for (int k = 0; k < 6; k++) A[k] = fmaf(ed[k], msk, A[k]);
      }
    }
  }
}

// the same gather for NX vectors at once (the Jacobian products of the solve's set-up: qvel, a_smooth and the warm start go
// through the contact Jacobian together): one unpacking of the path and one read of S[d] serve all of them; per vector the
// multiply-adds run in the order of path_gather, so every product is bit-identical to a separate pass
template <int NX> HD void path_gather_multi(const Work& w, const unsigned (&path)[3], const float* const (&x)[NX], float (&V)[NX][6], bool second, bool tail) {
  const unsigned pk[3] = {path[0], path[1], path[2]};
  LPTR(const float) Sb = (LPTR(const float))&w.S[0][0];      // (base registers: see path_gather)
  LPTR(const float) xb[NX];
#pragma unroll
  for (int v = 0; v < NX; v++) { xb[v] = (LPTR(const float))x[v]; asm volatile("" : "+s"(xb[v])); }
  asm volatile("" : "+s"(Sb));
#pragma unroll
  for (int h = 0; h < 2; h++) {
    if (h == 1) { __builtin_amdgcn_sched_barrier(0); if (!second) break; }
#pragma unroll
    for (int i = 6 * h; i < 6 * h + 6; i++) {
      if (i == 10 && !tail) break;
      const unsigned e = (pk[i >> 2] >> (8 * (i & 3))) & 0xFFu;
      const bool on = e != 0xFFu;
      const int d = on ? (int)e : 0;
      const float msk = on ? 1.f : 0.f;
      float Sd[6];
      LPTR(const float) Sp = Sb + 6 * d;
#pragma unroll
      for (int k = 0; k < 6; k++) Sd[k] = Sp[k];
#pragma unroll
      for (int v = 0; v < NX; v++) {
        const float xd = xb[v][d] * msk;
#pragma unroll
        for (int k = 0; k < 6; k++) V[v][k] = fmaf(Sd[k], xd, V[v][k]);
      }
    }
  }
}

// ---- kinematics: body frames, geoms, motion axes S, body inertias about the origin
template <class W> __device__ __forceinline__ void dev_kinematics(const DevModel& m, W& w, const float* q) {
  const int tid = opaque(threadIdx.x);
  const bool isb = tid < m.nbody;
  float P[3] = {0.f, 0.f, 0.f}, Q[4] = {1.f, 0.f, 0.f, 0.f};
  // 0. lane = joint: joint rotation quaternion / slide displacement (all sincos calls in parallel, constants by
  //    independent loads); parked in the inertia scratch, which is not written before step 3
  float (*jrec)[12] = reinterpret_cast<float (*)[12]>(&w.sc.dyn.I10[0][0]);   // [axis 3 | pos 3 | quat 4 or disp | type]
  // The model constants of steps 1 and 2 are fetched here, under step 0, and those of step 3 under the pointer-jumping rounds:
  // behind the hand-over point that precedes their use, each group was one more exposed global-load latency of the stage's chain
  // (a fence keeps a load on its side of it).
  int ja = 0, jn = 0, jump[MAXROUND];
  float bpos[3] = {0.f, 0.f, 0.f}, bquat[4] = {1.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < MAXROUND; r++) jump[r] = -1;
  if (isb) {
    ja = m.body_jntadr[tid]; jn = m.body_jntnum[tid];
    for (int i = 0; i < 3; i++) bpos[i] = m.body_pos[tid][i];
    for (int i = 0; i < 4; i++) bquat[i] = m.body_quat[tid][i];
#pragma unroll
    for (int r = 0; r < MAXROUND; r++) jump[r] = m.body_jump[r][tid];
  }
  if (tid < m.njnt) {
    const int j = tid, ty = m.jnt_type[j], qa = m.jnt_qposadr[j];
    float* r = jrec[j];
    for (int i = 0; i < 3; i++) { r[i] = m.jnt_axis[j][i]; r[3 + i] = m.jnt_pos[j][i]; }
    float sn = 0.f, cs = 1.f;
    const float qq = (ty == HOIC_JNT_FREE) ? 0.f : q[qa] - m.qpos0[qa];
    if (ty == HOIC_JNT_HINGE) sincos_pi(0.5f * qq, &sn, &cs);
    r[6] = (ty == HOIC_JNT_SLIDE) ? qq : cs; r[7] = sn * r[0]; r[8] = sn * r[1]; r[9] = sn * r[2];
    r[10] = __int_as_float(ty); r[11] = __int_as_float(qa);
  }
  wsync();
  // 1. transform of each body relative to its parent, with the joint axes / anchors in the parent frame
  if (isb) {
    const int b = tid;
    const bool jz = m.jnt_poszero != 0;       // wave-uniform
    for (int i = 0; i < 3; i++) P[i] = bpos[i];
    for (int i = 0; i < 4; i++) Q[i] = bquat[i];
    for (int j = ja; j < ja + jn; j++) {
      const float* r = jrec[j];
      const int ty = __float_as_int(r[10]);
      if (ty == HOIC_JNT_FREE) {
        const int qa = __float_as_int(r[11]);
        for (int i = 0; i < 3; i++) P[i] = q[qa + i];
        for (int i = 0; i < 4; i++) Q[i] = q[qa + 3 + i];
        normquat(Q);
      } else if (jz) {
        // joint at the body's origin (every joint of the HOIC hand): its anchor is the body position, which a rotation about it
        // does not move -- the two rotations of the anchor offset drop out (they would rotate the zero vector: same results)
        const float jax[3] = {r[0], r[1], r[2]};
        float ax[3];
        qrot(Q, jax, ax);
        for (int i = 0; i < 3; i++) { w.sc.dyn.u.j.jax[j][i] = ax[i]; w.sc.dyn.u.j.janc[j][i] = P[i]; }
        if (ty == HOIC_JNT_SLIDE) {
          for (int i = 0; i < 3; i++) P[i] += ax[i] * r[6];
        } else {
          const float ql[4] = {r[6], r[7], r[8], r[9]};
          float qn[4];
          mulquat(Q, ql, qn);
          for (int i = 0; i < 4; i++) Q[i] = qn[i];
        }
      } else {
        const float jax[3] = {r[0], r[1], r[2]}, jps[3] = {r[3], r[4], r[5]};
        float ax[3], t[3], an[3];
        qrot(Q, jax, ax); qrot(Q, jps, t);
        for (int i = 0; i < 3; i++) { an[i] = P[i] + t[i]; w.sc.dyn.u.j.jax[j][i] = ax[i]; w.sc.dyn.u.j.janc[j][i] = an[i]; }
        if (ty == HOIC_JNT_SLIDE) {
          for (int i = 0; i < 3; i++) P[i] += ax[i] * r[6];
        } else {
          const float ql[4] = {r[6], r[7], r[8], r[9]};
          float qn[4];
          mulquat(Q, ql, qn);
          for (int i = 0; i < 4; i++) Q[i] = qn[i];
          qrot(Q, jps, t);
          for (int i = 0; i < 3; i++) P[i] = an[i] - t[i];
        }
      }
    }
    normquat(Q);
    for (int i = 0; i < 3; i++) w.xpos[b][i] = P[i];
    for (int i = 0; i < 4; i++) w.xquat[b][i] = Q[i];
  }
  wsync();
  // (constants of step 3, in flight during the rounds below)
  float ip[3] = {0.f, 0.f, 0.f}, iq[4] = {1.f, 0.f, 0.f, 0.f}, bmass = 0.f, binr[3] = {0.f, 0.f, 0.f};
  if (isb) {
    for (int i = 0; i < 3; i++) { ip[i] = m.body_ipos[tid][i]; binr[i] = m.body_inertia[tid][i]; }
    for (int i = 0; i < 4; i++) iq[i] = m.body_iquat[tid][i];
    bmass = m.body_mass[tid];
  }
  int dj = 0, dty = 0, dbody = 0, dk = 0, dpar = 0;
  if (tid < m.nv) { dj = m.dof_jntid[tid]; dty = m.dof_jtype[tid]; dbody = m.dof_bodyid[tid]; dk = m.dof_k[tid]; dpar = m.dof_parentbody[tid]; }
  int gbody = 0; float gp[3] = {0.f, 0.f, 0.f}, gq[4] = {1.f, 0.f, 0.f, 0.f};
  {
    const int g = tid - 32;
    if (g >= 0 && g < m.ngeom) {
      gbody = m.geom_bodyid[g];
      for (int i = 0; i < 3; i++) gp[i] = m.geom_pos[g][i];
      for (int i = 0; i < 4; i++) gq[i] = m.geom_quat[g][i];
    }
  }
  // 2. pointer jumping: after round r a body's transform is relative to its ancestor 2^(r+1) levels up
#pragma unroll
  for (int r = 0; r < MAXROUND; r++) {
    if (r >= m.nround) break;
    const int src = jump[r];
    if (src >= 0) {
      float Ps[3], Qs[4], t[3], qn[4];
      for (int i = 0; i < 3; i++) Ps[i] = w.xpos[src][i];
      for (int i = 0; i < 4; i++) Qs[i] = w.xquat[src][i];
      qrot(Qs, P, t);
      for (int i = 0; i < 3; i++) P[i] = Ps[i] + t[i];
      mulquat(Qs, Q, qn);
      for (int i = 0; i < 4; i++) Q[i] = qn[i];
    }
    wsync();
    if (src >= 0) {
      for (int i = 0; i < 3; i++) w.xpos[tid][i] = P[i];
      for (int i = 0; i < 4; i++) w.xquat[tid][i] = Q[i];
    }
    wsync();
  }
  if (isb) {
    normquat(Q);
    for (int i = 0; i < 4; i++) w.xquat[tid][i] = Q[i];
  }
  wsync();
  PT(21);
  // 3a. per body: spatial inertia about the world origin (m, h = m c, Io: xx yy zz xy xz yz)
  if (isb) {
    const int b = tid;
    float c[3], t[3], qi[4], Ri[9];
    qrot(Q, ip, t);
    for (int i = 0; i < 3; i++) c[i] = P[i] + t[i];
    mulquat(Q, iq, qi);
    quat2mat(qi, Ri);
    const float mass = bmass, p0 = binr[0], p1 = binr[1], p2 = binr[2];
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
  // 3b. per dof: motion axis about the origin (the joint frames of step 1 moved to the world by the parent pose)
  if (tid < m.nv) {
    const int d = tid, j = dj, ty = dty, b = dbody;
    float* S = w.S[d];
    if (ty == HOIC_JNT_FREE) {
      const int kk = dk;
      if (kk < 3) { for (int i = 0; i < 6; i++) S[i] = (i == 3 + kk) ? 1.f : 0.f; }
      else {
        const float e[3] = {kk == 3 ? 1.f : 0.f, kk == 4 ? 1.f : 0.f, kk == 5 ? 1.f : 0.f};
        float ax[3];
        qrot(w.xquat[b], e, ax);
        for (int i = 0; i < 3; i++) S[i] = ax[i];
        cross3(w.xpos[b], ax, S + 3);
      }
    } else {
      const int p = dpar;
      const float la[3] = {w.sc.dyn.u.j.jax[j][0], w.sc.dyn.u.j.jax[j][1], w.sc.dyn.u.j.jax[j][2]};
      float ax[3];
      qrot(w.xquat[p], la, ax);
      if (ty == HOIC_JNT_SLIDE) {
        for (int i = 0; i < 3; i++) { S[i] = 0.f; S[3 + i] = ax[i]; }
      } else {
        const float lc[3] = {w.sc.dyn.u.j.janc[j][0], w.sc.dyn.u.j.janc[j][1], w.sc.dyn.u.j.janc[j][2]};
        float an[3];
        qrot(w.xquat[p], lc, an);
        for (int i = 0; i < 3; i++) { an[i] += w.xpos[p][i]; S[i] = ax[i]; }
        cross3(an, ax, S + 3);
      }
    }
  }
  PT(22);
  // 3c. per geom (upper half-wave)
  {
    const int g = tid - 32;
    if (g >= 0 && g < m.ngeom) {
      const int b = gbody;
      float t[3], qg[4];
      qrot(w.xquat[b], gp, t);
      for (int i = 0; i < 3; i++) w.gxpos[g][i] = w.xpos[b][i] + t[i];
      mulquat(w.xquat[b], gq, qg);
      quat2mat(qg, w.gxmat[g]);
    }
  }
  wsync();
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

// ---- joint-space inertia: composite rigid body sums over index ranges, then row (lane & 31) of M in registers:
// M[i][j] = S_j . (Ic_body(i) S_i) for j an ancestor-or-self dof of i, mirrored for descendants, 0 elsewhere
__device__ __forceinline__ void dev_mass_matrix(const DevModel& m, Work& w, MReg& M) {
  const int tid = opaque(threadIdx.x);
  // composite inertias in two rounds (DevModel::body_sum): small subtrees directly, then the bodies above them from their
  // children's composites -- the longest loop is SUM_DIRECT / the largest child count instead of the whole hand
  const int smode = tid < m.nbody ? m.body_sum[tid] : 0;
  // (every model constant of the stage is fetched here: behind the hand-over points below each would be one more exposed
  //  global-load latency)
  const int dlane = opaque(tid & 31);
  const bool vdl = dlane < m.nv;
  const unsigned k0 = smode == 2 ? m.body_kids[tid][0] : 0xFFFFFFFFu, k1 = smode == 2 ? m.body_kids[tid][1] : 0xFFFFFFFFu;
  const int dbody_ = vdl ? m.dof_bodyid[dlane] : 0;
  const unsigned am = vdl ? (m.dof_amask[dlane] | (1u << dlane)) : 0u, dm = vdl ? m.dof_dmask[dlane] : 0u;
  const float arm = vdl ? m.dof_armature[dlane] : 0.f;
  if (smode == 1) {
    // (at most SUM_DIRECT bodies: unrolled with the rows beyond the subtree masked off, so that all reads are in flight at once --
    //  a loop with a per-lane trip count paid one LDS round trip per body)
    float acc[10];
    for (int i = 0; i < 10; i++) acc[i] = 0.f;
    const int nsub = m.body_subtree[tid];
#pragma unroll
    for (int k = 0; k < SUM_DIRECT; k++) {
      const bool on = k < nsub;
      const float* row = w.sc.dyn.I10[on ? tid + k : tid];
      const float msk = on ? 1.f : 0.f;
      for (int i = 0; i < 10; i++) acc[i] = fmaf(row[i], msk, acc[i]);
    }
    for (int b = tid + SUM_DIRECT; b < tid + nsub; b++)      // (a tree the two-round plan does not fit: plain range sums, build_model)
      for (int i = 0; i < 10; i++) acc[i] += w.sc.dyn.I10[b][i];
    for (int i = 0; i < 10; i++) w.sc.dyn.Ic[tid][i] = acc[i];
  }
  wsync();
  if (smode == 2) {
    float acc[10];
    for (int i = 0; i < 10; i++) acc[i] = w.sc.dyn.I10[tid][i];
#pragma unroll
    for (int h = 0; h < 2; h++) {          // the children four at a time (masked), their reads in flight together
      const unsigned kk = h ? k1 : k0;
      if (kk == 0xFFFFFFFFu) break;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned c = (kk >> (8 * k)) & 0xFFu;
        const bool on = c != 0xFFu;
        const float* row = w.sc.dyn.Ic[on ? c : (kk & 0xFFu)];      // (masked-off slots re-read the batch's first child: a finite row)
        const float msk = on ? 1.f : 0.f;
        for (int i = 0; i < 10; i++) acc[i] = fmaf(row[i], msk, acc[i]);
      }
    }
    for (int i = 0; i < 10; i++) w.sc.dyn.Ic[tid][i] = acc[i];
  }
  wsync();   // (also: the joint frames in sc.dyn.u.j are dead from here on, u.f takes their place)
  const int d = dlane;
  const bool vd = vdl;
  float Si[6], fSi[6];
  for (int i = 0; i < 6; i++) { Si[i] = vd ? w.S[d][i] : 0.f; fSi[i] = 0.f; }
  if (vd) inert_mul(w.sc.dyn.Ic[dbody_], Si, fSi);
  if (tid < 32) for (int i = 0; i < 6; i++) w.sc.dyn.u.f.fS[d][i] = fSi[i];
  wsync();
  const int hi = tid >> 5;
  // Row j = MREG_ROW(reg, hi) = jc(reg) + 4 hi: the half-wave's share goes into per-lane bases (and pre-shifted masks), so that
  // every entry's LDS reads carry their offsets as immediates (one address add per read otherwise: the two tables sit 8 KB apart)
  // and its mask tests are bit extracts at constant positions.  Per half-wave uniform addresses: LDS broadcasts.
  LPTR(const float) Sb = (LPTR(const float))&w.S[4 * hi][0];
  LPTR(const float) fb = (LPTR(const float))&w.sc.dyn.u.f.fS[4 * hi][0];
  asm volatile("" : "+v"(Sb), "+v"(fb));      // (kept as two base registers: folded back into one base + 8 KB constants otherwise)
  const unsigned amh = am >> (4 * hi), dmh = dm >> (4 * hi);
  const int dh = d - 4 * hi;
#pragma unroll
  for (int rp = 0; rp < 16; rp += 2) {       // two entries at a time: their 24 LDS reads share one round trip
    float vv[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int jc = ((rp + q) & 3) + 8 * ((rp + q) >> 2);
      float sj[6], fj[6];
#pragma unroll
      for (int k = 0; k < 6; k++) { sj[k] = Sb[6 * jc + k]; fj[k] = fb[6 * jc + k]; }
      const float a = dot6(sj, fSi), bb = dot6(Si, fj);
      float v = ((amh >> jc) & 1u) ? a : (((dmh >> jc) & 1u) ? bb : 0.f);
      if (jc == dh) v += arm;
      vv[q] = v;
    }
    // The entries are pinned where they are computed: on the path that leaves the substep loop after a failed substep M is not
    // read again, so the optimiser would otherwise sink the 16 x 12 multiply-adds behind the collision stage (past that exit) and
    // keep the 192 LDS values they read alive across it -- 231 spilled registers at the 168-register budget.  (The armature used
    // to be a conditional LDS read per entry, which happened to hold the arithmetic in place.)
    asm volatile("" : "+v"(vv[0]), "+v"(vv[1]));
    M.r[rp] = vv[0]; M.r[rp + 1] = vv[1];
  }
  wsync();
}

// (M x)[lane & 31] on every lane, x in LDS: each half-wave sums its 16 columns, one cross-half add
HD float dev_Mx(const MReg& M, const float* x) {
  const int hi = opaque(threadIdx.x) >> 5;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const float* xp = x + 8 * g + 4 * hi;
    s0 = fmaf(M.r[4 * g], xp[0], s0); s1 = fmaf(M.r[4 * g + 1], xp[1], s1);
    s0 = fmaf(M.r[4 * g + 2], xp[2], s0); s1 = fmaf(M.r[4 * g + 3], xp[3], s1);
  }
  const float s = s0 + s1;
  const unsigned v = __float_as_uint(s);
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // .x: low-half value on both halves, .y: high-half value
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
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

// ---- bias forces (Coriolis, centrifugal, gravity): Newton-Euler with zero joint acceleration, without a tree walk:
//   cdd_d = (velocity of the chain above dof d) x S_d * qvel_d          (lane = dof)
//   V_b = sum_{d on path(b)} S_d qvel_d,  A_b = a_world + sum cdd_d       (lane = body)
//   f_b = I_b A_b + V_b x* I_b V_b;  subtree range sums;  bias_d = S_d . fsub_body(d)
__device__ __forceinline__ void dev_bias(const DevModel& m, Work& w, const float* qvel) {
  const int tid = opaque(threadIdx.x);
  // (the stage's model constants, fetched ahead of the hand-over points: see dev_mass_matrix)
  const int smode = tid < m.nbody ? m.body_sum[tid] : 0;
  const int sub_end = smode == 1 ? tid + m.body_subtree[tid] : 0;
  const unsigned k0 = smode == 2 ? m.body_kids[tid][0] : 0xFFFFFFFFu, k1 = smode == 2 ? m.body_kids[tid][1] : 0xFFFFFFFFu;
  const int dbody_ = tid < m.nv ? m.dof_bodyid[tid] : 0;
  if (tid < m.nv) {
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sd[6];
    const unsigned pth[3] = {m.dof_bpath[tid][0], m.dof_bpath[tid][1], m.dof_bpath[tid][2]};
    path_gather<false>(w, pth, qvel, nullptr, tid, v, nullptr, true, m.max_path > 10);   // dofs above `tid` on its path
    cross_motion(v, w.S[tid], sd);
    const float qd = qvel[tid];
    for (int i = 0; i < 6; i++) w.sc.dyn.u.f.fS[tid][i] = sd[i] * qd;
  }
  wsync();
  if (tid < m.nbody) {
    float f[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned bp[3] = {w.k_bpath[tid][0], w.k_bpath[tid][1], w.k_bpath[tid][2]};
    if (bp[0] != 0xFFFFFFFFu) {
      float V[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, A[6] = {0.f, 0.f, 0.f, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
      path_gather<true>(w, bp, qvel, w.sc.dyn.u.f.fS, 0xFF, V, A, true, m.max_path > 10);
      float Iv[6], Ia[6], x[6];
      inert_mul(w.sc.dyn.I10[tid], V, Iv); inert_mul(w.sc.dyn.I10[tid], A, Ia); cross_force(V, Iv, x);
      for (int i = 0; i < 6; i++) f[i] = Ia[i] + x[i];
    }
    for (int i = 0; i < 6; i++) w.sc.dyn.u.f.cfrc[tid][i] = f[i];
  }
  wsync();
  {   // subtree force sums, kept in the Ic slots: the two rounds of the composite inertias
    if (smode == 1) {
      float sub[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int nsub = sub_end - tid;
#pragma unroll
      for (int k = 0; k < SUM_DIRECT; k++) {      // (unrolled and masked: see dev_mass_matrix)
        const bool on = k < nsub;
        const float* row = w.sc.dyn.u.f.cfrc[on ? tid + k : tid];
        const float msk = on ? 1.f : 0.f;
        for (int i = 0; i < 6; i++) sub[i] = fmaf(row[i], msk, sub[i]);
      }
      for (int b = tid + SUM_DIRECT; b < sub_end; b++)
        for (int i = 0; i < 6; i++) sub[i] += w.sc.dyn.u.f.cfrc[b][i];
      for (int i = 0; i < 6; i++) w.sc.dyn.Ic[tid][i] = sub[i];
    }
    wsync();
    if (smode == 2) {
      float sub[6];
      for (int i = 0; i < 6; i++) sub[i] = w.sc.dyn.u.f.cfrc[tid][i];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const unsigned kk = h ? k1 : k0;
        if (kk == 0xFFFFFFFFu) break;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned c = (kk >> (8 * k)) & 0xFFu;
          const bool on = c != 0xFFu;
          const float* row = w.sc.dyn.Ic[on ? c : (kk & 0xFFu)];      // (masked-off slots re-read the batch's first child: a finite row)
          const float msk = on ? 1.f : 0.f;
          for (int i = 0; i < 6; i++) sub[i] = fmaf(row[i], msk, sub[i]);
        }
      }
      for (int i = 0; i < 6; i++) w.sc.dyn.Ic[tid][i] = sub[i];
    }
  }
  wsync();
  if (tid < NV) w.bias[tid] = (tid < m.nv) ? dot6(w.S[tid], w.sc.dyn.Ic[dbody_]) : 0.f;
  wsync();
}

// J^T (f at point, torque) of a body into qfrc, with the kinematics currently in the workspace (mj_applyFT,
// call sites uhc/envs/ho_im4.py:492-500,527-535)
HD float dev_apply_ft_dof(const DevModel& m, const Work& w, int dof, int body, const float* f, const float* tq, const float* point) {
  if (!((m.body_dofmask[body] >> dof) & 1u)) return 0.f;
  float wxp[3];
  cross3(w.S[dof], point, wxp);
  float r = 0.f;
  for (int k = 0; k < 3; k++) r += (wxp[k] + w.S[dof][3 + k]) * f[k] + w.S[dof][k] * tq[k];
  return r;
}
