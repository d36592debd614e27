// hoic_types.h — device-side constant tables and per-env LDS workspace (gfx950).
//
// One workgroup of ONE wavefront (64 lanes) owns one environment.  Everything an env needs during an
// env-step (15 fused substeps) lives in this workgroup's LDS; HBM is touched only for the persistent state
// (qpos/qvel/warm-start/lagged state), the action, the expert window and the outputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hoic.h"

#define NT 64            // lanes per env (one wavefront)
#define NB HOIC_MAX_BODY // 28
#define NJ HOIC_MAX_JNT  // 28
#define NQP 36           // padded nq
#define NV HOIC_MAX_NV   // 32
#define NU HOIC_MAX_NU   // 26
#define NG HOIC_MAX_GEOM // 28
#define NPAIR 128
#define MAXCON 32        // contacts kept per env per substep
#define NBASIS 4         // contact-frame Jacobian rows stored per contact: n, t1, t2, spin(n)
#define MAXLIM 32
#define NROW (NV + MAXLIM + MAXCON * 6)
#define LD 33            // padded leading dimension of 32-wide LDS matrices (bank-conflict free)
#define NHB HOIC_NHANDBODY
#define NHG 19
#define MAXMESHV 256
#define MAXMESHP HOIC_MAX_MESHPLANE

struct DevModel {
  int nbody, njnt, nq, nv, nu, ngeom, npair, nlevel, nM;
  int hand_body0, obj_body, hand_geom0, hand_geom1, obj_geom0, obj_geom1, hand_nq, hand_nv;
  float timestep, gravity[3], meaninertia, hand_mass;
  // bodies (depth-first order: a body's subtree is the index range [b, b + body_subtree[b]) )
  int body_parent[NB], body_depth[NB], body_jntadr[NB], body_jntnum[NB], body_dofadr[NB], body_dofnum[NB];
  int body_subtree[NB];
  unsigned body_dofmask[NB];  // dofs on the path root -> body
  float body_pos[NB][3], body_quat[NB][4], body_ipos[NB][3], body_iquat[NB][4], body_mass[NB], body_inertia[NB][3];
  // joints / dofs
  int jnt_type[NJ], jnt_qposadr[NJ], jnt_dofadr[NJ], jnt_bodyid[NJ], jnt_limited[NJ];
  float jnt_pos[NJ][3], jnt_axis[NJ][3], jnt_range[NJ][2], jnt_margin[NJ], jnt_K[NJ], jnt_B[NJ], jnt_solimp[NJ][5];
  float jnt_diag[NJ];  // dof_invweight0 of the joint's dof (limit row diagApprox)
  float qpos0[NQP];
  int dof_bodyid[NV], dof_jntid[NV];
  float dof_armature[NV], dof_damping[NV], dof_frictionloss[NV];
  float dof_flR[NV], dof_flB[NV];  // friction-loss row regulariser R and damping B (K = 0)
  int act_dofid[NU];
  // geoms
  int geom_type[NG], geom_bodyid[NG], geom_meshid[NG];
  float geom_size[NG][3], geom_pos[NG][3], geom_quat[NG][4], geom_rbound[NG];
  // static collision pair list with mixed parameters
  int pair_geom1[NPAIR], pair_geom2[NPAIR], pair_condim[NPAIR], pair_b1[NPAIR], pair_b2[NPAIR];
  float pair_mu[NPAIR][3], pair_K[NPAIR], pair_B[NPAIR], pair_solimp[NPAIR][5], pair_margin[NPAIR], pair_gap[NPAIR];
  float pair_Rscale[NPAIR];  // R = max(MINVAL,(1-imp)/imp) * Rscale  (pyramidal: 2 mu^2 tran (1+mu^2); condim 1: tran)
  // mass-matrix sparsity: entries (i, j) with j an ancestor-or-self dof of i
  unsigned char mi[256], mj[256];
  // convex meshes (hull vertices in the geom frame)
  int mesh_vertadr[HOIC_MAX_MESH], mesh_vertnum[HOIC_MAX_MESH], mesh_planeadr[HOIC_MAX_MESH], mesh_planenum[HOIC_MAX_MESH];
  float mesh_vert[MAXMESHV][3];
  float mesh_plane[MAXMESHP][4];   // hull faces n.x <= d in the geom frame
};

struct DevConfig {
  hoic_env_config c;
  float base_pose[NU], ctrl_scale[NU];
  hoic_reward_params rp;
  int mode_train;
};

struct DevExpert {
  int n_seq, total;
  const int* seq_off;   // [n_seq]
  const int* seq_len;   // [n_seq]
  const float *hand_dof, *hand_dof_vel, *obj_pose, *obj_vel, *obj_angvel, *body_pos, *body_quat;
};

// persistent per-env state in HBM (row per env)
struct DevState {
  float* qpos;   // [n, NQP]
  float* qvel;   // [n, NV]
  float* warm;   // [n, NV]
  float* qlag;   // [n, NQP]  state before the last integration (one-substep lag)
  float* vlag;   // [n, NV]
  int* cur_t;    // [n]
  int* start;    // [n]
  int* seq;      // [n]
  float* rfc_score;  // [n]
  int* overflow; // [n] contact-cap overflow counter
  long long* phase;  // [n, 24] per-phase cycle counters (HOIC_PHASE_TIMING builds only)
};

// lane-resident model constants: loaded from DevModel ONCE per launch, so the 15 fused substeps never chase
// dependent global loads for them (lane = body / geom(lane-32) / dof / joint / collision pair)
struct LaneK {
  // lane = body
  int b_parent, b_depth, b_jntadr, b_jntnum, b_dofadr, b_dofnum, b_subtree; unsigned b_mask;
  float b_pos[3], b_quat[4], b_ipos[3], b_iquat[4], b_mass, b_inertia[3];
  // lane - 32 = geom
  int g_body; float g_pos[3], g_quat[4];
  // lane = dof
  int d_body, d_jnt, d_jtype, d_k, d_act; float d_arm, d_damp, d_floss, d_flR, d_flB;
  // lane = joint
  int j_type, j_qadr, j_dadr, j_limited; float j_lo, j_hi, j_margin, j_K, j_B, j_diag, j_solimp[5];
  // lane = collision pair (pass 0: pairs 0..63, pass 1: pairs 64..127)
  int p_g1[2], p_g2[2], p_t1[2], p_t2[2], p_mesh[2]; float p_s1[2][3], p_s2[2][3], p_bound[2], p_margin[2];
  // mass-matrix entries handled by this lane
  int m_i[4], m_j[4]; float m_arm[4];
};

// per-env LDS workspace
struct Work {
  // joint-indexed constants read by the body lanes during kinematics
  float k_jaxis[NJ][3], k_jpos[NJ][3], k_jq0[NJ]; unsigned char k_jtype[NJ], k_jqadr[NJ];
  unsigned k_bmask[NB];
  float qpos[NQP], qvel[NV], qacc[NV], warm[NV], qlag[NQP], vlag[NV], action[NV];
  float ctrl[NV], applied[NV], bias[NV], passive[NV], fsmooth[NV], asmooth[NV], fcon[NV];
  float grad[NV], search[NV], Ma[NV], Ms[NV], tv[NV], tv2[NV];
  float xpos[NB][3], xquat[NB][4], xmat[NB][9], xipos[NB][3];
  // scratch shared by phases that never overlap: CRBA/RNE temporaries of the forward pass, and the 32x33
  // transpose buffer of the matrix-core solves
  union {
    struct { float I10[NB][10], Ic[NB][10], cvel[NB][6], cacc[NB][6], cfrc[NB][6], fS[NV][6]; } dyn;
    float T[NV * LD];
  } sc;
  float xanchor[NJ][3], xaxis[NJ][3];
  float S[NV][6];
  float gxpos[NG][3], gxmat[NG][9], old_gxpos[NG][3], old_gxmat[NG][9];
  float old_objvel[6];
  float M[NV * LD];
  // contacts of the current forward pass
  int ncon, nlim, nrow, pad0;
  float c_pos[MAXCON][3], c_frame[MAXCON][9], c_dist[MAXCON], c_mu[MAXCON][3], c_D[MAXCON], c_aref0[MAXCON], c_B[MAXCON];
  int c_pair[MAXCON], c_nrow[MAXCON], c_row0[MAXCON], c_b1[MAXCON], c_b2[MAXCON];
  // Jacobian-free contacts: row (c,k) of the contact Jacobian is  sg(dof,c) * S[dof] . c_W[c][k]  with the
  // wrench basis c_W[c][k] = [p x f_k ; f_k] (k = n,t1,t2) and [f_n ; 0] (spin) about the world origin
  float c_W[MAXCON][NBASIS][6], c_G[MAXCON][6];
  unsigned c_mpos[MAXCON], c_mneg[MAXCON];   // dofs moving body2 only (+1) / body1 only (-1)
  float u[MAXCON * NBASIS];
  float bV[NB][6];                           // body spatial velocities for J.x products
  // limit rows
  int lim_dof[MAXLIM]; float lim_sign[MAXLIM], lim_D[MAXLIM];
  // all constraint rows: [0,nv) friction loss, [nv, nv+nlim) limits, then contact rows
  float r_aref[NROW], r_jar[NROW], r_force[NROW], r_curv[NROW];
  unsigned char r_con[NROW], r_edge[NROW];
  // contact bookkeeping over the env step (record_contact / classify_contact)
  float rec_sum[NHG][12]; int rec_cnt[NHG];
  float avg_cps[NHG][12]; int avg_geom[NHG]; float avg_ts[NHG]; int n_avg;
  float gvel[NG][3], gangvel[NG][3], obj_avg_acc[6];
  float red[8];
  int solver_iter, fail;
#ifdef HOIC_PHASE_TIMING
  long long pt[24], pt_last;
#endif
};

#ifdef HOIC_PHASE_TIMING
#define PT(i) do { long long t_ = (long long)__builtin_readcyclecounter(); if (threadIdx.x == 0) { w.pt[i] += t_ - w.pt_last; w.pt_last = t_; } } while (0)
#else
#define PT(i) do {} while (0)
#endif
