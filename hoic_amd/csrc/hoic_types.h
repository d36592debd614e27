// hoic_types.h — device-side constant tables and per-env LDS workspace (gfx950).
//
// One workgroup of ONE wavefront (64 lanes) owns one environment.  Everything an env needs during an
// env-step (15 fused substeps) lives in this workgroup's LDS or in the wave's registers; HBM is touched only
// for the persistent state (qpos/qvel/warm-start/lagged state), the action, the expert window, the hand-over
// record between the two kernels of a step and the outputs.
//
// Occupancy is the design constraint: the work is a long chain of small dependent operations, so throughput
// comes from resident waves.  The workspace is kept under 20 KB (8 envs per CU = 2 waves per SIMD) and the
// kernels under 256 registers: the joint-space inertia matrix lives in registers (one row per lane), model
// constants are re-read from the (L1/L2 resident) DevModel instead of being pinned in registers, scratch of
// phases that never overlap is overlaid, and the float64 residual-force QP runs in a second kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstddef>
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
#define NCROW (MAXCON * 4)   // contact rows (pyramid edges): 4 per condim-3 contact, 6 per condim-4, 1 per condim-1; contacts whose rows do not fit are cut
#define NCSLOT (NCROW / NT)  // contact rows handled per lane
static_assert(NCROW <= 256 && MAXCON <= 32, "Work::c_row0 holds a row index in 8 bits, Work::cr_ce a contact index in 5 (contact << 3 | edge)");
#define LD 33            // padded leading dimension of 32-wide LDS matrices (bank-conflict free)
#define NHB HOIC_NHANDBODY
#define NHG 19
#define MAXMESHV HOIC_MAX_MESHVERT
#define MAXMESHP HOIC_MAX_MESHPLANE
#define MAXROUND 3       // pointer-jumping rounds of the kinematics (tree depth <= 8)
#define SUM_DIRECT 4     // subtrees of at most this many bodies are summed directly (DevModel::body_sum)
#define MAXVRUN (MAXMESHV / HOIC_HULL_RUN_VERTS + HOIC_MAX_MESH)
#define MAXFRUN (MAXMESHP / HOIC_HULL_RUN_FACES + HOIC_MAX_MESH)

struct DevModel {
  int nbody, njnt, nq, nv, nu, ngeom, npair, nlevel, nround;
  int hand_body0, obj_body, hand_geom0, hand_geom1, obj_geom0, obj_geom1, hand_nq, hand_nv;
  float timestep, gravity[3], meaninertia, hand_mass;
  // bodies (depth-first order: a body's subtree is the index range [b, b + body_subtree[b]) )
  int body_parent[NB], body_depth[NB], body_jntadr[NB], body_jntnum[NB], body_dofadr[NB], body_dofnum[NB];
  int body_subtree[NB];
  int body_jump[MAXROUND][NB];  // round r composes body b with body_jump[r][b] (-1: already in the world frame)
  // composite (subtree) sums in two rounds instead of one range sum per body (the hand's root body would walk all 21 bodies
  // while the other lanes idle): body_sum[b] = 0 not needed (no dof looks at this body's composite), 1 = range sum over the
  // subtree (<= SUM_DIRECT bodies), 2 = own value + the composites of the children body_kids[b] (all of them mode 1)
  int body_sum[NB];
  unsigned body_kids[NB][2];    // up to 8 children, one byte each (0xFF = none)
  unsigned body_dofmask[NB];    // dofs on the path root -> body
  unsigned body_path[NB][3];    // the same path as up to 12 packed dof indices (0xFF = none), root first
  int max_path;                 // dofs on the longest path (the gathers skip entries 10, 11 when none has more than ten)
  float body_pos[NB][3], body_quat[NB][4], body_ipos[NB][3], body_iquat[NB][4], body_mass[NB], body_inertia[NB][3];
  // joints / dofs
  int jnt_type[NJ], jnt_qposadr[NJ], jnt_dofadr[NJ], jnt_bodyid[NJ], jnt_limited[NJ];
  float jnt_pos[NJ][3], jnt_axis[NJ][3], jnt_range[NJ][2], jnt_margin[NJ], jnt_K[NJ], jnt_B[NJ], jnt_solimp[NJ][5];
  float jnt_diag[NJ];  // dof_invweight0 of the joint's dof (limit row diagApprox)
  int jnt_poszero;     // every hinge / slide joint sits at its body's origin (jnt_pos == 0, the HOIC hand): the kinematics skip the anchor rotations
  float qpos0[NQP];
  int dof_bodyid[NV], dof_jntid[NV], dof_actid[NV];
  // per-dof copies of the joint / body tables (flat: one load level, no joint -> dof -> body index chains)
  int dof_jtype[NV], dof_k[NV], dof_parentbody[NV], dof_qadr[NV], dof_limited[NV];
  unsigned dof_bpath[NV][3];   // body_path of the dof's body
  float dof_range[NV][2], dof_margin[NV], dof_limK[NV], dof_limB[NV], dof_limdiag[NV], dof_solimp[NV][5];
  unsigned dof_amask[NV];   // dofs strictly above dof d on its path (ancestors)
  unsigned dof_dmask[NV];   // dofs strictly below dof d (every dof whose path contains d)
  float dof_armature[NV], dof_damping[NV], dof_frictionloss[NV];
  float dof_flR[NV], dof_flB[NV];  // friction-loss row regulariser R and damping B (K = 0)
  int act_dofid[NU];
  // geoms
  int geom_type[NG], geom_bodyid[NG], geom_meshid[NG];
  float geom_size[NG][3], geom_pos[NG][3], geom_quat[NG][4], geom_rbound[NG];
  // static collision pair list with mixed parameters
  int pair_geom1[NPAIR], pair_geom2[NPAIR], pair_condim[NPAIR], pair_b1[NPAIR], pair_b2[NPAIR];
  int pair_type1[NPAIR], pair_type2[NPAIR], pair_mesh[NPAIR];
  float pair_bound[NPAIR];   // rbound1 + rbound2 + margin
  float pair_size1[NPAIR][3], pair_size2[NPAIR][3];
  unsigned pair_mpos[NPAIR], pair_mneg[NPAIR];   // dofs moving body2 only / body1 only
  float pair_mu[NPAIR][3], pair_K[NPAIR], pair_B[NPAIR], pair_solimp[NPAIR][5], pair_margin[NPAIR], pair_gap[NPAIR];
  float pair_Rscale[NPAIR];  // R = max(MINVAL,(1-imp)/imp) * Rscale  (pyramidal: 2 mu^2 tran (1+mu^2); condim 1: tran)
  int pair_pool[NPAIR];      // pool entry of a pair that can produce more than COLSLOT contacts (index within its pass of 64 pairs), else -1
  // convex meshes (hull vertices and face planes in the geom frame)
  int mesh_vertadr[HOIC_MAX_MESH], mesh_vertnum[HOIC_MAX_MESH], mesh_planeadr[HOIC_MAX_MESH], mesh_planenum[HOIC_MAX_MESH];
  __attribute__((aligned(16))) float mesh_vert[MAXMESHV][4];    // hull vertices (x, y, z, 0), geom frame; float4 loads
  __attribute__((aligned(16))) float mesh_plane[MAXMESHP][4];   // hull faces n.x <= d in the geom frame
  // ---- run bounds of the hull tables (build_model; exact pruning of the narrow phase's hull queries, hoic_collide.h)
  int mesh_prune;                                  // 0: stream every table entry (HOIC_MESH_STREAM=1, A/B and the identity test)
  int obb_reject;                                  // 0: no oriented-box reject in the collision driver (HOIC_NO_OBB_REJECT=1, A/B and its test)
  int warm_shift;                                  // 1: the Newton solve's warm start is a_smooth + (what the constraints added last substep); 0: MuJoCo's
                                                   //    plain qacc_warmstart = the last substep's qacc (HOIC_PLAIN_WARMSTART=1, A/B)
  int mesh_vrunadr[HOIC_MAX_MESH], mesh_vrunnum[HOIC_MAX_MESH], mesh_frunadr[HOIC_MAX_MESH], mesh_frunnum[HOIC_MAX_MESH];
  __attribute__((aligned(16))) float mesh_aabb[HOIC_MAX_MESH][8];      // lo xyz _, hi xyz _ of the hull vertices (mesh frame)
  __attribute__((aligned(16))) float mesh_vrun[MAXVRUN][4];            // bounding sphere of a run of HOIC_HULL_RUN_VERTS vertices: centre, radius
  __attribute__((aligned(16))) float mesh_frun[MAXFRUN][12];           // run of HOIC_HULL_RUN_FACES faces: (c, emax) (nlo, 0) (nhi, 0):
                                                                       //   n.x - d <= sum_i max(nlo_i w_i, nhi_i w_i) + emax, w = x - c
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

// hand-over record written by the substep kernel and read by the post-step kernel (floats per env)
#define PB_OK 0
#define PB_ITER 1
#define PB_XPOS 2
#define PB_XQUAT (PB_XPOS + NB * 3)
#define PB_GXPOS (PB_XQUAT + NB * 4)
#define PB_REC (PB_GXPOS + NG * 3)
#define PB_RECCNT (PB_REC + NHG * 12)
#define PB_GVEL (PB_RECCNT + NHG)
#define PB_GANGVEL (PB_GVEL + NG * 3)
#define PB_OBJACC (PB_GANGVEL + NG * 3)
// split post-step (hoic_set_async_reward): what the reward part needs of the state BEFORE the in-launch reset
#define PB_QPOS 704                 // final qpos / qvel of the step
#define PB_QVEL (PB_QPOS + NQP)
#define PB_EV (PB_QVEL + NV)        // expert view as int bits: sequence offset, length, start, cur_t (already advanced)
#define PB_DONE (PB_EV + 4)         // 1: the env was reset in the launch (clear the QP's warm-start flag)
#define PB_SIZE 784
static_assert(PB_OBJACC + 6 <= PB_QPOS && PB_DONE + 1 <= PB_SIZE, "post buffer layout");
// geom poses at launch start (for the 15-substep finite differences, ho_im4.py:553-559)
#define OG_SIZE (NG * 12)
// quantities of an env's last forward pass (the pass on the state before the last integration) that the first substep
// of the next env step reads (one-substep lag): M in the MReg layout, bias, motion axes, contacts
#define LG_M 0
#define LG_BIAS (LG_M + 16 * NT)
#define LG_S (LG_BIAS + NV)
#define LG_NCON (LG_S + NV * 6)
#define LG_CPOS (LG_NCON + 4)
#define LG_CFRAME (LG_CPOS + MAXCON * 3)
#define LG_CGEOM (LG_CFRAME + MAXCON * 9)
#define LG_SIZE (LG_CGEOM + MAXCON)

// persistent per-env state in HBM (row per env)
struct DevState {
  float* qpos;   // [n, NQP]
  float* qvel;   // [n, NV]
  float* warm;   // [n, NV]  warm start of the next solve: qacc - a_smooth of the last one (DevModel::warm_shift), or MuJoCo's plain qacc
  float* qlag;   // [n, NQP]  state before the last integration (one-substep lag)
  float* vlag;   // [n, NV]
  int* cur_t;    // [n]
  int* start;    // [n]
  int* seq;      // [n]
  float* rfc_score;  // [n]
  int* diag;     // [n, 2] running counters: forward passes that found more than MAXCON contacts (the list is cut after
                 // MAXCON), substeps whose Newton loop used up cfg.solver_iterations without meeting a stop criterion
  float* post;   // [2, n, PB_SIZE]: two buffers, alternated per step by the split post-step (buffer 0 otherwise)
  float* oldg;   // [n, OG_SIZE]
  float* lagrec; // [n, LG_SIZE]
  int* lag_valid; // [n] 0: the record does not belong to (qlag, vlag) (after a reset / set_state / failed step)
  float* qpcol;   // [n, QP_COL_FLOATS] columns of the residual-force QP (post-step kernel scratch)
  double* qp_lam; // [n, 8] multipliers of the last residual-force QP: lambda[6], valid flag, pad (diagnostic; the active-set solve starts cold)
  long long* phase;  // [n, 24] per-phase cycle counters (HOIC_PHASE_TIMING builds only)
  unsigned* cost;    // [2, n] shader-clock duration (>> 6) of the env's last substep / post-step pass
  int* order;        // [2, n] launch order of the next step: workgroup b runs env order[b], longest first
};

// joint-space inertia matrix in the layout of a v_mfma_f32_32x32x2_f32 accumulator: lane (col + 32*hi) holds, in
// register reg, M[row][col] with row = (reg & 3) + 8 * (reg >> 2) + 4 * hi.  M is symmetric, so the same 16 values are
// also one half of ROW col (columns `row`): matrix-vector products need one cross-half add, the solves a plain copy.
struct MReg { float r[16]; };
#define MREG_ROW(reg, hi) (((reg) & 3) + 8 * ((reg) >> 2) + 4 * (hi))

struct DofK { float floss, flR; };   // friction-loss row constants of dof (lane & 31), held in registers only inside the solve

// ---- per-env LDS workspace of the substep kernel (and of the probe kernel): 12.5 KB, i.e. TWELVE environments per CU =
// three wavefronts per SIMD (the gfx950 LDS allocation granule is 1280 B: <= 12800 B buys the twelfth workgroup).
// Round 3's struct held 19.98 KB (eight per CU).  What went where:
//   * the state before the last integration (qlag, vlag) and the contact sums of record_contact live in global memory only
//     (written once per substep / touched only when a hand-object contact exists); the clipped action is re-read from it;
//   * qacc doubles as the warm start of the next solve (they were copies of each other);
//   * per-dof model constants other than the body paths are re-read from the L1/L2-resident DevModel;
//   * vectors that live only between two matrix-core solves (a_smooth, the search direction, the PD torques) sit in the
//     solves' own scratch (sc.vec overlays sc.T);
//   * contact rows: four per contact (NCROW = 128; the rare condim-4 contacts take six, the list is cut when the rows run out,
//     counted in diag[0] like a contact overflow), reference accelerations in registers, p x f_k formed on the fly from c_pos;
//   * the collision phase stages two contacts per lane (+ a pool for the pairs that can produce four) and the geom rotation
//     matrices in the region that the dynamics scratch and the solver arrays occupy at other times.
// scratch of phases that never overlap: dynamics temporaries / the columns of L during a matrix-core solve / vectors that
// live only between two solves
union WorkScratch {
  struct {
    float I10[NB][10], Ic[NB][10];   // body / composite spatial inertias about the origin (Ic later: subtree forces)
    union {
      struct { float fS[NV][6], cfrc[NB][6]; } f;   // Ic*S (later: velocity-product accelerations), body bias forces
      struct { float jax[NJ][3], janc[NJ][3]; } j;  // joint axes / anchors in the parent-body frame
    } u;
  } dyn;
  float T[NV * LD];                  // columns of L during the matrix-core solves
  struct { float x[NV], ctrl[NV]; } vec;   // between two solves: a_smooth / the search direction; the PD torques
  // the solve's set-up (between the M^-1 f solve and the first Newton solve: T is dead, vec is live): body velocities and
  // contact-frame products of the THREE vectors that go through the contact Jacobian there (qvel, a_smooth, the warm start)
  struct { float keep[2 * NV]; float bV[3][NB][6]; float u[3][MAXCON * 4]; } mv;
};
static_assert(sizeof(((WorkScratch*)0)->mv) <= sizeof(((WorkScratch*)0)->T), "the set-up's buffers must fit into the solves' scratch");
#define COLSLOT 2                // contacts staged per lane (= pair) in col_lc
#define COLPOOL 32               // pairs per pass that can produce more than COLSLOT contacts (DevModel::pair_pool): 2 more each
struct Work {
  float qpos[NQP], qvel[NV], qacc[NV];        // qacc: result of the last solve = warm start of the next one
  // kinematics of the last forward pass
  float xpos[NB][3], xquat[NB][4];
  float gxpos[NG][3];
  union {
    struct {
      WorkScratch sc;                      // scratch shared by phases that never overlap
      // ---- solver phase: derived per-contact data, contact rows.
      // Jacobian-free contacts: row (c,k) of the contact Jacobian is
      //   sg(dof,c) * S[dof] . W[c][k],   W[c][k] = [p x f_k ; f_k] (k = n,t1,t2),  [f_n ; 0] (spin)
      // with p = c_pos and f_k = c_frame; p x f_k is never stored (f_k . (v + w x p) instead).
      float c_mu[MAXCON][3], c_D[MAXCON], c_aref0[MAXCON], c_B[MAXCON];
      unsigned char c_nrow[MAXCON], c_b1[MAXCON], c_b2[MAXCON], c_row0[MAXCON];
      unsigned c_mpos[MAXCON], c_mneg[MAXCON];   // dofs moving body2 only (+1) / body1 only (-1)
      union {
        struct { float bV[NB][6]; float u[MAXCON * 4]; };   // body spatial velocities for J.x products; u = (contact-frame J) x
        float c_G[MAXCON][6];                                // per-contact wrench of J'f (after the rows that read u)
      };
      float cr_force[NCROW], cr_curv[NCROW];     // contact rows (pyramid edges)
      unsigned char cr_ce[NCROW];                // contact << 3 | edge
    };
    // ---- collision phase: every lane (= pair) stages up to COLSLOT contacts of 7 floats (dist, pos, normal), element
    // (q, k) of lane l at col_lc[(q * 7 + k) * NT + l]; pairs that can produce more own a pool entry for contacts 2, 3;
    // clipping scratch of the wave-cooperative box-box; geom rotation matrices (kinematics -> collision only)
    struct { float col_lc[COLSLOT * 7 * NT]; float col_pool[COLPOOL][2 * 7]; float col_poly[32]; float gxmat[NG][9]; };
  };
  // contacts of the current forward pass
  int ncon, nrow, solver_iter, capped;
  float acon[NV];                      // qacc - a_smooth of the last solve: the acceleration the constraints added (warm start of the next solve)
  unsigned cbod;                       // bodies that take part in a contact (bit b), bit 31: one of them has a path of more than six dofs
  float c_pos[MAXCON][3], c_frame[MAXCON][9], c_dist[MAXCON];
  unsigned char c_pair[MAXCON], c_g1[MAXCON], c_g2[MAXCON];
  unsigned k_bpath[NB][3];             // packed dof paths of the bodies (model constant, loaded once per launch)
  float k_damp[NV];                    // joint damping per dof (model constant, loaded once per launch: read at the head of two of a
                                       // substep's solves, where a global read was an exposed latency each time)
  float applied[NV], bias[NV], ftot[NV];   // applied + actuator forces; bias forces; f_smooth + J'f of the last solve
  float S[NV][6];                      // motion axes [angular; linear at the world origin]
#ifdef HOIC_PHASE_TIMING
  long long pt[24], pt_last;
#endif
};
#ifndef HOIC_PHASE_TIMING
static_assert(sizeof(Work) <= 12800, "Work must stay within 10 LDS granules of 1280 B: 12 environments per CU");
#endif
static_assert(offsetof(Work, gxmat) >= offsetof(Work, c_mu), "gxmat must not overlay the dynamics scratch (written while it is live)");
static_assert(offsetof(Work, col_poly) + sizeof(((Work*)0)->col_poly) <= offsetof(Work, gxmat), "collision staging layout");

// ---- workspace of the post-step kernel, the reset kernel and the QP probe (float64 QP, kinematics of a reset): 8 KB.  The
// residual-force QP's columns (up to 19 contacts x 5 points x 4 edges, 7 floats each, column c of component k at
// [k * QP_MAXCOL + c]) live in global memory (DevState::qpcol), not here
#define QP_MAXCOL (NHG * 5 * 4)
#define QP_COL_FLOATS (7 * QP_MAXCOL)
struct PostWork {
  float qpos[NQP], qvel[NV], qacc[NV], action[NV];
  float xpos[NB][3], xquat[NB][4];
  float gxpos[NG][3];
  union {
    struct {
      float I10[NB][10], Ic[NB][10];
      union {
        struct { float fS[NV][6], cfrc[NB][6]; } f;
        struct { float jax[NJ][3], janc[NJ][3]; } j;
      } u;
    } dyn;
    struct {
      double qp_G[36];                 // residual-force QP: Gram matrix of the passive columns (packed lower 8 x 8)
      float qp_a[8][8];                //                    the passive columns themselves (a[6], c, pad)
      float avg_cps[NHG][12]; int avg_geom[NHG]; float avg_ts[NHG]; int n_avg;
      float gvel[NG][3], gangvel[NG][3], obj_avg_acc[6];
    } post;
  } sc;
  float S[NV][6];
  float gxmat[NG][9];
  float rec_sum[NHG][12]; int rec_cnt[NHG];
#ifdef HOIC_PHASE_TIMING
  long long pt[24], pt_last;
#endif
};

#ifdef HOIC_PHASE_TIMING
#define PT(i) do { long long t_ = (long long)__builtin_readcyclecounter(); if (threadIdx.x == 0) { w.pt[i] += t_ - w.pt_last; w.pt_last = t_; } } while (0)
#define PTC(i) do { if (threadIdx.x == 0) w.pt[i] += 1; } while (0)      // event counters in the slots the substep kernel has no phase for
#else
#define PT(i) do {} while (0)
#define PTC(i) do {} while (0)
#endif
