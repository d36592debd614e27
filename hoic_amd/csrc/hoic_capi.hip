// hoic_capi.hip — kernels + the C-ABI of include/hoic.h (libhoic_hip.so, gfx950 only).
//
// One workgroup = one wavefront = one environment; a launch covers all envs (grid = n_envs).  Inside a
// launch the whole env step (HandObjMimic4.step, uhc/envs/ho_im4.py:611-662) runs as two kernels: 15 substeps of
// control glue + dynamics + contact solve + integration, then contact averaging, the residual-force QP,
// termination, reward and the 617-float observation.  There is no CPU path in this library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#ifdef HOIC_TRACE_DISPATCH
__device__ int g_trace_bb[1 << 16];     // development aid: box-box turns per workgroup
__device__ int g_trace_qp[4 << 16];     // QP iterations / line-search steps / columns / warm per workgroup
#endif
#include "hoic_env.h"
#include "hoic_zfilter.h"

static thread_local std::string g_err;
static void set_err(const std::string& s) { g_err = s; }
void hoic_set_error(const std::string& s) { g_err = s; }     // for the other translation units of the library (hoic_mlp.hip)
extern "C" const char* hoic_last_error(void) { return g_err.c_str(); }
#ifndef HOIC_BUILD_ID
#define HOIC_BUILD_ID "unknown"
#endif
extern "C" const char* hoic_build_id(void) { return HOIC_BUILD_ID; }

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(std::string(#x) + ": " + hipGetErrorString(e_)); return HOIC_ERR_DEVICE; } } while (0)

// ------------------------------------------------------------------------------------------------ kernels
__device__ __forceinline__ void load_state(const DevModel& m, const DevState& st, Work& w, int env) {
  const int tid = threadIdx.x;
  if (tid < NQP) w.qpos[tid] = as_global(st.qpos)[(size_t)env * NQP + tid];
  if (tid < NV) {
    w.qvel[tid] = as_global(st.qvel)[(size_t)env * NV + tid];
    // the warm start of the first solve: the constraint acceleration of the env's last solve (dev_solve; zero after a reset), or
    // MuJoCo's plain qacc_warmstart
    const float wrow = as_global(st.warm)[(size_t)env * NV + tid];
    w.qacc[tid] = m.warm_shift ? 0.f : wrow; w.acon[tid] = m.warm_shift ? wrow : 0.f;
    w.applied[tid] = 0.f;
  }
  if (tid == 0) { w.ncon = 0; w.nrow = 0; w.solver_iter = 0; w.cbod = 0u; w.capped = 0; }
  wsync();
}
// lag_too: the state is also the state before the last integration (after a reset)
template <class W> __device__ __forceinline__ void store_state(const DevState& st, const W& w, int env, bool lag_too) {
  const int tid = opaque(threadIdx.x);
  if (tid < NQP) { as_global(st.qpos)[(size_t)env * NQP + tid] = w.qpos[tid]; if (lag_too) as_global(st.qlag)[(size_t)env * NQP + tid] = w.qpos[tid]; }
  if (tid < NV) {
    as_global(st.qvel)[(size_t)env * NV + tid] = w.qvel[tid]; if (lag_too) as_global(st.vlag)[(size_t)env * NV + tid] = w.qvel[tid];
    as_global(st.warm)[(size_t)env * NV + tid] = w.qacc[tid];
  }
}

// velocity / acceleration stages of mj_forward on the state in w.qpos/w.qvel with w.applied (applied + actuator forces) and the
// warm start (w.qacc) set (dev_forward_kin has run on the same state); a0_out: the unconstrained acceleration of dof lane & 31
__device__ __forceinline__ bool dev_forward_dyn(const DevModel& m, const DevConfig& cfg, Work& w, const MReg& M, float* a0_out, bool shift_warm) {
  const int tid = opaque(threadIdx.x), d = tid & 31;
  float fs = 0.f;
  if (d < m.nv) fs = -w.k_damp[d] * w.qvel[d] - w.bias[d] + w.applied[d];   // passive (joint damping) - bias + applied + actuation
  PT(20);
  float a0 = dev_hsolve<false>(m, w, M, 0.f, m.nv, fs); PT(8);     // unconstrained acceleration
  a0 = (d < m.nv) ? a0 : 0.f;
  if (a0_out) *a0_out = a0;
  if (tid < NV) w.sc.vec.x[tid] = a0;           // a_smooth as an LDS vector for the first row evaluation (dead once Newton starts)
  wsync();
  RowK rk;
  DofK dk;
  dev_make_constraint(m, w, rk, dk, w.qpos, w.qvel); PT(7);
  dev_solve(m, w, M, rk, dk, w.qvel, fs, a0, cfg.c.solver_iterations, shift_warm); PT(9);
  // mj_checkPos / mj_checkVel / mj_checkAcc [MJ-doc]: a non-finite or huge (> 1e10) entry of qpos, qvel or qacc is MuJoCo's
  // "Nan, Inf or huge value" warning, which mujoco_py raises and the env turns into fail = True (ho_im4.py:635-637)
  float bad = 0.f;
  if (tid < m.nv) {
    const float a = w.qacc[tid], v = w.qvel[tid];
    bad = (isfinite(a) && fabsf(a) < 1e10f && isfinite(v) && fabsf(v) < 1e10f) ? 0.f : 1.f;
  }
  for (int i = tid; i < m.nq; i += NT) { const float q = w.qpos[i]; if (!(isfinite(q) && fabsf(q) < 1e10f)) bad = 1.f; }
  return !(wave_max(bad) > 0.f);
}

// semi-implicit Euler with implicit joint damping; the pre-integration state goes to (gqlag, gvlag) in global memory (the env's
// rows of DevState::qlag / vlag; null: not recorded), the acceleration stays in w.qacc as the next warm start
__device__ __forceinline__ void dev_euler(const DevModel& m, Work& w, const MReg& M, GPTR(float) gqlag, GPTR(float) gvlag) {
  const int tid = opaque(threadIdx.x), d = tid & 31;
  const float h = m.timestep;
  const float rhs = (d < m.nv) ? w.ftot[d] : 0.f;
  const float damp = w.k_damp[d];
  PT(20);
  const float acc = dev_hsolve<false>(m, w, M, h * damp, m.nv, rhs);
  if (gqlag) {
    if (tid < NQP) gqlag[tid] = w.qpos[tid];
    if (tid < NV) gvlag[tid] = w.qvel[tid];
  }
  if (tid < m.nv) w.qvel[tid] += h * acc;
  wsync();
  if (tid < m.njnt) {
    const int qa = m.jnt_qposadr[tid], da = m.jnt_dofadr[tid];
    if (m.jnt_type[tid] == HOIC_JNT_FREE) {
      for (int i = 0; i < 3; i++) w.qpos[qa + i] += h * w.qvel[da + i];
      float wv[3] = {w.qvel[da + 3], w.qvel[da + 4], w.qvel[da + 5]};
      const float ang = normalize3(wv) * h;
      if (ang != 0.f) {
        float s, c;
        sincos_pi(0.5f * ang, &s, &c);
        float dq[4] = {c, s * wv[0], s * wv[1], s * wv[2]}, qo[4] = {w.qpos[qa + 3], w.qpos[qa + 4], w.qpos[qa + 5], w.qpos[qa + 6]}, qn[4];
        mulquat(qo, dq, qn);
        normquat(qn);
        for (int i = 0; i < 4; i++) w.qpos[qa + 3 + i] = qn[i];
      }
    } else w.qpos[qa] += h * w.qvel[da];
  }
  wsync();
}

// ---- what follows the substeps of an env step (HandObjMimic4.step after do_simulation, ho_im4.py:631-662): contact
// averaging, the residual-force QP (float64), termination, reward, the optional in-launch reset and the 617-float
// observation.  Expects in the workspace: final qpos / qvel, body / geom poses of the last forward pass and -- for the reward
// parts -- the clipped action, the contact sums and the 15-substep finite differences in sc.post.
// PART: POST_ALL = all of it (one kernel behind the substeps); the split form (hoic_set_async_reward) runs POST_A --
// termination, in-launch reset, observation: what the next policy forward waits for -- at the end of the substep kernel (on
// its slim workspace W = Work) and POST_B -- contact classification, residual-force QP, reward: needed only when the
// rollout's rewards are read -- in the post-step kernel on a side stream, from the hand-over record `rec` (which then also
// carries the pre-reset state and expert view).  Same arithmetic in every form; POST_B cannot fail a step retroactively
// (a non-finite QP score, never observed, zeroes the score in every form and fails the step in POST_ALL only).
enum { POST_ALL = 0, POST_A = 1, POST_B = 2 };
template <int PART, class W>
__device__ __forceinline__ void dev_poststep(const DevModel& m, const DevConfig& cfg, W& w, const DevExpert& ex, const DevState& st,
                                             ExpertView& ev, int env, int io, bool ok, int solver_iter, const float* vf, const float* vt,
                                             float* __restrict__ obs, float* __restrict__ reward, float* __restrict__ reward_info,
                                             int* __restrict__ flags, float* __restrict__ percent, const int* __restrict__ next_seq,
                                             const int* __restrict__ next_start, GPTR(float) rec) {
  const int tid = threadIdx.x;
  float rfc_score = 0.f;
  if constexpr (PART != POST_A) {
    if (ok) {
      dev_classify_contact(m, w);                                                                // :562
      if (cfg.c.residual_force) rfc_score = dev_solve_rfc(m, cfg, w, vf, vt, (double*)(as_global(st.qp_lam) + (size_t)env * 8),
                                                          as_global(st.qpcol) + (size_t)env * QP_COL_FLOATS);   // :631
      if (!isfinite(rfc_score)) { if (PART == POST_ALL) ok = false; rfc_score = 0.f; }
    }
    asm volatile("" ::: "memory");   // keep the expert-frame loads of the reward / observation below the QP (register peak)
  }
  if (PART != POST_B) ev.cur_t += 1;                                                            // :641 (POST_B: the record's view is advanced already)
  const int expert_len = ev.len - ev.start;
  const bool end = ev.cur_t >= expert_len - cfg.c.future_w_size - 1;                            // :657
  if constexpr (PART != POST_A) {
    float rw[10];
    dev_reward(m, cfg, w, ev, rfc_score, rw);
    float r = rw[0];
    if (cfg.rp.use_end_reward && end) r += cfg.rp.end_reward;                                   // agent_handmimic.py:479-480
    if (tid == 0) { reward[io] = r; as_global(st.rfc_score)[env] = rfc_score; }
    if (tid < HOIC_NREWARD_INFO) reward_info[(size_t)io * HOIC_NREWARD_INFO + tid] = rw[1 + tid];
    if (PART == POST_B) {
      if (tid == 6 && rec[PB_DONE] != 0.f) as_global(st.qp_lam)[(size_t)env * 8 + 6] = 0.0;
      return;
    }
  }
  float df[5];
  dev_ho_diff(m, w, ev, df);
  const bool body_fail = df[0] > cfg.c.pos_diff_thresh || df[1] > cfg.c.rot_diff_thresh || df[2] > cfg.c.jpos_diff_thresh ||
                         df[3] > cfg.c.obj_pos_diff_thresh || df[4] > cfg.c.obj_rot_diff_thresh;
  bool fail = !ok;
  if (cfg.mode_train) fail = fail || body_fail;                                                 // :655-656
  const bool done = fail || end;
  if (tid == 0) {
    flags[4 * io] = fail; flags[4 * io + 1] = end; flags[4 * io + 2] = done; flags[4 * io + 3] = solver_iter;
    percent[io] = (float)ev.cur_t / (float)(expert_len - 1);                                   // :660
  }
  const bool reset = done && next_seq != nullptr;
  if (PART == POST_A) {        // the reward part's inputs, before the reset below replaces them
    if (tid < NQP) rec[PB_QPOS + tid] = w.qpos[tid];
    if (tid < NV) rec[PB_QVEL + tid] = w.qvel[tid];
    if (tid == 0) {
      rec[PB_EV] = __int_as_float(ev.off); rec[PB_EV + 1] = __int_as_float(ev.len); rec[PB_EV + 2] = __int_as_float(ev.start);
      rec[PB_EV + 3] = __int_as_float(ev.cur_t); rec[PB_DONE] = reset ? 1.f : 0.f;
    }
  }
  if (reset) {   // the sampler's next episode (agent_handmimic.py:444-454) in the same launch
    const int ns = min(max(next_seq[io], 0), ex.n_seq - 1);
    const int nst = min(max(next_start[io], 0), as_global(ex.seq_len)[ns] - 2);
    wsync();
    dev_reset_state(m, w, ex, ns, nst);
    dev_kinematics(m, w, w.qpos);
    if (PART == POST_ALL && tid == 6) as_global(st.qp_lam)[(size_t)env * 8 + 6] = 0.0;
    ev.off = as_global(ex.seq_off)[ns]; ev.len = as_global(ex.seq_len)[ns]; ev.start = nst; ev.cur_t = 0;
    if (tid == 0) { as_global(st.seq)[env] = ns; as_global(st.start)[env] = nst; as_global(st.lag_valid)[env] = 0; }
    store_state(st, w, env, true);
  }
  dev_write_obs(m, w, ev, obs + (size_t)io * HOIC_OBS_DIM);
  if (tid == 0) as_global(st.cur_t)[env] = ev.cur_t;
}

// ---- kernel 1 of a step: the 15 substeps (control glue + dynamics + contact solve + integration), f32.
// Leaves the new state in HBM and a hand-over record (lagged body / geom poses, contact sums, 15-substep finite
// differences) for the post-step kernel.
// THREE wavefronts per SIMD: the 12.5 KB workspace puts 12 envs on a CU and the kernel is held to 168 registers (round 3:
// 19.98 KB, 200 registers, two wavefronts); the work is a long dependent chain per env, throughput follows resident waves.
template <int MODE>      // 0: substeps + hand-over record, 2: + the post-step part POST_A (split form)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3, 3))) void hoic_substep_kernel(const DevModel* __restrict__ mp, const DevConfig* __restrict__ cp,
                                                          const DevExpert* __restrict__ exq, const DevState* __restrict__ stq,
                                                          const float* __restrict__ action, int first, int use_order, int use_lag,
                                                          float* __restrict__ obs, float* __restrict__ reward, float* __restrict__ reward_info,
                                                          int* __restrict__ flags, float* __restrict__ percent, const int* __restrict__ next_seq,
                                                          const int* __restrict__ next_start, int n_envs, int post_buf) {
  // the expert / state pointer tables stay in device memory (22 pointers would otherwise be pinned in SGPRs)
  __shared__ Work w;
  const DevModel& m = *mp; const DevConfig& cfg = *cp; const DevExpert& ex = *exq; const DevState& st = *stq;
  // workgroup b runs env first + b (I/O row b), or env order[b] of a whole-batch step in longest-first order
  const int env = use_order ? as_global(st.order)[first + blockIdx.x] : first + (int)blockIdx.x, io = env - first, tid = threadIdx.x;
  const long long clk0 = (long long)__builtin_readcyclecounter();
#ifdef HOIC_TRACE_DISPATCH
  const long long trace_t0 = (long long)__builtin_amdgcn_s_memrealtime();
  int trace_ncon = 0, trace_iter = 0, trace_ncon_max = 0, trace_late = 0, trace_last = 0;
  if (tid == 0) g_trace_bb[blockIdx.x] = 0;
#endif
  load_state(m, st, w, env);
  dev_load_constants(m, w);
  MReg M;
#ifdef HOIC_PHASE_TIMING
  if (tid == 0) { for (int i = 0; i < 24; i++) w.pt[i] = 0; w.pt_last = (long long)__builtin_readcyclecounter(); }
  wsync();
#endif
  GPTR(const float) act = as_global(action) + (size_t)io * HOIC_ACT_DIM;       // clipped where it is read (ho_im4.py:613)
  const int seq = as_global(st.seq)[env];
  ExpertView ev{&ex, as_global(ex.seq_off)[seq], as_global(ex.seq_len)[seq], as_global(st.start)[env], as_global(st.cur_t)[env]};
  int* ovf = (int*)&as_global(st.diag)[2 * env];
  int ncapped = 0;
  GPTR(float) post = as_global(st.post) + ((size_t)post_buf * n_envs + env) * PB_SIZE;
  GPTR(float) oldg = as_global(st.oldg) + (size_t)env * OG_SIZE;
  GPTR(float) gqlag = as_global(st.qlag) + (size_t)env * NQP;
  GPTR(float) gvlag = as_global(st.vlag) + (size_t)env * NV;
  // record_contact's sums of this env step accumulate in the record itself
  for (int k = tid; k < NHG * 13; k += NT) post[PB_REC + k] = 0.f;
  static_assert(PB_RECCNT == PB_REC + NHG * 12, "contact sums and counts are contiguous");
  // One loop, three modes, so that every stage has a single (inlined) call site:
  //   mode 0  the forward pass on the lagged state: quantities of the previous forward pass (one-substep lag,
  //           SURVEY.md row Q1) are recomputed instead of being persisted
  //   mode 1  a substep: control glue, forward dynamics, Euler
  //   mode 2  after a failed substep: kinematics of the restored state for the observation / reward
  float old_objvel = 0.f;
  bool ok = true;
  const int nsub = cfg.c.sim_step;
  int mode = 0, done_sub = 0;
  // The quantities of the previous forward pass that the first substep reads (M and bias for the PD torque, motion axes
  // for the applied forces, contacts for the bookkeeping) were left behind by the previous launch; only an env whose
  // state was replaced since (reset, set_state, failed step) recomputes them with a forward pass on (qlag, vlag).
  GPTR(float) lag = as_global(st.lagrec) + (size_t)env * LG_SIZE;
  if (use_lag && as_global(st.lag_valid)[env] != 0 && nsub > 0) {
#pragma unroll
    for (int reg = 0; reg < 16; reg++) M.r[reg] = lag[LG_M + reg * NT + tid];
    if (tid < NV) w.bias[tid] = lag[LG_BIAS + tid];
    for (int k = tid; k < NV * 6; k += NT) w.S[k / 6][k % 6] = lag[LG_S + k];
    const int nc = (int)lag[LG_NCON];
    if (tid == 0) w.ncon = nc;
    if (tid < nc) {
      for (int i = 0; i < 3; i++) w.c_pos[tid][i] = lag[LG_CPOS + tid * 3 + i];
      for (int i = 0; i < 9; i++) w.c_frame[tid][i] = lag[LG_CFRAME + tid * 9 + i];
      const int gg = (int)lag[LG_CGEOM + tid];
      w.c_g1[tid] = (unsigned char)(gg & 255); w.c_g2[tid] = (unsigned char)(gg >> 8);
    }
    if (tid < 3) w.gxpos[2][tid] = oldg[2 * 12 + tid];
    old_objvel = (tid < 6) ? w.qvel[m.nv - 6 + tid] : 0.f;
    mode = 1;
    wsync();
  }
  while (true) {
    // the model / config pointers are laundered every pass: otherwise the compiler hoists the (loop-invariant)
    // model-constant loads out of the loop and pins them in registers
    // (the laundered pointers keep their global address space: a generic pointer would turn every model read into a
    // flat_load, which counts on the LDS counter as well and forces s_waitcnt vmcnt(0) lgkmcnt(0) drains)
    GPTR(const DevModel) mq = (GPTR(const DevModel))mp; GPTR(const DevConfig) cq = (GPTR(const DevConfig))cp;
    asm volatile("" : "+s"(mq), "+s"(cq));
    const DevModel& ml = *(const DevModel*)mq; const DevConfig& cl = *(const DevConfig*)cq;
    PT(0);
    if (mode == 1) {
      const PdFetch pf = dev_pd_fetch(ml, cl, ev, act);
      dev_record_contact(ml, w, post); PT(2);          // :543 (contacts of the previous forward pass)
      dev_pd_torque(ml, cl, w, M, pf); PT(1);          // :518-523
      dev_applied(ml, cl, w, act);                     // :526-540
    }
    if (mode == 0) {   // the lagged pass runs on (qlag, vlag): into the workspace's state for the pass, the state proper comes back below
      const int tl = opaque(tid);      // (the lane's global addresses are formed here, not kept in registers across the loop)
      if (tl < NQP) w.qpos[tl] = gqlag[tl];
      if (tl < NV) w.qvel[tl] = gvlag[tl];
      wsync();
    }
    dev_forward_kin(ml, cl, w, M, w.qpos, w.qvel, ovf);      // (ovf: wave-uniform; only lane 0 writes through it)
    if (mode == 0) {
      for (int g = tid; g < ml.ngeom; g += NT) {
        for (int i = 0; i < 3; i++) oldg[g * 12 + i] = w.gxpos[g][i];
        for (int i = 0; i < 9; i++) oldg[g * 12 + 3 + i] = w.gxmat[g][i];
      }
      const int tl = opaque(tid);
      if (tl < NQP) w.qpos[tl] = as_global(st.qpos)[(size_t)env * NQP + tl];
      if (tl < NV) w.qvel[tl] = as_global(st.qvel)[(size_t)env * NV + tl];
      wsync();
      old_objvel = (tid < 6) ? w.qvel[ml.nv - 6 + tid] : 0.f;
      mode = 1;
      if (nsub <= 0) break;
      continue;
    }
    if (mode == 2) break;
    ok = dev_forward_dyn(ml, cl, w, M, nullptr, ml.warm_shift != 0);       // :545 mj_step = forward ...
    ncapped += w.capped;
#ifdef HOIC_TRACE_DISPATCH
    trace_ncon += w.ncon; trace_iter += w.solver_iter; trace_ncon_max = max(trace_ncon_max, w.ncon);
    if (done_sub >= nsub - 5) trace_late += w.solver_iter;
    if (done_sub >= nsub - 1) trace_last = w.solver_iter * 100 + w.ncon;
#endif
    if (!ok) {  // the reference converts the MuJoCo exception into fail=True (:635-637); keep the last finite state
      const int tl = opaque(tid);
      if (tl < NQP) w.qpos[tl] = gqlag[tl];
      if (tl < NV) { w.qvel[tl] = gvlag[tl]; w.qacc[tl] = 0.f; }
      wsync();
      mode = 2;
      continue;
    }
    dev_euler(ml, w, M, gqlag, gvlag); PT(10);       //              ... + Euler
    if (++done_sub >= nsub) break;
  }
  PT(0);
  const int te = opaque(tid);       // (lane id of the epilogue: its global addresses are formed here, not carried through the loop)
  {   // 15-substep finite differences (:554-559) -> the hand-over record; the geom poses of the last forward pass -> oldg
    const float dt = (float)nsub * m.timestep, idt = ok ? 1.f / dt : 0.f;
    const bool hand_over = ok && nsub > 0;
    if (te < 6) post[PB_OBJACC + te] = ok ? (w.qvel[m.nv - 6 + te] - old_objvel) * idt : 0.f;                   // :554
    for (int g = te; g < m.ngeom; g += NT) {
      float gv[3] = {0.f, 0.f, 0.f}, ga[3] = {0.f, 0.f, 0.f};
      if (ok) {
        // the geom's rotation matrix from its body's pose (the LDS copy of the forward pass lived only until the collision
        // stage: same arithmetic as dev_kinematics step 3c)
        const int b = m.geom_bodyid[g];
        const float gq[4] = {m.geom_quat[g][0], m.geom_quat[g][1], m.geom_quat[g][2], m.geom_quat[g][3]};
        float qg[4], Rg[9];
        mulquat(w.xquat[b], gq, qg);
        quat2mat(qg, Rg);
        for (int i = 0; i < 3; i++) gv[i] = (w.gxpos[g][i] - oldg[g * 12 + i]) * idt;                  // :555
        float Rd[9], aa[3], Ro[9];
        for (int i = 0; i < 9; i++) Ro[i] = oldg[g * 12 + 3 + i];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) {
            float sm = 0.f;
            for (int k = 0; k < 3; k++) sm += Rg[3 * i + k] * Ro[3 * j + k];
            Rd[3 * i + j] = sm;
          }
        dev_matrix_to_axis_angle(Rd, aa);                                                         // :556-559
        for (int i = 0; i < 3; i++) ga[i] = aa[i] * idt;
        if (hand_over) {     // (it ran on the state that is now qlag, vlag: the next launch's "poses at launch start")
          for (int i = 0; i < 3; i++) oldg[g * 12 + i] = w.gxpos[g][i];
          for (int i = 0; i < 9; i++) oldg[g * 12 + 3 + i] = Rg[i];
        }
      }
      for (int i = 0; i < 3; i++) { post[PB_GVEL + g * 3 + i] = gv[i]; post[PB_GANGVEL + g * 3 + i] = ga[i]; }
    }
  }
  if (te == 0) { post[PB_OK] = ok ? 1.f : 0.f; post[PB_ITER] = (float)w.solver_iter; }
  for (int k = te; k < m.nbody * 3; k += NT) post[PB_XPOS + k] = w.xpos[k / 3][k % 3];
  for (int k = te; k < m.nbody * 4; k += NT) post[PB_XQUAT + k] = w.xquat[k / 4][k % 4];
  for (int k = te; k < m.ngeom * 3; k += NT) post[PB_GXPOS + k] = w.gxpos[k / 3][k % 3];
  if (m.warm_shift) {          // what persists as the warm start is the constraint acceleration (zero after a failed substep)
    wsync();
    if (te < NV) w.qacc[te] = ok ? w.acon[te] : 0.f;
    wsync();
  }
  store_state(st, w, env, false);
  if (ok && nsub > 0) {       // hand the last forward pass over to the next launch (it ran on the state that is now qlag, vlag)
#pragma unroll
    for (int reg = 0; reg < 16; reg++) lag[LG_M + reg * NT + te] = M.r[reg];
    if (te < NV) lag[LG_BIAS + te] = w.bias[te];
    for (int k = te; k < NV * 6; k += NT) lag[LG_S + k] = w.S[k / 6][k % 6];
    const int nc = w.ncon;
    if (te == 0) lag[LG_NCON] = (float)nc;
    if (te < nc) {
      for (int i = 0; i < 3; i++) lag[LG_CPOS + te * 3 + i] = w.c_pos[te][i];
      for (int i = 0; i < 9; i++) lag[LG_CFRAME + te * 9 + i] = w.c_frame[te][i];
      lag[LG_CGEOM + te] = (float)((int)w.c_g1[te] | ((int)w.c_g2[te] << 8));
    }
  }
  if (tid == 0) { as_global(st.lag_valid)[env] = (ok && nsub > 0) ? 1 : 0; if (ncapped) as_global(st.diag)[2 * env + 1] += ncapped; }
  const long long clk1 = (long long)__builtin_readcyclecounter();
  if (tid == 0) as_global(st.cost)[env] = (unsigned)((clk1 - clk0) >> 6);
  if (MODE != 0) {
    wsync();
    dev_poststep<POST_A>(m, cfg, w, ex, st, ev, env, io, ok, w.solver_iter, nullptr, nullptr, obs, reward, reward_info, flags, percent,
                         next_seq, next_start, post);
  }
#ifdef HOIC_TRACE_DISPATCH   // development aid: when and where (XCC / SE / CU / SIMD) each env ran, constant 100 MHz clock
  if (tid == 0) {
    GPTR(long long) tr = as_global(st.phase) + (size_t)env * 24;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    tr[0] = trace_t0; tr[1] = (long long)__builtin_amdgcn_s_memrealtime(); tr[2] = hw; tr[3] = xcc; tr[4] = blockIdx.x;
    tr[5] = trace_ncon; tr[6] = trace_iter; tr[7] = g_trace_bb[blockIdx.x]; tr[8] = trace_ncon_max; tr[9] = trace_late; tr[10] = trace_last;
  }
#endif
  PT(13);
#ifdef HOIC_PHASE_TIMING
  if (tid < 24) as_global(st.phase)[(size_t)env * 24 + tid] = w.pt[tid];
#endif
}

// ---- kernel 2 of a step: contact averaging, the residual-force QP (float64), termination, reward, the optional
// in-launch reset and the 617-float observation (HandObjMimic4.step after do_simulation, ho_im4.py:631-662)
// WPE: resident wavefronts per SIMD the register budget is set for: 3 = 168 registers (a post-step workgroup costs what a substep
// workgroup costs; the float64 active-set QP spills 146 values at that budget, inside the branch the envs with a hand-object
// contact take), 2 = 256 registers (80 B of scratch left; such a wavefront needs a SIMD with at most one substep wavefront on it).
// HOIC_POSTB_WIDE=1 selects the wide build for the split form's reward part (A/B on the contact-rich workload).
template <int PART, int WPE = 3>      // POST_ALL: everything after the substeps; POST_B: the reward part of the split form (record-only inputs)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void hoic_poststep_kernel(const DevModel* __restrict__ mp, const DevConfig* __restrict__ cp,
                                                           DevExpert ex, DevState st, const float* __restrict__ action,
                                                           float* __restrict__ obs, float* __restrict__ reward,
                                                           float* __restrict__ reward_info, int* __restrict__ flags,
                                                           float* __restrict__ percent, const int* __restrict__ next_seq,
                                                           const int* __restrict__ next_start, int first, int use_order, int n_envs, int post_buf) {
  __shared__ PostWork w;
  const DevModel& m = *mp; const DevConfig& cfg = *cp;
  const int env = use_order ? as_global(st.order)[n_envs + blockIdx.x] : first + (int)blockIdx.x, io = env - first, tid = threadIdx.x;
  const long long clk0 = (long long)__builtin_readcyclecounter();
#ifdef HOIC_TRACE_DISPATCH
  if (tid == 0) for (int i = 0; i < 4; i++) g_trace_qp[blockIdx.x * 4 + i] = 0;
#endif
  GPTR(float) post = as_global(st.post) + ((size_t)post_buf * n_envs + env) * PB_SIZE;
  // POST_B: the state and the expert view of the step come from the record (the env may have been reset since)
  if (tid < NQP) w.qpos[tid] = PART == POST_B ? post[PB_QPOS + tid] : as_global(st.qpos)[(size_t)env * NQP + tid];
  if (tid < NV) {
    w.qvel[tid] = PART == POST_B ? post[PB_QVEL + tid] : as_global(st.qvel)[(size_t)env * NV + tid];
    w.qacc[tid] = PART == POST_B ? 0.f : as_global(st.warm)[(size_t)env * NV + tid];     // (goes back with the state after an in-launch reset only)
    w.action[tid] = fminf(fmaxf(action[(size_t)io * HOIC_ACT_DIM + tid], -1.f), 1.f);
  }
  for (int k = tid; k < m.nbody * 3; k += NT) w.xpos[k / 3][k % 3] = post[PB_XPOS + k];
  for (int k = tid; k < m.nbody * 4; k += NT) w.xquat[k / 4][k % 4] = post[PB_XQUAT + k];
  for (int k = tid; k < m.ngeom * 3; k += NT) w.gxpos[k / 3][k % 3] = post[PB_GXPOS + k];
  for (int k = tid; k < NHG * 12; k += NT) w.rec_sum[k / 12][k % 12] = post[PB_REC + k];
  if (tid < NHG) w.rec_cnt[tid] = (int)post[PB_RECCNT + tid];
  for (int k = tid; k < m.ngeom * 3; k += NT) { w.sc.post.gvel[k / 3][k % 3] = post[PB_GVEL + k]; w.sc.post.gangvel[k / 3][k % 3] = post[PB_GANGVEL + k]; }
  if (tid < 6) w.sc.post.obj_avg_acc[tid] = post[PB_OBJACC + tid];
  bool ok = post[PB_OK] != 0.f;
  const int solver_iter = (int)post[PB_ITER];
  wsync();
  const int seq = as_global(st.seq)[env];
  ExpertView ev{&ex, as_global(ex.seq_off)[seq], as_global(ex.seq_len)[seq], as_global(st.start)[env], as_global(st.cur_t)[env]};
  if (PART == POST_B) {
    ev.off = __float_as_int(post[PB_EV]); ev.len = __float_as_int(post[PB_EV + 1]); ev.start = __float_as_int(post[PB_EV + 2]);
    ev.cur_t = __float_as_int(post[PB_EV + 3]);
  }
  float vf[3], vt[3];
  for (int i = 0; i < 3; i++) {
    vf[i] = cfg.c.residual_force ? cfg.c.residual_force_scale * w.action[m.nu + i] : 0.f;     // :622-623
    vt[i] = cfg.c.residual_force ? cfg.c.residual_torque_scale * w.action[m.nu + 3 + i] : 0.f;
  }
  dev_poststep<PART>(m, cfg, w, ex, st, ev, env, io, ok, solver_iter, vf, vt, obs, reward, reward_info, flags, percent, next_seq, next_start, post);
  if (tid == 0) {
    as_global(st.cost)[n_envs + env] = (unsigned)(((long long)__builtin_readcyclecounter() - clk0) >> 6);
#ifdef HOIC_TRACE_DISPATCH
    GPTR(long long) tr = as_global(st.phase) + (size_t)env * 24;
    for (int i = 0; i < 4; i++) tr[12 + i] = g_trace_qp[blockIdx.x * 4 + i];
#endif
  }
}

// ---- launch order of the next step (optional, HOIC_REORDER=1).  An env's pass takes between ~0.7x and ~1.8x the
// mean (contacts, Newton and QP iterations), and 4096 envs are only two rounds of the 2048 resident wavefronts, so
// the last round ends with a tail of half-empty CUs (27 % of the wave slots idle, tools/dispatch_trace.py).
// Longest-first dispatch by the measured duration of the previous pass shortens that tail when durations persist from
// step to step: a counting sort on 1024 duration bins, one workgroup per kernel (blockIdx 0: substep order, 1:
// post-step order).  The order only changes which CU runs an env, never a result.
// (256 bins / threads: 2 KB of LDS, so the workgroup fits beside the eight substep workgroups of a CU instead of waiting for one
//  of their slots -- inside the rollout the 1024-bin version took 51 us per call, most of it waiting)
#define ORDER_NT 256
// (first, count): the env range to order (a rollout range or the whole batch); order[first + k] = k-th longest env of the range
__global__ __launch_bounds__(ORDER_NT) void hoic_order_kernel(const unsigned* __restrict__ cost_all, int* __restrict__ order_all, int n_all, int first, int n) {
  __shared__ unsigned hist[ORDER_NT], scan[ORDER_NT];
  __shared__ unsigned lo, hi;
  const unsigned* cost = cost_all + (size_t)blockIdx.x * n_all + first;
  int* order = order_all + (size_t)blockIdx.x * n_all + first;
  const int tid = threadIdx.x;
  if (tid == 0) { lo = 0xFFFFFFFFu; hi = 0u; }
  hist[tid] = 0u;
  __syncthreads();
  unsigned mn = 0xFFFFFFFFu, mx = 0u;
  for (int i = tid; i < n; i += ORDER_NT) { const unsigned c = cost[i]; mn = min(mn, c); mx = max(mx, c); }
  atomicMin(&lo, mn); atomicMax(&hi, mx);
  __syncthreads();
  const unsigned base = lo, span = max(hi - lo, 1u);
  for (int i = tid; i < n; i += ORDER_NT) {
    const unsigned b = (ORDER_NT - 1) - (unsigned)(((unsigned long long)(cost[i] - base) * (ORDER_NT - 1)) / span);   // bin 0 = longest
    atomicAdd(&hist[b], 1u);
  }
  __syncthreads();
  // exclusive prefix sum over the bins
  unsigned v = hist[tid];
  scan[tid] = v;
  __syncthreads();
  for (int off = 1; off < ORDER_NT; off <<= 1) {
    const unsigned a = tid >= off ? scan[tid - off] : 0u;
    __syncthreads();
    scan[tid] += a;
    __syncthreads();
  }
  hist[tid] = scan[tid] - v;
  __syncthreads();
  for (int i = tid; i < n; i += ORDER_NT) {
    const unsigned b = (ORDER_NT - 1) - (unsigned)(((unsigned long long)(cost[i] - base) * (ORDER_NT - 1)) / span);
    order[atomicAdd(&hist[b], 1u)] = first + i;
  }
}

__global__ __launch_bounds__(NT) void hoic_reset_kernel(const DevModel* __restrict__ mp, DevExpert ex, DevState st,
                                                        const int* __restrict__ env_ids, const int* __restrict__ seqs,
                                                        const int* __restrict__ starts, float* __restrict__ obs, int n_envs) {
  __shared__ PostWork w;
  const DevModel& m = *mp;
  const int k = blockIdx.x, tid = threadIdx.x;
  const int env = env_ids ? env_ids[k] : k;
  if (env < 0 || env >= n_envs) return;              // ids outside the batch are ignored (wave-uniform)
  // device-side draws are clamped into the table: sequence in [0, n_seq), start so that two frames remain
  const int seq = min(max(seqs[k], 0), ex.n_seq - 1);
  const int start = min(max(starts[k], 0), as_global(ex.seq_len)[seq] - 2);
  if (tid < NQP) w.qpos[tid] = 0.f;
  wsync();
  dev_reset_state(m, w, ex, seq, start);
  dev_kinematics(m, w, w.qpos);
  ExpertView ev{&ex, as_global(ex.seq_off)[seq], as_global(ex.seq_len)[seq], start, 0};
  if (obs) dev_write_obs(m, w, ev, obs + (size_t)env * HOIC_OBS_DIM);
  store_state(st, w, env, true);
  if (tid == 0) { as_global(st.lag_valid)[env] = 0; as_global(st.cur_t)[env] = 0; as_global(st.start)[env] = start; as_global(st.seq)[env] = seq; as_global(st.rfc_score)[env] = 0.f; as_global(st.qp_lam)[(size_t)env * 8 + 6] = 0.0; }
}

__global__ __launch_bounds__(NT) void hoic_set_state_kernel(const DevModel* __restrict__ mp, DevState st,
                                                            const float* __restrict__ qpos, const float* __restrict__ qvel) {
  const int env = blockIdx.x, tid = threadIdx.x, nq = mp->nq, nv = mp->nv;
  if (tid < NQP) { const float v = tid < nq ? qpos[(size_t)env * nq + tid] : 0.f; as_global(st.qpos)[(size_t)env * NQP + tid] = v; as_global(st.qlag)[(size_t)env * NQP + tid] = v; }
  if (tid < NV) {
    const float v = tid < nv ? qvel[(size_t)env * nv + tid] : 0.f;
    as_global(st.qvel)[(size_t)env * NV + tid] = v; as_global(st.vlag)[(size_t)env * NV + tid] = v; as_global(st.warm)[(size_t)env * NV + tid] = 0.f;
  }
  if (tid == 0) as_global(st.lag_valid)[env] = 0;
}
__global__ __launch_bounds__(NT) void hoic_get_state_kernel(const DevModel* __restrict__ mp, DevState st, float* __restrict__ qpos,
                                                            float* __restrict__ qvel, int* __restrict__ cur_t) {
  const int env = blockIdx.x, tid = threadIdx.x, nq = mp->nq, nv = mp->nv;
  if (qpos && tid < nq) qpos[(size_t)env * nq + tid] = as_global(st.qpos)[(size_t)env * NQP + tid];
  if (qvel && tid < nv) qvel[(size_t)env * nv + tid] = as_global(st.qvel)[(size_t)env * NV + tid];
  if (cur_t && tid == 0) cur_t[env] = as_global(st.cur_t)[env];
}

// after hoic_set_expert: every env's (sequence, start, cur_t) must index the new table (the caller is expected to reset)
__global__ void hoic_clamp_episode_kernel(DevExpert ex, DevState st, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  int sq = st.seq[e]; sq = sq < 0 ? 0 : (sq >= ex.n_seq ? ex.n_seq - 1 : sq);
  const int len = ex.seq_len[sq];
  int sa = st.start[e]; sa = sa < 0 ? 0 : (sa > len - 2 ? len - 2 : sa);
  int ct = st.cur_t[e]; ct = ct < 0 ? 0 : (ct > len - 1 - sa ? len - 1 - sa : ct);
  st.seq[e] = sq; st.start[e] = sa; st.cur_t[e] = ct; st.lag_valid[e] = 0;
}

struct ProbeArgs {
  const float *qpos, *qvel, *ctrl, *applied, *warm;
  int do_step;
  float *xpos, *xquat, *gxpos, *gxmat, *qM, *bias, *contacts, *asmooth, *qacc, *qpos_out, *qvel_out, *cforce;
  int *ncon, *iters;
};
__global__ __launch_bounds__(NT) void hoic_probe_kernel(const DevModel* __restrict__ mp, const DevConfig* __restrict__ cp, ProbeArgs a) {
  __shared__ Work w;
  const DevModel& m = *mp; const DevConfig& cfg = *cp;
  const int env = blockIdx.x, tid = threadIdx.x;
  if (tid < NQP) w.qpos[tid] = tid < m.nq ? a.qpos[(size_t)env * m.nq + tid] : 0.f;
  if (tid < NV) {
    w.qvel[tid] = tid < m.nv ? a.qvel[(size_t)env * m.nv + tid] : 0.f;
    float f = (a.applied && tid < m.nv) ? a.applied[(size_t)env * m.nv + tid] : 0.f;
    const int ai = tid < m.nv ? m.dof_actid[tid] : -1;
    if (a.ctrl && ai >= 0) f += a.ctrl[(size_t)env * m.nu + ai];
    w.applied[tid] = f;                                                              // applied + actuator forces, as dev_applied leaves them
    w.qacc[tid] = (a.warm && tid < m.nv) ? a.warm[(size_t)env * m.nv + tid] : 0.f;      // warm start
  }
  if (tid == 0) { w.ncon = 0; w.nrow = 0; w.solver_iter = 0; w.cbod = 0u; w.capped = 0; }
  wsync();
  dev_load_constants(m, w);
  MReg M;
  dev_forward_kin(m, cfg, w, M, w.qpos, w.qvel, nullptr);
  // (the geom rotation matrices live in the collision stage's region: read them before the solver reuses it)
  if (a.gxmat) for (int k = tid; k < m.ngeom * 9; k += NT) a.gxmat[(size_t)env * m.ngeom * 9 + k] = w.gxmat[k / 9][k % 9];
  wsync();
  float a0 = 0.f;
  const bool ok = dev_forward_dyn(m, cfg, w, M, &a0, false);
  if (a.xpos) for (int k = tid; k < m.nbody * 3; k += NT) a.xpos[(size_t)env * m.nbody * 3 + k] = w.xpos[k / 3][k % 3];
  if (a.xquat) for (int k = tid; k < m.nbody * 4; k += NT) a.xquat[(size_t)env * m.nbody * 4 + k] = w.xquat[k / 4][k % 4];
  if (a.gxpos) for (int k = tid; k < m.ngeom * 3; k += NT) a.gxpos[(size_t)env * m.ngeom * 3 + k] = w.gxpos[k / 3][k % 3];
  if (a.qM && (tid & 31) < m.nv) {
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int r = MREG_ROW(reg, tid >> 5);
      if (r < m.nv) a.qM[((size_t)env * m.nv + r) * m.nv + (tid & 31)] = M.r[reg];
    }
  }
  if (a.bias && tid < m.nv) a.bias[(size_t)env * m.nv + tid] = w.bias[tid];
  if (a.asmooth && tid < m.nv) a.asmooth[(size_t)env * m.nv + tid] = a0;
  if (a.qacc && tid < m.nv) a.qacc[(size_t)env * m.nv + tid] = w.qacc[tid];
  if (a.ncon && tid == 0) a.ncon[env] = w.ncon;
  if (a.iters && tid == 0) a.iters[env] = ok ? w.solver_iter : -1;
  if (a.contacts) {
    for (int c = tid; c < HOIC_PROBE_MAXCON; c += NT) {
      float* r = a.contacts + ((size_t)env * HOIC_PROBE_MAXCON + c) * 16;
      if (c < w.ncon) {
        const int p = w.c_pair[c];
        r[0] = w.c_dist[c];
        for (int i = 0; i < 3; i++) r[1 + i] = w.c_pos[c][i];
        for (int i = 0; i < 9; i++) r[4 + i] = w.c_frame[c][i];
        r[13] = (float)m.pair_geom1[p]; r[14] = (float)m.pair_geom2[p]; r[15] = (float)m.pair_condim[p];
      } else for (int i = 0; i < 16; i++) r[i] = 0.f;
    }
  }
  if (a.cforce) {     // mj_contactForce: decode the pyramid's edge forces (the solver's row forces are intact: nothing ran since)
    for (int c = tid; c < HOIC_PROBE_MAXCON; c += NT) {
      float* r = a.cforce + ((size_t)env * HOIC_PROBE_MAXCON + c) * 6;
      float f[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (ok && c < w.ncon) {
        const int nr = w.c_nrow[c], r0 = w.c_row0[c];
        for (int e = 0; e < nr; e++) f[0] += w.cr_force[r0 + e];
        for (int k = 0; 2 * k + 1 < nr; k++) f[1 + k] = w.c_mu[c][k] * (w.cr_force[r0 + 2 * k] - w.cr_force[r0 + 2 * k + 1]);
      }
      for (int i = 0; i < 6; i++) r[i] = f[i];
    }
  }
  if (a.do_step && ok) dev_euler(m, w, M, (GPTR(float))nullptr, (GPTR(float))nullptr);
  if (a.qpos_out && tid < m.nq) a.qpos_out[(size_t)env * m.nq + tid] = w.qpos[tid];
  if (a.qvel_out && tid < m.nv) a.qvel_out[(size_t)env * m.nv + tid] = w.qvel[tid];
}

// ------------------------------------------------------------------------------------------------ host side
struct hoic_sim {
  bool reorder = false;
  bool use_lag = true;       // HOIC_NO_LAGREC=1: recompute the lagged forward pass at every launch (development aid)
  int n_envs = 0, device = 0;
  DevModel hm;            // host copy
  DevModel* d_model = nullptr;
  DevConfig hcfg;
  DevConfig* d_cfg = nullptr;
  DevExpert ex{};
  DevExpert* d_ex = nullptr;   // device copies of the pointer tables (substep kernel)
  DevState* d_st = nullptr;
  std::vector<void*> ex_allocs;
  DevState st{};
  int *d_iota_seq = nullptr, *d_iota_start = nullptr;
  bool timing = false;
  static const int NEV = 64;          // ring of timing events: steps are timed without a host sync per step
  hipEvent_t ev[NEV][3] = {};
  long long n_timed = 0, n_drained = 0;
  bool has_expert = false;
  // split post-step (hoic_set_async_reward): per env range a side stream for the reward part, the record buffer of the next
  // step and the events "reward part of the step that used buffer b has finished"
  struct AsyncRange { int first = 0, count = 0, next_buf = 0; hipStream_t side = nullptr; hipEvent_t sub_done = nullptr, ord_done = nullptr, rew_done[2] = {nullptr, nullptr};
                      bool pending[2] = {false, false}; bool order_ready = false;
                      hipStream_t sub = nullptr; int sub_reserve = 0; hipEvent_t in_ready = nullptr; };   // CU-masked substep stream (hoic_set_cu_reserve)
  int reserve_cus = 0;
  bool async_reward = false;
  bool postb_wide = false;   // HOIC_POSTB_WIDE=1: the split form's reward part on its 256-register build (hoic_poststep_kernel<POST_B, 2>)
  std::vector<AsyncRange> ranges;
  int expert_reserve = 0, expert_cap = 0;   // streaming: room for more frames behind the last sequence
  std::vector<int> h_seq_len, h_seq_off;
};

static int32_t drain_rewards(hoic_sim* s, hipStream_t stream);      // outstanding reward parts (asynchronous-reward mode) -> dependencies of `stream`

namespace {
struct Blob {
  const char* p; size_t n;
  const hoic_blob_entry* find(const char* name) const {
    const hoic_blob_header* h = (const hoic_blob_header*)p;
    const hoic_blob_entry* e = (const hoic_blob_entry*)(p + sizeof(hoic_blob_header));
    for (int i = 0; i < h->nentries; i++) if (strncmp(e[i].name, name, 32) == 0) return &e[i];
    return nullptr;
  }
  bool f64(const char* name, std::vector<double>& out) const {
    const hoic_blob_entry* e = find(name);
    if (!e || e->dtype != 0) { set_err(std::string("model blob: missing float64 array ") + name); return false; }
    out.resize(e->nbytes / 8);
    memcpy(out.data(), p + e->offset, e->nbytes);
    return true;
  }
  bool i32(const char* name, std::vector<int>& out) const {
    const hoic_blob_entry* e = find(name);
    if (!e || e->dtype != 1) { set_err(std::string("model blob: missing int32 array ") + name); return false; }
    out.resize(e->nbytes / 4);
    memcpy(out.data(), p + e->offset, e->nbytes);
    return true;
  }
};
template <size_t N> bool cpf(const Blob& b, const char* name, float (&dst)[N]) {
  std::vector<double> v;
  if (!b.f64(name, v) || v.size() > N) { if (v.size() > N) set_err(std::string("model blob: too large: ") + name); return false; }
  for (size_t i = 0; i < v.size(); i++) dst[i] = (float)v[i];
  return true;
}
template <size_t N, size_t K> bool cpf2(const Blob& b, const char* name, float (&dst)[N][K]) {
  std::vector<double> v;
  if (!b.f64(name, v) || v.size() > N * K) { if (v.size() > N * K) set_err(std::string("model blob: too large: ") + name); return false; }
  for (size_t i = 0; i < v.size(); i++) dst[i / K][i % K] = (float)v[i];
  return true;
}
template <size_t N> bool cpi(const Blob& b, const char* name, int (&dst)[N]) {
  std::vector<int> v;
  if (!b.i32(name, v) || v.size() > N) { if (v.size() > N) set_err(std::string("model blob: too large: ") + name); return false; }
  for (size_t i = 0; i < v.size(); i++) dst[i] = v[i];
  return true;
}
bool geti(const Blob& b, const char* name, int& dst) { std::vector<int> v; if (!b.i32(name, v) || v.empty()) return false; dst = v[0]; return true; }
bool getf(const Blob& b, const char* name, float& dst) { std::vector<double> v; if (!b.f64(name, v) || v.empty()) return false; dst = (float)v[0]; return true; }

void kb_from_solref(const double* solref_in, const double* solimp, double timestep, bool friction, float& K, float& B) {
  double ref[2] = {solref_in[0], solref_in[1]};
  double dmax = std::min(std::max(solimp[1], 0.0001), 0.9999);
  if ((ref[0] > 0) != (ref[1] > 0)) { ref[0] = 0.02; ref[1] = 1; }
  if (ref[0] > 0 && ref[0] < 2 * timestep) ref[0] = 2 * timestep;
  double k, bb;
  if (friction) k = 0;
  else if (ref[0] > 0) k = 1.0 / std::max(1e-15, dmax * dmax * ref[0] * ref[0] * ref[1] * ref[1]);
  else k = -ref[0] / std::max(1e-15, dmax * dmax);
  if (ref[1] > 0) bb = 2.0 / std::max(1e-15, dmax * ref[0]);
  else bb = -ref[1] / std::max(1e-15, dmax);
  K = (float)k; B = (float)bb;
}

bool build_model(const void* blob, size_t nbytes, DevModel& m) {
  if (nbytes < sizeof(hoic_blob_header) || memcmp(blob, HOIC_BLOB_MAGIC, 8) != 0) { set_err("model blob: bad magic"); return false; }
  Blob b{(const char*)blob, nbytes};
  memset(&m, 0, sizeof(m));
  if (!geti(b, "nbody", m.nbody) || !geti(b, "njnt", m.njnt) || !geti(b, "nq", m.nq) || !geti(b, "nv", m.nv) ||
      !geti(b, "nu", m.nu) || !geti(b, "ngeom", m.ngeom) || !geti(b, "npair", m.npair)) return false;
  if (m.nbody > NB || m.njnt > NJ || m.nq > HOIC_MAX_NQ || m.nv > NV || m.nu > NU || m.ngeom > NG || m.npair > NPAIR) {
    set_err("model blob: exceeds compiled capacities"); return false;
  }
  if (!geti(b, "hand_body0", m.hand_body0) || !geti(b, "obj_body", m.obj_body) || !geti(b, "hand_geom0", m.hand_geom0) ||
      !geti(b, "hand_geom1", m.hand_geom1) || !geti(b, "obj_geom0", m.obj_geom0) || !geti(b, "obj_geom1", m.obj_geom1) ||
      !geti(b, "hand_nq", m.hand_nq) || !geti(b, "hand_nv", m.hand_nv)) return false;
  if (m.hand_geom1 - m.hand_geom0 + 1 != NHG) { set_err("model blob: expected 19 hand collision geoms"); return false; }
  if (!getf(b, "timestep", m.timestep) || !getf(b, "meaninertia", m.meaninertia) || !getf(b, "hand_mass", m.hand_mass)) return false;
  if (!cpf(b, "gravity", m.gravity)) return false;
  if (!cpi(b, "body_parent", m.body_parent) || !cpi(b, "body_depth", m.body_depth) || !cpi(b, "body_jntadr", m.body_jntadr) ||
      !cpi(b, "body_jntnum", m.body_jntnum) || !cpi(b, "body_dofadr", m.body_dofadr) || !cpi(b, "body_dofnum", m.body_dofnum)) return false;
  if (!cpf2(b, "body_pos", m.body_pos) || !cpf2(b, "body_quat", m.body_quat) || !cpf2(b, "body_ipos", m.body_ipos) ||
      !cpf2(b, "body_iquat", m.body_iquat) || !cpf(b, "body_mass", m.body_mass) || !cpf2(b, "body_inertia", m.body_inertia)) return false;
  if (!cpi(b, "jnt_type", m.jnt_type) || !cpi(b, "jnt_qposadr", m.jnt_qposadr) || !cpi(b, "jnt_dofadr", m.jnt_dofadr) ||
      !cpi(b, "jnt_bodyid", m.jnt_bodyid) || !cpi(b, "jnt_limited", m.jnt_limited)) return false;
  if (!cpf2(b, "jnt_pos", m.jnt_pos) || !cpf2(b, "jnt_axis", m.jnt_axis) || !cpf2(b, "jnt_range", m.jnt_range) ||
      !cpf(b, "jnt_margin", m.jnt_margin) || !cpf2(b, "jnt_solimp", m.jnt_solimp) || !cpf(b, "qpos0", m.qpos0)) return false;
  if (!cpi(b, "dof_bodyid", m.dof_bodyid) || !cpi(b, "dof_jntid", m.dof_jntid)) return false;
  if (!cpf(b, "dof_armature", m.dof_armature) || !cpf(b, "dof_damping", m.dof_damping) || !cpf(b, "dof_frictionloss", m.dof_frictionloss)) return false;
  if (!cpi(b, "act_dofid", m.act_dofid)) return false;
  if (!cpi(b, "geom_type", m.geom_type) || !cpi(b, "geom_bodyid", m.geom_bodyid) || !cpi(b, "geom_meshid", m.geom_meshid)) return false;
  if (!cpf2(b, "geom_size", m.geom_size) || !cpf2(b, "geom_pos", m.geom_pos) || !cpf2(b, "geom_quat", m.geom_quat) ||
      !cpf(b, "geom_rbound", m.geom_rbound)) return false;
  if (!cpi(b, "pair_geom1", m.pair_geom1) || !cpi(b, "pair_geom2", m.pair_geom2) || !cpi(b, "pair_condim", m.pair_condim)) return false;
  if (!cpf2(b, "pair_solimp", m.pair_solimp) || !cpf(b, "pair_margin", m.pair_margin) || !cpf(b, "pair_gap", m.pair_gap)) return false;
  std::vector<double> dsolref, dsolimp, jsolref, jsolimp, psolref, psolimp, pfric, invw, binvw;
  std::vector<int> dparent, weld;
  if (!b.f64("dof_solref", dsolref) || !b.f64("dof_solimp", dsolimp) || !b.f64("jnt_solref", jsolref) || !b.f64("jnt_solimp", jsolimp) ||
      !b.f64("pair_solref", psolref) || !b.f64("pair_solimp", psolimp) || !b.f64("pair_friction", pfric) ||
      !b.f64("dof_invweight0", invw) || !b.f64("body_invweight0", binvw) || !b.i32("dof_parentid", dparent)) return false;
  std::vector<double> impr;
  double impratio = 1.0;
  if (b.f64("impratio", impr) && !impr.empty()) impratio = impr[0];
  m.jnt_poszero = 1;
  for (int j = 0; j < m.njnt; j++)
    if (m.jnt_type[j] != HOIC_JNT_FREE && (m.jnt_pos[j][0] != 0.f || m.jnt_pos[j][1] != 0.f || m.jnt_pos[j][2] != 0.f)) m.jnt_poszero = 0;
  // tree bookkeeping
  m.nlevel = 0;
  for (int i = 0; i < m.nbody; i++) m.nlevel = std::max(m.nlevel, m.body_depth[i]);
  for (int i = 0; i < m.nbody; i++) {
    int e = i + 1;
    while (e < m.nbody && m.body_depth[e] > m.body_depth[i]) e++;
    m.body_subtree[i] = e - i;
  }
  {   // plan of the composite sums (DevModel::body_sum)
    auto plan = [&](bool two_round) {
      bool ok = true;
      for (int i = 0; i < m.nbody; i++) {
        m.body_kids[i][0] = m.body_kids[i][1] = 0xFFFFFFFFu;
        bool needed = false;            // some dof reads this body's composite: the body carries dofs itself
        needed = m.body_dofnum[i] > 0;
        m.body_sum[i] = needed ? 1 : 0;
        if (two_round && needed && m.body_subtree[i] > SUM_DIRECT) {
          int nk = 0;
          for (int c = i + 1; c < i + m.body_subtree[i]; c += m.body_subtree[c]) {
            if (nk >= 8 || m.body_subtree[c] > SUM_DIRECT) { ok = false; break; }
            m.body_kids[i][nk >> 2] = (m.body_kids[i][nk >> 2] & ~(0xFFu << (8 * (nk & 3)))) | ((unsigned)c << (8 * (nk & 3)));
            nk++;
          }
          m.body_sum[i] = 2;
        }
      }
      if (ok)      // children of a mode-2 body are summed in the first round whether or not they carry dofs
        for (int i = 0; i < m.nbody; i++)
          if (m.body_sum[i] == 2)
            for (int k = 0; k < 8; k++) { const unsigned c = (m.body_kids[i][k >> 2] >> (8 * (k & 3))) & 0xFFu; if (c != 0xFFu) m.body_sum[c] = 1; }
      return ok;
    };
    if (!plan(true)) plan(false);       // a tree this plan does not fit (a big subtree below a big subtree): plain range sums
  }
  std::vector<int> lastdof;
  if (!b.i32("body_lastdof", lastdof)) return false;
  for (int i = 0; i < m.nbody; i++) {
    unsigned mask = 0;
    for (int d = lastdof[i]; d >= 0; d = dparent[d]) mask |= 1u << d;
    m.body_dofmask[i] = mask;
  }
  // pointer-jumping schedule of the kinematics: round r composes a body with its ancestor 2^r levels up
  {
    int anc[NB];
    for (int i = 0; i < m.nbody; i++) anc[i] = m.body_parent[i];
    m.nround = 0;
    for (int r = 0; r < MAXROUND; r++) {
      bool any = false;
      int nxt[NB];
      for (int i = 0; i < m.nbody; i++) {
        m.body_jump[r][i] = (i > 0 && anc[i] != 0) ? anc[i] : -1;
        nxt[i] = (i > 0 && anc[i] != 0) ? anc[anc[i]] : 0;
        any = any || m.body_jump[r][i] >= 0;
      }
      for (int i = 0; i < m.nbody; i++) anc[i] = nxt[i];
      if (any) m.nround = r + 1;
    }
    for (int i = 0; i < m.nbody; i++) if (anc[i] != 0) { set_err("model blob: kinematic tree deeper than 8 levels"); return false; }
  }
  for (int i = 0; i < m.nbody; i++) {
    int path[32], np = 0;
    for (int d = lastdof[i]; d >= 0; d = dparent[d]) path[np++] = d;
    if (np > 12) { set_err("model blob: more than 12 dofs on a body's path"); return false; }
    m.max_path = std::max(m.max_path, np);
    for (int k = 0; k < 3; k++) m.body_path[i][k] = 0xFFFFFFFFu;
    for (int k = 0; k < np; k++) {   // root first
      const int d = path[np - 1 - k];
      m.body_path[i][k >> 2] = (m.body_path[i][k >> 2] & ~(0xFFu << (8 * (k & 3)))) | ((unsigned)d << (8 * (k & 3)));
    }
  }
  for (int d = 0; d < m.nv; d++) {
    unsigned am = 0;
    for (int e = dparent[d]; e >= 0; e = dparent[e]) am |= 1u << e;
    m.dof_amask[d] = am;
    m.dof_actid[d] = -1;
    for (int u = 0; u < m.nu; u++) if (m.act_dofid[u] == d) m.dof_actid[d] = u;
  }
  for (int d = 0; d < m.nv; d++) {
    const int j = m.dof_jntid[d], b = m.dof_bodyid[d];
    m.dof_jtype[d] = m.jnt_type[j]; m.dof_k[d] = d - m.jnt_dofadr[j]; m.dof_parentbody[d] = m.body_parent[b];
    m.dof_qadr[d] = m.jnt_qposadr[j];
    m.dof_limited[d] = (m.jnt_limited[j] && m.jnt_type[j] != HOIC_JNT_FREE) ? 1 : 0;
    for (int k = 0; k < 3; k++) m.dof_bpath[d][k] = m.body_path[b][k];
    m.dof_range[d][0] = m.jnt_range[j][0]; m.dof_range[d][1] = m.jnt_range[j][1]; m.dof_margin[d] = m.jnt_margin[j];
    for (int k = 0; k < 5; k++) m.dof_solimp[d][k] = m.jnt_solimp[j][k];
  }
  for (int d = 0; d < m.nv; d++) {
    unsigned dm = 0;
    for (int e = 0; e < m.nv; e++) if (e != d && ((m.dof_amask[e] >> d) & 1u)) dm |= 1u << e;
    m.dof_dmask[d] = dm;
  }
  // The solver's elimination order (hoic_solver.h hs_factor) is built for the HOIC hand: palm dofs 0..5, five fingers of four
  // dofs each (6 + 4 f .. 9 + 4 f) that hang on the palm and on nothing else, a free object on dofs 26..31.
  {
    bool ok = m.nv == 32 && m.hand_nv == 26;
    for (int d = 0; ok && d < 32; d++) {
      const unsigned rel = m.dof_amask[d] | m.dof_dmask[d];          // every dof this one shares a chain with
      unsigned allowed;
      if (d < 6) allowed = 0x03FFFFFFu;                               // palm: itself and all fingers
      else if (d < 26) allowed = 0x3Fu | (0xFu << (6 + 4 * ((d - 6) / 4)));      // a finger: the palm and its own four
      else allowed = 0xFC000000u;                                     // the object: only itself
      if (rel & ~allowed) ok = false;
      if (d >= 26 && m.dof_jtype[d] != HOIC_JNT_FREE) ok = false;
    }
    if (!ok) { set_err("model blob: the joint structure is not the hand (palm 6 + 5 x 4 finger dofs) + free object the solver's elimination order is built for"); return false; }
  }
  // constraint constants
  for (int i = 0; i < m.nv; i++) {
    float K;
    kb_from_solref(&dsolref[2 * i], &dsolimp[5 * i], m.timestep, true, K, m.dof_flB[i]);
    double s0 = std::min(std::max(dsolimp[5 * i], 0.0001), 0.9999), s1 = std::min(std::max(dsolimp[5 * i + 1], 0.0001), 0.9999);
    double imp = (s0 == s1 || dsolimp[5 * i + 2] <= 1e-15) ? 0.5 * (s0 + s1) : s0;   // pos = margin = 0 -> x = 0
    m.dof_flR[i] = (float)std::max(1e-15, (1 - imp) * invw[i] / imp);
  }
  for (int j = 0; j < m.njnt; j++) {
    kb_from_solref(&jsolref[2 * j], &jsolimp[5 * j], m.timestep, false, m.jnt_K[j], m.jnt_B[j]);
    m.jnt_diag[j] = (float)invw[m.jnt_dofadr[j]];
    if (m.jnt_type[j] != HOIC_JNT_FREE) { const int d = m.jnt_dofadr[j]; m.dof_limK[d] = m.jnt_K[j]; m.dof_limB[d] = m.jnt_B[j]; m.dof_limdiag[d] = m.jnt_diag[j]; }
    if (m.jnt_limited[j] && m.jnt_type[j] != HOIC_JNT_FREE && m.jnt_range[j][1] - m.jnt_range[j][0] <= 2 * m.jnt_margin[j]) {
      set_err("model blob: joint range narrower than twice its margin is not supported"); return false;
    }
  }
  for (int p = 0; p < m.npair; p++) {
    kb_from_solref(&psolref[2 * p], &psolimp[5 * p], m.timestep, false, m.pair_K[p], m.pair_B[p]);
    const double* f = &pfric[5 * p];
    m.pair_mu[p][0] = (float)f[0]; m.pair_mu[p][1] = (float)f[1]; m.pair_mu[p][2] = (float)f[2];
    const int b1 = m.geom_bodyid[m.pair_geom1[p]], b2 = m.geom_bodyid[m.pair_geom2[p]];
    const double tran = binvw[2 * b1] + binvw[2 * b2];
    if (m.pair_condim[p] == 1) m.pair_Rscale[p] = (float)tran;
    else {
      const double mu = f[0] * std::sqrt(1.0 / std::max(1e-15, impratio));
      m.pair_Rscale[p] = (float)(2 * mu * mu * (tran + f[0] * f[0] * tran));
    }
    m.pair_b1[p] = b1; m.pair_b2[p] = b2;
    const int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p];
    m.pair_type1[p] = m.geom_type[g1]; m.pair_type2[p] = m.geom_type[g2]; m.pair_mesh[p] = m.geom_meshid[g2];
    m.pair_bound[p] = m.geom_rbound[g1] + m.geom_rbound[g2] + m.pair_margin[p];
    for (int k = 0; k < 3; k++) { m.pair_size1[p][k] = m.geom_size[g1][k]; m.pair_size2[p][k] = m.geom_size[g2][k]; }
    m.pair_mpos[p] = m.body_dofmask[b2] & ~m.body_dofmask[b1]; m.pair_mneg[p] = m.body_dofmask[b1] & ~m.body_dofmask[b2];
    if (m.pair_type1[p] == HOIC_GEOM_MESH) { set_err("model blob: a mesh must be the second geom of a pair"); return false; }
  }
  {   // pool entries of the collision staging (hoic_collide.h LaneContacts): the pair types whose narrow phase can produce more
      // than COLSLOT contacts -- plane-box (4), box-box (4), box-mesh (4), plane-mesh (3) -- numbered within their pass of 64 pairs
    int used[(NPAIR + NT - 1) / NT] = {};
    for (int p = 0; p < m.npair; p++) {
      const int t1 = m.pair_type1[p], t2 = m.pair_type2[p];
      const bool big = ((t1 == HOIC_GEOM_PLANE || t1 == HOIC_GEOM_BOX) && (t2 == HOIC_GEOM_BOX || t2 == HOIC_GEOM_MESH));
      m.pair_pool[p] = big ? used[p / NT]++ : -1;
      if (used[p / NT] > COLPOOL) { set_err("model blob: more than 32 pairs of one pass can produce four contacts (collision staging pool)"); return false; }
    }
  }
  std::vector<int> mva, mvn, mpa, mpn; std::vector<double> mv, mpl;
  if (b.i32("mesh_vertadr", mva) && b.i32("mesh_vertnum", mvn) && b.f64("mesh_vert", mv) &&
      b.i32("mesh_planeadr", mpa) && b.i32("mesh_planenum", mpn) && b.f64("mesh_plane", mpl)) {
    if (mva.size() > HOIC_MAX_MESH || mv.size() > (size_t)MAXMESHV * 3 || mpl.size() > (size_t)MAXMESHP * 4) {
      set_err("model blob: mesh tables exceed compiled capacities"); return false;
    }
    for (size_t i = 0; i < mva.size(); i++) { m.mesh_vertadr[i] = mva[i]; m.mesh_vertnum[i] = mvn[i]; m.mesh_planeadr[i] = mpa[i]; m.mesh_planenum[i] = mpn[i]; }
    for (size_t i = 0; i < mv.size(); i++) m.mesh_vert[i / 3][i % 3] = (float)mv[i];     // [.][3] stays 0
    for (size_t i = 0; i < mpl.size(); i++) m.mesh_plane[i / 4][i % 4] = (float)mpl[i];
    // ---- run bounds (float32 table values, double arithmetic, rounded outward): see hoic_collide.h
    int vr = 0, fr = 0;
    for (size_t i = 0; i < mva.size(); i++) {
      const int va = mva[i], vn = mvn[i], pa = mpa[i], pn = mpn[i];
      m.mesh_vrunadr[i] = vr; m.mesh_vrunnum[i] = (vn + HOIC_HULL_RUN_VERTS - 1) / HOIC_HULL_RUN_VERTS;
      m.mesh_frunadr[i] = fr; m.mesh_frunnum[i] = (pn + HOIC_HULL_RUN_FACES - 1) / HOIC_HULL_RUN_FACES;
      if (m.mesh_vrunnum[i] > NT || m.mesh_frunnum[i] > 2 * NT) { set_err("model blob: a mesh has more runs than the narrow phase handles"); return false; }
      for (int k = 0; k < 3; k++) { m.mesh_aabb[i][k] = 1e30f; m.mesh_aabb[i][4 + k] = -1e30f; }
      for (int v = va; v < va + vn; v++)
        for (int k = 0; k < 3; k++) { m.mesh_aabb[i][k] = std::min(m.mesh_aabb[i][k], m.mesh_vert[v][k]); m.mesh_aabb[i][4 + k] = std::max(m.mesh_aabb[i][4 + k], m.mesh_vert[v][k]); }
      for (int r = 0; r < m.mesh_vrunnum[i]; r++, vr++) {
        const int v0 = va + r * HOIC_HULL_RUN_VERTS, v1 = std::min(va + vn, v0 + HOIC_HULL_RUN_VERTS);
        double c[3] = {0, 0, 0}, rad = 0;
        for (int v = v0; v < v1; v++) for (int k = 0; k < 3; k++) c[k] += m.mesh_vert[v][k] / (v1 - v0);
        float cf[3] = {(float)c[0], (float)c[1], (float)c[2]};
        for (int v = v0; v < v1; v++) { double d2 = 0; for (int k = 0; k < 3; k++) { const double d = (double)m.mesh_vert[v][k] - cf[k]; d2 += d * d; } rad = std::max(rad, std::sqrt(d2)); }
        for (int k = 0; k < 3; k++) m.mesh_vrun[vr][k] = cf[k];
        m.mesh_vrun[vr][3] = std::nextafterf((float)(rad * (1 + 1e-6) + 1e-9), 1e30f);
      }
      for (int r = 0; r < m.mesh_frunnum[i]; r++, fr++) {
        const int f0 = pa + r * HOIC_HULL_RUN_FACES, f1 = std::min(pa + pn, f0 + HOIC_HULL_RUN_FACES);
        // reference point of the run: the mean of the feet of the origin's perpendiculars onto the faces (near the patch,
        // inside or on the hull), so that n.c - d is a small non-positive number for the run's own faces
        double c[3] = {0, 0, 0};
        float lo[3] = {2.f, 2.f, 2.f}, hi[3] = {-2.f, -2.f, -2.f};
        for (int f = f0; f < f1; f++)
          for (int k = 0; k < 3; k++) {
            c[k] += (double)m.mesh_plane[f][k] * m.mesh_plane[f][3] / (f1 - f0);
            lo[k] = std::min(lo[k], m.mesh_plane[f][k]); hi[k] = std::max(hi[k], m.mesh_plane[f][k]);
          }
        float cf[3] = {(float)c[0], (float)c[1], (float)c[2]};
        double emax = -1e300;
        for (int f = f0; f < f1; f++) emax = std::max(emax, (double)m.mesh_plane[f][0] * cf[0] + (double)m.mesh_plane[f][1] * cf[1] + (double)m.mesh_plane[f][2] * cf[2] - m.mesh_plane[f][3]);
        float* o = m.mesh_frun[fr];
        for (int k = 0; k < 3; k++) { o[k] = cf[k]; o[4 + k] = lo[k]; o[8 + k] = hi[k]; }
        o[3] = std::nextafterf((float)emax, 1e30f); o[7] = 0.f; o[11] = 0.f;
      }
    }
    m.mesh_prune = getenv("HOIC_MESH_STREAM") == nullptr ? 1 : 0;
    m.obb_reject = getenv("HOIC_NO_OBB_REJECT") == nullptr ? 1 : 0;
    m.warm_shift = getenv("HOIC_PLAIN_WARMSTART") == nullptr ? 1 : 0;
  } else { set_err("model blob: mesh tables missing"); return false; }
  return true;
}
}  // namespace

extern "C" hoic_sim* hoic_create(const void* model_blob, size_t nbytes, int32_t n_envs, int32_t device_id) {
  if (!model_blob || n_envs <= 0) { set_err("hoic_create: bad arguments"); return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err("hoic_create: no HIP device (this library has no CPU path)"); return nullptr; }
  if (hipSetDevice(device_id) != hipSuccess) { set_err("hoic_create: hipSetDevice failed"); return nullptr; }
  hoic_sim* s = new hoic_sim();
  s->n_envs = n_envs; s->device = device_id;
  if (!build_model(model_blob, nbytes, s->hm)) { delete s; return nullptr; }
  memset(&s->hcfg, 0, sizeof(s->hcfg));
  hoic_env_config& c = s->hcfg.c;
  c.pos_diff_thresh = 0.1f; c.rot_diff_thresh = 1.0f; c.jpos_diff_thresh = 0.1f; c.obj_pos_diff_thresh = 0.1f; c.obj_rot_diff_thresh = 1.0f;
  c.residual_force_scale = 2.5f; c.residual_torque_scale = 0.125f; c.sim_step = 15; c.future_w_size = 5;
  c.residual_force = 1; c.explain_force = 1; c.surface_contact = 1; c.pd_rel = 1; c.solver_iterations = 20;   // the hand MJCF's <option iterations="20">
  for (int i = 0; i < NU; i++) { c.jkp[i] = i < 3 ? 50.f : (i < 6 ? 5.f : 1.f); c.jkd[i] = 0.1f * c.jkp[i]; c.torque_lim[i] = c.jkp[i]; }
  for (int j = 0; j < s->hm.hand_nq && j < NU; j++) {   // ho_im4.py:103-107
    const float lo = s->hm.jnt_range[j][0], hi = s->hm.jnt_range[j][1];
    s->hcfg.base_pose[j] = 0.5f * (hi + lo);
    s->hcfg.ctrl_scale[j] = (hi - s->hcfg.base_pose[j]) * (j >= 6 ? 1.2f : 1.f);
  }
  const float wk0[16] = {0.25f, 0.2f, 0.1f, 0.45f, 0.2f, 0.4f, 0.1f, 0.5f, 3.f, 3.f, 0.05f, 6.f, 10.f, 1.f, 0.05f, 1.f};
  memcpy(s->hcfg.rp.wk, wk0, sizeof(wk0));
  s->hcfg.mode_train = 1;
  bool ok = hipMalloc(&s->d_model, sizeof(DevModel)) == hipSuccess && hipMalloc(&s->d_cfg, sizeof(DevConfig)) == hipSuccess;
  const size_t n = (size_t)n_envs;
  ok = ok && hipMalloc(&s->st.qpos, n * NQP * 4) == hipSuccess && hipMalloc(&s->st.qlag, n * NQP * 4) == hipSuccess &&
       hipMalloc(&s->st.qvel, n * NV * 4) == hipSuccess && hipMalloc(&s->st.vlag, n * NV * 4) == hipSuccess &&
       hipMalloc(&s->st.warm, n * NV * 4) == hipSuccess && hipMalloc(&s->st.cur_t, n * 4) == hipSuccess &&
       hipMalloc(&s->st.start, n * 4) == hipSuccess && hipMalloc(&s->st.seq, n * 4) == hipSuccess &&
       hipMalloc(&s->st.rfc_score, n * 4) == hipSuccess && hipMalloc(&s->st.diag, n * 8) == hipSuccess &&
       hipMalloc(&s->st.phase, n * 24 * 8) == hipSuccess && hipMalloc(&s->st.post, 2 * n * PB_SIZE * 4) == hipSuccess &&
       hipMalloc(&s->st.oldg, n * OG_SIZE * 4) == hipSuccess && hipMalloc(&s->st.qp_lam, n * 8 * 8) == hipSuccess && hipMalloc(&s->st.qpcol, n * (size_t)QP_COL_FLOATS * 4) == hipSuccess &&
       hipMalloc(&s->st.cost, 2 * n * 4) == hipSuccess && hipMalloc(&s->st.order, 2 * n * 4) == hipSuccess &&
       hipMalloc(&s->st.lagrec, n * LG_SIZE * 4) == hipSuccess && hipMalloc(&s->st.lag_valid, n * 4) == hipSuccess;
  if (!ok) { set_err("hoic_create: hipMalloc failed"); hoic_destroy(s); return nullptr; }
  ok = hipMalloc(&s->d_ex, sizeof(DevExpert)) == hipSuccess && hipMalloc(&s->d_st, sizeof(DevState)) == hipSuccess;
  if (!ok) { set_err("hoic_create: hipMalloc failed"); hoic_destroy(s); return nullptr; }
  hipMemcpy(s->d_st, &s->st, sizeof(DevState), hipMemcpyHostToDevice);
  hipMemcpy(s->d_model, &s->hm, sizeof(DevModel), hipMemcpyHostToDevice);
  hipMemcpy(s->d_cfg, &s->hcfg, sizeof(DevConfig), hipMemcpyHostToDevice);
  hipMemset(s->st.qpos, 0, n * NQP * 4); hipMemset(s->st.qlag, 0, n * NQP * 4); hipMemset(s->st.qvel, 0, n * NV * 4);
  hipMemset(s->st.vlag, 0, n * NV * 4); hipMemset(s->st.warm, 0, n * NV * 4); hipMemset(s->st.cur_t, 0, n * 4);
  hipMemset(s->st.start, 0, n * 4); hipMemset(s->st.seq, 0, n * 4); hipMemset(s->st.rfc_score, 0, n * 4);
  hipMemset(s->st.diag, 0, n * 8); hipMemset(s->st.phase, 0, n * 24 * 8);
  hipMemset(s->st.post, 0, 2 * n * PB_SIZE * 4); hipMemset(s->st.oldg, 0, n * OG_SIZE * 4); hipMemset(s->st.qp_lam, 0, n * 8 * 8);
  hipMemset(s->st.cost, 0, 2 * n * 4);
  hipMemset(s->st.lagrec, 0, n * LG_SIZE * 4); hipMemset(s->st.lag_valid, 0, n * 4);
  hipLaunchKernelGGL(hoic_order_kernel, dim3(2), dim3(ORDER_NT), 0, 0, s->st.cost, s->st.order, n_envs, 0, n_envs);   // a valid permutation from the start
  // Longest-first dispatch by each env's previous duration, per launch range (HOIC_REORDER=0 switches it off).  The
  // step-to-step correlation of an env's duration is only 0.4-0.5 (tools/duration_predictors.py; the contact count before the
  // step predicts even less: 0.2-0.25), but in the two-range rollout the workgroups of a launch start one by one as the
  // other range's wavefronts retire, and starting the likely-long ones first shortens the launch: 2.10 -> 1.95 ms per
  // 2048-env launch, rollout +3 % (+4 % on a tracking policy).  Whole-batch launches gain nothing measurable.
  s->reorder = !(getenv("HOIC_REORDER") != nullptr && getenv("HOIC_REORDER")[0] == '0');
  s->use_lag = getenv("HOIC_NO_LAGREC") == nullptr;
  s->postb_wide = getenv("HOIC_POSTB_WIDE") != nullptr;
  hipDeviceSynchronize();
  return s;
}

extern "C" void hoic_destroy(hoic_sim* s) {
  if (!s) return;
  hipSetDevice(s->device);
  for (void* p : s->ex_allocs) hipFree(p);
  void* ptrs[] = {s->d_model, s->d_cfg, s->st.qpos, s->st.qlag, s->st.qvel, s->st.vlag, s->st.warm, s->st.cur_t,
                  s->st.start, s->st.seq, s->st.rfc_score, s->st.diag, s->st.phase, s->st.post, s->st.oldg, s->st.qp_lam, s->st.qpcol, s->st.cost, s->st.order, s->st.lagrec, s->st.lag_valid, s->d_iota_seq, s->d_iota_start, s->d_ex, s->d_st};
  for (void* p : ptrs) if (p) hipFree(p);
  for (int i = 0; i < hoic_sim::NEV; i++) for (int k = 0; k < 3; k++) if (s->ev[i][k]) hipEventDestroy(s->ev[i][k]);
  for (auto& r : s->ranges) {
    if (r.side) { hipStreamSynchronize(r.side); hipStreamDestroy(r.side); }
    if (r.sub) { hipStreamSynchronize(r.sub); hipStreamDestroy(r.sub); }
    if (r.in_ready) hipEventDestroy(r.in_ready);
    if (r.sub_done) hipEventDestroy(r.sub_done);
    if (r.ord_done) hipEventDestroy(r.ord_done);
    for (int b = 0; b < 2; b++) if (r.rew_done[b]) hipEventDestroy(r.rew_done[b]);
  }
  delete s;
}

extern "C" int32_t hoic_num_envs(const hoic_sim* s) { return s ? s->n_envs : 0; }
extern "C" int32_t hoic_obs_dim(const hoic_sim* s) { (void)s; return HOIC_OBS_DIM; }
extern "C" int32_t hoic_action_dim(const hoic_sim* s) { (void)s; return HOIC_ACT_DIM; }

// The configuration block is rewritten with a synchronous copy while kernels of ANY stream of this handle may still read it
// (the reward parts of the split post-step run on non-blocking side streams and read cfg.rp): every setter first waits for
// the whole device, so a step launched before the call sees the old block and a step launched after it the new one.
static int32_t write_config(hoic_sim* s) {
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipDeviceSynchronize());
  for (auto& r : s->ranges) r.pending[0] = r.pending[1] = false;      // every reward part has finished
  HIPCHK(hipMemcpy(s->d_cfg, &s->hcfg, sizeof(DevConfig), hipMemcpyHostToDevice));
  return HOIC_OK;
}
extern "C" int32_t hoic_set_config(hoic_sim* s, const hoic_env_config* cfg) {
  if (!s || !cfg) { set_err("hoic_set_config: null"); return HOIC_ERR_ARG; }
  if (cfg->future_w_size != 5) { set_err("hoic_set_config: future_w_size must be 5 (obs layout)"); return HOIC_ERR_ARG; }
  if (cfg->sim_step <= 0 || cfg->solver_iterations <= 0) { set_err("hoic_set_config: sim_step/solver_iterations"); return HOIC_ERR_ARG; }
  s->hcfg.c = *cfg;
  return write_config(s);
}
extern "C" int32_t hoic_set_reward_params(hoic_sim* s, const hoic_reward_params* rp) {
  if (!s || !rp) { set_err("hoic_set_reward_params: null"); return HOIC_ERR_ARG; }
  s->hcfg.rp = *rp;
  return write_config(s);
}
// the stream-ordered form: the new parameters take effect for every launch enqueued on (or ordered behind) `stream` after
// this call and nothing waits on the host -- the caller's training loop runs one phase ahead of the GPU
__global__ void hoic_set_reward_params_kernel(DevConfig* cfg, hoic_reward_params rp) {
  if (threadIdx.x == 0) cfg->rp = rp;
}
extern "C" int32_t hoic_set_reward_params_async(hoic_sim* s, const hoic_reward_params* rp, void* stream) {
  if (!s || !rp) { set_err("hoic_set_reward_params_async: null"); return HOIC_ERR_ARG; }
  HIPCHK(hipSetDevice(s->device));
  // reward parts still running on the ranges' side streams read the old parameters: order them before the write
  const int32_t rc = drain_rewards(s, (hipStream_t)stream);
  if (rc != HOIC_OK) return rc;
  s->hcfg.rp = *rp;
  hipLaunchKernelGGL(hoic_set_reward_params_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, s->d_cfg, *rp);
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}
extern "C" int32_t hoic_set_mode(hoic_sim* s, int32_t train) {
  if (!s) return HOIC_ERR_ARG;
  s->hcfg.mode_train = train ? 1 : 0;
  return write_config(s);
}

template <typename T> static T* upload(hoic_sim* s, const T* h, size_t n, size_t extra = 0) {
  T* d = nullptr;
  if (hipMalloc(&d, (n + extra) * sizeof(T)) != hipSuccess) return nullptr;
  hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice);
  s->ex_allocs.push_back(d);
  return d;
}
extern "C" int32_t hoic_set_expert(hoic_sim* s, int32_t n_seq, const int32_t* seq_len, const float* hand_dof,
                                   const float* hand_dof_vel, const float* obj_pose, const float* obj_vel,
                                   const float* obj_angvel, const float* body_pos, const float* body_quat) {
  if (!s || n_seq <= 0 || !seq_len || !hand_dof || !hand_dof_vel || !obj_pose || !obj_vel || !obj_angvel || !body_pos || !body_quat) {
    set_err("hoic_set_expert: bad arguments"); return HOIC_ERR_ARG;
  }
  // validate everything before the previous tables are released
  std::vector<int> off(n_seq);
  size_t T = 0;
  for (int i = 0; i < n_seq; i++) { if (seq_len[i] < 2) { set_err("hoic_set_expert: sequence shorter than 2 frames"); return HOIC_ERR_ARG; } off[i] = (int)T; T += seq_len[i]; }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipDeviceSynchronize());
  for (void* p : s->ex_allocs) hipFree(p);
  s->ex_allocs.clear();
  s->has_expert = false;               // until the new tables are complete, step / reset refuse to run
  const int nh = s->hm.hand_nq;
  DevExpert& x = s->ex;
  x = DevExpert{};
  x.n_seq = n_seq; x.total = (int)T;
  x.seq_off = upload(s, off.data(), n_seq); x.seq_len = upload(s, seq_len, n_seq);
  const size_t R = (size_t)s->expert_reserve;
  x.hand_dof = upload(s, hand_dof, T * nh, R * nh); x.hand_dof_vel = upload(s, hand_dof_vel, T * nh, R * nh);
  x.obj_pose = upload(s, obj_pose, T * 7, R * 7); x.obj_vel = upload(s, obj_vel, T * 3, R * 3); x.obj_angvel = upload(s, obj_angvel, T * 3, R * 3);
  x.body_pos = upload(s, body_pos, T * NHB * 3, R * NHB * 3); x.body_quat = upload(s, body_quat, T * NHB * 4, R * NHB * 4);
  s->expert_cap = (int)(T + R);
  s->h_seq_off = off; s->h_seq_len.assign(seq_len, seq_len + n_seq);
  if (!x.seq_off || !x.seq_len || !x.hand_dof || !x.hand_dof_vel || !x.obj_pose || !x.obj_vel || !x.obj_angvel || !x.body_pos || !x.body_quat) {
    for (void* p : s->ex_allocs) hipFree(p);
    s->ex_allocs.clear(); x = DevExpert{};
    set_err("hoic_set_expert: hipMalloc failed"); return HOIC_ERR_DEVICE;
  }
  HIPCHK(hipMemcpy(s->d_ex, &s->ex, sizeof(DevExpert), hipMemcpyHostToDevice));
  // envs still point at (sequence, start) of the previous table: bring them inside the new one until the caller resets
  hipLaunchKernelGGL(hoic_clamp_episode_kernel, dim3((s->n_envs + 255) / 256), dim3(256), 0, 0, s->ex, s->st, s->n_envs);
  HIPCHK(hipDeviceSynchronize());
  s->has_expert = true;
  return HOIC_OK;
}

extern "C" int32_t hoic_set_expert_reserve(hoic_sim* s, int32_t frames) {
  if (!s || frames < 0) { set_err("hoic_set_expert_reserve: bad arguments"); return HOIC_ERR_ARG; }
  s->expert_reserve = frames;
  return HOIC_OK;
}
extern "C" int32_t hoic_append_expert_frame(hoic_sim* s, const float* hand_dof, const float* hand_dof_vel, const float* obj_pose,
                                            const float* obj_vel, const float* obj_angvel, const float* body_pos,
                                            const float* body_quat, void* stream) {
  if (!s || !hand_dof || !hand_dof_vel || !obj_pose || !obj_vel || !obj_angvel || !body_pos || !body_quat) { set_err("hoic_append_expert_frame: null"); return HOIC_ERR_ARG; }
  if (!s->has_expert) { set_err("hoic_append_expert_frame: set_expert has not been called"); return HOIC_ERR_STATE; }
  DevExpert& x = s->ex;
  if (x.total >= s->expert_cap) { set_err("hoic_append_expert_frame: reserve used up (hoic_set_expert_reserve)"); return HOIC_ERR_STATE; }
  hipStream_t st = (hipStream_t)stream;
  const int nh = s->hm.hand_nq;
  const size_t t = (size_t)x.total;
  const int last = x.n_seq - 1;
  HIPCHK(hipMemcpyAsync((float*)x.hand_dof + t * nh, hand_dof, nh * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync((float*)x.hand_dof_vel + t * nh, hand_dof_vel, nh * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync((float*)x.obj_pose + t * 7, obj_pose, 7 * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync((float*)x.obj_vel + t * 3, obj_vel, 3 * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync((float*)x.obj_angvel + t * 3, obj_angvel, 3 * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync((float*)x.body_pos + t * NHB * 3, body_pos, NHB * 3 * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync((float*)x.body_quat + t * NHB * 4, body_quat, NHB * 4 * 4, hipMemcpyHostToDevice, st));
  s->h_seq_len[last] += 1; x.total += 1;
  HIPCHK(hipMemcpyAsync((int*)x.seq_len + last, &s->h_seq_len[last], 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(s->d_ex, &s->ex, sizeof(DevExpert), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));      // the host staging values above must outlive the copies
  return HOIC_OK;
}

extern "C" int32_t hoic_reset(hoic_sim* s, const int32_t* d_env_ids, int32_t n, const int32_t* d_seq, const int32_t* d_start,
                              float* d_obs_out, void* stream) {
  if (!s || !d_seq || !d_start || n <= 0 || n > s->n_envs) { set_err("hoic_reset: bad arguments"); return HOIC_ERR_ARG; }
  if (!s->has_expert) { set_err("hoic_reset: set_expert has not been called"); return HOIC_ERR_STATE; }
  HIPCHK(hipSetDevice(s->device));
  { const int32_t rc = drain_rewards(s, (hipStream_t)stream); if (rc != HOIC_OK) return rc; }      // the reset writes rfc_score / the QP's flag
  hipLaunchKernelGGL(hoic_reset_kernel, dim3(n), dim3(NT), 0, (hipStream_t)stream, s->d_model, s->ex, s->st, d_env_ids, d_seq, d_start, d_obs_out, s->n_envs);
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}

static int32_t step_range(hoic_sim* s, int first, int count, const float* d_action, float* d_obs, float* d_reward, float* d_reward_info,
                          int32_t* d_flags, float* d_percent, const int32_t* d_next_seq, const int32_t* d_next_start, void* stream,
                          const char* who) {
  if (!s || !d_action || !d_obs || !d_reward || !d_reward_info || !d_flags || !d_percent) { set_err(std::string(who) + ": null pointer"); return HOIC_ERR_ARG; }
  if ((d_next_seq == nullptr) != (d_next_start == nullptr)) { set_err(std::string(who) + ": next_seq/next_start must come together"); return HOIC_ERR_ARG; }
  if (first < 0 || count <= 0 || first + count > s->n_envs) { set_err(std::string(who) + ": env range outside [0, n_envs)"); return HOIC_ERR_ARG; }
  if (!s->has_expert) { set_err(std::string(who) + ": set_expert has not been called"); return HOIC_ERR_STATE; }
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipSetDevice(s->device));     // the launches go to the handle's device whatever the caller's current one is
  hipEvent_t* e = s->timing ? s->ev[s->n_timed % hoic_sim::NEV] : nullptr;
  // longest-first dispatch (HOIC_REORDER=1): the substep workgroups of the launch start in the order of their env's previous
  // duration; the post-step order only exists for whole-batch launches of the default form
  const int use_order = s->reorder ? 1 : 0;
  const bool whole = first == 0 && count == s->n_envs;
  const bool split = s->async_reward;
  // (split form: the order of a range's next launch is made on the side stream right behind the substeps that measured the
  //  durations, off the caller's chain; see below)
  if (use_order && !split) {
    // this launch rewrites order[first, first + count): a split-mode range that overlaps it must not trust the order made
    // behind its previous step any more (it would run foreign envs: rows outside its I/O buffers, envs stepped twice)
    for (auto& q : s->ranges) if (q.first < first + count && first < q.first + q.count) q.order_ready = false;
    hipLaunchKernelGGL(hoic_order_kernel, dim3(whole ? 2 : 1), dim3(ORDER_NT), 0, st, s->st.cost, s->st.order, s->n_envs, first, count);
  }
  if (split) {
    // Split form: termination, reset and observation at the end of the substep kernel (what the caller's next policy forward
    // needs), the reward part (contact classification, residual-force QP, reward) on the range's side stream from the
    // hand-over record.  The record buffers alternate, so the reward part of step t only has to finish before the
    // substeps of step t + 2 of the same range overwrite its record; d_action / d_reward / d_reward_info of a step must
    // stay alive and unread until hoic_sync_rewards.
    hoic_sim::AsyncRange* r = nullptr;
    for (auto& q : s->ranges) if (q.first == first && q.count == count) r = &q;
    if (!r) {
      if (s->ranges.size() >= 16) { set_err(std::string(who) + ": too many distinct env ranges in asynchronous-reward mode"); return HOIC_ERR_STATE; }
      s->ranges.emplace_back(); r = &s->ranges.back(); r->first = first; r->count = count;
      HIPCHK(hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&r->sub_done, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&r->ord_done, hipEventDisableTiming));
      for (int b = 0; b < 2; b++) HIPCHK(hipEventCreateWithFlags(&r->rew_done[b], hipEventDisableTiming));
    }
    // the order kernel behind this step rewrites order[first, first + count): other ranges that overlap it lose theirs
    for (auto& q : s->ranges) if (&q != r && q.first < first + count && first < q.first + q.count) q.order_ready = false;
    const int buf = r->next_buf; r->next_buf ^= 1;
    // hoic_set_cu_reserve: the substep kernel goes to the range's CU-masked stream `ks`, behind everything the caller has
    // enqueued so far; the caller's stream continues behind the kernel (sub_done below)
    hipStream_t ks = st;
    if (s->reserve_cus > 0) {
      if (r->sub && r->sub_reserve != s->reserve_cus) { HIPCHK(hipStreamSynchronize(r->sub)); HIPCHK(hipStreamDestroy(r->sub)); r->sub = nullptr; }
      if (!r->sub) {
        int ncu = 0;
        HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, s->device));
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int i = s->reserve_cus; i < ncu; i++) mask[i / 32] |= 1u << (i % 32);     // bit i = CU (i / 8) of XCD (i % 8): tools/probe/cu_mask.hip
        HIPCHK(hipExtStreamCreateWithCUMask(&r->sub, (uint32_t)mask.size(), mask.data()));
        r->sub_reserve = s->reserve_cus;
        if (!r->in_ready) HIPCHK(hipEventCreateWithFlags(&r->in_ready, hipEventDisableTiming));
      }
      HIPCHK(hipEventRecord(r->in_ready, st));
      HIPCHK(hipStreamWaitEvent(r->sub, r->in_ready, 0));
      ks = r->sub;
    }
    if (r->pending[buf]) HIPCHK(hipStreamWaitEvent(ks, r->rew_done[buf], 0));      // the reward part of step t - 2 read this record
    const int have_order = (use_order && r->order_ready) ? 1 : 0;
    if (have_order) HIPCHK(hipStreamWaitEvent(ks, r->ord_done, 0));                // this range's launch order (made behind its previous step)
    if (e) hipEventRecord(e[0], ks);
    hipLaunchKernelGGL(hoic_substep_kernel<2>, dim3(count), dim3(NT), 0, ks, s->d_model, s->d_cfg, s->d_ex, s->d_st, d_action, first, have_order,
                       s->use_lag ? 1 : 0, d_obs, d_reward, d_reward_info, d_flags, d_percent, d_next_seq, d_next_start, s->n_envs, buf);
    if (e) hipEventRecord(e[1], ks);
    HIPCHK(hipEventRecord(r->sub_done, ks));
    if (ks != st) HIPCHK(hipStreamWaitEvent(st, r->sub_done, 0));
    HIPCHK(hipStreamWaitEvent(r->side, r->sub_done, 0));
    if (use_order) {
      hipLaunchKernelGGL(hoic_order_kernel, dim3(1), dim3(ORDER_NT), 0, r->side, s->st.cost, s->st.order, s->n_envs, first, count);
      HIPCHK(hipEventRecord(r->ord_done, r->side)); r->order_ready = true;
    }
    if (s->postb_wide)
      hipLaunchKernelGGL((hoic_poststep_kernel<POST_B, 2>), dim3(count), dim3(NT), 0, r->side, s->d_model, s->d_cfg, s->ex, s->st, d_action, d_obs,
                         d_reward, d_reward_info, d_flags, d_percent, d_next_seq, d_next_start, first, 0, s->n_envs, buf);
    else
      hipLaunchKernelGGL(hoic_poststep_kernel<POST_B>, dim3(count), dim3(NT), 0, r->side, s->d_model, s->d_cfg, s->ex, s->st, d_action, d_obs,
                         d_reward, d_reward_info, d_flags, d_percent, d_next_seq, d_next_start, first, 0, s->n_envs, buf);
    HIPCHK(hipEventRecord(r->rew_done[buf], r->side)); r->pending[buf] = true;
    if (e) { hipEventRecord(e[2], r->side); s->n_timed++; }      // "post-step" time of this form: end of the substeps -> end of the reward part
    HIPCHK(hipGetLastError());
    return HOIC_OK;
  }
  if (e) hipEventRecord(e[0], st);
  hipLaunchKernelGGL(hoic_substep_kernel<0>, dim3(count), dim3(NT), 0, st, s->d_model, s->d_cfg, s->d_ex, s->d_st, d_action, first, use_order,
                     s->use_lag ? 1 : 0, (float*)nullptr, (float*)nullptr, (float*)nullptr, (int*)nullptr, (float*)nullptr,
                     (const int*)nullptr, (const int*)nullptr, s->n_envs, 0);
  if (e) hipEventRecord(e[1], st);
  hipLaunchKernelGGL(hoic_poststep_kernel<POST_ALL>, dim3(count), dim3(NT), 0, st, s->d_model, s->d_cfg, s->ex, s->st, d_action, d_obs,
                     d_reward, d_reward_info, d_flags, d_percent, d_next_seq, d_next_start, first, (use_order && whole) ? 1 : 0, s->n_envs, 0);
  if (e) { hipEventRecord(e[2], st); s->n_timed++; }
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}

extern "C" int32_t hoic_step(hoic_sim* s, const float* d_action, float* d_obs, float* d_reward, float* d_reward_info,
                             int32_t* d_flags, float* d_percent, const int32_t* d_next_seq, const int32_t* d_next_start,
                             void* stream) {
  if (!s) { set_err("hoic_step: null handle"); return HOIC_ERR_ARG; }
  return step_range(s, 0, s->n_envs, d_action, d_obs, d_reward, d_reward_info, d_flags, d_percent, d_next_seq, d_next_start, stream, "hoic_step");
}

// every outstanding reward part becomes a dependency of `stream`
static int32_t drain_rewards(hoic_sim* s, hipStream_t stream) {
  for (auto& r : s->ranges)
    for (int b = 0; b < 2; b++)
      if (r.pending[b]) { HIPCHK(hipStreamWaitEvent(stream, r.rew_done[b], 0)); r.pending[b] = false; }
  return HOIC_OK;
}
extern "C" int32_t hoic_set_async_reward(hoic_sim* s, int32_t enable, void* stream) {
  if (!s) { set_err("hoic_set_async_reward: null handle"); return HOIC_ERR_ARG; }
  HIPCHK(hipSetDevice(s->device));
  if (!enable) { const int32_t rc = drain_rewards(s, (hipStream_t)stream); if (rc != HOIC_OK) return rc; }
  s->async_reward = enable != 0;
  return HOIC_OK;
}
extern "C" int32_t hoic_set_cu_reserve(hoic_sim* s, int32_t n_cus) {
  if (!s) { set_err("hoic_set_cu_reserve: null handle"); return HOIC_ERR_ARG; }
  int ncu = 0;
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, s->device));
  if (n_cus < 0 || n_cus % 8 != 0 || n_cus > ncu - 8) { set_err("hoic_set_cu_reserve: n_cus must be a multiple of 8 in [0, CUs - 8]"); return HOIC_ERR_ARG; }
  s->reserve_cus = n_cus;
  return HOIC_OK;
}
extern "C" int32_t hoic_sync_rewards(hoic_sim* s, void* stream) {
  if (!s) { set_err("hoic_sync_rewards: null handle"); return HOIC_ERR_ARG; }
  HIPCHK(hipSetDevice(s->device));
  return drain_rewards(s, (hipStream_t)stream);
}

extern "C" int32_t hoic_step_range(hoic_sim* s, int32_t first, int32_t count, const float* d_action, float* d_obs, float* d_reward,
                                   float* d_reward_info, int32_t* d_flags, float* d_percent, const int32_t* d_next_seq,
                                   const int32_t* d_next_start, void* stream) {
  return step_range(s, first, count, d_action, d_obs, d_reward, d_reward_info, d_flags, d_percent, d_next_seq, d_next_start, stream, "hoic_step_range");
}

extern "C" int32_t hoic_get_state(hoic_sim* s, float* d_qpos, float* d_qvel, int32_t* d_cur_t, void* stream) {
  if (!s) return HOIC_ERR_ARG;
  HIPCHK(hipSetDevice(s->device));
  hipLaunchKernelGGL(hoic_get_state_kernel, dim3(s->n_envs), dim3(NT), 0, (hipStream_t)stream, s->d_model, s->st, d_qpos, d_qvel, d_cur_t);
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}
extern "C" int32_t hoic_set_state(hoic_sim* s, const float* d_qpos, const float* d_qvel, void* stream) {
  if (!s || !d_qpos || !d_qvel) { set_err("hoic_set_state: null"); return HOIC_ERR_ARG; }
  HIPCHK(hipSetDevice(s->device));
  { const int32_t rc = drain_rewards(s, (hipStream_t)stream); if (rc != HOIC_OK) return rc; }
  hipLaunchKernelGGL(hoic_set_state_kernel, dim3(s->n_envs), dim3(NT), 0, (hipStream_t)stream, s->d_model, s->st, d_qpos, d_qvel);
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}
extern "C" int32_t hoic_get_rfc_score(hoic_sim* s, float* d_score, void* stream) {
  if (!s || !d_score) return HOIC_ERR_ARG;
  HIPCHK(hipMemcpyAsync(d_score, s->st.rfc_score, (size_t)s->n_envs * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return HOIC_OK;
}

extern "C" int32_t hoic_probe_forward(hoic_sim* s, int32_t n, const float* d_qpos, const float* d_qvel, const float* d_ctrl,
                                      const float* d_applied, const float* d_warm, int32_t do_step, float* d_xpos,
                                      float* d_xquat, float* d_geom_xpos, float* d_geom_xmat, float* d_qM, float* d_bias,
                                      int32_t* d_ncon, float* d_contacts, float* d_qacc_smooth, float* d_qacc,
                                      float* d_qpos_out, float* d_qvel_out, int32_t* d_solver_iter, float* d_contact_force, void* stream) {
  if (!s || n <= 0 || !d_qpos || !d_qvel) { set_err("hoic_probe_forward: bad arguments"); return HOIC_ERR_ARG; }
  ProbeArgs a{d_qpos, d_qvel, d_ctrl, d_applied, d_warm, do_step, d_xpos, d_xquat, d_geom_xpos, d_geom_xmat, d_qM, d_bias,
              d_contacts, d_qacc_smooth, d_qacc, d_qpos_out, d_qvel_out, d_contact_force, d_ncon, d_solver_iter};
  HIPCHK(hipSetDevice(s->device));
  hipLaunchKernelGGL(hoic_probe_kernel, dim3(n), dim3(NT), 0, (hipStream_t)stream, s->d_model, s->d_cfg, a);
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}

// the residual-force QP on given columns (tests: the solver against the oracle's on hard instances)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void hoic_probe_qp_kernel(const float* __restrict__ cols, const int* __restrict__ ncols,
                                                                                                const double* __restrict__ rhs, int max_col,
                                                                                                double* __restrict__ lam_out, int* __restrict__ stat_out) {
  __shared__ PostWork w;
  __shared__ float qcols[QP_COL_FLOATS];
  const int k = blockIdx.x, tid = threadIdx.x, ncol = ncols[k];
  float* qc = qcols;
  const float* src = cols + (size_t)k * max_col * 7;
  for (int c = tid; c < ncol; c += NT)
    for (int i = 0; i < 7; i++) qc[i * QP_MAXCOL + c] = src[(size_t)c * 7 + i];
  wsync();
  double b[6], lam[6];
  for (int i = 0; i < 6; i++) b[i] = rhs[(size_t)k * 6 + i];
  int stat[2];
  dev_nnqp(w, (const float*)qc, ncol, b, lam, stat);
  if (tid < 6) lam_out[(size_t)k * 6 + tid] = lam[tid];
  if (tid < 2) stat_out[(size_t)k * 2 + tid] = stat[tid];
}

extern "C" int32_t hoic_probe_qp(hoic_sim* s, int32_t n, const float* d_cols, const int32_t* d_ncols, const double* d_rhs,
                                 int32_t max_col, double* d_lambda, int32_t* d_stat, void* stream) {
  if (!s || n <= 0 || !d_cols || !d_ncols || !d_rhs || !d_lambda || !d_stat || max_col <= 0 || max_col > QP_MAXCOL) {
    set_err("hoic_probe_qp: bad arguments (max_col must be in 1..380)"); return HOIC_ERR_ARG;
  }
  HIPCHK(hipSetDevice(s->device));
  hipLaunchKernelGGL(hoic_probe_qp_kernel, dim3(n), dim3(NT), 0, (hipStream_t)stream, d_cols, d_ncols, d_rhs, max_col, d_lambda, d_stat);
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}

extern "C" int32_t hoic_gae(int32_t T, int32_t N, const float* d_rewards, const float* d_masks, const float* d_values,
                            const float* d_next_values, float gamma, float tau, float* d_adv, float* d_returns, void* stream) {
  if (T <= 0 || N <= 0 || !d_rewards || !d_masks || !d_values || !d_adv || !d_returns) { set_err("hoic_gae: bad arguments"); return HOIC_ERR_ARG; }
  const float gamma_tau = (float)((double)gamma * (double)tau);
  hipLaunchKernelGGL(hoic_gae_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, N, d_rewards, d_masks, d_values,
                     d_next_values, gamma, gamma_tau, d_adv, d_returns);
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}
extern "C" int32_t hoic_normalize_advantages(int64_t n, float* d_adv, double* d_scratch, void* stream) {
  if (n < 2 || !d_adv || !d_scratch) { set_err("hoic_normalize_advantages: needs n >= 2, the advantages and 512 doubles of scratch"); return HOIC_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(hoic_adv_moments_kernel, dim3(ADV_BLOCKS), dim3(256), 0, st, d_adv, (long long)n, d_scratch);
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(hoic_adv_apply_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, d_adv, (long long)n, d_scratch);
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}
extern "C" int64_t hoic_rollout_stats_scratch_doubles(int32_t n_info) { return (int64_t)RS_BLOCKS * (4 + (n_info > 0 ? n_info : 0)) + 1; }
extern "C" int32_t hoic_rollout_stats(int64_t n, const float* d_rewards, const int32_t* d_flags, const float* d_reward_info, int32_t n_info,
                                      float end_bonus, float* d_masks, double* d_scratch, double* d_stats, void* stream) {
  if (n <= 0 || !d_rewards || !d_flags || n_info < 0 || n_info > RS_MAXINFO || (n_info && !d_reward_info) || !d_scratch || !d_stats) {
    set_err("hoic_rollout_stats: bad arguments (at most 16 reward terms)"); return HOIC_ERR_ARG;
  }
  // the last double of the scratch buffer is the ticket (zero before the first launch; every launch leaves it zero)
  unsigned* ticket = (unsigned*)(d_scratch + (size_t)RS_BLOCKS * (4 + n_info));
  hipLaunchKernelGGL(hoic_rollout_stats_kernel, dim3(RS_BLOCKS), dim3(256), 0, (hipStream_t)stream, (long long)n, d_rewards, d_flags, d_reward_info,
                     n_info, end_bonus, d_masks, d_scratch, ticket, d_stats);
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}
extern "C" int64_t hoic_zfilter_scratch_doubles(int32_t n, int32_t dim) {
  if (n <= 0 || dim <= 0) return 0;
  return (int64_t)((n + ZF_ROWS - 1) / ZF_ROWS) * dim * 2;
}
extern "C" int32_t hoic_zfilter_absorb(int32_t dim, const double* d_state, const double* const* d_fork_states, int32_t n_forks, double* d_state_out,
                                       void* stream) {
  if (dim <= 0 || !d_state || !d_fork_states || n_forks < 0 || n_forks > ZF_MAXFORK || !d_state_out || d_state_out == d_state) {
    set_err("hoic_zfilter_absorb: bad arguments (at most 8 forks; state_out must not alias state)"); return HOIC_ERR_ARG;
  }
  ZfForks f{};
  f.n = n_forks;
  for (int i = 0; i < n_forks; i++) { if (!d_fork_states[i]) { set_err("hoic_zfilter_absorb: null fork state"); return HOIC_ERR_ARG; } f.st[i] = d_fork_states[i]; }
  hipLaunchKernelGGL(hoic_zfilter_absorb_kernel, dim3((dim + ZF_NT - 1) / ZF_NT), dim3(ZF_NT), 0, (hipStream_t)stream, d_state, f, dim, d_state_out);
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}
extern "C" int32_t hoic_zfilter(int32_t n, int32_t dim, const float* d_x, const double* d_state_in, double* d_state_out,
                                int32_t update, float clip, float* d_y, double* d_scratch, void* stream) {
  if (n <= 0 || dim <= 0 || !d_x || !d_state_in) { set_err("hoic_zfilter: bad arguments"); return HOIC_ERR_ARG; }
  if (update && (!d_state_out || !d_scratch || d_state_out == d_state_in)) { set_err("hoic_zfilter: update needs a scratch buffer and a state_out that is not state_in"); return HOIC_ERR_ARG; }
  const dim3 grid((dim + ZF_NT - 1) / ZF_NT, (n + ZF_ROWS - 1) / ZF_ROWS);
  hipStream_t st = (hipStream_t)stream;
  if (update) hipLaunchKernelGGL(hoic_zfilter_moments_kernel, grid, dim3(ZF_NT), 0, st, d_x, n, dim, d_scratch);
  hipLaunchKernelGGL(hoic_zfilter_apply_kernel, grid, dim3(ZF_NT), 0, st, d_x, n, dim, d_scratch, d_state_in, d_state_out, update, clip, d_y);
  HIPCHK(hipGetLastError());
  return HOIC_OK;
}

extern "C" int32_t hoic_get_diagnostics(hoic_sim* s, int64_t* contact_overflow_total, int64_t* solver_cap_hits, int32_t* envs_with_overflow,
                                        int32_t reset) {
  if (!s) { set_err("hoic_get_diagnostics: null handle"); return HOIC_ERR_ARG; }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipDeviceSynchronize());
  std::vector<int> d((size_t)s->n_envs * 2);
  HIPCHK(hipMemcpy(d.data(), s->st.diag, d.size() * 4, hipMemcpyDeviceToHost));
  long long ov = 0, cap = 0; int nenv = 0;
  for (int e = 0; e < s->n_envs; e++) { ov += d[2 * e]; cap += d[2 * e + 1]; nenv += d[2 * e] != 0; }
  if (contact_overflow_total) *contact_overflow_total = ov;
  if (solver_cap_hits) *solver_cap_hits = cap;
  if (envs_with_overflow) *envs_with_overflow = nenv;
  if (reset) HIPCHK(hipMemset(s->st.diag, 0, d.size() * 4));
  return HOIC_OK;
}

extern "C" int32_t hoic_enable_timing(hoic_sim* s, int32_t enable) {
  if (!s) return HOIC_ERR_ARG;
  if (enable && !s->ev[0][0])
    for (int i = 0; i < hoic_sim::NEV; i++) for (int k = 0; k < 3; k++) HIPCHK(hipEventCreate(&s->ev[i][k]));
  s->timing = enable != 0;
  s->n_drained = s->n_timed;
  return HOIC_OK;
}
static float ev_ms(hipEvent_t a, hipEvent_t b) {
  if (hipEventSynchronize(b) != hipSuccess) return -1.f;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.f;
  return ms;
}
extern "C" float hoic_last_step_ms(hoic_sim* s) {
  if (!s || !s->ev[0][0] || s->n_timed == 0) return -1.f;
  hipEvent_t* e = s->ev[(s->n_timed - 1) % hoic_sim::NEV];
  return ev_ms(e[0], e[1]);
}
extern "C" float hoic_last_poststep_ms(hoic_sim* s) {
  if (!s || !s->ev[0][0] || s->n_timed == 0) return -1.f;
  hipEvent_t* e = s->ev[(s->n_timed - 1) % hoic_sim::NEV];
  return ev_ms(e[1], e[2]);
}
extern "C" int32_t hoic_env_durations(hoic_sim* s, uint32_t* h_substep, uint32_t* h_poststep) {
  if (!s || !h_substep || !h_poststep) return HOIC_ERR_ARG;
  hipSetDevice(s->device);
  HIPCHK(hipMemcpy(h_substep, s->st.cost, (size_t)s->n_envs * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(h_poststep, s->st.cost + s->n_envs, (size_t)s->n_envs * 4, hipMemcpyDeviceToHost));
  return HOIC_OK;
}
extern "C" int32_t hoic_step_times(hoic_sim* s, float* substep_ms, float* poststep_ms, int32_t max_n) {
  if (!s || !substep_ms || !poststep_ms || max_n <= 0) return HOIC_ERR_ARG;
  if (!s->ev[0][0]) return 0;
  long long first = s->n_drained;
  if (s->n_timed - first > hoic_sim::NEV) first = s->n_timed - hoic_sim::NEV;     // older ones were overwritten
  int n = 0;
  for (long long i = first; i < s->n_timed && n < max_n; i++, n++) {
    hipEvent_t* e = s->ev[i % hoic_sim::NEV];
    substep_ms[n] = ev_ms(e[0], e[1]); poststep_ms[n] = ev_ms(e[1], e[2]);
  }
  s->n_drained = first + n;
  return n;
}

extern "C" int32_t hoicdbg_env_ncon(hoic_sim* s, float* out) {      // development aid: contacts of every env's last forward pass (lag record)
  if (!s || !out) return HOIC_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy2D(out, 4, s->st.lagrec + LG_NCON, (size_t)LG_SIZE * 4, 4, s->n_envs, hipMemcpyDeviceToHost));
  return HOIC_OK;
}
extern "C" int32_t hoicdbg_phase_raw(hoic_sim* s, long long* out) {     // development aid: the raw [n_envs, 24] counters
  if (!s || !out) return HOIC_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, s->st.phase, (size_t)s->n_envs * 24 * 8, hipMemcpyDeviceToHost));
  return HOIC_OK;
}
// development aid (not part of include/hoic.h): per-phase shader-cycle counters of the last step, averaged over
// envs; all zeros unless the library was built with -DHOIC_PHASE_TIMING
extern "C" int32_t hoicdbg_phase_cycles(hoic_sim* s, double* out24, int32_t* overflow_total) {
  if (!s || !out24) return HOIC_ERR_ARG;
  std::vector<long long> h((size_t)s->n_envs * 24);
  std::vector<int> ov((size_t)s->n_envs * 2);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(h.data(), s->st.phase, h.size() * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(ov.data(), s->st.diag, ov.size() * 4, hipMemcpyDeviceToHost));
  for (int k = 0; k < 24; k++) { double a = 0; for (int e = 0; e < s->n_envs; e++) a += (double)h[(size_t)e * 24 + k]; out24[k] = a / s->n_envs; }
  if (overflow_total) { long long t = 0; for (int e = 0; e < s->n_envs; e++) t += ov[2 * e]; *overflow_total = (int32_t)t; }
  return HOIC_OK;
}
