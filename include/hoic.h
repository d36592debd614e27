/* hoic.h — C-ABI of the MI355X-native batched HandObjMimic simulator (libhoic_hip.so).
 *
 * The reference has no FFI of its own for this path: its boundary is the mujoco_py call set underneath
 * HandObjMimic4.  Each entry point below names the reference interface it replaces.
 * All pointers named d_* are DEVICE pointers (HBM, e.g. torch tensor .data_ptr()); everything else is
 * host memory.  `stream` is a hipStream_t passed as void* (NULL = default stream).  Functions return 0 on
 * success, a negative hoic_status otherwise; they never throw.  Per-env simulation failures are reported
 * through the `fail` output flags, like the reference's try/except around do_simulation
 * (uhc/envs/ho_im4.py:627-637).
 */
#ifndef HOIC_H
#define HOIC_H
#include <stddef.h>
#include <stdint.h>
#include "hoic_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hoic_sim hoic_sim;

enum hoic_status {
  HOIC_OK = 0,
  HOIC_ERR_ARG = -1,
  HOIC_ERR_MODEL = -2,
  HOIC_ERR_DEVICE = -3,
  HOIC_ERR_STATE = -4
};

/* Environment configuration: the fields HandObjMimic4.__init__ reads from cfg
 * (uhc/envs/ho_im4.py:46-107; values from config/release/*.yml via handmimic_config.py:75-143). */
typedef struct hoic_env_config {
  float jkp[HOIC_MAX_NU], jkd[HOIC_MAX_NU], torque_lim[HOIC_MAX_NU]; /* joint_params, yml:116-143 */
  float pos_diff_thresh, rot_diff_thresh, jpos_diff_thresh, obj_pos_diff_thresh, obj_rot_diff_thresh;
  float residual_force_scale, residual_torque_scale;
  int32_t sim_step;        /* frame_skip, 15 */
  int32_t future_w_size;   /* 5 (obs layout is compiled for 5) */
  int32_t residual_force, explain_force, surface_contact;
  int32_t pd_rel;          /* pd_type == "rel" */
  int32_t solver_iterations; /* Newton iteration cap per substep (fp32) */
  int32_t pd_ref_offset;     /* 0: training env; 1: streaming env (uhc/envs/ho_im_test.py: the new frame is inserted before
                              * env.step, InferenceServer/RLTest.py:298-299, so the PD target is expert frame t+1) */
  int32_t mesh_single_contact; /* 0 (default): a convex-mesh pair yields every contact point this narrow phase finds (up to 3 plane-mesh,
                              * 2 capsule-mesh, 4 box-mesh); 1: only the deepest one, the contact COUNT of MuJoCo 2.1's convex
                              * collider (libccd/MPR: one contact per mesh pair; assets/SingleDepth/bottle_light.xml:12-19,
                              * banana_light.xml:13-22) -- the switch a MuJoCo capture will be compared under */
} hoic_env_config;

/* Reward parameters of ho_mimic_reward_9 (uhc/envs/ho_reward.py:943-967); refreshed every epoch by
 * Config.update_adaptive_params (handmimic_config.py:157-195). Order:
 * w_p,w_wp,w_v,w_j,w_op,w_or,w_ov,w_orfc, k_p,k_wp,k_v,k_j,k_op,k_or,k_ov,k_orfc */
typedef struct hoic_reward_params {
  float wk[16];
  float end_reward;   /* env.end_reward, added when info['end'] (agent_handmimic.py:479-480) */
  int32_t use_end_reward;
} hoic_reward_params;

/* ---- lifetime -----------------------------------------------------------------------------------
 * replaces mujoco_py.load_model_from_path + MjSim(model) (mujoco_env.py:18-34) for n_envs environments */
hoic_sim* hoic_create(const void* model_blob, size_t nbytes, int32_t n_envs, int32_t device_id);
void hoic_destroy(hoic_sim* s);
int32_t hoic_num_envs(const hoic_sim* s);
int32_t hoic_obs_dim(const hoic_sim* s);     /* observation_space.shape[0] = 617 */
int32_t hoic_action_dim(const hoic_sim* s);  /* action_space.shape[0] = 32 */
const char* hoic_last_error(void);
/* first 16 hex digits of the SHA-256 over the sources this library was built from (csrc/Makefile BUILD_ID; no reference
 * analogue): measurement files under profiles/ carry it, bench.py quotes a counter pass only for the library it was taken on */
const char* hoic_build_id(void);

int32_t hoic_set_config(hoic_sim* s, const hoic_env_config* cfg);
int32_t hoic_set_reward_params(hoic_sim* s, const hoic_reward_params* rp);
/* The same refresh ordered on a HIP stream instead of behind a device synchronisation: launches enqueued on (or ordered
 * behind) `stream` after the call see the new parameters, earlier ones the old; the host does not wait.  For a training
 * loop that enqueues iteration k+1's rollout while the GPU still runs iteration k's update (agent_handmimic.py:311-336
 * makes the same refresh once per epoch, between the two). */
int32_t hoic_set_reward_params_async(hoic_sim* s, const hoic_reward_params* rp, void* stream);
int32_t hoic_set_mode(hoic_sim* s, int32_t train); /* set_mode('train'|'test'), ho_im4.py:132 */

/* ---- expert motions: set_expert for every sequence at once (ho_im4.py:135; arrays are the output of
 * DatasetSingleDepth.preprocess_seq, dataset_singledepth.py:78-142).  Host float32 arrays, sequences
 * concatenated along time: hand_dof[T,26], hand_dof_vel[T,26], obj_pose[T,7], obj_vel[T,3],
 * obj_angvel[T,3], body_pos[T,21,3], body_quat[T,21,4]; seq_len[n_seq] sums to T. */
int32_t hoic_set_expert(hoic_sim* s, int32_t n_seq, const int32_t* seq_len, const float* hand_dof,
                        const float* hand_dof_vel, const float* obj_pose, const float* obj_vel,
                        const float* obj_angvel, const float* body_pos, const float* body_quat);

/* ---- streaming (HandObjMimicTest.insert_new_frame, uhc/envs/ho_im_test.py:34-38): the LAST sequence of the expert
 * table may grow at run time.  hoic_set_expert_reserve (before hoic_set_expert) reserves room for `frames` more
 * frames; hoic_append_expert_frame appends one frame — host float32 arrays hand_dof[26], hand_dof_vel[26],
 * obj_pose[7], obj_vel[3], obj_angvel[3], body_pos[21*3], body_quat[21*4] — to the last sequence (asynchronous on
 * `stream`; HOIC_ERR_STATE when the reserve is used up).  With an ever-growing sequence and pd_ref_offset = 1 the
 * absolute frame index cur_t + k of the kernels equals the reference's sliding-window index k. */
int32_t hoic_set_expert_reserve(hoic_sim* s, int32_t frames);
int32_t hoic_append_expert_frame(hoic_sim* s, const float* hand_dof, const float* hand_dof_vel, const float* obj_pose,
                                 const float* obj_vel, const float* obj_angvel, const float* body_pos,
                                 const float* body_quat, void* stream);

/* ---- reset: MujocoEnv.reset + HandObjMimic4.reset_model (mujoco_env.py:95-114, ho_im4.py:690-716) for the
 * envs listed in d_env_ids (n entries, int32; NULL = all envs in order).  d_seq / d_start: per listed env,
 * the sequence index and start frame (agent_handmimic.py:444-452).  d_obs_out: [n_envs,617] rows of the
 * listed envs are overwritten.  Ids outside [0, n_envs) are ignored; sequence / start values are clamped into the
 * expert table (the same holds for d_next_seq / d_next_start of hoic_step). */
int32_t hoic_reset(hoic_sim* s, const int32_t* d_env_ids, int32_t n, const int32_t* d_seq, const int32_t* d_start,
                   float* d_obs_out, void* stream);

/* ---- step: HandObjMimic4.step for every env (ho_im4.py:611-662) fused with ho_mimic_reward_9.
 * d_action [n_envs,32]; outputs: d_obs [n_envs,617], d_reward [n_envs] (custom reward incl. end reward),
 * d_reward_info [n_envs,9], d_flags [n_envs,4] int32 = fail,end,done,reserved; d_percent [n_envs].
 * If d_next_seq/d_next_start are non-NULL, envs that are done are reset in the same launch to that
 * sequence/start (the sampler's next-episode draw) and d_obs holds the reset observation for them. */
int32_t hoic_step(hoic_sim* s, const float* d_action, float* d_obs, float* d_reward, float* d_reward_info,
                  int32_t* d_flags, float* d_percent, const int32_t* d_next_seq, const int32_t* d_next_start,
                  void* stream);
/* The same env step for the envs [first, first + count) only; every device array holds `count` rows, row b belongs to
 * env first + b.  Ranges are independent: stepping disjoint ranges on different streams lets the tail of one range's
 * launch (few long-running envs) overlap the other range's work; the host mirror's rollout pipelines two half-batches
 * this way (hoic_amd/agent.py).  State, outputs and results are identical to hoic_step on the whole batch. */
int32_t hoic_step_range(hoic_sim* s, int32_t first, int32_t count, const float* d_action, float* d_obs, float* d_reward,
                        float* d_reward_info, int32_t* d_flags, float* d_percent, const int32_t* d_next_seq,
                        const int32_t* d_next_start, void* stream);


/* ---- state access: MjSim.get_state / set_state (mujoco_env.py:109-113) and the data.* reads of the env.
 * d_qpos [n_envs,33], d_qvel [n_envs,32]; set_state also clears warm start and makes lagged == current. */
int32_t hoic_get_state(hoic_sim* s, float* d_qpos, float* d_qvel, int32_t* d_cur_t, void* stream);
int32_t hoic_set_state(hoic_sim* s, const float* d_qpos, const float* d_qvel, void* stream);
/* rfc_score of the last step (env.rfc_score, ho_im4.py:631) */
int32_t hoic_get_rfc_score(hoic_sim* s, float* d_score, void* stream);

/* ---- probe: one mj_forward (+ optional Euler step) at given state, dumping the mjData fields the env reads
 * (sim.forward()/sim.step() + data.qM, qfrc_bias, body_xpos, body_xquat, geom_xpos, contact[], qacc;
 * call sites ho_im4.py:383,398-401,545,810-829,884).  Used by the parity tests.  All device pointers, any
 * output may be NULL.  d_ctrl [n,26], d_applied [n,32], d_warm [n,32] may be NULL (zeros).
 * d_contacts [n,HOIC_PROBE_MAXCON,16] rows: dist,pos[3],frame[9],geom1,geom2,dim.
 * d_contact_force [n,HOIC_PROBE_MAXCON,6]: mj_contactForce of every contact (ho_im4.py:866-881 get_contact, test mode): the
 * contact-frame 6-vector decoded from the pyramid's edge forces of the constraint solve -- [0] normal force (sum of the edges),
 * [1], [2] tangential forces mu_i (f_2i - f_2i+1), [3] the torsional moment of a condim-4 contact, [4], [5] zero. */
#define HOIC_PROBE_MAXCON 32
int32_t hoic_probe_forward(hoic_sim* s, int32_t n, const float* d_qpos, const float* d_qvel, const float* d_ctrl,
                           const float* d_applied, const float* d_warm, int32_t do_step, float* d_xpos,
                           float* d_xquat, float* d_geom_xpos, float* d_geom_xmat, float* d_qM, float* d_bias,
                           int32_t* d_ncon, float* d_contacts, float* d_qacc_smooth, float* d_qacc,
                           float* d_qpos_out, float* d_qvel_out, int32_t* d_solver_iter, float* d_contact_force, void* stream);

/* Generalized advantage estimation (khrylib core/common.py:12-19, called from agent_pg.py:46) over a time-major
 * rollout, no handle needed: d_rewards, d_masks (0 at episode ends), d_values [T, N] float32, d_next_values [N] (value
 * after the last collected step; null: 0).  delta_t = r_t + gamma V_{t+1} m_t - V_t, A_t = delta_t + gamma tau A_{t+1} m_t
 * -> d_adv [T, N] (not normalised), d_returns = V + A. */
int32_t hoic_gae(int32_t T, int32_t N, const float* d_rewards, const float* d_masks, const float* d_values,
                 const float* d_next_values, float gamma, float tau, float* d_adv, float* d_returns, void* stream);

/* Normalisation of the batch's advantages in place (khrylib core/common.py:22: (A - A.mean()) / A.std(), torch's unbiased
 * standard deviation over the whole batch): d_adv [n] float32 as hoic_gae wrote them, d_scratch 512 float64.  Sums in float64
 * in a fixed order; two launches. */
int32_t hoic_normalize_advantages(int64_t n, float* d_adv, double* d_scratch, void* stream);

/* The logger's statistics of a fixed-horizon rollout and the batch's masks in one launch (LoggerRL as the reference's sampler
 * fills it step by step: agent_handmimic.py:476-482 -- c_reward is the step's reward WITHOUT the end bonus -- and
 * uhc/khrylib/rl/core/logger_rl.py; masks: agent_handmimic.py:484, mask = 0 if done else 1).  n = T x N entries of the rollout's
 * storage: d_rewards [n], d_flags [n, 4] (fail, end, done, .) and d_reward_info [n, n_info] as hoic_step wrote them; end_bonus =
 * the bonus the step kernel added on 'end' steps.  d_stats (float64) = sum, min, max of c_reward, number of done entries,
 * n_info sums of the reward terms; d_masks [n] float32 (may be null).  d_scratch: hoic_rollout_stats_scratch_doubles(n_info)
 * float64, ZEROED once by the caller before the first launch (its last entry is a ticket that every launch returns to zero).
 * Sums in float64 in a fixed order. */
int64_t hoic_rollout_stats_scratch_doubles(int32_t n_info);
int32_t hoic_rollout_stats(int64_t n, const float* d_rewards, const int32_t* d_flags, const float* d_reward_info, int32_t n_info,
                           float end_bonus, float* d_masks, double* d_scratch, double* d_stats, void* stream);

/* The running observation filter of the sampler (khrylib ZFilter / RunningStat, uhc/khrylib/utils/zfilter.py:8-73,
 * called on every observation at agent_handmimic.py:463) on a batch, no handle needed.  d_x [n, dim] float32; state =
 * (count, mean[dim], S[dim]) as 1 + 2 dim float64 on the device.  update != 0: the batch is merged into the state
 * first (128-row chunks, Chan's pairwise update in a fixed order: the statistics of pushing the rows one by one) and
 * the new state goes to d_state_out (must not alias d_state_in); d_scratch holds hoic_zfilter_scratch_doubles(n, dim)
 * float64.  Then, if d_y is not null, y = clip((x - mean) / (sqrt(var) + 1e-8), +-clip) with var = S / (count - 1)
 * (mean^2 while count == 1) of the state AFTER the batch -> d_y [n, dim] float32. */
int64_t hoic_zfilter_scratch_doubles(int32_t n, int32_t dim);
int32_t hoic_zfilter(int32_t n, int32_t dim, const float* d_x, const double* d_state_in, double* d_state_out,
                     int32_t update, float clip, float* d_y, double* d_scratch, void* stream);
/* Merge of the per-range forks of the filter after a pipelined rollout (each env range updates its own copy of the running
 * statistics, like the reference's sampler threads: uhc/khrylib/rl/agents/agent.py:64-120; here all copies are merged, the
 * reference keeps worker 0's): d_state = the filter the forks were copied from (unchanged since), d_fork_states = host array of
 * n_forks (<= 8) device pointers to the forks' (count, mean[dim], S[dim]); d_state_out (not aliasing d_state) receives the
 * statistics of pushing every fork's new rows into d_state.  One launch; bit-identical to hoic_amd.rl.BatchZFilter.absorb. */
int32_t hoic_zfilter_absorb(int32_t dim, const double* d_state, const double* const* d_fork_states, int32_t n_forks, double* d_state_out,
                            void* stream);
/* The same filter step for the fixed-horizon sampler's env ranges together with what follows it in a range's chain
 * (agent_handmimic.py:463-465: running_state(state) then policy_net.select_action): the launch that normalises the rows also
 * writes them in the rollout forward's operand format T (hoic_mlp_pack_tiled; d_T [n x Kp] at 2^d_exps[slot_x]) and refreshes the
 * forward engine's delayed exponents (slots in `mask`) from d_amax as hoic_mlp_update_exps does -- two launches instead of four.
 * States and filter are bit-identical to hoic_zfilter.  n must be a multiple of 128; d_scratch as for hoic_zfilter; update = 0
 * normalises with d_state_in (d_state_out, d_scratch unused).  d_P (may be null): these n rows of the UPDATE's first-layer
 * operand as well -- hoic_mlp_pack's row format [n x 2 KpP] at the same exponent (agent_pg.py:39-49 hands the stacked states
 * of the rollout to update_params; here they arrive packed); columns >= Kp of d_P are not written (the caller zeroes them once). */
int32_t hoic_zfilter_tiled(int32_t n, int32_t dim, const float* d_x, const double* d_state_in, double* d_state_out, int32_t update,
                           float clip, float* d_y, double* d_scratch, void* d_T, int32_t Kp, int32_t* d_exps, int32_t slot_x,
                           float* d_amax, int32_t nslots, uint64_t mask, int32_t target, int32_t* d_overflow, void* d_P, int32_t KpP,
                           void* stream);

/* The residual-force QP of HandObjMimic4.get_rfc_score (ho_im4.py:1040-1083) on caller-supplied data, n independent
 * problems:  min_x |A x - b|^2 + c.x + 1e-7/2 |x|^2, x >= 0.  d_cols [n, max_col, 7] float32: column k of problem i
 * is (a_k[6], c_k); d_ncols [n] columns in use (<= max_col <= 380); d_rhs [n, 6] float64.  Outputs: d_lambda [n, 6]
 * float64 = 2 (A x - b) (rfc_score = |lambda[0:3]| / 2 + |lambda[3:6]| / 2), d_stat [n, 2] = active-set iterations,
 * dual-Newton fallback iterations.  Test / diagnostic entry: hoic_step runs the same routine on its own columns. */
int32_t hoic_probe_qp(hoic_sim* s, int32_t n, const float* d_cols, const int32_t* d_ncols, const double* d_rhs,
                      int32_t max_col, double* d_lambda, int32_t* d_stat, void* stream);

/* Run-time guards of the two compiled capacities (the reference model allows nconmax = 100 contacts and 20 solver
 * iterations, assets/hand_model/spheremesh/sphere_mesh_hand_add_geom.xml:4-8; MuJoCo reports either limit as a warning
 * in mjData.warning, which the env never reads).  Counters accumulate over all steps since creation or the last reset:
 *   contact_overflow_total  forward passes in which the narrow phase produced more than 32 contacts for one env; the
 *                           list is then cut after the first 32 in pair order (floor pairs, table, hand-object, hand-hand)
 *   solver_cap_hits         substeps whose Newton loop used hoic_env_config.solver_iterations iterations without
 *                           meeting a stop criterion (gradient, step or cost-improvement tolerance)
 *   envs_with_overflow      number of envs with at least one overflow
 * Any output may be NULL; reset != 0 clears the counters.  Synchronises the device. */
/* Split post-step for pipelined samplers.  By default hoic_step / hoic_step_range deliver all outputs in stream order.  With
 * hoic_set_async_reward(s, 1, .) a step delivers d_obs, d_flags and d_percent in stream order (termination, the in-launch
 * reset and the observation run at the end of the substep kernel: all the next policy forward needs) while d_reward,
 * d_reward_info and the stored rfc_score come from the rest of the post-step work (contact classification, residual-force
 * QP in float64, ho_mimic_reward_9) on a side stream per env range, off the caller's critical path: they are valid for
 * `stream` after hoic_sync_rewards(s, stream), and the d_action / d_reward / d_reward_info buffers of every step since
 * the last synchronisation must stay alive and untouched until then.  Same results as the default form (bit-identical;
 * tests/test_gpu_parity.py::test_async_reward_matches_the_default_step).  hoic_reset, hoic_set_state, hoic_set_expert and
 * switching the mode off synchronise by themselves.  Reference: HandObjMimic4.step returns obs, reward and done together
 * (ho_im4.py:611-662); the sampler only needs obs to continue (agent_handmimic.py:463-482). */
int32_t hoic_set_async_reward(hoic_sim* s, int32_t enable, void* stream);
int32_t hoic_sync_rewards(hoic_sim* s, void* stream);
/* Scheduling knob of the split form (no reference analogue; results never depend on it): keep `n_cus` compute units (a
 * multiple of 8: the same number on each of the 8 XCDs; 0 = off) free of substep workgroups.  The substep kernel of a range
 * then runs on a CU-masked stream of the library (hipExtStreamCreateWithCUMask) behind an event of the caller's stream, and
 * the caller's stream continues behind the kernel's end.  With three 168-register substep wavefronts on every SIMD nothing
 * else becomes resident on a CU, so the kernels of the OTHER ranges' policy chains (filter, tiled forward, action head) and
 * the reward parts otherwise wait for substep wavefronts to retire; on the reserved CUs they always find room.  Takes effect
 * for ranges stepped after the call. */
int32_t hoic_set_cu_reserve(hoic_sim* s, int32_t n_cus);
int32_t hoic_get_diagnostics(hoic_sim* s, int64_t* contact_overflow_total, int64_t* solver_cap_hits,
                             int32_t* envs_with_overflow, int32_t reset);

/* ---- the dense GEMMs of the PPO update (AgentPPO.update_policy / AgentPG.update_value: forward + backward of the two
 * GELU MLPs, uhc/khrylib/rl/agents/agent_ppo.py:16-56, agent_pg.py:18-25, models/mlp.py:24-27) at float32 accuracy on
 * the f16 matrix cores; no handle needed.  Operands are "packed" tensors: a float32 matrix [R x C] scaled by a power of
 * two 2^e and split error-free into float16 pairs x 2^e = hi + lo, stored as R rows of 2 C halves in groups of eight
 * columns [h0 .. h7 l0 .. l7].  Per-tensor exponents live in a device int32 table d_exps (slot indices are
 * arguments), running maxima in d_amax (float32 per slot).
 *
 * hoic_mlp_pack: float32 d_x [R x C] (row stride ld; optional elementwise factor d_mul of the same layout) -> d_P
 *   [Rp x 2 Cp] and / or the transpose d_PT [Cp x 2 Rp], zero padded, scaled by 2^d_exps[slot].
 * hoic_mlp_amax / hoic_mlp_update_exps: d_amax[slot] = max(d_amax[slot], max |x (* mul)|); then for the slots of `mask`
 *   e = target - ceil(log2 amax) (2^e amax in [2^(target-1), 2^target)), amax cleared; *d_overflow += 1 for a slot whose
 *   measured maximum exceeded the float16 range under its previous exponent (exact != 0 skips THAT test: the slot was
 *   measured on the tensor that is packed NEXT, nothing was packed under the old exponent) and, always, for a slot whose
 *   maximum is not finite (hoic_mlp_amax / hoic_mlp_amax_colsum report Inf for an array holding any Inf or NaN).
 * hoic_mlp_gemm: C[m][n] = extra_scale 2^-(e_a + e_b) sum_k A[m][k] B[n][k], A [M x 2K], B [N x 2K] packed (M % 256 ==
 *   N % 128 == K % 32 == 0), three f16 MFMAs (hi.hi + hi.lo + lo.hi) into one float32 accumulator.  epi 0: float32 d_C
 *   [splits][M x N] (split-K slabs over blockIdx.y);  epi 1 (forward layer): v = gelu(C + bias[n]) -> optional
 *   float32 d_hf32, packed d_P [M x 2N] and transposed d_PT [N x 2M] at 2^d_exps[slot_out], gelu'(.) -> d_gout;
 *   epi 2 (data gradient): v = C * d_gin[m][n] -> d_P / d_PT likewise.  Epilogues 1, 2 fold max |v| into d_amax[slot_out].
 *   d_colpart (optional, epi 2 in pipeline mode 3 without d_PT): [M / 128][N] float32, the column sums of v over each
 *   128-row chunk (the bias gradient's partial sums; every entry written by exactly one wavefront in fixed order).
 * hoic_mlp_slab_reduce: out[r][c] = scale * sum_s slabs[s][r][c] for c < out_cols (fixed order: deterministic).
 * hoic_mlp_rowsum_packed: out[r] = 2^-e * sum_c (hi + lo)[r][c] of a packed [rows x 2 Cp] tensor (bias gradients). */
int32_t hoic_mlp_pack(const float* d_x, const float* d_mul, int32_t R, int32_t C, int64_t ld, void* d_P, void* d_PT, int32_t Rp,
                      int32_t Cp, const int32_t* d_exps, int32_t slot, void* stream);
int32_t hoic_mlp_amax(const float* d_x, const float* d_mul, int64_t n, float* d_amax, int32_t slot, void* stream);
int32_t hoic_mlp_update_exps(int32_t* d_exps, float* d_amax, int32_t nslots, uint64_t mask, int32_t target, int32_t exact,
                             int32_t* d_overflow, void* stream);
/* the same relative to a reference slot whose exponent is exact (the loss-side gradient): e_i = target - ceil(log2 amax_i) +
 * (e_ref - *d_ref_prev), then *d_ref_prev = e_ref -- the hidden-layer gradients keep their head-room when the whole
 * gradient's scale jumps between passes. */
int32_t hoic_mlp_update_exps_rel(int32_t* d_exps, float* d_amax, int32_t nslots, uint64_t mask, int32_t target, int32_t ref_slot,
                                 int32_t* d_ref_prev, int32_t* d_overflow, void* stream);
int32_t hoic_mlp_gemm(int32_t epi, int32_t M, int32_t N, int32_t K, const void* d_A, const void* d_B, const int32_t* d_exps,
                      float* d_amax, int32_t slot_a, int32_t slot_b, int32_t slot_out, float extra_scale, int32_t splits, float* d_C,
                      const float* d_bias, const float* d_gin, float* d_gout, float* d_hf32, void* d_P, void* d_PT, float* d_colpart,
                      void* stream);
/* hoic_mlp_gemm_tn: C[i][j] = extra_scale 2^-(e_a + e_b) sum_m A[m][i] B[m][j] with BOTH operands row-major over the
 * contraction (sample) index m: A [K x 2M], B [K x 2N] packed along their columns (M % 256 == N % 128 == K % 32 == 0) ->
 * float32 slabs d_C [splits][M x N].  The weight gradient dW = dZ^T H from the activations / gradients as the other
 * epilogues wrote them (LDS transposing reads, ds_read_b64_tr_b16).
 * hoic_mlp_colsum_packed: out[c] = 2^-e sum_r (hi + lo)[r][c] of a packed row-major [R x 2C] tensor (bias gradients);
 * d_scratch holds ceil(R / 512) * C floats; fixed summation order.
 * hoic_mlp_amax_colsum: one pass over two float32 arrays [R x C]: d_amax[slot] = max(., max |x * mul|) and
 *   d_part[ceil(R / 128)][C] = column sums of x * mul per 128-row chunk (the last layer's dZ = dH * gelu').
 * hoic_mlp_colpart_finish: out[c] = sum over the chunks of d_part[chunk][c] (fixed order: deterministic).
 * hoic_mlp_set_pipeline: kernel variant of hoic_mlp_gemm / hoic_mlp_gemm_tn (measurement aid; 3 = default): 0 plain loop,
 * 1 software-pipelined 8-wavefront kernel, 2 4-wavefront 256 x 128 kernel with K stages of 16 and two workgroups per CU,
 * 3 = 2 with D[m][n] accumulators for epilogues 1, 2 when no transposed output is requested (full-line stores, column
 * partial sums) and the weight gradient on the same 4-wavefront main loop (modes 0-2: its 8-wavefront predecessor). */
int32_t hoic_mlp_gemm_tn(int32_t M, int32_t N, int32_t K, const void* d_A, const void* d_B, const int32_t* d_exps, int32_t slot_a,
                         int32_t slot_b, float extra_scale, int32_t splits, float* d_C, void* stream);
int32_t hoic_mlp_colsum_packed(const void* d_P, int32_t R, int32_t C, float* d_out, float* d_scratch, const int32_t* d_exps,
                               int32_t slot, void* stream);
int32_t hoic_mlp_amax_colsum(const float* d_x, const float* d_mul, int32_t R, int32_t C, float* d_amax, int32_t slot, float* d_part,
                             void* stream);
int32_t hoic_mlp_colpart_finish(const float* d_part, int32_t nchunks, int32_t C, float* d_out, void* stream);
int32_t hoic_mlp_set_pipeline(int32_t mode);
/* The policy's forward pass during the rollout (replaces torch.nn.Linear + GELU of PolicyGaussian.select_action,
 * uhc/khrylib/rl/core/policy_gaussian.py + uhc/khrylib/models/mlp.py:24-27, for the batched sampler): an f16x3 GEMM with
 * fused bias + GELU that uses NO LDS and <= 128 registers, so that its wavefronts run beside the simulator's substep kernel
 * (whose workgroups hold all of every CU's LDS) instead of queueing behind it.  Operands in the tiled format T of a matrix
 * [R x K]: tile (a, s) = rows 32 a .. + 31, k = 16 s .. + 15 at byte ((a K/16 + s) * 2048): hi plane [64 lanes x 16 B], lo
 * plane [64 x 16 B]; lane 32 hf + l holds row 32 a + l, k = 16 s + 8 hf .. + 7 (one contiguous KB per operand load).
 * hoic_mlp_pack_tiled: float32 [R x C] (row stride ld) -> T [Rp x Kp] (Rp % 32 == Kp % 16 == 0, zero padded) at 2^d_exps[slot].
 * hoic_mlp_forward_tiled: out = gelu(2^-(e_x + e_w) X W^T + bias), X = T [M x K], W = T [N x K] (M % 32 == N % 64 == K % 16
 *   == 0) -> d_outT (format T [M x N] at 2^d_exps[slot_out], max |out| folded into d_amax[slot_out]: the next layer's
 *   input) or d_outF (float32 row-major [M x N]: the last hidden layer); exactly one of the two. */
int32_t hoic_mlp_pack_tiled(const float* d_x, int32_t R, int32_t C, int64_t ld, void* d_T, int32_t Rp, int32_t Kp, const int32_t* d_exps,
                            int32_t slot, void* stream);
int32_t hoic_mlp_forward_tiled(int32_t M, int32_t N, int32_t K, const void* d_X, const void* d_W, const int32_t* d_exps, float* d_amax,
                               int32_t slot_x, int32_t slot_w, int32_t slot_out, const float* d_bias, void* d_outT, float* d_outF,
                               void* stream);
/* hoic_mlp_head: d_out[m][n] = sum_k d_h[m][k] d_W[n][k] + d_bias[n] (+ d_std[n] * d_eps[m][n] when d_eps is given): the action
 * head on the policy body's output and the Gaussian sample in one LDS-free float32 launch (PolicyGaussian.forward /
 * select_action, uhc/khrylib/rl/core/policy_gaussian.py:27-33; mean + std * N(0, 1), distributions.py:11-13); with N = 1 the
 * value head (the value MLP's last nn.Linear, uhc/khrylib/rl/core/critic.py).  K % 16 == 0, N <= 32; row strides ldh / lde /
 * ldo in floats; d_bias, d_std, d_eps may be NULL.
 * hoic_mlp_head_backward: the same head's backward pass in one pass over d_h, given d_g = dLoss/d_out [M x N]:
 *   d_dh[m][k] = sum_n d_g[m][n] d_W[n][k];  d_grad[n K + k] = sum_m d_g[m][n] d_h[m][k] (weight gradient, [N x K] row-major),
 *   d_grad[N K + n] = sum_m d_g[m][n] (bias gradient).  What torch.autograd does for nn.Linear in the reference's
 *   loss.backward() (agent_ppo.py:46-56, agent_pg.py:18-25), in float32 FMAs with a fixed summation order.  K even, N <= 32;
 *   d_grad holds S = (N K + N + 1) & ~1 floats, d_part (scratch) nblocks * S floats; nblocks = row blocks of the partial sums. */
int32_t hoic_mlp_head(int32_t M, int32_t K, int32_t N, const float* d_h, int64_t ldh, const float* d_W, const float* d_bias,
                      const float* d_std, const float* d_eps, int64_t lde, float* d_out, int64_t ldo, void* stream);
int32_t hoic_mlp_head_backward(int32_t M, int32_t K, int32_t N, const float* d_h, int64_t ldh, const float* d_W, const float* d_g, int64_t ldg,
                               float* d_dh, int64_t lddh, float* d_grad, float* d_part, int32_t nblocks, void* stream);
/* The update's two losses with their backward pass, one launch each (plus the fixed-order finish of the per-block sums):
 * hoic_mlp_ppo_loss: the PPO-clip surrogate on the action head's output d_mean [M x N <= 32] -- Gaussian log-probability of
 *   d_act under (d_mean, exp(d_log_std)) (policy_gaussian.py get_log_prob, distributions.py log_prob), ratio to d_fixed
 *   [M], clipped surrogate with d_adv [M] (uhc/khrylib/rl/agents/agent_ppo.py:58-64) -> d_g = d(weight L)/d_mean [M x N],
 *   d_sums[0..N) = d(weight L)/d_log_std, d_sums[32] = L (unweighted).  d_fixed == NULL: epoch 0 (agent_ppo.py:18-20: the
 *   old policy is the current one, ratio = 1), the log-probabilities are written to d_logp_out [M].  d_sums: 34 floats,
 *   d_part: nblocks * 34 floats of scratch.
 * hoic_mlp_value_loss: L = mean((d_v - d_ret)^2) (agent_pg.py:18-25) -> d_g = d(weight L)/d_v [M], d_loss[0] = L;
 *   d_part: nblocks floats. */
int32_t hoic_mlp_ppo_loss(int32_t M, int32_t N, const float* d_mean, int64_t ldm, const float* d_act, int64_t lda, const float* d_adv,
                          const float* d_fixed, const float* d_log_std, float clip, float weight, float* d_g, int64_t ldg, float* d_logp_out,
                          float* d_sums, float* d_part, int32_t nblocks, void* stream);
int32_t hoic_mlp_value_loss(int32_t M, const float* d_v, const float* d_ret, float weight, float* d_g, float* d_loss, float* d_part,
                            int32_t nblocks, void* stream);
int32_t hoic_mlp_slab_reduce(const float* d_slabs, int32_t S, int32_t rows, int32_t cols, float* d_out, int32_t out_cols, int64_t ldo,
                             float scale, void* stream);
int32_t hoic_mlp_rowsum_packed(const void* d_P, int32_t rows, int32_t Cp, float* d_out, const int32_t* d_exps, int32_t slot,
                               void* stream);

/* hoic_step is two launches: the substep kernel (15 fused substeps, the dominant kernel) and the post-step
 * kernel (contact averaging, residual-force QP, termination, reward, observation).  Durations in milliseconds of
 * the most recent launches, measured with HIP events on the launch stream (negative if timing was not enabled
 * via hoic_enable_timing): hoic_last_step_ms = substep kernel, hoic_last_poststep_ms = post-step kernel. */
int32_t hoic_enable_timing(hoic_sim* s, int32_t enable);
float hoic_last_step_ms(hoic_sim* s);
float hoic_last_poststep_ms(hoic_sim* s);
/* Durations of the launches recorded since the previous call (a ring of 64 event sets, so that a rollout is timed
 * without one host synchronisation per step); blocks until the last of them has finished.  Returns the number of
 * entries written (<= max_n), or a negative hoic_status. */
int32_t hoic_step_times(hoic_sim* s, float* substep_ms, float* poststep_ms, int32_t max_n);
/* Per-env duration of the last step's two passes in units of 64 shader clocks (HOST arrays of n_envs entries each).
 * These are the keys of the longest-first launch order of the next step (the order never changes a result);
 * synchronises the device. */
int32_t hoic_env_durations(hoic_sim* s, uint32_t* h_substep, uint32_t* h_poststep);

#ifdef __cplusplus
}
#endif
#endif
