/*
 * deepmimic_hip.h — C-ABI of libdeepmimic_hip.so (MI355X / gfx950).
 *
 * The reference has no FFI boundary of its own on this path: DPEnv.step() calls
 * gym's MujocoEnv.do_simulation -> mujoco_py.MjSim.step (Cython) -> libmujoco
 * (src/deepmimic_env.py:301,357,362).  This header is the boundary the build adds
 * underneath the reference's two Python protocol surfaces (gym.Env per-env,
 * SB3 VecEnv batched).  Each entry point cites what it replaces.
 *
 * Conventions: plain pointers and sizes only; returns 0 on success or a negative
 * DM_E* code, never throws; every buffer pointer is CALLER-OWNED DEVICE memory
 * (e.g. torch.Tensor.data_ptr()) unless the name says `host_`; work is
 * stream-ordered on the hipStream_t passed as `void* stream` (NULL = default
 * stream); the library owns only its internal per-env state.  Re-entrant per
 * handle, no global mutable state (an eval env may live on another thread,
 * src/sb3_ppo.py:173-174).
 */
#ifndef DEEPMIMIC_HIP_H
#define DEEPMIMIC_HIP_H

#include <stdint.h>

#include "dm_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct DmEngine *DmHandle;

#define DM_OK 0
#define DM_EINVAL (-22)
#define DM_ENOMEM (-12)
#define DM_EHIP (-5)
#define DM_ENODEV (-19)

#define DM_MAX_CLIPS 8

/* done_reason codes (src/deepmimic_env.py:424,438,442 strings; :366-378 and :465-476 early-outs) */
#define DM_REASON_NONE 0
#define DM_REASON_LOW_Z 1
#define DM_REASON_HIGH_Z 2
#define DM_REASON_MAX_EP_LEN 3
#define DM_REASON_ACYCLIC_END 4
#define DM_REASON_SIM_ERROR 5
#define DM_REASON_OBS_BOUNDS 6
#define DM_REASON_FALLEN_NO_AMNESTY 7 /* DPCombinedEnv "fallen without amnesty" (src/combined_env.py:436) */

/* Integrator selector (DmConfig.integrator) */
#define DM_CFG_INT_MODEL 0
#define DM_CFG_INT_EULER 1
#define DM_CFG_INT_RK4 2

/* Task selector (DmConfig.task) */
#define DM_TASK_DPENV 0     /* DPEnv            (src/deepmimic_env.py:272-484): obs 67, terms 5 */
#define DM_TASK_COMBINED 1  /* DPCombinedEnv    (src/combined_env.py:101-533) on humanoid3d: obs 72, terms 8;
                             * clips 0,1,2 = walk, run, getup; per-env motion id 0 walk 1 run 2 getup 3 to_getup */

/* Task configuration — DPEnvConfig (src/deepmimic_env.py:258-270), RobotConfig.low_z
 * (src/config.py:13) and the reward weights hard-coded in step() (:400-404). */
typedef struct DmConfig {
  int32_t num_envs;
  int32_t max_ep_length;   /* 1000 */
  float vel_obs_scale;     /* 0.1 */
  float low_z, high_z;     /* 0.7, 2.0 */
  float w_pose, w_vel, w_end_eff, w_com, w_joint_limit; /* 0.75 0.1 0.15 0.0 -0.1 */
  float obs_bound;         /* 100.0 */
  uint64_t seed;           /* counter-based RNG seed for reference-state-init resets */
  int32_t auto_reset;      /* 1: SubprocVecEnv worker semantics (reset inside step when done) */
  int32_t device;          /* HIP device ordinal */
  int32_t lpt_schedule;    /* 1 (default): dm_step launches envs longest-first by last step's work estimate */
  int32_t task;            /* DM_TASK_DPENV (default) or DM_TASK_COMBINED */
  int32_t amnesty_steps;   /* 150  DPCombinedEnvConfig.AMNESTY_STEPS (combined_env.py:34) */
  int32_t to_getup_len;    /* 180  MTToGetup.length (combined_env.py:97) */
  int32_t integrator;      /* DM_CFG_INT_MODEL (default): <option integrator> of the XML (RK4, xml :9);
                            * DM_CFG_INT_EULER: MuJoCo's semi-implicit Euler with implicit joint damping [EXT mj_Euler]
                            * (one forward evaluation per step instead of four); DM_CFG_INT_RK4 */
  int32_t stale_contact_slots; /* 0 (default): the foot-contact observation reads the ACTIVE contacts [0, ncon).  1: SURVEY F8, the
                            * reference's literal behaviour (src/deepmimic_env.py:88,113): mujoco-py's `mjdata.contact` is the whole
                            * contact array, so slots >= ncon still show the pair an earlier evaluation left there (until a later
                            * evaluation with more contacts overwrites it, or a simulator error resets the data). */
} DmConfig;

void dm_default_config(DmConfig *cfg);

/* Replaces: MujocoEnv.__init__(xml, 6) -> load_model_from_path + MjSim (deepmimic_env.py:301),
 * one instance per batch instead of one per SubprocVecEnv worker (src/sb3_ppo.py:275). */
int dm_create(const DmModel *model, const DmConfig *cfg, DmHandle *out);
int dm_destroy(DmHandle h);
const char *dm_last_error(DmHandle h);
int dm_num_envs(DmHandle h);
/* Row widths of the obs / terms buffers for the configured task: 67 / 5 (DPEnv) or 72 / 8 (DPCombinedEnv:
 * the five calc_imitation_reward terms, info["imitation_reward"], info["task_reward"], debug_n_bad_angles). */
int dm_obs_dim(DmHandle h);
int dm_terms_dim(DmHandle h);

/* Replaces: DPEnv.load_mocap -> MocapDM tables (deepmimic_env.py:321-324; mocap_v2.py:338-348).
 * HOST float64 tables: qpos[L*35], qvel[L*34], body_xpos[L*14*3], geom_xpos[L*16*3]. */
int dm_load_clip(DmHandle h, int clip_id, int L, const double *host_qpos, const double *host_qvel,
                 const double *host_body_xpos, const double *host_geom_xpos);
/* Clip flags: MotionConfig.floor_motions / acyclical_motions membership (src/config.py:36-37):
 * DM_CLIP_FLOOR skips the low/high COM termination (deepmimic_env.py:420), DM_CLIP_ACYCLIC ends the episode
 * when the last frame is reached (:440-442). */
#define DM_CLIP_FLOOR 1
#define DM_CLIP_ACYCLIC 2
int dm_set_clip_flags(DmHandle h, int clip_id, int flags);
/* Per-env clip assignment (device int32[N]; NULL = all envs use clip 0). */
int dm_set_env_clips(DmHandle h, const int32_t *clip_ids, void *stream);
/* Reads the per-env clip id back; under DM_TASK_COMBINED it is the motion id (0 walk, 1 run, 2 getup,
 * 3 to_getup), i.e. `env.current_motion_mocap` of src/combined_env.py:190, and dm_set_env_clips sets it. */
int dm_get_env_clips(DmHandle h, int32_t *clip_ids, void *stream);

/* Replaces: DPEnv.reset()/reset_model(idx_init) (deepmimic_env.py:496-510) for the envs with
 * mask[i] != 0 (mask NULL = all).  idx_init NULL = random frame from the engine RNG
 * (reference_state_init, :312-316).  obs_out float[N*67] (rows of unmasked envs untouched). */
int dm_reset(DmHandle h, const uint8_t *mask, const int32_t *idx_init, float *obs_out, void *stream);

/* Replaces: DPEnv.step(action) for every env (deepmimic_env.py:335-484) plus the VecEnv worker's
 * auto-reset.  actions float[N*28]; obs float[N*67]; rew float[N]; done uint8[N];
 * terms float[N*5] = reward_config, reward_qvel, reward_end_eff, reward_com, reward_joint_limit
 * (:251-255); reason int32[N]; terminal_obs float[N*67] (written for done envs).  terms, reason,
 * terminal_obs may be NULL. */
int dm_step(DmHandle h, const float *actions, float *obs, float *rew, uint8_t *done, float *terms,
            int32_t *reason, float *terminal_obs, void *stream);

/* Replaces: `self.sim.step()` alone (deepmimic_env.py:362, frame_skip 1): ctrl <- actions, one mj_step-equivalent, state and
 * warm start advance; NO observation, reward, termination, counters or auto-reset.  The "physics-only" figure SURVEY 8(d)
 * asks for beside the full step(), and the building block of a frame_skip > 1 loop.  Envs are launched longest-first
 * by the work estimates of the last dm_step, like dm_step itself.  actions float[N*28]. */
int dm_physics_step(DmHandle h, const float *actions, void *stream);

/* Replaces: DPEnv.step(action, force_state=(qpos, qvel)) (deepmimic_env.py:355-357): set_state +
 * sim.forward, then obs/reward/done exactly as step().  qpos float[N*35], qvel float[N*34].
 * No auto-reset on this path. */
int dm_step_forced(DmHandle h, const float *qpos, const float *qvel, float *obs, float *rew,
                   uint8_t *done, float *terms, int32_t *reason, void *stream);

/* Replaces: MujocoEnv.set_state (+ sim.forward) and sim.get_state for n envs listed in env_ids
 * (device int32[n]; NULL = envs 0..n-1).  qacc_warmstart/ctrl may be NULL (left unchanged / not
 * returned).  Used for teacher-forced parity tests and checkpointing of the env batch. */
int dm_set_state(DmHandle h, const int32_t *env_ids, int n, const float *qpos, const float *qvel,
                 const float *qacc_warmstart, const float *ctrl, int run_forward, void *stream);
/* Replaces: sim.forward() on the current state (src/deepmimic_env.py:491): recomputes every derived quantity and the
 * warm start from the stored qpos / qvel without changing them.  env_ids NULL = envs 0..n-1. */
int dm_forward(DmHandle h, const int32_t *env_ids, int n, void *stream);
int dm_get_state(DmHandle h, const int32_t *env_ids, int n, float *qpos, float *qvel,
                 float *qacc_warmstart, float *ctrl, void *stream);
/* task counters: idx_curr, episode_length int32[N]; episode_reward float[N] (deepmimic_env.py:452-455) */
int dm_get_counters(DmHandle h, int32_t *idx_curr, int32_t *episode_length, float *episode_reward,
                    void *stream);
int dm_set_counters(DmHandle h, const int32_t *idx_curr, const int32_t *episode_length, void *stream);

/* Replaces: gym.Env.seed() (random.seed for reference_state_init, src/deepmimic_env.py:313): re-keys the counter-based
 * generator of the random-frame resets (and of dm_fill_random_actions).  Takes effect from the next launch. */
int dm_set_seed(DmHandle h, uint64_t seed);

/* Derived quantities of the LAST forward evaluation of every env, for parity tests:
 * sim.data.body_xpos/geom_xpos/cvel/qacc and the contact list.  Layout per env (floats):
 *   [0:42) xpos 14x3 | [42:90) geom_xpos 16x3 | [90:174) cvel 14x6 | [174:208) qacc |
 *   [208:242) qacc_smooth | 242 ncon | 243 nefc | 244 solver_iter | 245 nlimit | 246 overflow |
 *   247/248 per-RK-stage ncon / nefc packed one byte per stage (int32 bit pattern) |
 *   [256:256+3*32) contacts: (geom1, geom2, dist) x 32 | [352:416) efc_force
 * => DM_DEBUG_STRIDE floats per env. Enabled by dm_set_debug(h, buf) with buf float[N*stride] or NULL. */
#define DM_DEBUG_STRIDE 416
int dm_set_debug(DmHandle h, float *debug_buf);

/* Longest-first scheduling (DmConfig.lpt_schedule): 4096 envs are only ~2 rounds of resident wavefronts, so the
 * step time is set by stragglers (envs with many active constraints / an auto-reset).  Every dm_step records a
 * work estimate per env; the next dm_step bucket-sorts it on device (one tiny kernel) and hands the heavy envs to
 * the first workgroups.  This only changes which env a workgroup picks up, never any result (tests: permutation
 * invariance).  No reference counterpart: SubprocVecEnv workers are scheduled by the OS.
 * dm_get_work copies the last estimates (device int32[N]) for inspection. */
int dm_get_work(DmHandle h, int32_t *work_out, void *stream);

/* Device-side uniform random actions in [-2,2) from the counter-based generator shared with the
 * oracle's baseline driver (oracle/dm_oracle.c: hash32), for bench.py config 2. */
int dm_fill_random_actions(DmHandle h, float *actions, uint32_t step_index, void *stream);

/* Timing of the step kernel on the engine's stream with HIP events (ms of the last dm_step). */
int dm_last_step_ms(DmHandle h, float *ms);
/* Mean kernel duration over the launches recorded since dm_enable_timing(h, n) (ring of 512 event pairs, read
 * without a host sync in between: the caller synchronises once, after its timed region).  n = 1: every launch carries an
 * event pair; n > 1: every n-th launch (two event records cost ~4 us of stream time each); 0: off. */
int dm_mean_step_ms(DmHandle h, float *ms, int32_t *count);
int dm_enable_timing(DmHandle h, int enable);

/* ---- learner side of the rollout loop --------------------------------------------------------------------------
 * Fused PPO clipped-surrogate loss, forward + backward (csrc/dm_ppo.hip).  Replaces the elementwise / reduction tail of
 * SB3's PPO.train [EXT] (called from src/sb3_ppo.py:307-313): log_prob and entropy of the diagonal Gaussian, ratio,
 * clipped surrogate, F.mse_loss value loss, per-minibatch advantage normalisation, and their gradients.
 * Device pointers: mean[B*A], log_std[A], value[B], act[B*A], old_logp[B], adv[B], ret[B] in; grad_mean[B*A],
 * grad_log_std[A], grad_value[B] out; out8 = {loss, policy_loss, value_loss, entropy, approx_kl, clip_fraction,
 * adv_mean, 1/(adv_std + 1e-8)}; scratch >= 2 floats.  A <= 32.  Stream-ordered; returns 0 or a negative DM_E* code. */
int dm_ppo_loss(const float *mean, const float *log_std, const float *value, const float *act, const float *old_logp,
                const float *adv, const float *ret, int B, int A, float clip_range, float vf_coef, float ent_coef,
                int normalize_advantage, float *grad_mean, float *grad_log_std, float *grad_value, float *out8,
                float *scratch, void *stream);

/* Weight / bias gradient of one linear layer of the policy or value MLP (csrc/dm_ppo.hip, MFMA split-K):
 * dW[O x I] += dY^T X, db[O] += column sums of dY, for dY [B x O], X [B x I] row-major, B a multiple of 64.
 * dW and db must be zero on entry.  Replaces the weight-gradient GEMM + bias reduction of torch.nn.Linear's backward
 * inside SB3's PPO.train [EXT] for the launch-bound [256,128] network of src/sb3_ppo.py:265. */
int dm_linear_wgrad(const float *dY, const float *X, float *dW, float *db, int B, int O, int I, void *stream);

/* Minibatch gather of the rollout buffer for PPO.train [EXT]: out_x[r] = x[idx[r]], r < B, for obs [n x D], act [n x A],
 * adv, ret, old log-prob in one launch (idx: int64 as produced by torch.randperm). */
int dm_ppo_gather(const long long *idx, int B, const float *obs, int D, const float *act, int A, const float *adv,
                  const float *ret, const float *logp, float *o_obs, float *o_act, float *o_adv, float *o_ret, float *o_logp,
                  void *stream);

/* torch.nn.utils.clip_grad_norm_(max_norm) + torch.optim.Adam.step() [EXT, as used by SB3's PPO.train] on one flat
 * parameter / gradient / moment buffer of n floats.  state2 = {scratch, step count, DM_ADAM_PARTIALS partial sums} on the
 * device, zero-initialised by the caller once; its length is passed (state2_floats >= 2 + DM_ADAM_PARTIALS, else -22: r2's
 * two-float buffer would be overrun, hence also the new names).  grad_scale > 0 multiplies the gradient on the fly: 1 on one
 * GPU, 1 / world after the sum all-reduce of the data-parallel learner (no separate division launch).  No weight decay, no
 * amsgrad.  The gradient norm is reduced in a fixed order, so replicas holding the same (all-reduced) gradient stay
 * bit-identical. */
#define DM_ADAM_PARTIALS 1024
int dm_flat_adam_step(float *p, const float *g, float *m, float *v, int n, float lr, float beta1, float beta2, float eps,
                      float max_norm, float grad_scale, float *state2, int state2_floats, void *stream);
/* the same without the begin launch (state2 prepared by dm_ppo_mlp_grad's adam_state2 fold): two launches */
int dm_flat_adam_update(float *p, const float *g, float *m, float *v, int n, float lr, float beta1, float beta2, float eps,
                        float max_norm, float grad_scale, float *state2, int state2_floats, void *stream);

/* the same (begin != 0: dm_flat_adam_step, else dm_flat_adam_update) with the gather of the NEXT minibatch riding on the first
 * launch: dm_ppo_gather(next) as extra blocks of the norm launch — SB3's RolloutBuffer.get indexing [EXT] for minibatch j + 1
 * beside clip_grad_norm_ of minibatch j, one launch less per optimizer step.  next == NULL: no gather. */
typedef struct DmGatherSpec {
  const long long *idx; int B, D, A, reserved;
  const float *obs, *act, *adv, *ret, *logp;
  float *o_obs, *o_act, *o_adv, *o_ret, *o_logp;
} DmGatherSpec;
int dm_flat_adam_step_gather(float *p, const float *g, float *m, float *v, int n, float lr, float beta1, float beta2, float eps,
                             float max_norm, float grad_scale, float *state2, int state2_floats, int begin, const DmGatherSpec *next,
                             void *stream);

/* out[o] += sum_b Y[b][o] for a row-major [B x O] matrix (out zeroed by the caller, stream-ordered): the bias gradient of the
 * nn.Linear layers of PPO.train [EXT] whose weight gradient stays on the library GEMM (layers beyond 256 units). */
int dm_colsum(const float *Y, int B, int O, float *out, void *stream);

/* Wide trunks ([1024,512], BASELINE configs 3-5) in SB3's PPO.train [EXT] (policy_kwargs of src/sb3_ppo.py:265), three fusions
 * around the library GEMMs (csrc/dm_ppo.hip):
 * dm_linear_tanh: Y[B x O] = tanh(X W^T + b) for the FIRST layer (X = observations [B x I], I <= 128): nn.Linear + nn.Tanh in one
 *   launch, Y written once;
 * dm_tanh_linear_wgrad: that layer's backward, dW[O x I] += (dY * (1 - Y^2))^T X, db[O] += its column sums (zero on
 *   entry) — no dZ intermediate, no input gradient (observations need none);
 * dm_tanh_bwd_colsum: dZ = dY * (1 - Y^2) (dZ may alias dY) and db[O] += column sums of dZ for the deeper layers. */
int dm_linear_tanh(const float *X, const float *W, const float *bias, float *Y, int B, int O, int I, void *stream);
int dm_tanh_linear_wgrad(const float *dY, const float *Y, const float *X, float *dW, float *db, int B, int O, int I, void *stream);
int dm_tanh_bwd_colsum(const float *dY, const float *Y, float *dZ, float *db, int B, int O, void *stream);

/* Rollout side of SB3's collect_rollouts [EXT] (driven by src/sb3_ppo.py:307-313), two launches per env step:
 * dm_policy_sample: act = mean + exp(log_std) * N(0,1) (counter-based generator: seed, env, counter[0], action index),
 *   logp of the diagonal Gaussian, act_env = clamp(act, lo, hi) (what DPEnv.step receives);
 * dm_rollout_store: rollout-buffer row <- (obs the policy saw, action, value, logp, reward, done), last_obs <- new obs,
 *   counter[0] += 1.  All pointers are device pointers; b_* point at row t of the [T, N, ...] buffers. */
int dm_policy_sample(const float *mean, const float *log_std, int N, int A, unsigned long long seed, const unsigned *counter,
                     const float *lo, const float *hi, float *act, float *act_env, float *logp, void *stream);
int dm_rollout_store(int N, int D, int A, const float *last_obs, const float *act, const float *val, const float *logp,
                     const float *rew, const unsigned char *done, const float *new_obs, float *b_obs, float *b_act, float *b_val,
                     float *b_logp, float *b_rew, float *b_done, float *last_obs_out, unsigned *counter, void *stream);

/* The policy side of a rollout step as ONE launch (csrc/dm_policy.hip): both trunks of SB3's actor-critic MLP
 * (obs -> H1 -> H2 -> A / 1, tanh; what [EXT] ActorCriticPolicy.forward computes for src/sb3_ppo.py:307-313), the
 * sampling head of dm_policy_sample (same draws: seed, env, counter[0] + draw_offset, action index) and the
 * rollout-buffer writes of the policy's outputs (pass row t of the [T, N, ...] buffers as act / logp / val / obs_copy).
 * dm_policy_pack re-orders the three nn.Linear weights ([out, in] row-major) of ONE trunk into MFMA operand order
 * (dm_policy_packed_floats floats, 16-byte aligned; A = 1 for the value trunk); call it again when the weights change.
 * H1, H2 multiples of 32, A <= 32, 32 (H1 + 4 + max(D8 + 4, 132)) floats of LDS <= 160 KB.  mean_out and obs_copy may be
 * NULL; deterministic != 0 returns act = mean (logp of the mean). */
long long dm_policy_packed_floats(int D, int H1, int H2, int A);
int dm_policy_pack(const float *W1, const float *W2, const float *W3, int D, int H1, int H2, int A, float *packed, void *stream);
int dm_policy_forward(const float *obs, int N, int D, int H1, int H2, int A, const float *pi_packed, const float *pi_b1,
                      const float *pi_b2, const float *pi_b3, const float *vf_packed, const float *vf_b1, const float *vf_b2,
                      const float *vf_b3, const float *log_std, unsigned long long seed, const unsigned *counter,
                      unsigned draw_offset, int deterministic, const float *lo, const float *hi, float *mean_out, float *act,
                      float *act_env, float *logp, float *val, float *obs_copy, void *stream);

/* One PPO minibatch gradient of SB3's MlpPolicy with two hidden layers (net_arch [H1, H2], tanh, separate policy / value
 * trunks) in three launches (csrc/dm_ppo_mlp.hip): what zero_grad + evaluate_actions + the dm_ppo_loss loss + backward
 * compute for src/sb3_ppo.py:254-271,307-312 -> [EXT] PPO.train.  Index 0 = policy trunk (layers D->H1, H1->H2, H2->A),
 * 1 = value trunk (.., H2->1); W are nn.Linear weights [out, in] row-major.  Gradients are ACCUMULATED into gW / gb /
 * g_log_std, which the caller zeroes on the same stream (the optimizer's flat arena); out8 as in dm_ppo_loss.
 * B a multiple of 64, H1 / H2 multiples of 32 and <= 256, A <= 32; workspace >= dm_ppo_mlp_workspace_floats floats,
 * 16-byte aligned. */
typedef struct DmPpoMlpStep {
  int32_t B, D, H1, H2, A, normalize_advantage;
  float clip_range, vf_coef, ent_coef;
  int32_t reserved;
  const float *obs, *act, *adv, *ret, *old_logp, *log_std;
  const float *W[2][3];
  const float *b[2][3];
  float *gW[2][3];
  float *gb[2][3];
  float *g_log_std;
  float *out8;
  float *workspace;
  long long workspace_floats;
  /* optional folds (NULL / 0 to skip), each saving one launch of the optimizer step: */
  float *zero_ptr;              /* zero_floats floats cleared by the first launch (the flat gradient arena that holds gW / gb / g_log_std) */
  long long zero_floats;
  float *adam_state2;           /* dm_flat_adam_step's begin (state2[0] = 0, state2[1] += 1): follow with dm_flat_adam_update */
  float *loss_acc;              /* loss_acc[0] += loss, loss_acc[1] += 1 (running mean of the loss without a host-side add) */
} DmPpoMlpStep;
long long dm_ppo_mlp_workspace_floats(int B, int D, int H1, int H2, int A);
int dm_ppo_mlp_grad(const DmPpoMlpStep *step, void *stream);

/* One PPO minibatch gradient of the WIDE actor-critic MLP (net_arch up to [1024, 512], the net BASELINE configs 3-5 name) on the bf16
 * matrix pipe: replaces optimizer.zero_grad(), evaluate_actions, the loss and loss.backward() of SB3's PPO.train [EXT]
 * (src/sb3_ppo.py:254-271,307-312).  Three launches: weights -> bf16 operand layouts (+ advantage statistics and the optional folds);
 * the fused forward / loss / input-gradient chain of both trunks (v_mfma_f32_32x32x16_bf16, fp32 accumulation, activations in LDS as
 * bf16); the six weight gradients dW = dZ^T X (+ the loss scalars).  fp32 master weights, loss arithmetic and gradients; gradients
 * are ACCUMULATED into the caller's fp32 buffers (zero on entry, or pass the arena as zero_ptr); out8 as dm_ppo_loss.
 * trunk 0 = policy (pi, action_net), 1 = value (vf, value_net); W / b / gW / gb in torch layout [out][in].
 * Supported: B % 64 == 0, D <= 112, H1 in {256, 512, 768, 1024}, H2 in {128, 256, 384, 512}, A <= 32 (dm_ppo_wide_supported). */
typedef struct DmPpoWideStep {
  int32_t B, D, H1, H2, A, normalize_advantage;
  float clip_range, vf_coef, ent_coef;
  int32_t reserved;
  const float *obs, *act, *adv, *ret, *old_logp, *log_std;
  const float *W[2][3], *b[2][3];
  float *gW[2][3], *gb[2][3];
  float *g_log_std;
  void *wpk[2];                 /* bf16 scratch, dm_ppo_wide_packed_elems(D, H1, H2) elements per trunk */
  void *xbT;                    /* bf16 scratch, (dm_ppo_wide_dp(D) rounded up to 32) * B elements: the observations, transposed; this and
                                   the five arrays below are opaque (MFMA fragment order, csrc/dm_ppo_wide.hip) */
  void *h1T[2], *dz1T[2];       /* bf16 scratch [H1][B]: tanh output of layer 1, d loss / d (pre-activation of layer 1), transposed */
  void *h2T[2], *dz2T[2];       /* bf16 scratch [H2][B] */
  void *dz3T[2];                /* bf16 scratch [32][B] */
  float *part;                  /* scratch, 2 * (B / 32) * 40 floats */
  float *stats8, *out8;         /* 8 floats each */
  float *zero_ptr;              /* optional folds, as DmPpoMlpStep */
  long long zero_floats;
  float *adam_state2, *loss_acc;
} DmPpoWideStep;
long long dm_ppo_wide_packed_elems(int D, int H1, int H2);
int dm_ppo_wide_dp(int D);
int dm_ppo_wide_supported(int B, int D, int H1, int H2, int A);
int dm_ppo_wide_grad(const DmPpoWideStep *step, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPMIMIC_HIP_H */
