/*
 * deepmimic_g1_hip.h — C-ABI of the Unitree G1 engine inside libdeepmimic_hip.so (MI355X / gfx950).
 *
 * Second robot of the reference: DPEnv(robot="unitree_g1") (src/deepmimic_env.py:272-484 with the unitree_g1 branches
 * at :204-211, :244-246, :303-307, :348-351, :426-433; model src/mujoco/humanoid_deepmimic/envs/asset/
 * deepmimic_unitree_g1.xml).  Same conventions as deepmimic_hip.h: plain pointers and sizes, 0 or a negative DM_E* code,
 * caller-owned DEVICE buffers unless the name says host_, stream-ordered on the hipStream_t passed as void*.
 *
 * The model argument is `struct DmModel` of include/dm_model.h compiled with -DDM_ROBOT_G1 (nq 44, nv 43, 94 geoms, the
 * trailing G1 fields: friction loss, mesh hull vertices, task constants); model_bytes must equal its sizeof.
 */
#ifndef DEEPMIMIC_G1_HIP_H
#define DEEPMIMIC_G1_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct DmG1Engine *DmG1Handle;

#define DMG1_NQ 44
#define DMG1_NV 43
#define DMG1_NU 37
#define DMG1_NACT 23   /* policy actions: the 14 hand motors are held at 0 (src/deepmimic_env.py:303-307) */
#define DMG1_NOBS 85   /* qpos[7:] 37 | 0.1 qvel[6:] 37 | torso 8 | foot contacts 2 | phase 1 (src/deepmimic_env.py:33-45) */
#define DMG1_NOBS_COMBINED 98 /* DPCombinedEnv: 82 + extra contacts 8 + phase 1 + player-action obs 7 (src/combined_env.py:495-505) */
#define DMG1_NBODY 39
#define DMG1_NGEOM 94
#define DMG1_MAXCON 48  /* contact slots per forward evaluation (MuJoCo: nconmax 200, xml :10) */
#define DMG1_MAXROW 256 /* constraint rows per forward evaluation: 37 friction-loss + limits + 4 per contact */
#define DMG1_DEBUG_STRIDE 1024
#define DMG1_MAX_CLIPS 8   /* clip slots per engine (as DM_MAX_CLIPS of deepmimic_hip.h) */

typedef struct DmG1Config {
  int32_t num_envs;
  int32_t max_ep_length;   /* 1000 (DPEnvConfig.MAX_EP_LENGTH) */
  float vel_obs_scale;     /* 0.1 */
  float high_z;            /* 2.0; low_z comes from the model (RobotConfig.low_z = 0.4) */
  float obs_bound;         /* 100.0 */
  uint64_t seed;
  int32_t auto_reset;
  int32_t device;
  int32_t task;            /* 0: DPEnv (src/deepmimic_env.py); 1: DPCombinedEnv (src/combined_env.py:101-533) as the reference runs
                            * it: clips 0, 1, 2 = walk, run, getup_facedown_towalk, per-env motion 0 walk 1 run 2 getup 3 to_getup,
                            * obs 98, terms 8 (the five imitation terms, imitation_reward, task_reward, debug_n_bad_angles),
                            * max_ep_length 2000 */
  int32_t amnesty_steps;   /* 150 (DPCombinedEnvConfig.AMNESTY_STEPS, combined_env.py:34) */
  int32_t to_getup_len;    /* 180 (MTToGetup.length, combined_env.py:97) */
  int32_t pipeline;        /* how dmg1_step runs: 0 (default) = split pipeline from 512 envs up, else monolithic; 1 = monolithic (one wave
                            * runs its env's whole step); 2 = split (per evaluation a per-env launch and a batch-wide narrowphase launch with
                            * one wave per colliding PAIR; ~135 KB of device memory per env).  Same results either way. */
} DmG1Config;

void dmg1_default_config(DmG1Config *cfg);
size_t dmg1_model_sizeof(void);

/* Replaces MujocoEnv.__init__(xml, 6) (src/deepmimic_env.py:301) for robot="unitree_g1", one instance per batch. */
int dmg1_create(const void *model, size_t model_bytes, const DmG1Config *cfg, DmG1Handle *out);
int dmg1_destroy(DmG1Handle h);
const char *dmg1_last_error(DmG1Handle h);

/* Replaces DPEnv.load_mocap (src/deepmimic_env.py:321-324): HOST float64 tables of MocapDM(robot="unitree_g1"):
 * qpos[L*44], qvel[L*43], body_xpos[L*39*3], geom_xpos[L*94*3].  flags: 1 floor motion, 2 acyclical motion
 * (src/config.py:36-37), 4 the "run" roll / pitch rule (src/deepmimic_env.py:426-433).  clip_id 0..DMG1_MAX_CLIPS-1; the
 * DPCombinedEnv task reads slots 0, 1, 2 = walk, run, getup. */
int dmg1_load_clip(DmG1Handle h, int clip_id, int L, const double *host_qpos, const double *host_qvel, const double *host_body_xpos,
                   const double *host_geom_xpos, int flags);

/* Replaces DPEnv.reset() / reset_model(idx_init) (src/deepmimic_env.py:496-510).  mask NULL = all envs; idx_init NULL =
 * random frame (reference_state_init).  obs_out float[N*85]. */
int dmg1_reset(DmG1Handle h, const uint8_t *mask, const int32_t *idx_init, float *obs_out, void *stream);

/* Replaces DPEnv.step(action) for every env (src/deepmimic_env.py:335-484) plus the VecEnv worker's auto-reset.
 * actions float[N*23] (scaled by 20 and padded with 14 zeros inside, :348-351); obs float[N*85]; rew float[N]; done uint8[N];
 * terms float[N*5]; reason int32[N] (DM_REASON_* of deepmimic_hip.h, 8 = run angle rule); terminal_obs float[N*85].
 * terms, reason, terminal_obs may be NULL. */
int dmg1_step(DmG1Handle h, const float *actions, float *obs, float *rew, uint8_t *done, float *terms, int32_t *reason,
              float *terminal_obs, void *stream);

/* Replaces DPEnv.step(action, force_state=(qpos, qvel)) (src/deepmimic_env.py:355-357).  qpos float[N*44], qvel float[N*43]. */
int dmg1_step_forced(DmG1Handle h, const float *qpos, const float *qvel, float *obs, float *rew, uint8_t *done,
                     float *terms, int32_t *reason, void *stream);

/* MujocoEnv.set_state (+ sim.forward) / sim.get_state for all N envs; qacc_warmstart may be NULL. */
int dmg1_set_state(DmG1Handle h, const float *qpos, const float *qvel, const float *qacc_warmstart, int run_forward,
                   void *stream);
int dmg1_get_state(DmG1Handle h, float *qpos, float *qvel, float *qacc_warmstart, void *stream);
int dmg1_get_counters(DmG1Handle h, int32_t *idx_curr, int32_t *episode_length, float *episode_reward, void *stream);
int dmg1_set_counters(DmG1Handle h, const int32_t *idx_curr, const int32_t *episode_length, void *stream);

/* Row width of the obs buffers for the configured task (85 / 98); per-env motion id of the DPCombinedEnv task
 * (env.current_motion_mocap, combined_env.py:190; under that task dmg1_get/set_counters' idx_curr is current_motion_n_steps). */
/* Replaces: gym.Env.seed() / VecEnv.seed() (random.seed for reference_state_init, src/deepmimic_env.py:313): re-keys the
 * counter-based generator of the random-frame (RSI) resets.  Takes effect from the next launch. */
int dmg1_set_seed(DmG1Handle h, uint64_t seed);
int dmg1_obs_dim(DmG1Handle h);
int dmg1_get_motion(DmG1Handle h, int32_t *motion, void *stream);
/* DPEnv task: per-env clip id (one `DPEnv(motion=...)` per SubprocVecEnv worker in the reference; multi-clip batches as in
 * BASELINE config 5) — the same state slot as the motion id of the combined task.  clip_ids device int32[N]. */
int dmg1_set_env_clips(DmG1Handle h, const int32_t *clip_ids, void *stream);
int dmg1_get_env_clips(DmG1Handle h, int32_t *clip_ids, void *stream);
int dmg1_set_motion(DmG1Handle h, const int32_t *motion, void *stream);

/* Parity-test hook: per-env dump of the last forward evaluation, float[N*DMG1_DEBUG_STRIDE] (NULL switches it off):
 *  [0:117) xpos | [117:160) qacc_smooth | [160:203) qacc | 203 ncon | 204 nefc | 205 solver_iter | 206 nlimit | 207 overflow |
 *  [208:208+48*9) per contact: dist, geom1, geom2, pos3, normal3 | [640:640+256) efc_force |
 *  after a step, per RK stage k = 0..3: 1000+k contacts, 1004+k rows (low byte), 1012+k a 24-bit hash of the stage's contact
 *  list h <- (131 h + 97 geom1 + geom2 + 1) mod 2^24 (the oracle keeps the same: "stage_chash<k>") */
int dmg1_set_debug(DmG1Handle h, float *debug);

/* Diagnostics of the split pipeline: per round r = 0..5 of the LAST dmg1_step, host_out[4 r + 0 / 1 / 2] = tickets of
 * support-query pairs (MPR, plane-mesh), of analytic pairs, and tickets pulled.  Synchronises the device.  Zeros when monolithic. */
int dmg1_queue_counters(DmG1Handle h, int32_t *host_out24);

/* Kernel time of the last dmg1_step in ms (HIP events on the launch stream), or < 0. */
float dmg1_last_kernel_ms(DmG1Handle h);

#ifdef __cplusplus
}
#endif
#endif
