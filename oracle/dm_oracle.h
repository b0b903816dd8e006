/*
 * dm_oracle.h — CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * A scalar fp64 restatement, one environment at a time, of the reference hot
 * path: DPEnv.step() (src/deepmimic_env.py:335-484) down through
 * MujocoEnv.do_simulation -> mujoco_py.MjSim.step -> libmujoco mj_step.
 *
 * PARITY STATUS: the physics arithmetic of the reference lives in MuJoCo
 * 2.0/2.1.0 (closed binary at those versions, absent from /root/reference,
 * README.md:24, src/sb3_ppo.py:345).  It is restated here from MuJoCo's
 * published "Computation" documentation.  No golden
 * vector produced by MuJoCo exists in the reference tree, so the physics part
 * is "PARITY UNPINNED"; the env semantics, mocap pipeline and policy MLP are
 * pinned by fixtures under tests/golden.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library.
 */
#ifndef DM_ORACLE_H
#define DM_ORACLE_H

#include "../include/dm_model.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifdef DM_ROBOT_G1
#define DMO_MAXCON 200 /* deepmimic_unitree_g1.xml :10 nconmax="200" */
#else
#define DMO_MAXCON 100 /* MuJoCo default nconmax [EXT] */
#endif
#define DMO_MAXROW 500 /* MuJoCo default njmax [EXT] */

typedef struct DmoContact {
  double dist;
  double pos[3];
  double frame[9]; /* rows: normal (geom1->geom2), tangent1, tangent2 */
  double includemargin;
  double mu;
  int32_t geom1, geom2, dim, pair;
  int32_t efc_address, pad;
} DmoContact;

typedef struct DmoData {
  /* ---- state (what persists between steps) ---- */
  double qpos[DM_NQ];
  double qvel[DM_NV];
  double ctrl[DM_NU];
  double qacc_warmstart[DM_NV];
  double time;
  /* ---- caps (runtime; default MuJoCo's, tests may lower to match the HIP build) ---- */
  int32_t maxcon, maxrow;
  /* ---- derived by the last forward evaluation ---- */
  double xpos[DM_NBODY][3], xquat[DM_NBODY][4], xmat[DM_NBODY][9], xipos[DM_NBODY][3];
  double xanchor[DM_NJNT][3], xaxis[DM_NJNT][3];
  double geom_xpos[DM_NGEOM][3], geom_xmat[DM_NGEOM][9];
  double subtree_com[3]; /* COM of the whole kinematic tree (root body 1) */
  double cinert[DM_NBODY][10], crb[DM_NBODY][10];
  double cdof[DM_NV][6], cdof_dot[DM_NV][6];
  double cvel[DM_NBODY][6];
  double qM[DM_NM], qLD[DM_NM], qLDiagInv[DM_NV], qLDiagSqrtInv[DM_NV];
  double qfrc_bias[DM_NV], qfrc_passive[DM_NV], qfrc_actuator[DM_NV], qfrc_smooth[DM_NV];
  double qfrc_constraint[DM_NV];
  double qacc_smooth[DM_NV], qacc[DM_NV];
  int32_t ncon, nefc, solver_iter, nlimit, nfriction, pad_i;
  int32_t overflow_con, overflow_row; /* counts of dropped contacts / rows */
  int32_t stage_ncon[4], stage_nefc[4]; /* per RK stage of the last dmo_step (test diagnostics) */
  int32_t stale_contact_slots;          /* F8 (src/deepmimic_env.py:88,113): the foot-contact observation scans ALL maxcon slots of the
                                         * contact array, stale ones included, as mujoco-py's `mjdata.contact` does; 0 (default): [0, ncon) */
  int32_t pad_f8;
  int32_t stage_chash[4];               /* 24-bit hash of the stage's contact list: h <- (131 h + 97 geom1 + geom2 + 1) mod 2^24 */
  DmoContact contact[DMO_MAXCON];
  int32_t efc_type[DMO_MAXROW], efc_id[DMO_MAXROW]; /* 0 limit, 1 frictionless, 2 pyramidal */
  double efc_pos[DMO_MAXROW], efc_margin[DMO_MAXROW], efc_diagApprox[DMO_MAXROW];
  double efc_R[DMO_MAXROW], efc_D[DMO_MAXROW], efc_vel[DMO_MAXROW], efc_aref[DMO_MAXROW];
  double efc_b[DMO_MAXROW], efc_force[DMO_MAXROW], efc_frictionloss[DMO_MAXROW];
  double *efc_J;  /* maxrow x nv  (heap) */
  double *efc_AR; /* maxrow x maxrow (heap) */
} DmoData;

/* motion clip tables (src/mujoco/mocap_v2.py:338-348 getters) */
typedef struct DmoClip {
  int32_t L;
  const double *qpos;      /* L x 35 */
  const double *qvel;      /* L x 34 */
  const double *body_xpos; /* L x 14 x 3 */
  const double *geom_xpos; /* L x 16 x 3 */
  int32_t flags;           /* 1 floor motion, 2 acyclical motion (src/config.py:36-37) */
  int32_t pad;
} DmoClip;

/* per-env task state (DPEnv attributes idx_curr, episode_length, ...) */
typedef struct DmoEnv {
  int32_t idx_curr, episode_length;
  double episode_reward;
} DmoEnv;

enum { DMO_REASON_NONE = 0, DMO_REASON_LOW_Z = 1, DMO_REASON_HIGH_Z = 2,
       DMO_REASON_MAX_EP_LEN = 3, DMO_REASON_ACYCLIC_END = 4,
       DMO_REASON_SIM_ERROR = 5, DMO_REASON_OBS_BOUNDS = 6,
       DMO_REASON_RUN_ANGLE = 8 /* G1 "run" clip only (src/deepmimic_env.py:426-433) */ };

DmoData *dmo_data_new(const DmModel *m);
void dmo_data_free(DmoData *d);
void dmo_data_reset(const DmModel *m, DmoData *d); /* qpos0, zeros */

/* mj_forward / mj_step equivalents; return 0, or 1 if a NaN/huge value was met
 * (MuJoCo would reset and mujoco-py raise MujocoException [EXT]). */
int dmo_forward(const DmModel *m, DmoData *d);
int dmo_step(const DmModel *m, DmoData *d);

/* MujocoEnv.set_state + sim.forward (src/deepmimic_env.py:355-357,508) */
int dmo_set_state(const DmModel *m, DmoData *d, const double *qpos, const double *qvel);

/* src/deepmimic_env.py:33-45 (DPEnvConfig flags of :258-270) */
void dmo_get_obs(const DmModel *m, const DmoData *d, int idx_curr, int L, double *obs67);
/* src/deepmimic_env.py:193-256; terms5 = reward_config, qvel, end_eff, com, joint_limit */
double dmo_reward(const DmModel *m, const DmoData *d, const DmoClip *clip, int idx, double *terms5);

/* Full DPEnv.step (src/deepmimic_env.py:335-484).  force_qpos/force_qvel may be NULL.
 * Outputs: obs67, *reward, terms5, *reason; returns done (0/1). */
int dmo_env_step(const DmModel *m, DmoData *d, DmoEnv *e, const DmoClip *clip,
                 const double *action, const double *force_qpos, const double *force_qvel,
                 double *obs67, double *reward, double *terms5, int32_t *reason);
/* DPEnv.reset / reset_model(idx_init) (src/deepmimic_env.py:496-510) */
int dmo_env_reset(const DmModel *m, DmoData *d, DmoEnv *e, const DmoClip *clip, int idx_init,
                  double *obs67);

/* ---- DPCombinedEnv (src/combined_env.py) on the humanoid3d model: walk / run / getup / to_getup state machine.
 * The reference class is hard-wired to unitree_g1 (:165); this restates its step()/reset()/_get_obs() with the
 * humanoid3d RobotConfig (no action scale, no extra-contact geoms, low_z 0.7).  clips[3] = walk, run, getup. */
#ifdef DM_ROBOT_G1
#define DMO_NOBS_COMBINED 98 /* 37 + 37 + torso 8 + extra contacts 8 + phase + player-action obs 7 */
#else
#define DMO_NOBS_COMBINED 72
#endif
enum { DMO_MOTION_WALK = 0, DMO_MOTION_RUN = 1, DMO_MOTION_GETUP = 2, DMO_MOTION_TO_GETUP = 3 };
enum { DMO_REASON_FALLEN_NO_AMNESTY = 7 };
typedef struct DmoCombEnv {
  int32_t motion, n_steps, episode_length, pad;
  double episode_reward;
} DmoCombEnv;
/* _get_obs (:495-505): qpos[7:], S*qvel[6:], torso(8), phase(1), player-action obs (hx, hy, onehot3, getup2) */
void dmo_combined_obs(const DmModel *m, const DmoData *d, const DmoCombEnv *e, const DmoClip *clips, double *obs72);
/* step (:243-493).  terms8 = the five calc_imitation_reward terms, imitation_reward, task_reward, n_bad_angles */
int dmo_combined_step(const DmModel *m, DmoData *d, DmoCombEnv *e, const DmoClip *clips, const double *action,
                      const double *force_qpos, const double *force_qvel, double *obs72, double *reward,
                      double *terms8, int32_t *reason);
/* reset (:205-241) with the two random draws made explicit: motion in {walk, getup} and n_steps */
int dmo_combined_reset(const DmModel *m, DmoData *d, DmoCombEnv *e, const DmoClip *clips, int motion, int n_steps,
                       double *obs72);

/* narrowphase test hook (geom types of include/dm_model.h; sizes as in DmModel.geom_size; mats row-major 3x3) */
int dmo_narrowphase(int t1, const double *x1, const double *M1, const double *z1, int t2, const double *x2,
                    const double *M2, const double *z2, double margin, double *out_n_x_10);

/* Sensitivity-study switches (tests/sensitivity_extracted_policy.py only): names in dm_oracle.c; "reset" restores the
 * MuJoCo defaults.  Returns 0, or -1 for an unknown name. */
int dmo_set_tweak(const char *name, double value);

/* rotation helpers exposed for tests */
void dmo_quat_to_rpy(const double *wxyz, double *rpy);

/* Batched CPU baseline driver used by bench.py: runs `nsteps` random-torque
 * DPEnv.step()s on `nenv` envs (auto-reset to frame (env+step) %% L), single thread. */
double dmo_bench_steps(const DmModel *m, const DmoClip *clip, int nenv, int nsteps, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
