// dm_device.h — device-side tables and LDS layout of the fused step kernel (gfx950 only).
//
// Mapping: ONE 64-lane wavefront per environment.  Lanes take
// roles per phase: lane = body (kinematics, velocities), lane = dof (inertia rows,
// bias forces, triangular solves), lane = collision candidate pair, lane =
// constraint row (Jacobian row, A-matrix row, PGS residual).  All per-env
// intermediates live in LDS; the A = J M^-1 J^T row of each constraint lives in
// the VGPRs of its lane.  Control flow (contact count, row count, sweeps) is
// wave-uniform because a wave never mixes environments.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dm_model.h"

#define DMK_NQ DM_NQ
#define DMK_NV DM_NV
#define DMK_NU DM_NU
#define DMK_NB DM_NBODY
#define DMK_NG DM_NGEOM
#define DMK_MAXCON DM_MAXCON   // contact slots per forward evaluation
#define DMK_MAXROW DM_MAXROW   // constraint rows per forward evaluation
#define DMK_LANEROW 64         // one row per lane up to here; 65..128 rows take the two-rows-per-lane path
#define DMK_REGROW 32          // A-matrix columns kept in registers; columns 32..63 spill to ar_scratch
#define DMK_MSTRIDE 35         // dense tree-sparse M, odd stride: conflict-free rows and columns
#define DMK_MAXANC 12          // deepest dof has 12 ancestors (root 6 + hip 3 + knee 1 + ankle x,y)
#define DMK_STATE_STRIDE 144   // floats per env in the HBM state row (see DmState)
#define DMK_CLIP_ROW 80        // floats per clip frame row (73 used)

// One collision candidate: everything the prefilter and the narrowphase need except the poses.
struct DmPairDev {
  int16_t g1, g2;
  int8_t t1, t2, pad0, pad1;
  float margin, rbsum;   // max of the two margins; sum of the bounding radii (plane: radius of geom 2)
  float z1[3], z2[3];    // geom sizes
};

// Global-memory (read-only) model tables, fp32.  Built on the host from DmModel.
struct DmDev {
  // scalars
  float timestep, tolerance, pgs_scale, gravity[3];
  float K, B;               // reference-acceleration stiffness / damping (refsafe applied)
  float solimp[5];
  float total_mass_inv;
  float qpos0[36];
  int32_t iterations, npair;
  int32_t torso_body, rfoot_geom, lfoot_geom, floor_geom;
  int32_t ee_geom[4];
  // bodies
  int32_t b_parent[16], b_depth[16], b_dofadr[16], b_dofnum[16];
  float b_pos[16][3], b_ipos[16][3], b_inertia[16][6], b_mass[16], b_invw[16];
  uint32_t b_subtree[16];       // bit c: body c is in the subtree of b (incl. b)
  uint64_t b_chain[16];         // bit k: dof k moves body b
  uint32_t b_chainb[16];        // ancestor bodies root..self, one byte each (0 = none)
  // dofs
  int32_t d_body[DM_NV], d_nanc[DM_NV], d_act[DM_NV], d_limited[DM_NV];
  float d_axis[DM_NV][3];       // joint axis in the body frame (hinges)
  float d_arm[DM_NV], d_damp[DM_NV], d_invw[DM_NV], d_lo[DM_NV], d_hi[DM_NV];
  float d_gear[DM_NV], d_clo[DM_NV], d_chi[DM_NV];
  uint8_t d_anc[DM_NV][DMK_MAXANC];
  uint8_t d_ancabs[DM_NV][DMK_MAXANC + 4]; // ancestor dof at absolute depth d (0 if none)
  uint64_t d_desc[DM_NV];       // bit k: dof k is a strict descendant
  int32_t d_pbody[DM_NV];       // parent body of the dof's body
  uint64_t nanc_pack[3];        // d_nanc of every dof, 4 bits each
  uint64_t d_ancm[DM_NV];       // bit j: dof j is a strict ancestor
  int32_t d_madr[DM_NV];        // start of row i in the sparse factor
  // geoms
  int32_t g_body[16], g_type[16], g_condim[16];
  float g_pos[16][3], g_mat[16][9], g_size[16][3], g_rbound[16], g_margin[16], g_mu[16];
  // candidate pairs (canonical order)
  int16_t p_g1[DM_MAXPAIR], p_g2[DM_MAXPAIR];
  DmPairDev pairs[DM_MAXPAIR];
  uint8_t tri_a[80], tri_b[80]; // lower-triangle pair decode for the factorisation
};

// Per-env LDS working set (one wave).  13 008 B: twelve waves per CU (three per SIMD) need <= 13 653 B each.
struct EnvLds {
  float qpos[36], qvel[36], warm[36], ctrl[28];
  float qacc_smooth[36], qacc[36];
  float cs[36][2];                    // cos, sin of half joint angle per dof
  float xpos[DMK_NB][3], xquat[DMK_NB][4], xmat[DMK_NB][9], xipos[DMK_NB][3];
  float xaxis[DMK_NV][3];
  float gpos[DMK_NG][3], gmat[DMK_NG][9];
  float com[4];
  float cinert[DMK_NB][10];
  float cdof[DMK_NV][8];              // ang3, lin3, pad2 (16-byte rows)
  float cvel[DMK_NB][6];
  float M[DM_NM + 2];                 // sparse L^T D L factor, MuJoCo row layout (dm_topology.h)
  float dinv[36], dsqrtinv[36];
  float rk[5][36], rkq[4];            // RK4 bookkeeping per dof: X0 position, X0 velocity, sum b_i v_i, sum b_i a_i, stage velocity; X0 root quaternion
  // contacts of the current forward evaluation
  float c_dist[DMK_MAXCON], c_pos[DMK_MAXCON][3], c_frame[DMK_MAXCON][9];
  int32_t c_g1[DMK_MAXCON], c_g2[DMK_MAXCON];
#ifdef DM_PROFILE
  unsigned long long prof_t, prof_t0; unsigned prof[16], prof_stage[4];  // diagnostic stamps (-DDM_PROFILE build only)
#endif
  int32_t info[8];                    // ncon, nefc, nlimit, solver_iter, overflow (last forward evaluation)
  int16_t rowinfo[DMK_MAXROW];        // contact rows: (contact << 3) | edge | 0x4000 (whole pyramid kept); limit rows: -(2 dof + side + 1)
  // velocity-stage scratch (dead before the constraint stage) / box-box polygon scratch
  union {
    struct {
      float crb[DMK_NB][10];
      float qloc[DMK_NB][8];          // body rotation relative to its parent (4) + offset in the parent frame (3)
      union {
        struct { float cdofdot[DMK_NV][6]; float cacc[DMK_NB][6]; };
        alignas(16) float mbuf[DMK_NV][8];   // I_crb(body(k)) cdof_k per dof, broadcast while the columns of M are built
      };
      float cfrc[DMK_NB][6], cfrcsub[DMK_NB][6];
    } v;
    struct {
      float tr[20][DM_NV + 1];        // force-weighted constraint rows, transposed through LDS (finish)
    } fin;
    struct {
      float poly[2][16][3];
      float cand[8][8];               // dist, pos3, normal3, pad
      int32_t ncand;
    } bb;
  } u;
};

// HBM state row per env (AoS so one wave reads one contiguous 576-byte row):
//  [0:35) qpos | [35:69) qvel | [69:103) qacc_warmstart | [103:131) ctrl |
//  131 idx_curr(int) | 132 episode_length(int) | 133 episode_reward | 134 clip_id(int) |
//  135 reset_counter(int) | rest pad
#define DMS_QPOS 0
#define DMS_QVEL 35
#define DMS_WARM 69
#define DMS_CTRL 103
#define DMS_IDX 131
#define DMS_EPLEN 132
#define DMS_EPREW 133
#define DMS_CLIP 134
#define DMS_RCNT 135
#define DMS_F8R 136   // F8 option: bit k = contact slot k last held a (right foot, floor) pair; DMS_F8L: left foot
#define DMS_F8L 137

// clip row layout (DMK_CLIP_ROW floats): [0:28) qpos[7:] | [28:56) qvel[6:] | [56:60) root quat wxyz |
//  [60:72) end-effector geom xpos 4x3 | [72:75) mass-weighted body_xpos COM | [75:78) root pos | pad
// plus full reset rows kept separately: qpos[35] qvel[34] (DMK_RESET_ROW floats)
#define DMK_RESET_ROW 72

struct DmClipDev {
  const float *rows;    // L x DMK_CLIP_ROW
  const float *reset;   // L x DMK_RESET_ROW : qpos 35 | qvel 34
  int32_t L, flags;     // flags: DM_CLIP_FLOOR | DM_CLIP_ACYCLIC
};

enum { DMK_MODE_STEP = 0, DMK_MODE_FORCED = 1, DMK_MODE_RESET = 2, DMK_MODE_SETSTATE = 3,
       DMK_MODE_PHYSICS = 4 };   // PHYSICS: sim.step() alone (the mj_step-equivalent): no observation, reward, termination, counters

struct DmLaunch {
  const DmDev *T;
  float *state;                 // N x DMK_STATE_STRIDE
  float *ar_scratch;            // N x 128 x 128: A-matrix columns beyond the register-resident 32 (touched only when nefc > 32)
  DmClipDev clips[8];
  int32_t N, nslots, mode, auto_reset, max_ep_length;
  int32_t amnesty_steps, to_getup_len;   // DPCombinedEnv task only
  int32_t integrator, f8;                // DM_INT_RK4 (xml :9) or DM_INT_EULER; f8: DmConfig.stale_contact_slots
  float vel_obs_scale, low_z, high_z, obs_bound;
  float w_pose, w_vel, w_ee, w_com, w_jl;
  uint64_t seed;
  // inputs
  const float *actions;         // N x 28            (STEP)
  const float *in_qpos;         // N x 35            (FORCED / SETSTATE, indexed by slot)
  const float *in_qvel;         // N x 34
  const float *in_warm;         // N x 34 or null    (SETSTATE)
  const float *in_ctrl;         // N x 28 or null    (SETSTATE)
  const int32_t *env_ids;       // slot -> env or null
  int32_t *cost;                // per-env work estimate of this step (constraint-row updates + forwards), or null
  const uint8_t *mask;          // RESET: per-env mask or null
  const int32_t *idx_init;      // RESET: per-env frame or null (random)
  int32_t run_forward;          // SETSTATE
  // outputs
  float *obs, *rew, *terms, *terminal_obs, *debug;
  uint8_t *done;
  int32_t *reason;
};
