// dm_g1.hip — Unitree G1 engine (second robot of the reference: DPEnv(robot="unitree_g1"), src/deepmimic_env.py:272-484;
// model deepmimic_unitree_g1.xml: nq 44, nv 43, 39 bodies, 94 geoms of which 46 collide, 32 convex meshes, friction loss).
//
// First GPU version of SURVEY §8f-2.  Same mapping as the humanoid3d kernel — ONE 64-lane wavefront per environment, all
// per-env intermediates in LDS, lanes changing role per phase — but written for generality, not yet for speed: the dof
// tree is walked through tables (no compile-time topology), the Jacobian rows and A = J M^-1 J^T live in a per-env global
// scratch (L2), and the narrowphase runs wave-uniform in fp64: analytic routines for the pairs MuJoCo has them for,
// libccd's Minkowski Portal Refinement (restated from its published algorithm, as MuJoCo 2.x's mjc_Convex uses it) for
// everything that involves a cylinder or a mesh, with the support mapping of a mesh evaluated by all 64 lanes over its
// hull vertices.  Everything else is fp32.  Checked against oracle/libdm_oracle_g1.so (tests/test_g1_gpu.py).
#define DM_ROBOT_G1 1
#define DmModel DmModelG1
#include "../../include/dm_model.h"
#undef DmModel
#include "../../include/deepmimic_g1_hip.h"
#include "dm_g1_topology.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#define DM_OK 0
#define DM_EINVAL (-22)
#define DM_ENOMEM (-12)
#define DM_EHIP (-5)
#define DM_ENODEV (-19)

namespace g1 {

constexpr int NQ = 44, NV = 43, NU = 37, NB = 39, NG = 94, NJ = 38, NM = 434, NACT = 23, NOBS = 85, NREW = 23;
constexpr int MAXCON = DMG1_MAXCON, MAXROW = DMG1_MAXROW, MAXANC = 16, MAXSURV = 384, NCG = 48;   // NCG: geoms that collide (47)
constexpr int PAIRCAP = 128;    // split pipeline: surviving pairs per env and evaluation that the pair queue holds (monolithic: MAXSURV)
constexpr int STATE = 176;   // floats per env in the HBM state row
constexpr int S_QPOS = 0, S_QVEL = 44, S_WARM = 87, S_CTRL = 130, S_IDX = 167, S_EPLEN = 168, S_EPREW = 169, S_RCNT = 170, S_MOTION = 171;
constexpr int NOBS_C = DMG1_NOBS_COMBINED;   // DPCombinedEnv: 82 + extra contacts 8 + phase + player-action obs 7
constexpr int CLIP_ROW = 64;  // per frame: 23 reward qpos | 23 reward qvel | root quat 4 | ee geom xpos 12 | root velocity xy 2
constexpr float MINVALF = 1e-15f, MAXVALF = 1e10f;
constexpr double MINVAL = 1e-15;
enum { MODE_STEP = 0, MODE_FORCED = 1, MODE_RESET = 2, MODE_SETSTATE = 3 };
enum { ROW_LIMIT = 0, ROW_CONTACT = 2, ROW_FRICTION = 3 };

struct Dev {   // read-only model tables (global memory, fp32)
  float timestep, tolerance, pgs_scale, K, B, solimp[5], total_mass_inv, gravity[3];
  float low_z, action_scale;
  int32_t iterations, npair, maxdepth, torso_body, floor_geom, rfoot_geom, lfoot_geom, ee_geom[4], extra_geom[8];
  int32_t rew_q[NREW], rew_v[NREW], rew_j[NREW];
  float qpos0[NQ];
  int32_t b_parent[40], b_depth[40], b_dof[40];
  float b_pos[40][3], b_quat[40][4], b_ipos[40][3], b_inertia[40][6], b_mass[40], b_invw[40];
  uint64_t b_desc[40];   // bit d: body d is in the subtree of b (incl. b)
  uint64_t b_chain[40];  // bit k: dof k moves body b
  int32_t d_body[44], d_nanc[44], d_madr[44], d_act[44];
  uint8_t d_anc[44][MAXANC];   // ancestors of dof k: parent, grand-parent, ...
  float d_axis[44][3], d_arm[44], d_damp[44], d_invw[44], d_floss[44], d_lo[44], d_hi[44], d_clo[44], d_chi[44];
  int32_t g_body[96], g_type[96], g_mesh[96], g_ci[96];   // g_ci: index into Lds::gmat, -1 for visual geoms
  float g_pos[96][3], g_mat[96][9], g_size[96][3], g_rbound[96], g_mu[96];
  float g_bc[96][3], g_bh[96][3];   // local bounding box of the geom (centre, half extents) for the OBB filter
  int16_t p_g1[1024], p_g2[1024];
  int32_t m_vadr[32], m_vnum[32], m_cadr[32], m_cnum[32];   // vertex range and cluster range of each mesh
  double m_center[32][3];
  uint8_t tri_a[128], tri_b[128];
  uint32_t f_tgt[44][64];   // factorisation step k, ancestor pairs p = lane (low half) and lane + 64 (high half): the qLD entry updated
  // fp64 copies of everything the pose chain of a body and the narrowphase read: the narrowphase is fp64, and so are its inputs
  // (a convex-convex contact normal depends discontinuously on the poses; fp32 poses moved it in ~3 % of thrashing env-steps)
  double timestep_d, qpos0_d[NQ], b_pos_d[40][3], b_quat_d[40][4], d_axis_d[44][3], g_pos_d[96][3], g_mat_d[96][9], g_size_d[96][3];
};

struct Lds {   // per-env working set (one wave): 19.2 KB, eight waves per CU
  float qpos[NQ], qvel[44], warm[44], ctrl[40];
  float xpos[40][3], xquat[40][4], xmat[40][9], xipos[40][3];
  float xpos_lo[40][3], xquat_lo[40][4];   // body pose in fp64 = (double)x + (double)x_lo: the fp32 consumers read the rounded value
  float x0q_lo[4];                         // the same for the normalised root quaternion the RK stages start from
  float xaxis[44][3];
  float gpos[96][3], gmat[NCG][9];   // rotation only of the geoms that collide (Dev::g_ci)
  float com[4];
  float cdof[44][6];
  float cvel[40][6];
  float qLD[NM + 2], dinv[44], dsq[44];
  float bias[44], fsm[44], qas[44], qacc[44], qfc[44], tmp[44];
  float x0q[NQ], x0v[44], accq[44], accv[44];
  int32_t info[8];   // ncon, nefc, nlimit, solver_iter, overflow, nsurv
#ifdef G1_PROFILE
  long long prof_t; unsigned prof[18];   // 16 used; 80 bytes keep the head of the working set a multiple of 16
#endif
  union {
    struct {   // smooth-dynamics scratch: dead once qacc_smooth is known
      float cinert[40][10], crb[40][10], cdofdot[44][6], cacc[40][6], cfrc[40][6];
    } sm;
    struct {   // contacts of the evaluation (kept for the observation) and the narrowphase staging (fp64, wave-uniform)
      float c_dist[MAXCON], c_pos[MAXCON][3], c_frame[MAXCON][9], c_mu[MAXCON];
      int32_t c_g1[MAXCON], c_g2[MAXCON];
      int16_t surv[MAXSURV];
      double geo[2][18];     // the two geoms of the pair being processed: pos 3 | mat 9 | size 3 | centre 3
      int32_t geoi[2][6];    // type, nvert, nclus, vertex start, cluster start, pad
      double rc[8][7];       // its contacts: dist, pos 3, normal 3
      double poly[2][16][3]; // box-box polygons
      double mpr_ps[4][9];   // MPR portal (v0..v3: v, v1, v2): wave-uniform state, 72 VGPRs if kept per lane
    } co;
    struct {   // low words of the RK stage's qpos: written by integrate_pos, read at the top of kinematics — between two evaluations,
      char skip[5680];   // when `sm` is dead; placed behind the end of `co` (the last contacts stay readable for the observation)
      float qlo[NQ];
    } rk;
  } u;
};

// The per-env working set is a file-scope __shared__ object: every phase is its own (out-of-line) function with its own
// register allocation, and all of them address it as LDS (ds_* instructions), not through generic pointers.
__shared__ Lds g_S;
#define S g_S
static_assert(sizeof(Lds) <= 20480, "eight waves per CU need <= 20 480 B of LDS per env");
static_assert(sizeof(((Lds *)0)->u.co) <= 5680 && sizeof(((Lds *)0)->u.rk) <= sizeof(((Lds *)0)->u.sm), "rk.qlo sits behind co, inside sm");

struct ClipDev {
  const float *rows;    // L x CLIP_ROW
  const float *reset;   // L x 88 : qpos 44 | qvel 43
  const float *com;     // L x 4 : mass-weighted body_xpos COM of the frame
  int32_t L, flags;
};

struct Launch {
  const Dev *T;
  const double *mesh_vert;   // hull vertices of all meshes, xyz, reordered into clusters of 64 (see g1_build_meshes)
  const int32_t *mesh_oidx;  // original index of each reordered vertex (tie-break of equal support values)
  const double *mesh_clus;   // per cluster: bounding-sphere centre xyz, radius
  float *state;              // N x STATE
  float *jt, *bt, *ar;       // per-env scratch: J^T [44][MAXROW], (D^-1/2 L^-T J^T) [44][MAXROW], A [MAXROW][MAXROW]
  float *rows;               // per-env scratch: R, aref -> b, force, friction-loss bound, meta (type | id << 2): [5][MAXROW]
  float *sepc;               // per-env separating-direction cache of the MPR pairs: [SEPC + 1][4]
  const int32_t *order;      // STEP: workgroup -> env, heaviest envs first (g1_schedule_kernel), or null
  int32_t *cost;             // STEP: per-env work estimate of this step, or null
  ClipDev clips[DMG1_MAX_CLIPS];   // DPEnv: the env's clip id picks one (multi-clip batches); DPCombinedEnv: 0..2 = walk, run, getup
  int32_t task, amnesty_steps, to_getup_len, pad2;
  int32_t N, mode, auto_reset, max_ep_length, run_forward, pad;
  float vel_obs_scale, high_z, obs_bound;
  uint64_t seed;
  const float *actions, *in_qpos, *in_qvel, *in_warm;
  const uint8_t *mask;
  const int32_t *idx_init;
  float *obs, *rew, *terms, *terminal_obs, *debug;
  uint8_t *done;
  int32_t *reason;
  // ---- split pipeline (g1_env_kernel / g1_pair_kernel): per-env slot of the LDS working set + step context between launches, and
  // the pair queue of the collision stage
  char *ws;                  // N x WS_BYTES
  double *pq_geo;            // [N][PAIRCAP][36]: the two staged geoms of a surviving pair (fp64)
  int32_t *pq_pair;          // [N][PAIRCAP]: pair id of the env's s-th survivor (canonical order)
  int32_t *pq_cnt;           // [N][PAIRCAP]: contacts the pair kernel found
  float *pq_con;             // [N][PAIRCAP][8][16]: dist | pos 3 | frame 9 of each
  int32_t *tick;             // [2][N * PAIRCAP]: tickets (env << 8 | s): list 0 = support-query pairs (MPR, plane-mesh), 1 = analytic
  int32_t *qctr;             // per round: tickets in list 0, in list 1, queue head, pad
  float *sepc2;              // [N][1024][4]: separating-direction cache, one entry per (env, pair)
  int32_t round, pad3;
};

// ------------------------------------------------------------------------------------------ small math (fp32)
// sum over the 64 lanes, uniform in every lane: four DPP steps inside each row of 16, the four row sums through v_readlane
// (~12 instructions; six LDS-permute round trips would sit in every dependent chain that reduces)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wsum(float v) {
  v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);   // row_half_mirror
  v += dpp_f32<0x140>(v);   // row_mirror
  return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
}
__device__ __forceinline__ void cross3(float *r, const float *a, const float *b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void quat_mul(float *r, const float *a, const float *b) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
__device__ __forceinline__ void quat2mat(float *m, const float *q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
__device__ __forceinline__ void quat_rot(float *r, const float *q, const float *v) {
  float m[9];
  quat2mat(m, q);
  float x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
        z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void quat_normalize(float *q) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVALF) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { float s = 1.f / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
__device__ __forceinline__ void mat_vec(float *r, const float *m, const float *v) {
  float x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
        z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void mul_inert_vec(float *r, const float *I, const float *v) {   // [EXT] mju_mulInertVec
  r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
  r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + I[8] * v[3] - I[6] * v[5];
  r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
  r[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
  r[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
  r[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}
__device__ __forceinline__ void cross_motion(float *r, const float *vel, const float *v) {   // [EXT] mju_crossMotion
  r[0] = -vel[2] * v[1] + vel[1] * v[2];
  r[1] = vel[2] * v[0] - vel[0] * v[2];
  r[2] = -vel[1] * v[0] + vel[0] * v[1];
  r[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  r[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  r[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
__device__ __forceinline__ void cross_force(float *r, const float *vel, const float *f) {   // [EXT] mju_crossForce
  r[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  r[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  r[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  r[3] = -vel[2] * f[4] + vel[1] * f[5];
  r[4] = vel[2] * f[3] - vel[0] * f[5];
  r[5] = -vel[1] * f[3] + vel[0] * f[4];
}
__device__ __forceinline__ void quat_to_rpy(const float *q, float *rpy) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  rpy[0] = atan2f(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
  float s = 2 * (w * y - z * x);
  rpy[1] = asinf(fminf(fmaxf(s, -1.f), 1.f));
  rpy[2] = atan2f(2 * (w * z + x * y), 1 - 2 * (y * y + z * z));
}
__device__ __host__ __forceinline__ uint32_t hash32(uint64_t seed, uint32_t env, uint32_t step, uint32_t j) {
  uint64_t x = seed ^ ((uint64_t)env * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)step * 0xBF58476D1CE4E5B9ull) ^
               ((uint64_t)j * 0x94D049BB133111EBull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

#define SYNC() __syncthreads()
#ifdef G1_PROFILE
#define PROF(k) do { const long long t_ = clock64(); if (threadIdx.x == 0) { S.prof[k] += (unsigned)(t_ - S.prof_t); } S.prof_t = t_; } while (0)
#else
#define PROF(k) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------ position stage
// fp64 helpers of the pose chain
__device__ __forceinline__ void dquat_mul(double *r, const double *a, const double *b) {
  const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  const double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
__device__ __forceinline__ void dquat2mat(double *m, const double *q) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
__device__ __forceinline__ void dquat_normalize(double *q) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
__device__ __forceinline__ void split_hi_lo(const double x, float &hi, float &lo) { hi = (float)x; lo = (float)(x - (double)hi); }

// [EXT] mj_kinematics.  The pose chain (xpos, xquat) runs in fp64 and is kept as (fp32 value, fp32 remainder): the narrowphase
// is fp64 and takes its geom poses from it (stage_geo); everything else reads the rounded fp32 arrays.  `qlo`: the evaluation
// belongs to RK stage 2..4, whose qpos carries low words (integrate_pos).
__device__ __noinline__ void kinematics(const Dev &T, const int lane, const bool qlo) {
  if (lane == 0) {
    S.xpos[0][0] = S.xpos[0][1] = S.xpos[0][2] = 0;
    S.xquat[0][0] = 1; S.xquat[0][1] = S.xquat[0][2] = S.xquat[0][3] = 0;
    for (int i = 0; i < 3; i++) S.xpos_lo[0][i] = 0;
    for (int i = 0; i < 4; i++) S.xquat_lo[0][i] = 0;
    for (int i = 0; i < 9; i++) S.xmat[0][i] = (i % 4 == 0) ? 1.f : 0.f;
    S.xipos[0][0] = S.xipos[0][1] = S.xipos[0][2] = 0;
  }
  // the body's own joint rotation, all bodies at once (one fp64 sincos per lane, outside the level loop)
  double ql[4] = {1, 0, 0, 0}, rq[4] = {1, 0, 0, 0}, rp[3] = {0, 0, 0};
  if (lane >= 2 && lane < NB) {
    const int k = T.b_dof[lane];   // the body's single hinge; joint anchor = body origin
    const double ang = ((double)S.qpos[k + 1] + (qlo ? (double)S.u.rk.qlo[k + 1] : 0.0)) - T.qpos0_d[k + 1];
    double sn, cs;
    sincos(0.5 * ang, &sn, &cs);
    ql[0] = cs; ql[1] = T.d_axis_d[k][0] * sn; ql[2] = T.d_axis_d[k][1] * sn; ql[3] = T.d_axis_d[k][2] * sn;
  } else if (lane == 1) {   // free root: MuJoCo normalises the stored quaternion in place
    for (int i = 0; i < 4; i++) rq[i] = (double)S.qpos[3 + i] + (qlo ? (double)S.u.rk.qlo[3 + i] : 0.0);
    for (int i = 0; i < 3; i++) rp[i] = (double)S.qpos[i] + (qlo ? (double)S.u.rk.qlo[i] : 0.0);
    dquat_normalize(rq);
  }
  SYNC();
  for (int L = 1; L <= T.maxdepth; L++) {
    if (lane >= 1 && lane < NB && T.b_depth[lane] == L) {
      const int b = lane, p = T.b_parent[b];
      double pos[3], q[4];
      if (b == 1) {
        for (int i = 0; i < 4; i++) {   // x0q_lo: low words of the quaternion the RK stages start from (stage 1 only)
          float hi, l;
          split_hi_lo(rq[i], hi, l);
          q[i] = rq[i]; S.qpos[3 + i] = hi;
          if (!qlo) S.x0q_lo[i] = l;
        }
        for (int i = 0; i < 3; i++) pos[i] = rp[i];
      } else {
        double pq[4], pm[9];
        for (int i = 0; i < 4; i++) pq[i] = (double)S.xquat[p][i] + (double)S.xquat_lo[p][i];
        dquat2mat(pm, pq);
        for (int i = 0; i < 3; i++)
          pos[i] = ((double)S.xpos[p][i] + (double)S.xpos_lo[p][i]) +
                   (pm[3 * i] * T.b_pos_d[b][0] + pm[3 * i + 1] * T.b_pos_d[b][1] + pm[3 * i + 2] * T.b_pos_d[b][2]);
        dquat_mul(q, pq, T.b_quat_d[b]);
        const int k = T.b_dof[b];
        double qm[9];
        dquat2mat(qm, q);
        for (int i = 0; i < 3; i++)
          S.xaxis[k][i] = (float)(qm[3 * i] * T.d_axis_d[k][0] + qm[3 * i + 1] * T.d_axis_d[k][1] + qm[3 * i + 2] * T.d_axis_d[k][2]);
        double qn[4];
        dquat_mul(qn, q, ql);
        for (int i = 0; i < 4; i++) q[i] = qn[i];
      }
      dquat_normalize(q);
      double m[9];
      dquat2mat(m, q);
      for (int i = 0; i < 3; i++) split_hi_lo(pos[i], S.xpos[b][i], S.xpos_lo[b][i]);
      for (int i = 0; i < 4; i++) split_hi_lo(q[i], S.xquat[b][i], S.xquat_lo[b][i]);
      for (int i = 0; i < 9; i++) S.xmat[b][i] = (float)m[i];
      for (int i = 0; i < 3; i++)
        S.xipos[b][i] = (float)(pos[i] + ((double)T.b_ipos[b][0] * m[3 * i] + (double)T.b_ipos[b][1] * m[3 * i + 1] + (double)T.b_ipos[b][2] * m[3 * i + 2]));
    }
    SYNC();
  }
  for (int g = lane; g < NG; g += 64) {
    const int b = T.g_body[g];
    float gp[3] = {T.g_pos[g][0], T.g_pos[g][1], T.g_pos[g][2]}, t[3];
    mat_vec(t, S.xmat[b], gp);
    for (int i = 0; i < 3; i++) S.gpos[g][i] = S.xpos[b][i] + t[i];
    const int ci = T.g_ci[g];
    if (ci >= 0)
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        S.gmat[ci][3 * i + j] = S.xmat[b][3 * i] * T.g_mat[g][j] + S.xmat[b][3 * i + 1] * T.g_mat[g][3 + j] +
                               S.xmat[b][3 * i + 2] * T.g_mat[g][6 + j];
  }
  SYNC();
}

__device__ __noinline__ void com_pos(const Dev &T, const int lane) {   // [EXT] mj_comPos
  float m = (lane >= 1 && lane < NB) ? T.b_mass[lane] : 0.f, c[3];
  for (int i = 0; i < 3; i++) c[i] = wsum(m * ((lane < NB) ? S.xipos[lane][i] : 0.f)) * T.total_mass_inv;
  if (lane == 0) for (int i = 0; i < 3; i++) S.com[i] = c[i];
  if (lane < NB) {
    const int b = lane;
    float *ci = S.u.sm.cinert[b];
    if (b == 0) { for (int i = 0; i < 10; i++) ci[i] = 0; }
    else {
      const float *I = T.b_inertia[b], *R = S.xmat[b];
      float Ib[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, Tm[9], W[9];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Tm[3 * i + j] = R[3 * i] * Ib[j] + R[3 * i + 1] * Ib[3 + j] + R[3 * i + 2] * Ib[6 + j];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) W[3 * i + j] = Tm[3 * i] * R[3 * j] + Tm[3 * i + 1] * R[3 * j + 1] + Tm[3 * i + 2] * R[3 * j + 2];
      float o[3] = {S.xipos[b][0] - c[0], S.xipos[b][1] - c[1], S.xipos[b][2] - c[2]}, mass = T.b_mass[b];
      float oo = dot3(o, o);
      ci[0] = W[0] + mass * (oo - o[0] * o[0]); ci[1] = W[4] + mass * (oo - o[1] * o[1]); ci[2] = W[8] + mass * (oo - o[2] * o[2]);
      ci[3] = W[1] - mass * o[0] * o[1]; ci[4] = W[2] - mass * o[0] * o[2]; ci[5] = W[5] - mass * o[1] * o[2];
      ci[6] = mass * o[0]; ci[7] = mass * o[1]; ci[8] = mass * o[2]; ci[9] = mass;
    }
  }
  if (lane < NV) {   // cdof: (angular, linear) about the whole-body COM
    const int k = lane;
    float *cd = S.cdof[k];
    if (k < 3) { for (int i = 0; i < 6; i++) cd[i] = 0; cd[3 + k] = 1; }
    else {
      float ax[3], off[3];
      const int b = T.d_body[k];
      if (k < 6) { ax[0] = S.xmat[1][k - 3]; ax[1] = S.xmat[1][3 + k - 3]; ax[2] = S.xmat[1][6 + k - 3]; }
      else { ax[0] = S.xaxis[k][0]; ax[1] = S.xaxis[k][1]; ax[2] = S.xaxis[k][2]; }
      for (int i = 0; i < 3; i++) off[i] = c[i] - S.xpos[b][i];
      cd[0] = ax[0]; cd[1] = ax[1]; cd[2] = ax[2];
      cross3(cd + 3, ax, off);
    }
  }
  SYNC();
}

__device__ __forceinline__ float bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// x <- L^-T x for one constraint row held in registers: the dof tree is compile-time (dm_g1_topology.h), so every index is
// static; the factor entries are wave-uniform LDS reads
template <int I, int J, int OFF>
struct RowAnc {
  static __device__ __forceinline__ void run(float (&x)[NV], const float xi, const float *L) {
    if constexpr (J >= 0) {
      x[J] = fmaf(-L[g1topo::MADR[I] + OFF], xi, x[J]);
      RowAnc<I, (J >= 0 ? g1topo::PARENT[J >= 0 ? J : 0] : -1), OFF + 1>::run(x, xi, L);
    }
  }
};
template <int I>
struct RowSolve {
  static __device__ __forceinline__ void run(float (&x)[NV], const float *L) {
    RowAnc<I, g1topo::PARENT[I], 1>::run(x, x[I], L);
    if constexpr (I > 0) RowSolve<I - 1>::run(x, L);
  }
};
// compile-time loop with early exit: f(integral_constant<int, I>) returns false to stop
template <int I, int N>
struct StaticFor {
  template <class F>
  static __device__ __forceinline__ void run(F &&f) {
    if constexpr (I < N) {
      if (f(std::integral_constant<int, I>{})) StaticFor<I + 1, N>::run(f);
    }
  }
};

__device__ __noinline__ void crb_factor(const Dev &T, const int lane) {   // [EXT] mj_crb + mj_factorM
  if (lane < NB) {
    float acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t mk = T.b_desc[lane];
    if (lane == 0) mk = 0;
    while (mk) {
      const int d = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      for (int i = 0; i < 10; i++) acc[i] += S.u.sm.cinert[d][i];
    }
    for (int i = 0; i < 10; i++) S.u.sm.crb[lane][i] = acc[i];
  }
  SYNC();
  if (lane < NV) {
    const int i = lane;
    float buf[6], cd[6];
    for (int q = 0; q < 6; q++) cd[q] = S.cdof[i][q];
    mul_inert_vec(buf, S.u.sm.crb[T.d_body[i]], cd);
    int adr = T.d_madr[i];
    float v = T.d_arm[i];
    for (int q = 0; q < 6; q++) v += cd[q] * buf[q];
    S.qLD[adr] = v;
    const int n = T.d_nanc[i];
    const uint4 aw = *reinterpret_cast<const uint4 *>(T.d_anc[i]);   // the whole ancestor list in one load (a loop over T.d_anc[i][a]
    const uint32_t awv[4] = {aw.x, aw.y, aw.z, aw.w};                  // waits for one global load per ancestor)
#pragma unroll
    for (int a = 0; a < 14; a++) {
      if (a < n) {
        const int j = (awv[a >> 2] >> (8 * (a & 3))) & 255;
        float s = 0;
        for (int q = 0; q < 6; q++) s += S.cdof[j][q] * buf[q];
        S.qLD[adr + 1 + a] = s;
      }
    }
  }
  SYNC();
  // L^T D L, rows of dof k: [k, parent(k), grand-parent, ...]; step k updates the rows of its ancestors.  The dof tree is
  // compile-time (dm_g1_topology.h): per step the only run-time table is the target entry of the lane's ancestor pair, and
  // those (43 steps x up to 105 pairs) are requested in one batch before the first step — a loop over T.d_anc / T.d_madr was
  // three dependent global loads per step.
  uint32_t tgt[NV];
#pragma unroll
  for (int k = 1; k < NV; k++) tgt[k] = T.f_tgt[k][lane];
  const int ta0 = T.tri_a[lane], tb0 = T.tri_b[lane], ta1 = T.tri_a[lane + 64], tb1 = T.tri_b[lane + 64];
  StaticFor<0, NV - 1>::run([&](auto kc) {
    constexpr int k = NV - 1 - decltype(kc)::value;     // NV-1 .. 1
    constexpr int n = g1topo::NANC[k], kk = g1topo::MADR[k], np = n * (n + 1) / 2;
    const float dk = S.qLD[kk];
    if (lane < np) S.qLD[tgt[k] & 0xffff] -= S.qLD[kk + 1 + tb0] * (S.qLD[kk + 1 + ta0] / dk);
    if constexpr (np > 64) {
      if (lane + 64 < np) S.qLD[tgt[k] >> 16] -= S.qLD[kk + 1 + tb1] * (S.qLD[kk + 1 + ta1] / dk);
    }
    SYNC();
    if (lane < n) S.qLD[kk + 1 + lane] = S.qLD[kk + 1 + lane] / dk;
    SYNC();
    return true;
  });
  if (lane < NV) { const float d = S.qLD[T.d_madr[lane]]; S.dinv[lane] = 1.f / d; S.dsq[lane] = 1.f / sqrtf(d); }
  SYNC();
}

// x <- M^-1 x for an LDS vector, M = L^T D L on the compile-time dof tree: lane j keeps x_j in a register.  L^-T: dofs in
// descending order, x_i broadcast (v_readlane, static lane) to its ancestors — the ancestors of a dof are a chain, so the factor
// entry lane j needs is qLD[MADR[i] + NANC[i] - nanc_j] and "is an ancestor" is a compile-time lane mask.  L^-1: dofs in
// ascending order, the final x_j broadcast to its descendants (column-oriented, no reduction).  ~450 instructions with
// two-instruction dependent steps; the tree walk through T.d_nanc / T.d_anc it replaces (a table load, LDS read-modify-writes
// and a wave reduction per dof) took ~100 k cycles per call, eight calls per env-step.  (A static solve with the whole vector in
// 43 registers of every lane, as the constraint rows use, was slower still: +1 ms per launch.)
constexpr uint64_t g1_anc_mask(int i) {
  uint64_t m = 0;
  for (int j = g1topo::PARENT[i]; j >= 0; j = g1topo::PARENT[j]) m |= 1ull << j;
  return m;
}
constexpr uint64_t g1_desc_mask(int j) {
  uint64_t m = 0;
  for (int i = 0; i < g1topo::NV; i++) if ((g1_anc_mask(i) >> j) & 1) m |= 1ull << i;
  return m;
}
__device__ __noinline__ void solve_m(const Dev &T, float *xl, const int lane) {
  const int lk = lane < NV ? lane : 0;
  const int nl = T.d_nanc[lk], ml = T.d_madr[lk] + nl;
  float x = xl[lk];
  StaticFor<0, NV - 1>::run([&](auto ic) {
    constexpr int i = NV - 1 - decltype(ic)::value;            // NV-1 .. 1
    constexpr uint64_t mask = g1_anc_mask(i);
    constexpr int base = g1topo::MADR[i] + g1topo::NANC[i];
    const float xi = bcast(x, i);
    if ((mask >> lane) & 1) x = fmaf(-S.qLD[base - nl], xi, x);
    return true;
  });
  x *= S.dinv[lk];
  StaticFor<0, NV - 1>::run([&](auto jc) {
    constexpr int j = decltype(jc)::value;                     // 0 .. NV-2
    constexpr uint64_t mask = g1_desc_mask(j);
    if constexpr (mask != 0) {
      const float xj = bcast(x, j);
      if ((mask >> lane) & 1) x = fmaf(-S.qLD[ml - g1topo::NANC[j]], xj, x);
    }
    return true;
  });
  SYNC();
  if (lane < NV) xl[lane] = x;
  SYNC();
}

// ------------------------------------------------------------------------------------------ velocity stage
__device__ __noinline__ void fwd_smooth(const Dev &T, const int lane) {   // mj_comVel, mj_passive, mj_rne, mj_fwdActuation
  if (lane == 0) for (int i = 0; i < 6; i++) { S.cvel[0][i] = 0; S.u.sm.cacc[0][i] = (i == 5) ? -T.gravity[2] : 0.f; }
  SYNC();
  for (int L = 1; L <= T.maxdepth; L++) {
    if (lane >= 1 && lane < NB && T.b_depth[lane] == L) {
      const int b = lane, p = T.b_parent[b];
      float cv[6], ca[6];
      for (int i = 0; i < 6; i++) { cv[i] = S.cvel[p][i]; ca[i] = S.u.sm.cacc[p][i]; }
      if (b == 1) {
        for (int k = 0; k < 3; k++) {
          for (int i = 0; i < 6; i++) S.u.sm.cdofdot[k][i] = 0;
          for (int i = 0; i < 6; i++) cv[i] += S.cdof[k][i] * S.qvel[k];
        }
        for (int k = 3; k < 6; k++) {
          float dd[6], cd[6];
          for (int i = 0; i < 6; i++) cd[i] = S.cdof[k][i];
          cross_motion(dd, cv, cd);
          for (int i = 0; i < 6; i++) { S.u.sm.cdofdot[k][i] = dd[i]; ca[i] += dd[i] * S.qvel[k]; }
        }
        for (int k = 3; k < 6; k++)
          for (int i = 0; i < 6; i++) cv[i] += S.cdof[k][i] * S.qvel[k];
      } else {
        const int k = T.b_dof[b];
        float dd[6], cd[6];
        for (int i = 0; i < 6; i++) cd[i] = S.cdof[k][i];
        cross_motion(dd, cv, cd);
        const float qv = S.qvel[k];
        for (int i = 0; i < 6; i++) { S.u.sm.cdofdot[k][i] = dd[i]; ca[i] += dd[i] * qv; cv[i] += cd[i] * qv; }
      }
      float t[6], t1[6], f[6];
      mul_inert_vec(f, S.u.sm.cinert[b], ca);
      mul_inert_vec(t, S.u.sm.cinert[b], cv);
      cross_force(t1, cv, t);
      for (int i = 0; i < 6; i++) { S.cvel[b][i] = cv[i]; S.u.sm.cacc[b][i] = ca[i]; S.u.sm.cfrc[b][i] = f[i] + t1[i]; }
    }
    SYNC();
  }
  // subtree sums of cfrc into crb[b][0..5] (crb is dead after the factorisation)
  if (lane >= 1 && lane < NB) {
    float acc[6] = {0, 0, 0, 0, 0, 0};
    uint64_t mk = T.b_desc[lane];
    while (mk) {
      const int d = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      for (int i = 0; i < 6; i++) acc[i] += S.u.sm.cfrc[d][i];
    }
    for (int i = 0; i < 6; i++) S.u.sm.crb[lane][i] = acc[i];
  }
  SYNC();
  if (lane < NV) {
    const int k = lane;
    float bias = 0;
    for (int i = 0; i < 6; i++) bias += S.cdof[k][i] * S.u.sm.crb[T.d_body[k]][i];
    const float passive = -T.d_damp[k] * S.qvel[k];
    float act = 0;
    const int a = T.d_act[k];
    if (a >= 0) act = fminf(fmaxf(S.ctrl[a], T.d_clo[k]), T.d_chi[k]);   // gear 1
    const float fs = passive - bias + act;
    S.bias[k] = bias; S.fsm[k] = fs; S.qas[k] = fs;
  }
  SYNC();
  solve_m(T, S.qas, lane);
  SYNC();
}

// ------------------------------------------------------------------------------------------ narrowphase (fp64, wave-uniform)
// one contact of the pair being processed, and the view of one of its two geoms — both live in LDS (Lds::rc, Lds::geo):
// the narrowphase is wave-uniform scalar code, and per-lane copies of these would sit in scratch memory
struct Con { double dist, pos[3], n[3]; };
struct Geo {
  int type, nvert, nclus;
  const double *pos, *mat, *size, *center;   // into Lds::geo
  const double *vert, *clus;                 // global: clustered hull vertices, cluster bounds
  const int32_t *oidx;
};
__device__ __forceinline__ double ddot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void dcross(double *r, const double *a, const double *b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ double dnorm(const double *a) { return sqrt(ddot(a, a)); }
__device__ __forceinline__ double dnormalize(double *a) {
  double n = dnorm(a);
  if (n < MINVAL) { a[0] = 1; a[1] = a[2] = 0; return n; }   // [EXT] mju_normalize3
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
__device__ __forceinline__ void drot(double *r, const double *m, const double *v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
         z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void drot_t(double *r, const double *m, const double *v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
         z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ double dclamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ void dsub(double *r, const double *a, const double *b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }

// ---- support mapping of a hull (see mesh_support_pair below)
// wave maximum of a double, uniform in every lane: four DPP steps inside each row of 16 lanes (both dwords moved with the same
// control), then the four row maxima through v_readlane — ~25 instructions instead of six LDS-permute round trips
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wmax_f64(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));    // quad_perm [1,0,3,2]
  v = fmax(v, dpp_f64<0x4E>(v));    // quad_perm [2,3,0,1]
  v = fmax(v, dpp_f64<0x141>(v));   // row_half_mirror
  v = fmax(v, dpp_f64<0x140>(v));   // row_mirror
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
constexpr int NCH = 2;            // cluster chunks of 64 per hull: hulls of up to 128 clusters (checked at create)
// Support ties (oracle/dm_convex.h "ties"): MPR queries a hull along its portal normal, in which two hull vertices tie EXACTLY by
// construction at every edge-edge / face-vertex contact; "first strict maximum" is then decided by the last bit of two dot
// products and ends in a different contact normal.  Rule (both sides): values within SUP_TIE of the maximum are tied, the
// lowest ORIGINAL vertex index wins.  A lane keeps its best value and the lowest-index vertex within SUP_TIE of it.
constexpr double SUP_TIE = 1e-12;
struct MeshPick { double best, cv, x, y, z; int bo, bk; };   // best value | candidate: value, vertex, original index, slot
// every lane scans vertex `lane` of up to four clusters of hull A and four of hull B (-1: none): sixteen independent loads in
// one round trip, no cross-lane step; the pick keeps the vertex itself, so no fetch follows the reduction
struct Scan4 { double x[4], y[4], z[4]; int o[4]; };
__device__ __forceinline__ void scan4_load(const Geo &g, const int (&c)[4], Scan4 &v, const int lane) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int k = 64 * (c[q] < 0 ? 0 : c[q]) + lane;
    const bool ok = c[q] >= 0;
    v.x[q] = ok ? g.vert[3 * k] : 0.0; v.y[q] = ok ? g.vert[3 * k + 1] : 0.0; v.z[q] = ok ? g.vert[3 * k + 2] : 0.0;
    v.o[q] = ok ? g.oidx[k] : 0x7fffffff;
  }
}
__device__ __forceinline__ void scan4_pick(const double *dl, const int (&c)[4], const Scan4 &v, MeshPick &p, const int lane) {
#pragma unroll
  for (int q = 0; q < 4; q++)
    if (v.o[q] != 0x7fffffff) {
      const double sv = v.x[q] * dl[0] + v.y[q] * dl[1] + v.z[q] * dl[2];
      const bool up = sv > p.best;
      // new candidate: a new best whose window the old candidate left (or that has the lower index), or a lower index inside the window
      if (up ? (!(p.cv >= sv - SUP_TIE) || v.o[q] < p.bo) : (sv >= p.best - SUP_TIE && v.o[q] < p.bo)) {
        p.cv = sv; p.bo = v.o[q]; p.bk = 64 * c[q] + lane;
        p.x = v.x[q]; p.y = v.y[q]; p.z = v.z[q];
      }
      if (up) p.best = sv;
    }
}
// the lanes' picks -> the wave's pick: the maximum value, and among equal values the lowest original index (one lane in all
// but degenerate cases: a ballot and v_readlanes; ties walk the tied lanes)
__device__ __forceinline__ void wave_pick(MeshPick &p) {
  const double vmax = wmax_f64(p.best);
  unsigned long long eq = __ballot(p.cv >= vmax - SUP_TIE);   // never empty: the lane that holds the maximum has best - SUP_TIE <= cv <= best
  int l = __ffsll((long long)eq) - 1;
  int bo = __builtin_amdgcn_readlane(p.bo, l);
  eq &= eq - 1;
  while (eq) {
    const int l2 = __ffsll((long long)eq) - 1;
    eq &= eq - 1;
    const int o2 = __builtin_amdgcn_readlane(p.bo, l2);
    if (o2 < bo) { bo = o2; l = l2; }
  }
  p.best = vmax; p.bo = bo; p.bk = __builtin_amdgcn_readlane(p.bk, l);
  p.x = readlane_f64(p.x, l); p.y = readlane_f64(p.y, l); p.z = readlane_f64(p.z, l);
}
// the next (up to) four candidate clusters of a hull: bit sets of clusters lane + 64 m, consumed in ascending order
__device__ __forceinline__ bool next4(unsigned long long (&todo)[NCH], int (&c)[4]) {
  bool any = false;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    c[q] = -1;
#pragma unroll
    for (int m = 0; m < NCH; m++)
      if (c[q] < 0 && todo[m]) { c[q] = __ffsll((long long)todo[m]) - 1 + 64 * m; todo[m] &= todo[m] - 1; any = true; }
  }
  return any;
}
// cluster spheres of one hull for clusters lane + 64 m (loads only), then bounds ub[m] and the cluster with the largest centre . d
struct ClusLoad { double c[NCH][7]; };   // centre 3 | sphere radius | half extents of the box about the centre 3
__device__ __forceinline__ void cluster_load(const Geo &g, const bool on, ClusLoad &L, const int lane) {
#pragma unroll
  for (int m = 0; m < NCH; m++) {
    const int c = lane + 64 * m;
    const bool ok = on && c < g.nclus;
    const double *cl = g.clus + 8 * (ok ? c : 0);
#pragma unroll
    for (int i = 0; i < 7; i++) L.c[m][i] = ok ? cl[i] : 0.0;
  }
}
__device__ __forceinline__ void cluster_bounds(const Geo &g, const double *dl, const bool on, const ClusLoad &L, double (&ub)[NCH], int &top,
                                               const int lane) {
  if (!on) { for (int m = 0; m < NCH; m++) ub[m] = -1e300; top = 0; return; }   // (wave-uniform) not a mesh: nothing to reduce
  const double dn = sqrt(dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2]);
  double best = -1e300;
  top = 0x7fffffff;
#pragma unroll
  for (int m = 0; m < NCH; m++) {
    const int c = lane + 64 * m;
    ub[m] = -1e300;
    if (on && c < g.nclus) {
      const double dc = L.c[m][0] * dl[0] + L.c[m][1] * dl[1] + L.c[m][2] * dl[2];
      // the tighter of two bounds: the enclosing sphere, and the box about the same centre (flat patches of the hull surface)
      ub[m] = dc + fmin(L.c[m][3] * dn, L.c[m][4] * fabs(dl[0]) + L.c[m][5] * fabs(dl[1]) + L.c[m][6] * fabs(dl[2]));
      if (dc > best) { best = dc; top = c; }
    }
  }
  const double vmax = wmax_f64(best);
  unsigned long long eq = __ballot(best == vmax);
  int l = __ffsll((long long)eq) - 1, t = __builtin_amdgcn_readlane(top, l);
  eq &= eq - 1;
  while (eq) {
    l = __ffsll((long long)eq) - 1;
    eq &= eq - 1;
    const int t2 = __builtin_amdgcn_readlane(top, l);
    if (t2 < t) t = t2;
  }
  top = t;
}
// Support vertices of up to two hulls at once, in local directions dlA / dlB.  The vertices are stored in compact clusters
// (k-d leaves, padded to 64 slots) with a bounding sphere and box each.  Three dependent memory round trips for BOTH hulls together
// (the narrowphase is one wave's latency chain: round trips are what it costs; every load of a round is requested before the
// first reduction of that round): (1) every cluster's centre . d and bound centre . d + min(radius |d|, box term); (2) the cluster with
// the largest centre . d is scanned: a true support value to prune with; (3) the clusters whose bound still reaches it are
// scanned, four per trip, every lane keeping its own best and the vertex it belongs to, one wave reduction at the
// end.  Exact; equal support values resolve to the lowest ORIGINAL vertex index, like a serial first-maximum scan.
__device__ __forceinline__ void mesh_support_pair(const Geo &A, const double *dlA, const bool meshA, MeshPick &pa, const Geo &B,
                                                  const double *dlB, const bool meshB, MeshPick &pb, const int lane) {
  double ubA[NCH], ubB[NCH];
  int topA, topB;
  { ClusLoad LA, LB;
    cluster_load(A, meshA, LA, lane);
    cluster_load(B, meshB, LB, lane);
    cluster_bounds(A, dlA, meshA, LA, ubA, topA, lane);
    cluster_bounds(B, dlB, meshB, LB, ubB, topB, lane); }
  pa = {-1e300, -1e300, 0.0, 0.0, 0.0, 0x7fffffff, 0};
  pb = pa;
  { const int ca[4] = {meshA ? topA : -1, -1, -1, -1}, cb[4] = {meshB ? topB : -1, -1, -1, -1};
    Scan4 va, vb;
    scan4_load(A, ca, va, lane);
    scan4_load(B, cb, vb, lane);
    scan4_pick(dlA, ca, va, pa, lane);
    scan4_pick(dlB, cb, vb, pb, lane); }
  const double ba = meshA ? wmax_f64(pa.best) : 0.0, bb = meshB ? wmax_f64(pb.best) : 0.0;
  unsigned long long todoA[NCH], todoB[NCH];
#pragma unroll
  for (int m = 0; m < NCH; m++) {
    todoA[m] = meshA ? __ballot(ubA[m] >= ba - SUP_TIE && (lane + 64 * m) != topA) : 0ull;
    todoB[m] = meshB ? __ballot(ubB[m] >= bb - SUP_TIE && (lane + 64 * m) != topB) : 0ull;
  }
  // (both hulls' candidates in ONE loop — eight clusters per trip — measured slower: 34 more live doubles spill)
  for (;;) {
    int ca[4];
    if (!next4(todoA, ca)) break;
    Scan4 va;
    scan4_load(A, ca, va, lane);
    scan4_pick(dlA, ca, va, pa, lane);
  }
  for (;;) {
    int cb[4];
    if (!next4(todoB, cb)) break;
    Scan4 vb;
    scan4_load(B, cb, vb, lane);
    scan4_pick(dlB, cb, vb, pb, lane);
  }
  if (meshA) wave_pick(pa);
  if (meshB) wave_pick(pb);
}
// slot of the support vertex in the clustered array, and the vertex itself
__device__ __forceinline__ int mesh_support_index(const Geo &g, const double *dl, double *v, const int lane) {
  MeshPick i, j;
  mesh_support_pair(g, dl, true, i, g, dl, false, j, lane);
  v[0] = i.x; v[1] = i.y; v[2] = i.z;
  return i.bk;
}

// [EXT] mjccd_support for the analytic shapes (local direction dl -> local point p)
__device__ __forceinline__ void support_local(const Geo &g, const double *dl, double *p) {
  p[0] = p[1] = p[2] = 0;
  if (g.type == DM_GEOM_SPHERE) {
    const double n = dnorm(dl);
    if (n > 0) for (int i = 0; i < 3; i++) p[i] = dl[i] * g.size[0] / n;
  } else if (g.type == DM_GEOM_CYLINDER) {
    const double n = sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
    if (n > MINVAL) { p[0] = dl[0] * g.size[0] / n; p[1] = dl[1] * g.size[0] / n; }
    p[2] = dl[2] * g.size[1] >= -0.5 * SUP_TIE ? g.size[1] : -g.size[1];
  } else if (g.type == DM_GEOM_BOX) {
    for (int i = 0; i < 3; i++) p[i] = dl[i] * g.size[i] >= -0.5 * SUP_TIE ? g.size[i] : -g.size[i];
  }
}

// ---- libccd MPR (ccdMPRPenetration), restated; tolerance / iteration cap = MuJoCo's mpr_tolerance / mpr_iterations
#define CCD_EPS 2.220446049250313e-16
#define MPR_TOL 1e-6
#define MPR_ITER 50
struct Sup { double v[3], v1[3], v2[3]; };
__device__ __forceinline__ bool ccd_zero(double x) { return fabs(x) < CCD_EPS; }
__device__ __forceinline__ bool ccd_eq(double a, double b) {
  double ab = fabs(a - b);
  if (ab < CCD_EPS) return true;
  a = fabs(a); b = fabs(b);
  return b > a ? ab < CCD_EPS * b : ab < CCD_EPS * a;
}
#ifdef G1_PAIRSTATS
__device__ unsigned long long g_pairstats[8];   // 0 support evaluations, 1 tickets, 2 hint-separated, 3 contacts, 4 MPR misses
#define PSTAT(k) do { if (lane == 0) atomicAdd(&g_pairstats[k], 1ull); } while (0)
#else
#define PSTAT(k) do {} while (0)
#endif
// support point of A - B in direction dir: both supports together, so the two hull scans share their memory round trips
__device__ __forceinline__ void mpr_support(const Geo &a, const Geo &b, const double *dir, Sup &s, const int lane) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]}, dla[3], dlb[3], pa[3], pb[3];
  drot_t(dla, a.mat, dir);
  drot_t(dlb, b.mat, nd);
  const bool ma = a.type == DM_GEOM_MESH, mb = b.type == DM_GEOM_MESH;
  PSTAT(0);

  support_local(a, dla, pa);
  support_local(b, dlb, pb);
  if (ma || mb) {
    MeshPick ka, kb;
#ifdef G1_PROFILE
    const long long t0_ = clock64();
#endif
    mesh_support_pair(a, dla, ma, ka, b, dlb, mb, kb, lane);
#ifdef G1_PROFILE
    if (lane == 0) { S.prof[13] += (unsigned)(clock64() - t0_); S.prof[12] += 1; }
#endif
    if (ma) { pa[0] = ka.x; pa[1] = ka.y; pa[2] = ka.z; }
    if (mb) { pb[0] = kb.x; pb[1] = kb.y; pb[2] = kb.z; }
  }
  drot(s.v1, a.mat, pa);
  drot(s.v2, b.mat, pb);
  for (int i = 0; i < 3; i++) { s.v1[i] += a.pos[i]; s.v2[i] += b.pos[i]; s.v[i] = s.v1[i] - s.v2[i]; }
}
__device__ __forceinline__ void portal_dir(const Sup *ps, double *dir) {
  double a[3], b[3];
  dsub(a, ps[2].v, ps[1].v);
  dsub(b, ps[3].v, ps[1].v);
  dcross(dir, a, b);
  dnormalize(dir);
}
__device__ __forceinline__ bool reach_tol(const Sup *ps, const Sup &v4, const double *dir) {
  const double dv4 = ddot(v4.v, dir);
  double m = fmin(fmin(dv4 - ddot(ps[1].v, dir), dv4 - ddot(ps[2].v, dir)), dv4 - ddot(ps[3].v, dir));
  return ccd_eq(m, MPR_TOL) || m < MPR_TOL;
}
__device__ __forceinline__ void expand_portal(Sup *ps, const Sup &v4) {
  double c[3];
  dcross(c, v4.v, ps[0].v);
  if (ddot(ps[1].v, c) > 0) { if (ddot(ps[2].v, c) > 0) ps[1] = v4; else ps[3] = v4; }
  else { if (ddot(ps[3].v, c) > 0) ps[2] = v4; else ps[1] = v4; }
}
__device__ __forceinline__ double tri_closest_origin(const double *a, const double *b, const double *c, double *w) {
  double ab[3], ac[3], ap[3] = {-a[0], -a[1], -a[2]};
  dsub(ab, b, a); dsub(ac, c, a);
  const double d1 = ddot(ab, ap), d2 = ddot(ac, ap);
  if (d1 <= 0 && d2 <= 0) { for (int i = 0; i < 3; i++) w[i] = a[i]; return ddot(w, w); }
  double bp[3] = {-b[0], -b[1], -b[2]};
  const double d3 = ddot(ab, bp), d4 = ddot(ac, bp);
  if (d3 >= 0 && d4 <= d3) { for (int i = 0; i < 3; i++) w[i] = b[i]; return ddot(w, w); }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) {
    const double v = d1 / (d1 - d3);
    for (int i = 0; i < 3; i++) w[i] = a[i] + v * ab[i];
    return ddot(w, w);
  }
  double cp[3] = {-c[0], -c[1], -c[2]};
  const double d5 = ddot(ab, cp), d6 = ddot(ac, cp);
  if (d6 >= 0 && d5 <= d6) { for (int i = 0; i < 3; i++) w[i] = c[i]; return ddot(w, w); }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) {
    const double v = d2 / (d2 - d6);
    for (int i = 0; i < 3; i++) w[i] = a[i] + v * ac[i];
    return ddot(w, w);
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    const double v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for (int i = 0; i < 3; i++) w[i] = b[i] + v * (c[i] - b[i]);
    return ddot(w, w);
  }
  const double den = 1.0 / (va + vb + vc), v = vb * den, u = vc * den;
  for (int i = 0; i < 3; i++) w[i] = a[i] + ab[i] * v + ac[i] * u;
  return ddot(w, w);
}
// 0 = penetration (depth, dir a -> b, pos), -1 = none
// The routine is a state machine around ONE support evaluation site (discoverPortal's vertices 1, 2, 3.., refinePortal,
// findPenetr): the support mapping of two hulls is the bulk of the code, and one inlined copy costs neither the code size
// of five nor a function call (whose callee-saved registers would go through scratch memory a hundred times per env-step).
// The arithmetic and its order are those of the straight-line libccd routine.
// `hint` (or null): a direction that separated the pair at an earlier evaluation — tested first with one support evaluation;
// returns -2 if it still separates them (disjoint shapes: MPR would find no contact either, so the shortcut is result-neutral).
// On a `no intersection` verdict reached through a strictly negative support projection, `sep` receives that direction.
__device__ __forceinline__ int mpr_penetration(const Geo &a, const Geo &b, double *depth, double *dir, double *pos, const double *hint,
                                               double *sep, bool &sep_valid, Sup *ps, const int lane) {
  enum { V1, V2, DISCOVER, REFINE, PENETR, HINT };
  constexpr double SEP_EPS = 1e-9;
  sep_valid = false;
  // ps: the portal (v0..v3: v, v1, v2), wave-uniform state in LDS (every lane writes the same values; 72 VGPRs if kept per lane)
  Sup sv;
  double d[3], va[3], vb[3], dot;
  for (int i = 0; i < 3; i++) { ps[0].v1[i] = a.center[i]; ps[0].v2[i] = b.center[i]; ps[0].v[i] = a.center[i] - b.center[i]; }
  if (ccd_zero(ps[0].v[0]) && ccd_zero(ps[0].v[1]) && ccd_zero(ps[0].v[2])) ps[0].v[0] += CCD_EPS * 10;
  for (int i = 0; i < 3; i++) d[i] = -ps[0].v[i];
  dnormalize(d);
  int state = V1, guard = 0, it = 0;
  if (hint) { state = HINT; for (int i = 0; i < 3; i++) d[i] = hint[i]; }
#define MPR_MISS(DOT) do { if ((DOT) < -SEP_EPS) { sep[0] = d[0]; sep[1] = d[1]; sep[2] = d[2]; sep_valid = true; } return -1; } while (0)
  for (;;) {
    mpr_support(a, b, d, sv, lane);
    if (state == HINT) {
      if (ddot(sv.v, d) < -SEP_EPS) return -2;
      for (int i = 0; i < 3; i++) d[i] = -ps[0].v[i];   // the hint went stale: the routine proper
      dnormalize(d);
      state = V1;
    } else if (state == V1) {
      ps[1] = sv;
      dot = ddot(ps[1].v, d);
      if (ccd_zero(dot) || dot < 0) MPR_MISS(dot);
      dcross(d, ps[0].v, ps[1].v);
      if (ccd_zero(ddot(d, d))) {
        for (int i = 0; i < 3; i++) pos[i] = 0.5 * (ps[1].v1[i] + ps[1].v2[i]);
        if (ccd_zero(ps[1].v[0]) && ccd_zero(ps[1].v[1]) && ccd_zero(ps[1].v[2])) { *depth = 0; dir[0] = dir[1] = dir[2] = 0; return 0; }
        for (int i = 0; i < 3; i++) dir[i] = ps[1].v[i];
        *depth = dnormalize(dir);
        return 0;
      }
      dnormalize(d);
      state = V2;
    } else if (state == V2) {
      ps[2] = sv;
      dot = ddot(ps[2].v, d);
      if (ccd_zero(dot) || dot < 0) MPR_MISS(dot);
      dsub(va, ps[1].v, ps[0].v); dsub(vb, ps[2].v, ps[0].v);
      dcross(d, va, vb);
      dnormalize(d);
      if (ddot(d, ps[0].v) > 0) { Sup t = ps[1]; ps[1] = ps[2]; ps[2] = t; for (int i = 0; i < 3; i++) d[i] = -d[i]; }
      state = DISCOVER;
      guard = 0;
    } else if (state == DISCOVER) {
      if (guard > 1000) return -1;
      guard++;
      ps[3] = sv;
      dot = ddot(ps[3].v, d);
      if (ccd_zero(dot) || dot < 0) MPR_MISS(dot);
      bool cont = false;
      dcross(va, ps[1].v, ps[3].v);
      dot = ddot(va, ps[0].v);
      if (dot < 0 && !ccd_zero(dot)) { ps[2] = ps[3]; cont = true; }
      if (!cont) {
        dcross(va, ps[3].v, ps[2].v);
        dot = ddot(va, ps[0].v);
        if (dot < 0 && !ccd_zero(dot)) { ps[1] = ps[3]; cont = true; }
      }
      if (cont) {
        dsub(va, ps[1].v, ps[0].v); dsub(vb, ps[2].v, ps[0].v);
        dcross(d, va, vb);
        dnormalize(d);
      } else {   // portal found: refinePortal's first test, or straight on to findPenetr (same portal direction)
        portal_dir(ps, d);
        dot = ddot(d, ps[1].v);
        state = (ccd_zero(dot) || dot > 0) ? PENETR : REFINE;
        guard = 0;
      }
    } else if (state == REFINE) {
      dot = ddot(sv.v, d);
      if (!(ccd_zero(dot) || dot > 0)) MPR_MISS(dot);
      if (reach_tol(ps, sv, d)) return -1;
      expand_portal(ps, sv);
      if (++guard >= 10000) return -1;
      portal_dir(ps, d);
      dot = ddot(d, ps[1].v);
      if (ccd_zero(dot) || dot > 0) state = PENETR;
    } else {   // findPenetr
      if (reach_tol(ps, sv, d) || it > MPR_ITER) {
        *depth = sqrt(tri_closest_origin(ps[1].v, ps[2].v, ps[3].v, dir));
        if (ccd_zero(*depth)) dir[0] = dir[1] = dir[2] = 0; else dnormalize(dir);
        double bb[4], t[3], sum;
        portal_dir(ps, d);
        dcross(t, ps[1].v, ps[2].v); bb[0] = ddot(t, ps[3].v);
        dcross(t, ps[3].v, ps[2].v); bb[1] = ddot(t, ps[0].v);
        dcross(t, ps[0].v, ps[1].v); bb[2] = ddot(t, ps[3].v);
        dcross(t, ps[2].v, ps[1].v); bb[3] = ddot(t, ps[0].v);
        sum = bb[0] + bb[1] + bb[2] + bb[3];
        if (ccd_zero(sum) || sum < 0) {
          bb[0] = 0;
          dcross(t, ps[2].v, ps[3].v); bb[1] = ddot(t, d);
          dcross(t, ps[3].v, ps[1].v); bb[2] = ddot(t, d);
          dcross(t, ps[1].v, ps[2].v); bb[3] = ddot(t, d);
          sum = bb[1] + bb[2] + bb[3];
        }
        const double inv = 1.0 / sum;
        double p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
        for (int k = 0; k < 4; k++)
          for (int i = 0; i < 3; i++) { p1[i] += bb[k] * ps[k].v1[i]; p2[i] += bb[k] * ps[k].v2[i]; }
        for (int i = 0; i < 3; i++) pos[i] = 0.5 * inv * (p1[i] + p2[i]);
        return 0;
      }
      expand_portal(ps, sv);
      it++;
      portal_dir(ps, d);
    }
  }
#undef MPR_MISS
}

// [EXT] mjc_Convex at margin 0; spheres get their analytic normal afterwards (mjc_fixNormal)
// Separating-direction cache (per env, in global memory): pairs that passed the box filter but do not touch — adjacent links,
// typically — keep the direction that proved it, stored in geom 1's frame; the next evaluation re-tests it with ONE support
// evaluation instead of running the portal search (4..5 on average).  Entries: pair id, direction xyz; slot SEPC = next victim.
constexpr int SEPC = 16;
// `direct`: cache points at the ONE entry of this (env, pair) — the pair kernel's waves own a pair each, a shared 16-entry set would race.
__device__ __forceinline__ int np_convex(Con *c, const Geo &a, const Geo &b, float *cache, const bool direct, const int pair, Sup *ps,
                                         const int lane) {
  double depth, dir[3], pos[3], hint[3], sep[3];
  bool sep_valid;
  int slot = -1;
  if (cache) {
    const unsigned long long hit = direct ? (__float_as_int(cache[0]) == pair ? 1ull : 0ull)
                                          : __ballot(lane < SEPC && __float_as_int(cache[4 * (lane < SEPC ? lane : 0)]) == pair);
    if (hit) {
      slot = __ffsll((long long)hit) - 1;
      const double hl[3] = {cache[4 * slot + 1], cache[4 * slot + 2], cache[4 * slot + 3]};
      drot(hint, a.mat, hl);
      dnormalize(hint);
    }
  }
  const int res = mpr_penetration(a, b, &depth, dir, pos, slot >= 0 ? hint : nullptr, sep, sep_valid, ps, lane);
  PSTAT(1); if (res == -2) PSTAT(2); else if (res == 0) PSTAT(3); else PSTAT(4);
  if (cache && res == -1 && sep_valid) {   // remember what separated them (geom 1's frame)
    if (slot < 0) {
      if (direct) slot = 0;
      else { slot = __float_as_int(cache[4 * SEPC]) & (SEPC - 1); if (lane == 0) cache[4 * SEPC] = __int_as_float(slot + 1); }
    }
    double sl[3];
    drot_t(sl, a.mat, sep);
    if (lane == 0) { cache[4 * slot] = __int_as_float(pair); cache[4 * slot + 1] = (float)sl[0]; cache[4 * slot + 2] = (float)sl[1]; cache[4 * slot + 3] = (float)sl[2]; }
  } else if (cache && res == 0 && slot >= 0) {
    if (lane == 0) cache[4 * slot] = __int_as_float(-1);   // they touch now: drop the entry
  }
  if (res != 0) return 0;
  if (dir[0] == 0 && dir[1] == 0 && dir[2] == 0) return 0;
  c->dist = -depth;
  for (int i = 0; i < 3; i++) { c->pos[i] = pos[i]; c->n[i] = dir[i]; }
  double n1[3], n2[3];
  bool h1 = false, h2 = false;
  if (a.type == DM_GEOM_SPHERE) { dsub(n1, pos, a.pos); h1 = dnormalize(n1) > MINVAL; }
  if (b.type == DM_GEOM_SPHERE) { dsub(n2, b.pos, pos); h2 = dnormalize(n2) > MINVAL; }
  if (h1 && h2) { for (int i = 0; i < 3; i++) c->n[i] = n1[i] + n2[i]; dnormalize(c->n); }
  else if (h1) for (int i = 0; i < 3; i++) c->n[i] = n1[i];
  else if (h2) for (int i = 0; i < 3; i++) c->n[i] = n2[i];
  return 1;
}

__device__ __forceinline__ int np_plane_sphere(Con *c, const Geo &p, const double *spos, double r) {
  double n[3] = {p.mat[2], p.mat[5], p.mat[8]}, df[3];
  dsub(df, spos, p.pos);
  const double dist = ddot(df, n) - r;
  if (dist > 0) return 0;
  c->dist = dist;
  for (int i = 0; i < 3; i++) { c->n[i] = n[i]; c->pos[i] = spos[i] - n[i] * (r + 0.5 * dist); }
  return 1;
}
__device__ __forceinline__ int np_plane_box(Con *c, const Geo &p, const Geo &b) {
  double n[3] = {p.mat[2], p.mat[5], p.mat[8]}, df[3];
  dsub(df, b.pos, p.pos);
  const double dist = ddot(df, n);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double v[3] = {b.size[0] * ((i & 1) ? 1 : -1), b.size[1] * ((i & 2) ? 1 : -1), b.size[2] * ((i & 4) ? 1 : -1)}, corner[3];
    drot(corner, b.mat, v);
    const double ld = ddot(n, corner);
    if (dist + ld > 0 || ld > 0) continue;
    c[cnt].dist = dist + ld;
    for (int k = 0; k < 3; k++) { c[cnt].n[k] = n[k]; c[cnt].pos[k] = corner[k] + b.pos[k] - n[k] * 0.5 * c[cnt].dist; }
    if (++cnt >= 4) return 4;
  }
  return cnt;
}
__device__ __forceinline__ int np_plane_cylinder(Con *c, const Geo &p, const Geo &cy) {   // [EXT] mjc_PlaneCylinder
  double normal[3] = {p.mat[2], p.mat[5], p.mat[8]}, axis[3] = {cy.mat[2], cy.mat[5], cy.mat[8]};
  double prjaxis = ddot(normal, axis);
  if (prjaxis > 0) { for (int i = 0; i < 3; i++) axis[i] = -axis[i]; prjaxis = -prjaxis; }
  double dif[3], vec[3];
  dsub(dif, cy.pos, p.pos);
  const double dist0 = ddot(dif, normal);
  for (int i = 0; i < 3; i++) vec[i] = axis[i] * prjaxis - normal[i];
  const double len2 = ddot(vec, vec);
  if (len2 >= MINVAL * MINVAL) { const double s = cy.size[0] / sqrt(len2); for (int i = 0; i < 3; i++) vec[i] *= s; }
  else for (int i = 0; i < 3; i++) vec[i] = cy.mat[3 * i] * cy.size[0];
  const double prjvec = ddot(vec, normal);
  for (int i = 0; i < 3; i++) axis[i] *= cy.size[1];
  prjaxis *= cy.size[1];
  int n = 0;
  if (dist0 + prjaxis + prjvec > 0) return 0;
  { const double dd = dist0 + prjaxis + prjvec; c[n].dist = dd;
    for (int i = 0; i < 3; i++) { c[n].pos[i] = cy.pos[i] + vec[i] + axis[i] - normal[i] * dd * 0.5; c[n].n[i] = normal[i]; } n++; }
  if (dist0 - prjaxis + prjvec <= 0) {
    const double dd = dist0 - prjaxis + prjvec; c[n].dist = dd;
    for (int i = 0; i < 3; i++) { c[n].pos[i] = cy.pos[i] + vec[i] - axis[i] - normal[i] * dd * 0.5; c[n].n[i] = normal[i]; } n++;
  }
  const double prjvec1 = -0.5 * prjvec;
  if (dist0 + prjaxis + prjvec1 <= 0) {
    double vec1[3];
    dcross(vec1, vec, axis);
    dnormalize(vec1);
    for (int i = 0; i < 3; i++) vec1[i] *= cy.size[0] * sqrt(3.0) * 0.5;
    const double dd = dist0 + prjaxis + prjvec1;
    for (int sg = 1; sg >= -1; sg -= 2) {
      c[n].dist = dd;
      for (int i = 0; i < 3; i++) { c[n].pos[i] = cy.pos[i] + sg * vec1[i] + axis[i] - 0.5 * vec[i] - normal[i] * dd * 0.5; c[n].n[i] = normal[i]; }
      n++;
    }
  }
  return n;
}
// [EXT] mjc_PlaneConvex for a mesh: support vertex towards the plane + three directions tilted by 0.3 (low-confidence
// restatement of the multi-contact rule, identical to the oracle's)
__device__ __forceinline__ int np_plane_mesh_body(Con *c, const Geo &p, const Geo &g, const int lane) {
  double normal[3] = {p.mat[2], p.mat[5], p.mat[8]}, t1[3] = {p.mat[0], p.mat[3], p.mat[6]}, t2[3] = {p.mat[1], p.mat[4], p.mat[7]};
  int used[4], n = 0;
  for (int k = 0; k < 4; k++) {
    double dw[3], dl[3];
    if (k == 0) for (int i = 0; i < 3; i++) dw[i] = -normal[i];
    else {
      const double ang = 2.0 * 3.14159265358979323846 * (k - 1) / 3.0, ca = 0.3 * cos(ang), sa = 0.3 * sin(ang);
      for (int i = 0; i < 3; i++) dw[i] = -normal[i] + ca * t1[i] + sa * t2[i];
    }
    drot_t(dl, g.mat, dw);
    double vl[3];
    const int vi = mesh_support_index(g, dl, vl, lane);
    bool dup = false;
    for (int j = 0; j < n; j++) dup |= used[j] == vi;
    if (dup) continue;
    double v[3], dif[3];
    drot(v, g.mat, vl);
    for (int i = 0; i < 3; i++) v[i] += g.pos[i];
    dsub(dif, v, p.pos);
    const double dist = ddot(dif, normal);
    if (dist > 0) { if (k == 0) return 0; continue; }
    used[n] = vi;
    c[n].dist = dist;
    for (int i = 0; i < 3; i++) { c[n].pos[i] = v[i] - 0.5 * dist * normal[i]; c[n].n[i] = normal[i]; }
    n++;
  }
  return n;
}
__device__ __forceinline__ int np_sphere_sphere(Con *c, const double *p1, double r1, const double *p2, double r2) {
  double df[3];
  dsub(df, p2, p1);
  const double cd = dnorm(df), dist = cd - r1 - r2;
  if (dist > 0) return 0;
  c->dist = dist;
  if (cd < MINVAL) { c->n[0] = 1; c->n[1] = c->n[2] = 0; }
  else for (int i = 0; i < 3; i++) c->n[i] = df[i] / cd;
  for (int i = 0; i < 3; i++) c->pos[i] = p1[i] + c->n[i] * (r1 + 0.5 * dist);
  return 1;
}
__device__ __forceinline__ int np_sphere_box(Con *c, const Geo &s, const Geo &b) {
  double t[3], ctr[3], cl[3], nl[3];
  const double r = s.size[0];
  dsub(t, s.pos, b.pos);
  drot_t(ctr, b.mat, t);
  for (int i = 0; i < 3; i++) { cl[i] = dclamp(ctr[i], -b.size[i], b.size[i]); nl[i] = cl[i] - ctr[i]; }
  const double dd = dnorm(nl);
  double dist, pl[3];
  if (dd - r > 0) return 0;
  if (dd <= MINVAL) {
    double closest = 2 * (b.size[0] + b.size[1] + b.size[2]);
    int k = 0;
    for (int i = 0; i < 6; i++) {
      const double test = b.size[i / 2] - ((i % 2) ? -1.0 : 1.0) * ctr[i / 2];
      if (test < closest) { closest = test; k = i; }
    }
    nl[0] = nl[1] = nl[2] = 0;
    nl[k / 2] = (k % 2) ? 1.0 : -1.0;
    dist = -closest - r;
  } else {
    for (int i = 0; i < 3; i++) nl[i] /= dd;
    dist = dd - r;
  }
  for (int i = 0; i < 3; i++) pl[i] = ctr[i] + nl[i] * (r + 0.5 * dist);
  c->dist = dist;
  drot(c->n, b.mat, nl);
  drot(t, b.mat, pl);
  for (int i = 0; i < 3; i++) c->pos[i] = t[i] + b.pos[i];
  return 1;
}
// box-box (SAT + face clipping), same construction as the humanoid3d path (own construction, not MuJoCo's code)
__device__ __forceinline__ int np_box_box_body(Con *c, const Geo &A, const Geo &Bx, double (*poly)[3], double (*tmp)[3]) {
  const double *p1 = A.pos, *R1 = A.mat, *s1 = A.size, *p2 = Bx.pos, *R2 = Bx.mat, *s2 = Bx.size;
  double R[9], AR[9], t[3], tw[3];
  dsub(tw, p2, p1);
  drot_t(t, R1, tw);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      R[3 * i + j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];
      AR[3 * i + j] = fabs(R[3 * i + j]) + 1e-9;
    }
  double best = -1e30, bn[3] = {0, 0, 0};
  int code = -1;
  for (int i = 0; i < 3; i++) {
    const double s = fabs(t[i]) - (s1[i] + s2[0] * AR[3 * i] + s2[1] * AR[3 * i + 1] + s2[2] * AR[3 * i + 2]);
    if (s > 0) return 0;
    if (s > best) { best = s; code = i; }
  }
  for (int j = 0; j < 3; j++) {
    const double tj = t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j];
    const double s = fabs(tj) - (s2[j] + s1[0] * AR[j] + s1[1] * AR[3 + j] + s1[2] * AR[6 + j]);
    if (s > 0) return 0;
    if (s > best) { best = s; code = 3 + j; }
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double ei[3] = {0, 0, 0}, ej[3] = {R[j], R[3 + j], R[6 + j]}, ax[3];
      ei[i] = 1;
      dcross(ax, ei, ej);
      const double l = dnorm(ax);
      if (l < 1e-6) continue;
      for (int k = 0; k < 3; k++) ax[k] /= l;
      double ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) ra += s1[k] * fabs(ax[k]);
      for (int k = 0; k < 3; k++) { double ek[3] = {R[k], R[3 + k], R[6 + k]}; rb += s2[k] * fabs(ddot(ax, ek)); }
      const double s = fabs(ddot(t, ax)) - (ra + rb);
      if (s > 0) return 0;
      if (s > best + 0.05 * fabs(best) + 1e-6) { best = s; code = 6 + 3 * i + j; bn[0] = ax[0]; bn[1] = ax[1]; bn[2] = ax[2]; }
    }
  if (code < 0) return 0;
  if (code >= 6) {
    const int i = (code - 6) / 3, j = (code - 6) % 3;
    double n1[3] = {bn[0], bn[1], bn[2]};
    if (ddot(n1, t) < 0) for (int k = 0; k < 3; k++) n1[k] = -n1[k];
    double pa[3], pb[3];
    for (int k = 0; k < 3; k++) pa[k] = (k == i) ? 0 : ((n1[k] > 0) ? s1[k] : -s1[k]);
    for (int k = 0; k < 3; k++) pb[k] = t[k];
    for (int k = 0; k < 3; k++) {
      if (k == j) continue;
      double ek[3] = {R[k], R[3 + k], R[6 + k]};
      const double sg = (ddot(n1, ek) > 0) ? -s2[k] : s2[k];
      for (int q = 0; q < 3; q++) pb[q] += sg * ek[q];
    }
    double ua[3] = {0, 0, 0}, ub[3] = {R[j], R[3 + j], R[6 + j]}, w[3];
    ua[i] = 1;
    dsub(w, pb, pa);
    const double uaub = ddot(ua, ub), q1 = ddot(ua, w), q2 = -ddot(ub, w), dd = 1 - uaub * uaub;
    double alpha = 0, beta = 0;
    if (dd > 1e-12) { alpha = (q1 + uaub * q2) / dd; beta = (uaub * q1 + q2) / dd; }
    alpha = dclamp(alpha, -s1[i], s1[i]);
    beta = dclamp(beta, -s2[j], s2[j]);
    double mid[3], mw[3];
    for (int k = 0; k < 3; k++) mid[k] = 0.5 * ((pa[k] + ua[k] * alpha) + (pb[k] + ub[k] * beta));
    drot(mw, R1, mid);
    drot(c->n, R1, n1);
    for (int k = 0; k < 3; k++) c->pos[k] = mw[k] + p1[k];
    c->dist = best;
    return 1;
  }
  const double *Ra, *Rb, *sa, *sb, *pa, *pb;
  int ax, flip;
  if (code < 3) { Ra = R1; Rb = R2; sa = s1; sb = s2; pa = p1; pb = p2; ax = code; flip = 0; }
  else { Ra = R2; Rb = R1; sa = s2; sb = s1; pa = p2; pb = p1; ax = code - 3; flip = 1; }
  double nrm[3] = {Ra[ax], Ra[3 + ax], Ra[6 + ax]}, dab[3];
  dsub(dab, pb, pa);
  if (ddot(nrm, dab) < 0) for (int k = 0; k < 3; k++) nrm[k] = -nrm[k];
  int ib = 0;
  double bestd = -1, nb[3];
  drot_t(nb, Rb, nrm);
  for (int k = 0; k < 3; k++) if (fabs(nb[k]) > bestd) { bestd = fabs(nb[k]); ib = k; }
  const double sgn = (nb[ib] > 0) ? -1.0 : 1.0;
  const int u = (ib + 1) % 3, v = (ib + 2) % 3;
  int np = 4;
  for (int q = 0; q < 4; q++) {
    const double su = (q == 0 || q == 3) ? -sb[u] : sb[u], sv = (q < 2) ? -sb[v] : sb[v];
    double loc[3], w[3], rel[3];
    loc[ib] = sgn * sb[ib]; loc[u] = su; loc[v] = sv;
    drot(w, Rb, loc);
    for (int k = 0; k < 3; k++) rel[k] = w[k] + pb[k] - pa[k];
    drot_t(poly[q], Ra, rel);
  }
  const int axes[2] = {(ax + 1) % 3, (ax + 2) % 3};
  for (int e = 0; e < 2; e++)
    for (int sd = -1; sd <= 1; sd += 2) {
      const int a = axes[e];
      int nn = 0;
      for (int q = 0; q < np; q++) {
        const double *P = poly[q], *Q = poly[(q + 1) % np];
        const double dp = sd * P[a] - sa[a], dq = sd * Q[a] - sa[a];
        if (dp <= 0) { for (int k = 0; k < 3; k++) tmp[nn][k] = P[k]; nn++; }
        if ((dp < 0 && dq > 0) || (dp > 0 && dq < 0)) {
          const double f = dp / (dp - dq);
          for (int k = 0; k < 3; k++) tmp[nn][k] = P[k] + f * (Q[k] - P[k]);
          nn++;
        }
        if (nn >= 15) break;
      }
      np = nn;
      for (int q = 0; q < np; q++) for (int k = 0; k < 3; k++) poly[q][k] = tmp[q][k];
      if (np == 0) return 0;
    }
  double nl[3];
  drot_t(nl, Ra, nrm);
  int cnt = 0;
  for (int q = 0; q < np && cnt < 8; q++) {
    const double depth = ddot(nl, poly[q]) - sa[ax];
    if (depth > 0) continue;
    double pl[3], pw[3];
    for (int k = 0; k < 3; k++) pl[k] = poly[q][k] - nl[k] * 0.5 * depth;
    drot(pw, Ra, pl);
    for (int k = 0; k < 3; k++) { c[cnt].pos[k] = pw[k] + pa[k]; c[cnt].n[k] = flip ? -nrm[k] : nrm[k]; }
    c[cnt].dist = depth;
    cnt++;
  }
  while (cnt > 4) {
    int w = 0;
    for (int q = 1; q < cnt; q++) if (c[q].dist >= c[w].dist) w = q;
    for (int q = w; q < cnt - 1; q++) c[q] = c[q + 1];
    cnt--;
  }
  return cnt;
}

// Staging of the pair being processed (fp64, wave-uniform): the two geoms (pos 3 | mat 9 | size 3 | centre 3 each), its contacts,
// the box-box polygons and the MPR portal.  The monolithic kernel keeps it in LDS slack of the contact stage (S.u.co), the pair
// kernel of the split pipeline in a small LDS block of its own.
struct NpStage {
  double (*geo)[18];
  int32_t (*geoi)[6];     // type, nvert, nclus, vertex start, cluster start, pad
  Con *rc;
  double (*poly0)[3], (*poly1)[3];
  Sup *ps;
};
// entry l (0..17) of geom g's staged record: its pose in fp64 from the fp64 pose of its body (kinematics) and the fp64 tables
__device__ __forceinline__ double geo_entry(const Dev &T, const int g, const int l) {
  const int type = T.g_type[g], me = T.g_mesh[g], b = T.g_body[g];
  double bq[4], R[9], v;
  for (int i = 0; i < 4; i++) bq[i] = (double)S.xquat[b][i] + (double)S.xquat_lo[b][i];
  dquat2mat(R, bq);
  if (l >= 12 && l < 15) v = T.g_size_d[g][l - 12];
  else if (l >= 3 && l < 12) {
    const int i = (l - 3) / 3, j = (l - 3) % 3;
    v = R[3 * i] * T.g_mat_d[g][j] + R[3 * i + 1] * T.g_mat_d[g][3 + j] + R[3 * i + 2] * T.g_mat_d[g][6 + j];
  } else {
    const int i = l < 3 ? l : l - 15;
    v = ((double)S.xpos[b][i] + (double)S.xpos_lo[b][i]) +
        (R[3 * i] * T.g_pos_d[g][0] + R[3 * i + 1] * T.g_pos_d[g][1] + R[3 * i + 2] * T.g_pos_d[g][2]);
    if (l >= 15 && type == DM_GEOM_MESH) {
      double gm[3];
      for (int j = 0; j < 3; j++) gm[j] = R[3 * i] * T.g_mat_d[g][j] + R[3 * i + 1] * T.g_mat_d[g][3 + j] + R[3 * i + 2] * T.g_mat_d[g][6 + j];
      v += gm[0] * T.m_center[me][0] + gm[1] * T.m_center[me][1] + gm[2] * T.m_center[me][2];
    }
  }
  return v;
}
__device__ __forceinline__ void stage_geoi(const Dev &T, const NpStage &W, int g, int slot) {
  const int type = T.g_type[g], me = T.g_mesh[g];
  const bool mesh = type == DM_GEOM_MESH;
  W.geoi[slot][0] = type; W.geoi[slot][1] = mesh ? T.m_vnum[me] : 0; W.geoi[slot][2] = mesh ? T.m_cnum[me] : 0;
  W.geoi[slot][3] = mesh ? T.m_vadr[me] : 0; W.geoi[slot][4] = mesh ? T.m_cadr[me] : 0;
}
// stage geom g of the current pair in slot `slot` (lanes 0..17 write one double each)
__device__ __forceinline__ void stage_geo(const Dev &T, const NpStage &W, int g, int slot, const int lane) {
  if (lane < 18) W.geo[slot][lane] = geo_entry(T, g, lane);
  if (lane == 0) stage_geoi(T, W, g, slot);
}
struct MeshPtrs { const double *vert; const int32_t *oidx; const double *clus; };
__device__ __forceinline__ void view_geo(const MeshPtrs &M, const NpStage &W, int slot, Geo &o) {
  o.type = W.geoi[slot][0]; o.nvert = W.geoi[slot][1]; o.nclus = W.geoi[slot][2];
  o.pos = &W.geo[slot][0]; o.mat = &W.geo[slot][3]; o.size = &W.geo[slot][12]; o.center = &W.geo[slot][15];
  o.vert = M.vert + 3 * (size_t)W.geoi[slot][3];
  o.oidx = M.oidx + W.geoi[slot][3];
  o.clus = M.clus + 8 * (size_t)W.geoi[slot][4];
}
// pair classes (cost): 0 analytic, 1 plane - mesh (support queries), 2 MPR
__device__ __forceinline__ int pair_class(const int t1, const int t2) {
  if (t1 == DM_GEOM_PLANE && t2 == DM_GEOM_MESH) return 1;
  if (t1 == DM_GEOM_PLANE || (t1 == DM_GEOM_SPHERE && (t2 == DM_GEOM_SPHERE || t2 == DM_GEOM_BOX)) || (t1 == DM_GEOM_BOX && t2 == DM_GEOM_BOX)) return 0;
  return 2;
}

__device__ __forceinline__ void make_frame(float *f) {   // [EXT] mju_makeFrame
  float n = sqrtf(dot3(f, f));
  if (n < MINVALF) { f[0] = 1; f[1] = f[2] = 0; } else { f[0] /= n; f[1] /= n; f[2] /= n; }
  if (sqrtf(dot3(f + 3, f + 3)) < 0.5f) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5f && f[1] > -0.5f) f[4] = 1; else f[5] = 1;
  }
  const float t = dot3(f, f + 3);
  for (int i = 0; i < 3; i++) f[3 + i] -= t * f[i];
  n = sqrtf(dot3(f + 3, f + 3));
  if (n < MINVALF) { f[3] = 1; f[4] = f[5] = 0; } else { f[3] /= n; f[4] /= n; f[5] /= n; }
  cross3(f + 6, f, f + 3);
}

// Separating-axis test of two oriented boxes (15 axes), conservative by `slack`: true only if the boxes are at least that far
// apart.  The filter is result-neutral: disjoint bounding boxes cannot hold intersecting geoms.
__device__ __forceinline__ bool obb_separated(const float *c1, const float *R1, const float *h1, const float *c2, const float *R2, const float *h2,
                              const float slack) {
  float R[9], AR[9], t[3], tw[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
  for (int i = 0; i < 3; i++) t[i] = R1[i] * tw[0] + R1[3 + i] * tw[1] + R1[6 + i] * tw[2];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      R[3 * i + j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];
      AR[3 * i + j] = fabsf(R[3 * i + j]) + 1e-6f;
    }
  for (int i = 0; i < 3; i++)
    if (fabsf(t[i]) > h1[i] + h2[0] * AR[3 * i] + h2[1] * AR[3 * i + 1] + h2[2] * AR[3 * i + 2] + slack) return true;
  for (int j = 0; j < 3; j++)
    if (fabsf(t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j]) > h2[j] + h1[0] * AR[j] + h1[1] * AR[3 + j] + h1[2] * AR[6 + j] + slack)
      return true;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const float ra = h1[i1] * AR[3 * i2 + j] + h1[i2] * AR[3 * i1 + j], rb = h2[j1] * AR[3 * i + j2] + h2[j2] * AR[3 * i + j1];
      if (fabsf(t[i2] * R[3 * i1 + j] - t[i1] * R[3 * i2 + j]) > ra + rb + slack) return true;
    }
  return false;
}

// [EXT] mj_collision, part 1: candidate pairs in canonical order through the bounding-sphere + bounding-box filters; the survivors'
// pair ids go to S.u.co.surv in order.  Returns their number (capped; `overflow` set when the cap cut the list).
__device__ __forceinline__ int broadphase(const Dev &T, const int cap, int &overflow, const int lane) {
  int nsurv = 0;
  for (int base = 0; base < T.npair; base += 64) {
    const int p = base + lane;
    bool keep = false;
    if (p < T.npair) {
      const int g1 = T.p_g1[p], g2 = T.p_g2[p], ci1 = T.g_ci[g1], ci2 = T.g_ci[g2];
      float df[3] = {S.gpos[g2][0] - S.gpos[g1][0], S.gpos[g2][1] - S.gpos[g1][1], S.gpos[g2][2] - S.gpos[g1][2]};
      float c2[3], bc2[3] = {T.g_bc[g2][0], T.g_bc[g2][1], T.g_bc[g2][2]}, h2[3] = {T.g_bh[g2][0], T.g_bh[g2][1], T.g_bh[g2][2]};
      mat_vec(c2, S.gmat[ci2], bc2);
      for (int i = 0; i < 3; i++) c2[i] += S.gpos[g2][i];
      if (T.g_type[g1] != DM_GEOM_PLANE) {
        keep = sqrtf(dot3(df, df)) <= T.g_rbound[g1] + T.g_rbound[g2];
        if (keep) {
          float c1[3], bc1[3] = {T.g_bc[g1][0], T.g_bc[g1][1], T.g_bc[g1][2]}, h1[3] = {T.g_bh[g1][0], T.g_bh[g1][1], T.g_bh[g1][2]};
          mat_vec(c1, S.gmat[ci1], bc1);
          for (int i = 0; i < 3; i++) c1[i] += S.gpos[g1][i];
          keep = !obb_separated(c1, S.gmat[ci1], h1, c2, S.gmat[ci2], h2, 1e-4f);
        }
      } else {
        float nrm[3] = {S.gmat[ci1][2], S.gmat[ci1][5], S.gmat[ci1][8]};
        keep = !(T.g_rbound[g2] > 0) || dot3(df, nrm) <= T.g_rbound[g2];
        if (keep) {   // lowest point of the bounding box above the plane: no contact possible
          const float *M2 = S.gmat[ci2];
          float dc[3] = {c2[0] - S.gpos[g1][0], c2[1] - S.gpos[g1][1], c2[2] - S.gpos[g1][2]}, ext = 0;
          for (int j = 0; j < 3; j++) ext += fabsf(nrm[0] * M2[j] + nrm[1] * M2[3 + j] + nrm[2] * M2[6 + j]) * h2[j];
          keep = dot3(dc, nrm) - ext <= 1e-4f;
        }
      }
    }
    const unsigned long long m = __ballot(keep);
    const int slot = nsurv + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    if (keep) { if (slot < cap) S.u.co.surv[slot] = (int16_t)p; }
    nsurv += __popcll(m);
  }
  if (nsurv > cap) { overflow = 1; nsurv = cap; }
  return nsurv;
}

// [EXT] mj_collision, part 2: the narrowphase of ONE staged pair (geoms in W.geo / W.geoi): analytic routines for the pairs MuJoCo
// has them for, MPR for everything with a cylinder or a mesh.  Returns the number of contacts left in W.rc.
// out-of-line copies for the monolithic kernel (its register budget is the env phases'); the pair kernel inlines the bodies: the
// Geo views then stay in registers instead of one scratch copy per lane (they were passed by reference to the calls), and the
// kernel needs 190 instead of 226 VGPRs
__device__ __noinline__ int np_plane_mesh(Con *c, const Geo &p, const Geo &g, const int lane) { return np_plane_mesh_body(c, p, g, lane); }
__device__ __noinline__ int np_box_box(Con *c, const Geo &A, const Geo &Bx, double (*poly)[3], double (*tmp)[3]) { return np_box_box_body(c, A, Bx, poly, tmp); }

template <bool INL>
__device__ __forceinline__ int narrow_pair(const MeshPtrs &M, const NpStage &W, float *cache, const bool direct, const int pair,
                                           const int skip, const int lane) {
  // (in the monolithic kernel A and B go to scratch, one copy per lane, because the two out-of-line routines take them by reference.
  // Passing the staging pointers by value instead was 5 % SLOWER in the pair kernel — 5.07 against 4.81 ms per step — and needs the
  // pointers laundered, or hipcc 7.2 folds the constant LDS address into an unencodable `v_cmp_ne_u32 0, src_shared_base`.)
  Geo A, B;
  view_geo(M, W, 0, A);
  view_geo(M, W, 1, B);
  Con *rc = W.rc;
  int n = 0;
  const int t1 = A.type, t2 = B.type;
  if (t1 == DM_GEOM_PLANE) {
    if (t2 == DM_GEOM_SPHERE) n = np_plane_sphere(rc, A, B.pos, B.size[0]);
    else if (t2 == DM_GEOM_CYLINDER) n = np_plane_cylinder(rc, A, B);
    else if (t2 == DM_GEOM_BOX) n = np_plane_box(rc, A, B);
    else if (t2 == DM_GEOM_MESH && !(skip & 128)) n = INL ? np_plane_mesh_body(rc, A, B, lane) : np_plane_mesh(rc, A, B, lane);
  } else if (t1 == DM_GEOM_SPHERE && t2 == DM_GEOM_SPHERE) n = np_sphere_sphere(rc, A.pos, A.size[0], B.pos, B.size[0]);
  else if (t1 == DM_GEOM_SPHERE && t2 == DM_GEOM_BOX) n = np_sphere_box(rc, A, B);
  else if (t1 == DM_GEOM_BOX && t2 == DM_GEOM_BOX) n = INL ? np_box_box_body(rc, A, B, W.poly0, W.poly1) : np_box_box(rc, A, B, W.poly0, W.poly1);
  else if (!(skip & 64)) n = np_convex(rc, A, B, (skip & 256) ? nullptr : cache, direct, pair, W.ps, lane);
  return n;
}
// contact k of the staged pair -> the fp32 record the constraint stage reads: dist | pos 3 | frame 9 ([EXT] mju_makeFrame of the normal)
__device__ __forceinline__ void contact_record(const Con &c, float &dist, float *pos, float *fr) {
  dist = (float)c.dist;
  fr[0] = (float)c.n[0]; fr[1] = (float)c.n[1]; fr[2] = (float)c.n[2];
  for (int i = 3; i < 9; i++) fr[i] = 0;
  make_frame(fr);
  for (int i = 0; i < 3; i++) pos[i] = (float)c.pos[i];
}

// [EXT] mj_collision of the monolithic kernel: the survivors one after the other on this wave
__device__ __noinline__ int collide(const Dev &T, const Launch &P, const int env, const int lane) {
  int overflow = 0;
  int nsurv = broadphase(T, MAXSURV, overflow, lane);
  if (P.pad & 32) nsurv = 0;
  SYNC();
  PROF(3);
  int ncon = 0, n_pm = 0, n_an = 0, n_mpr = 0;
  const NpStage W = {S.u.co.geo, S.u.co.geoi, reinterpret_cast<Con *>(&S.u.co.rc[0][0]), S.u.co.poly[0], S.u.co.poly[1],
                     reinterpret_cast<Sup *>(&S.u.co.mpr_ps[0][0])};
  const MeshPtrs M = {P.mesh_vert, P.mesh_oidx, P.mesh_clus};
  for (int s = 0; s < nsurv; s++) {
    const int p = S.u.co.surv[s];
    const int g1 = T.p_g1[p], g2 = T.p_g2[p];
    stage_geo(T, W, g1, 0, lane);
    stage_geo(T, W, g2, 1, lane);
    SYNC();
    const int cls = pair_class(W.geoi[0][0], W.geoi[1][0]);
    n_an += cls == 0; n_pm += cls == 1; n_mpr += cls == 2;
    const int n = narrow_pair<false>(M, W, P.sepc + (size_t)env * (4 * SEPC + 4), false, p, P.pad, lane);
    SYNC();
    PROF(cls == 1 ? 5 : (cls == 0 ? 4 : 6));
    for (int k = 0; k < n; k++) {
      if (ncon >= MAXCON) { overflow = 1; continue; }
      if (lane == 0) {
        contact_record(W.rc[k], S.u.co.c_dist[ncon], S.u.co.c_pos[ncon], S.u.co.c_frame[ncon]);
        S.u.co.c_g1[ncon] = g1; S.u.co.c_g2[ncon] = g2;
        S.u.co.c_mu[ncon] = fmaxf(T.g_mu[g1], T.g_mu[g2]);
      }
      ncon++;
    }
    SYNC();
  }
  SYNC();
  if (lane == 0) { S.info[0] = ncon; S.info[4] = overflow; S.info[5] = nsurv; S.info[6] = n_an | (n_pm << 8) | (n_mpr << 16); }
  return ncon;
}

// ------------------------------------------------------------------------------------------ constraints
__device__ __forceinline__ float impedance(const float *solimp, float pos, float margin) {   // [EXT] getimpedance
  const float dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (dmin == dmax || width <= MINVALF) return 0.5f * (dmin + dmax);
  const float x = fabsf(pos - margin) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  float y;
  if (power == 1) y = x;
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1);
  else y = 1 - powf(1 - x, power) / powf(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}

// rows -> J^T (global, dof-major), R, aref; returns nefc.  Order: friction loss (dof order), limits (joint order), contacts.
__device__ __noinline__ int make_constraint(const Dev &T, float *JT, float *RW, const int ncon, const int lane) {
  float *e_R = RW, *e_b = RW + MAXROW, *e_f = RW + 2 * MAXROW, *e_lim = RW + 3 * MAXROW;
  int32_t *e_meta = (int32_t *)(RW + 4 * MAXROW);
  // friction-loss rows: dofs 6..42 -> rows 0..36
  int nefc = 0;
  for (int k = 6; k < NV; k++) {   // (all hinges of this model carry friction loss; the table says which)
    if (T.d_floss[k] > 0) nefc++;
  }
  const int nfric = nefc;
  if (lane < NV - 6) {
    const int k = 6 + lane;
    // row index = number of friction dofs before k
    int r = 0;
    for (int q = 6; q < k; q++) r += T.d_floss[q] > 0;
    if (T.d_floss[k] > 0) {
      e_meta[r] = ROW_FRICTION | (k << 2);
      e_lim[r] = T.d_floss[k];
      const float imp = impedance(T.solimp, 0.f, 0.f);
      e_R[r] = fmaxf(MINVALF, (1 - imp) * T.d_invw[k] / imp);
      e_b[r] = -T.B * S.qvel[k];   // aref (K = 0 for friction rows)
    }
  }
  // joint limits
  bool lo = false, hi = false;
  float dlo = 0, dhi = 0;
  if (lane < NV - 6) {
    const int k = 6 + lane;
    const float q = S.qpos[k + 1];
    dlo = q - T.d_lo[k]; dhi = T.d_hi[k] - q;
    lo = dlo < 0; hi = dhi < 0;
  }
  {
    const unsigned long long mlo = __ballot(lo), mhi = __ballot(hi);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int before = __popcll(mlo & lt) + __popcll(mhi & lt);
    if (lo || hi) {
      const int k = 6 + lane;
      int r = nefc + before;
      for (int side = 0; side < 2; side++) {
        if (!(side ? hi : lo)) continue;
        if (r < MAXROW) {
          const float dist = side ? dhi : dlo;
          e_meta[r] = ROW_LIMIT | (k << 2) | (side << 12);
          e_lim[r] = 0;
          const float imp = impedance(T.solimp, dist, 0.f);
          e_R[r] = fmaxf(MINVALF, (1 - imp) * T.d_invw[k] / imp);
          const float jv = side ? -1.f : 1.f;
          e_b[r] = -T.B * (jv * S.qvel[k]) - T.K * imp * dist;
        }
        r++;
      }
    }
    nefc += __popcll(mlo) + __popcll(mhi);
  }
  const int nlimit = nefc - nfric;
  // contacts: 4 pyramid rows each
  const int row0 = nefc;
  int kept = ncon;
  if (row0 + 4 * ncon > MAXROW) kept = (MAXROW - row0) / 4;
  nefc = row0 + 4 * kept;
  SYNC();
  // zero J^T for all rows, then fill
  for (int k = 0; k < NV; k++)
    for (int r = lane; r < nefc; r += 64) JT[k * MAXROW + r] = 0.f;
  SYNC();
  for (int r = lane; r < row0; r += 64) {
    const int meta = e_meta[r], k = (meta >> 2) & 0x3FF, type = meta & 3;
    JT[k * MAXROW + r] = (type == ROW_FRICTION) ? 1.f : (((meta >> 12) & 1) ? -1.f : 1.f);
  }
  for (int c = 0; c < kept; c++) {
    const int b1 = T.g_body[S.u.co.c_g1[c]], b2 = T.g_body[S.u.co.c_g2[c]];
    const float mu = S.u.co.c_mu[c];
    const int r0 = row0 + 4 * c;
    if (lane < NV) {
      const int k = lane;
      const bool in1 = (T.b_chain[b1] >> k) & 1, in2 = (T.b_chain[b2] >> k) & 1;
      float j[3] = {0, 0, 0};
      if (in1 != in2) {
        float off[3] = {S.u.co.c_pos[c][0] - S.com[0], S.u.co.c_pos[c][1] - S.com[1], S.u.co.c_pos[c][2] - S.com[2]}, t[3];
        cross3(t, S.cdof[k], off);
        for (int i = 0; i < 3; i++) j[i] = (in2 ? 1.f : -1.f) * (S.cdof[k][3 + i] + t[i]);
      }
      const float *fr = S.u.co.c_frame[c];
      const float j0 = fr[0] * j[0] + fr[1] * j[1] + fr[2] * j[2], j1 = fr[3] * j[0] + fr[4] * j[1] + fr[5] * j[2],
                  j2 = fr[6] * j[0] + fr[7] * j[1] + fr[8] * j[2];
      JT[k * MAXROW + r0] = j0 + mu * j1; JT[k * MAXROW + r0 + 1] = j0 - mu * j1;
      JT[k * MAXROW + r0 + 2] = j0 + mu * j2; JT[k * MAXROW + r0 + 3] = j0 - mu * j2;
    }
    if (lane == 0) {
      const float tran = T.b_invw[b1] + T.b_invw[b2], diag = tran + mu * mu * tran;
      const float imp = impedance(T.solimp, S.u.co.c_dist[c], 0.f);
      const float R0 = fmaxf(MINVALF, (1 - imp) * diag / imp), Rpy = 2.f * mu * mu * R0;
      for (int e = 0; e < 4; e++) {
        e_meta[r0 + e] = ROW_CONTACT | (c << 2);
        e_lim[r0 + e] = 0;
        e_R[r0 + e] = Rpy;
        e_f[r0 + e] = imp;   // parked: aref needs the row velocity, computed below
      }
    }
  }
  SYNC();
  // contact rows: vel = J qvel, aref = -B vel - K imp dist
  for (int r = row0 + lane; r < nefc; r += 64) {
    float vel = 0;
    for (int k = 0; k < NV; k++) vel += JT[k * MAXROW + r] * S.qvel[k];
    const int c = (e_meta[r] >> 2);
    e_b[r] = -T.B * vel - T.K * e_f[r] * S.u.co.c_dist[c];
  }
  if (lane == 0) { S.info[1] = nefc; S.info[2] = nlimit; if (kept < ncon) S.info[4] = 1; }
  SYNC();
  return nefc;
}


// A = J M^-1 J^T + R into the per-env scratch.  Up to 128 rows: lane r (and r + 64) keeps its row of B = D^-1/2 L^-T J^T in
// 43 registers (solved with static indices), and A[i][:] is 43 broadcasts of row i (v_readlane: scalar operands) times the
// lanes' own rows — no memory traffic except J in and A out.  More rows: the general path through the B^T scratch.
// (body shared by the out-of-line copy the monolithic kernel calls and the split env kernel, which inlines it: as a callee it
// saves and restores 112 VGPRs per call through scratch — 28 KB per env and evaluation, most of g1_env_kernel's HBM writes)
__device__ __forceinline__ void project_constraint_body(const Dev &T, const float *JT, float *BT, float *AR, const float *RW,
                                                const int nefc, const int lane) {
  const float *e_R = RW;
  if (nefc <= MAXROW) {
    // rows in chunks of 64 (up to four): the lanes' rows of B in 43 registers each; A[i][j] for i in the same chunk by
    // broadcasts of row i, for i in an EARLIER chunk from the rows that chunk stored in the B^T scratch (wave-uniform loads),
    // written both ways
    const int nchunk = (nefc + 63) >> 6;
    for (int cj = 0; cj < nchunk; cj++) {
      const int j = lane + 64 * cj;
      const bool on = j < nefc;
      float x[NV];
#pragma unroll
      for (int k = 0; k < NV; k++) x[k] = on ? JT[k * MAXROW + j] : 0.f;
      RowSolve<NV - 1>::run(x, S.qLD);
#pragma unroll
      for (int k = 0; k < NV; k++) x[k] *= S.dsq[k];
      if (nchunk > 1) {
#pragma unroll
        for (int k = 0; k < NV; k++) if (on) BT[k * MAXROW + j] = x[k];
        SYNC();
      }
      // A is stored SCALED for the PGS sweep: entry (i, j) x -1 / A_jj, the factor of the lane that owns column j (the sweep
      // carries g_j = -r_j / A_jj); the diagonal itself goes to the spare row 43 of the B^T scratch
      const float rj = on ? e_R[j] : 0.f;
      float ajj = rj;
#pragma unroll
      for (int k = 0; k < NV; k++) ajj = fmaf(x[k], x[k], ajj);
      const float nj = on ? -1.f / ajj : 0.f;
      if (on) BT[NV * MAXROW + j] = ajj;
      if (nchunk > 1) SYNC();
      // rows of the same chunk: the 64 x 64 block B B^T on the matrix pipe — four 32 x 32 sub-blocks of v_mfma_f32_32x32x2f32
      // over 22 k pairs.  The instruction wants lane (r, h) to supply B[row r][k0 + h]: one v_permlane32_swap of (x[k0], x[k0 + 1])
      // yields the operand of rows 0..31 (lower half keeps x[k0], upper half receives the lower half's x[k0 + 1]) AND that of rows
      // 32..63 (the other result).  88 MFMAs + 22 swaps instead of 64 x (43 v_readlane + 43 FMA).
      {
        typedef float f16v __attribute__((ext_vector_type(16)));
        f16v acc00, acc01, acc10, acc11;
#pragma unroll
        for (int v = 0; v < 16; v++) acc00[v] = acc01[v] = acc10[v] = acc11[v] = 0.f;
        StaticFor<0, (NV + 1) / 2>::run([&](auto kc) {
          constexpr int k0 = decltype(kc)::value * 2;
          const float xa = x[k0], xb = (k0 + 1 < NV) ? x[k0 + 1 < NV ? k0 + 1 : 0] : 0.f;
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xa), __float_as_uint(xb), false, false);
          const float op0 = __uint_as_float(sw[0]), op1 = __uint_as_float(sw[1]);
          acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(op0, op0, acc00, 0, 0, 0);
          acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(op0, op1, acc01, 0, 0, 0);
          acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(op1, op0, acc10, 0, 0, 0);
          acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(op1, op1, acc11, 0, 0, 0);
          return true;
        });
        // result fragment: lane (c, h) holds D[row = 8 (v / 4) + 4 h + v % 4][col = c] of its sub-block; column j's scale
        // -1 / A_jj lives in lane j: both halves fetched with one swap
        const auto nsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(nj), __float_as_uint(nj), false, false);
        const float n0 = __uint_as_float(nsw[0]), n1 = __uint_as_float(nsw[1]);   // of columns c and 32 + c
        const int c = lane & 31, h = lane >> 5, base = 64 * cj;
#pragma unroll
        for (int v = 0; v < 16; v++) {
          const int r = 8 * (v / 4) + 4 * h + (v % 4);
#pragma unroll
          for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int jb = 0; jb < 2; jb++) {
              const int i = base + 32 * ib + r, jj = base + 32 * jb + c;
              const float val = ib == 0 ? (jb == 0 ? acc00[v] : acc01[v]) : (jb == 0 ? acc10[v] : acc11[v]);
              if (i < nefc && jj < nefc) AR[i * MAXROW + jj] = (i == jj) ? -1.f : val * (jb == 0 ? n0 : n1);
            }
        }
      }
      // cross blocks: an earlier chunk's rows (operand fragments loaded straight from the B^T scratch in the layout the
      // instruction wants) against this chunk's, on the matrix pipe as well; pass 0 gives the block A[ci][cj], pass 1 its
      // transpose A[cj][ci] with the operands exchanged, so both are stored with coalesced rows
      for (int ci = 0; ci < cj; ci++) {
        typedef float f16v __attribute__((ext_vector_type(16)));
        const int c = lane & 31, h = lane >> 5;
        const auto nsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(nj), __float_as_uint(nj), false, false);
        const float ncj[2] = {__uint_as_float(nsw[0]), __uint_as_float(nsw[1])};          // columns of this chunk
        const float nci[2] = {-1.f / BT[NV * MAXROW + 64 * ci + c], -1.f / BT[NV * MAXROW + 64 * ci + 32 + c]};   // of chunk ci (full)
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
          f16v d00, d01, d10, d11;
#pragma unroll
          for (int v = 0; v < 16; v++) d00[v] = d01[v] = d10[v] = d11[v] = 0.f;
          StaticFor<0, (NV + 1) / 2>::run([&](auto kc) {
            constexpr int k0 = decltype(kc)::value * 2;
            const float xa = x[k0], xb = (k0 + 1 < NV) ? x[k0 + 1 < NV ? k0 + 1 : 0] : 0.f;
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xa), __float_as_uint(xb), false, false);
            const float oj0 = __uint_as_float(sw[0]), oj1 = __uint_as_float(sw[1]);
            const bool kv = k0 + h < NV;
            const float oi0 = kv ? BT[(k0 + h) * MAXROW + 64 * ci + c] : 0.f, oi1 = kv ? BT[(k0 + h) * MAXROW + 64 * ci + 32 + c] : 0.f;
            if (pass == 0) {
              d00 = __builtin_amdgcn_mfma_f32_32x32x2f32(oi0, oj0, d00, 0, 0, 0);
              d01 = __builtin_amdgcn_mfma_f32_32x32x2f32(oi0, oj1, d01, 0, 0, 0);
              d10 = __builtin_amdgcn_mfma_f32_32x32x2f32(oi1, oj0, d10, 0, 0, 0);
              d11 = __builtin_amdgcn_mfma_f32_32x32x2f32(oi1, oj1, d11, 0, 0, 0);
            } else {
              d00 = __builtin_amdgcn_mfma_f32_32x32x2f32(oj0, oi0, d00, 0, 0, 0);
              d01 = __builtin_amdgcn_mfma_f32_32x32x2f32(oj0, oi1, d01, 0, 0, 0);
              d10 = __builtin_amdgcn_mfma_f32_32x32x2f32(oj1, oi0, d10, 0, 0, 0);
              d11 = __builtin_amdgcn_mfma_f32_32x32x2f32(oj1, oi1, d11, 0, 0, 0);
            }
            return true;
          });
          const int rbase = 64 * (pass == 0 ? ci : cj), cbase = 64 * (pass == 0 ? cj : ci);
#pragma unroll
          for (int v = 0; v < 16; v++) {
            const int r = 8 * (v / 4) + 4 * h + (v % 4);
#pragma unroll
            for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
              for (int b2 = 0; b2 < 2; b2++) {
                const int i = rbase + 32 * a2 + r, jj = cbase + 32 * b2 + c;
                const float val = a2 == 0 ? (b2 == 0 ? d00[v] : d01[v]) : (b2 == 0 ? d10[v] : d11[v]);
                if (i < nefc && jj < nefc) AR[i * MAXROW + jj] = val * (pass == 0 ? ncj[b2] : nci[b2]);
              }
          }
        }
      }
    }
    SYNC();
    return;
  }
}

// [EXT] mj_fwdConstraint + mj_solPGS: dual PGS, rows unilateral (limits, pyramid edges) or boxed (friction loss)
__device__ __forceinline__ void fwd_constraint_body(const Dev &T, const float *JT, const float *BT, const float *AR, float *RW, const int nefc, const int lane,
                                             const int max_iter) {
  float *e_R = RW, *e_b = RW + MAXROW, *e_f = RW + 2 * MAXROW, *e_lim = RW + 3 * MAXROW;
  const int32_t *e_meta = (const int32_t *)(RW + 4 * MAXROW);
  if (nefc == 0) {
    if (lane < NV) { S.qacc[lane] = S.qas[lane]; S.warm[lane] = S.qas[lane]; S.qfc[lane] = 0; }
    if (lane == 0) S.info[3] = 0;
    SYNC();
    return;
  }
  // b = J qacc_smooth - aref; warm-start forces from J qacc_warmstart - aref
  for (int r = lane; r < nefc; r += 64) {
    float s = 0, w = 0;
    for (int k = 0; k < NV; k++) { const float j = JT[k * MAXROW + r]; s += j * S.qas[k]; w += j * S.warm[k]; }
    const float aref = e_b[r], jar = w - aref, D = 1.f / e_R[r];
    float f;
    if ((e_meta[r] & 3) == ROW_FRICTION) {
      const float fl = e_lim[r], rf = e_R[r] * fl;
      f = jar <= -rf ? fl : (jar >= rf ? -fl : -D * jar);
    } else f = jar < 0 ? -D * jar : 0.f;
    e_b[r] = s - aref;
    e_f[r] = f;
  }
  SYNC();
  const int nr = (nefc + 63) >> 6;   // rows per lane (<= 4)
  float res[4] = {0, 0, 0, 0}, fr[4] = {0, 0, 0, 0}, diag[4] = {1, 1, 1, 1};
  float cost = 0;
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int r = lane + 64 * m;
    if (m < nr && r < nefc) {
      float s = 0;
      for (int c0 = 0; c0 < nefc; c0 += 8) {             // eight column loads in flight (a plain loop waits for each)
        float a[8], fc[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const bool ok = c0 + q < nefc;
          a[q] = ok ? AR[(c0 + q) * MAXROW + r] : 0.f;
          fc[q] = ok ? e_f[c0 + q] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) s = fmaf(a[q], fc[q], s);
      }
      diag[m] = BT[NV * MAXROW + r];
      s *= -diag[m];                                     // A is stored x -1 / A_rr (project_constraint)
      fr[m] = e_f[r];
      cost += fr[m] * (0.5f * s + e_b[r]);
      res[m] = e_b[r] + s;
    }
  }
  cost = wsum(cost);
  if (cost > 0) {
#pragma unroll
    for (int m = 0; m < 4; m++) { const int r = lane + 64 * m; if (m < nr && r < nefc) { fr[m] = 0; res[m] = e_b[r]; } }
  }
  // the rows' type / bound live in registers of the owning lane; A rows are fetched sixteen sweep-steps ahead (the
  // per-env A scratch is L2-resident: one exposed round trip per block of rows instead of one per row)
  float lm[4] = {-1, -1, -1, -1}, dinv[4] = {1, 1, 1, 1};   // friction-loss bound (< 0: unilateral row), 1 / A_ii
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int r = lane + 64 * m;
    if (m < nr && r < nefc) { lm[m] = ((e_meta[r] & 3) == ROW_FRICTION) ? e_lim[r] : -1.f; dinv[m] = 1.f / diag[m]; }
  }
  PROF(11);
  constexpr int PF = 16;
  int iter = 0;
  if (nefc <= 64) {
    // up to 64 rows (nine evaluations in ten): lane j keeps its row of A in 64 registers, the sweep is unrolled over the rows
    // with static lanes and static register indices — no memory and no dynamic lane select in the sweep's dependent chain
    float arow[64];
#pragma unroll
    for (int i = 0; i < 64; i++) arow[i] = (i < nefc && lane < nefc) ? AR[i * MAXROW + lane] : 0.f;
    // Lane j's force changes only at step j of a sweep, so a sweep carries the residual alone, scaled: g = -r / A_jj with the
    // lane's row of A pre-multiplied by -1 / A_jj.  The change its row WOULD take is then med3(g, lo - f, hi - f) (unilateral
    // rows: lo = 0, hi = inf; friction-loss rows: -+ the bound), the two bounds constant within the sweep.  A step is
    // v_med3, v_readlane, fma (+ the owner recording the g it saw): one broadcast instead of five, a three-instruction
    // dependent chain.  Forces and the cost decrease (mj_solPGS "improvement") are committed once per sweep.
    float f0 = fr[0];
    const float aii0 = diag[0], fl0 = lm[0], nainv0 = -dinv[0];
    const float lo0 = fl0 >= 0.f ? -fl0 : 0.f, hi0 = fl0 >= 0.f ? fl0 : __builtin_inff();
    float g0 = res[0] * nainv0;
    while (iter < max_iter) {
      const float nlo = lo0 - f0, nhi = hi0 - f0;
      float seen = g0;
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));   // keeps the per-row lane compares inside the sweep (else 64 SGPR pairs of masks spill)
      StaticFor<0, 64>::run([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (i >= nefc) return false;
        const float dl_ = bcast(__builtin_amdgcn_fmed3f(g0, nlo, nhi), i);
        seen = (lane_s == i) ? g0 : seen;
        asm volatile("" : "+v"(seen));   // select NOW: left alone the compiler keeps all 64 intermediate g alive and spills A
        g0 = fmaf(arow[i], dl_, g0);
        return true;
      });
      const float dl0 = __builtin_amdgcn_fmed3f(seen, nlo, nhi);
      const float improvement = -wsum(lane < nefc ? dl0 * aii0 * (0.5f * dl0 - seen) : 0.f);   // dl (dl A_ii / 2 + r), r = -g A_ii
      f0 = (dl0 == nlo) ? lo0 : ((dl0 == nhi) ? hi0 : f0 + dl0);                                 // exactly on the bound when clamped
      iter++;
      if (improvement * T.pgs_scale < T.tolerance) break;
    }
    fr[0] = f0;
  } else if (nefc <= 128) {
    // 65..128 rows (bodies on the floor — the envs that set the launch time): rows j and j + 64 per lane; the first 64 sweep
    // steps run as above on two register rows of A (columns 0..63), the steps for rows 64.. take their A rows from the
    // scratch, prefetched a block ahead
    float ar0[64], ar1[64];
#pragma unroll
    for (int i = 0; i < 64; i++) { ar0[i] = AR[i * MAXROW + lane]; ar1[i] = (lane + 64 < nefc) ? AR[i * MAXROW + lane + 64] : 0.f; }
    float f0 = fr[0], f1 = fr[1];
    const float aii0 = diag[0], fl0 = lm[0], aii1 = diag[1], fl1 = lm[1], nainv0 = -dinv[0], nainv1 = -dinv[1];
    const float lo0 = fl0 >= 0.f ? -fl0 : 0.f, hi0 = fl0 >= 0.f ? fl0 : __builtin_inff();
    const float lo1 = fl1 >= 0.f ? -fl1 : 0.f, hi1 = fl1 >= 0.f ? fl1 : __builtin_inff();
    float g0 = res[0] * nainv0, g1 = res[1] * nainv1;
    const bool has1 = lane + 64 < nefc;
    while (iter < max_iter) {
      const float nlo0 = lo0 - f0, nhi0 = hi0 - f0, nlo1 = lo1 - f1, nhi1 = hi1 - f1;
      float seen0 = g0, seen1 = g1;
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));
      StaticFor<0, 64>::run([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const float dl_ = bcast(__builtin_amdgcn_fmed3f(g0, nlo0, nhi0), i);
        seen0 = (lane_s == i) ? g0 : seen0;
        asm volatile("" : "+v"(seen0));
        g0 = fmaf(ar0[i], dl_, g0);
        g1 = fmaf(ar1[i], dl_, g1);
        return true;
      });
      for (int i0 = 64; i0 < nefc; i0 += PF) {
        float a0[PF], a1[PF];
#pragma unroll
        for (int q = 0; q < PF; q++) {
          const int i = i0 + q;
          a0[q] = (i < nefc) ? AR[i * MAXROW + lane] : 0.f;
          a1[q] = (i < nefc && has1) ? AR[i * MAXROW + lane + 64] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
          const int src = i0 + q - 64;
          if (i0 + q < nefc) {
            const float dl_ = bcast(__builtin_amdgcn_fmed3f(g1, nlo1, nhi1), src);
            seen1 = (lane_s == src) ? g1 : seen1;
            asm volatile("" : "+v"(seen1));
            g0 = fmaf(a0[q], dl_, g0);
            g1 = fmaf(a1[q], dl_, g1);
          }
        }
      }
      const float dl0 = __builtin_amdgcn_fmed3f(seen0, nlo0, nhi0), dl1 = has1 ? __builtin_amdgcn_fmed3f(seen1, nlo1, nhi1) : 0.f;
      const float improvement = -wsum(dl0 * aii0 * (0.5f * dl0 - seen0) + (has1 ? dl1 * aii1 * (0.5f * dl1 - seen1) : 0.f));
      f0 = (dl0 == nlo0) ? lo0 : ((dl0 == nhi0) ? hi0 : f0 + dl0);
      if (has1) f1 = (dl1 == nlo1) ? lo1 : ((dl1 == nhi1) ? hi1 : f1 + dl1);
      iter++;
      if (improvement * T.pgs_scale < T.tolerance) break;
    }
    fr[0] = f0; fr[1] = f1;
  } else {
    // 129..256 rows (bodies lying on the floor with many mesh / floor contacts — the DPCombinedEnv getup phases): four rows per
    // lane, the same scaled-residual step; the four A columns of a row are requested a block of PF rows ahead (the block's
    // owner register set is uniform: PF divides 64)
    float g[4], f[4], lo[4], hi[4], nainv[4], aii[4];
    bool has[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      has[m] = m < nr && lane + 64 * m < nefc;
      f[m] = fr[m]; aii[m] = diag[m]; nainv[m] = -dinv[m];
      lo[m] = lm[m] >= 0.f ? -lm[m] : 0.f;
      hi[m] = lm[m] >= 0.f ? lm[m] : __builtin_inff();
      g[m] = res[m] * nainv[m];
    }
    while (iter < max_iter) {
      float nlo[4], nhi[4], seen[4];
#pragma unroll
      for (int m = 0; m < 4; m++) { nlo[m] = lo[m] - f[m]; nhi[m] = hi[m] - f[m]; seen[m] = g[m]; }
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));
      for (int i0 = 0; i0 < nefc; i0 += PF) {
        float a[4][PF];
#pragma unroll
        for (int q = 0; q < PF; q++) {
          const int i = i0 + q;
#pragma unroll
          for (int m = 0; m < 4; m++) a[m][q] = (i < nefc && has[m]) ? AR[i * MAXROW + lane + 64 * m] : 0.f;
        }
#define PGS_BLOCK(M)                                                                        \
        _Pragma("unroll") for (int q = 0; q < PF; q++) {                                    \
          const int src = (i0 + q) & 63;                                                    \
          if (i0 + q < nefc) {                                                              \
            const float dl_ = bcast(__builtin_amdgcn_fmed3f(g[M], nlo[M], nhi[M]), src);    \
            seen[M] = (lane_s == src) ? g[M] : seen[M];                                     \
            asm volatile("" : "+v"(seen[M]));                                               \
            _Pragma("unroll") for (int m = 0; m < 4; m++) g[m] = fmaf(a[m][q], dl_, g[m]); \
          }                                                                                 \
        }
        switch (i0 >> 6) {
          case 0: PGS_BLOCK(0) break;
          case 1: PGS_BLOCK(1) break;
          case 2: PGS_BLOCK(2) break;
          default: PGS_BLOCK(3) break;
        }
#undef PGS_BLOCK
      }
      float imp = 0.f;
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const float dl = has[m] ? __builtin_amdgcn_fmed3f(seen[m], nlo[m], nhi[m]) : 0.f;
        imp += dl * aii[m] * (0.5f * dl - seen[m]);
        if (has[m]) f[m] = (dl == nlo[m]) ? lo[m] : ((dl == nhi[m]) ? hi[m] : f[m] + dl);
      }
      const float improvement = -wsum(imp);
      iter++;
      if (improvement * T.pgs_scale < T.tolerance) break;
    }
#pragma unroll
    for (int m = 0; m < 4; m++) fr[m] = f[m];
  }
  PROF(12);
#pragma unroll
  for (int m = 0; m < 4; m++) { const int r = lane + 64 * m; if (m < nr && r < nefc) e_f[r] = fr[m]; }
  SYNC();
  // qfrc_constraint = J^T f: lane k owns dof k and walks its row of J^T (eight loads in flight; 43 wave reductions of row
  // products were 1 700 instructions per evaluation)
  if (lane < NV) {
    const float *jk = JT + lane * MAXROW;
    float s = 0;
    for (int r0 = 0; r0 < nefc; r0 += 8) {
      float a[8], fc[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const bool ok = r0 + q < nefc;
        a[q] = ok ? jk[r0 + q] : 0.f;
        fc[q] = ok ? e_f[r0 + q] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; q++) s = fmaf(a[q], fc[q], s);
    }
    S.qfc[lane] = s; S.tmp[lane] = s;
  }
  SYNC();
  solve_m(T, S.tmp, lane);
  SYNC();
  if (lane < NV) { const float a = S.tmp[lane] + S.qas[lane]; S.qacc[lane] = a; S.warm[lane] = a; }
  if (lane == 0) S.info[3] = iter;
  SYNC();
}

// One forward evaluation = a per-env first half (poses, inertia, factorisation, smooth dynamics), the collision stage, and a
// per-env second half (constraint rows, A, PGS).  The monolithic kernel runs all three on the env's wave; the split pipeline runs
// the halves in g1_env_kernel and the narrowphase of ALL envs' pairs in g1_pair_kernel (one wave per pair).
__device__ __forceinline__ void forward_pre(const Dev &T, const int lane, const bool qlo) {
  PROF(15);
  kinematics(T, lane, qlo);
  com_pos(T, lane);
  PROF(0);
  crb_factor(T, lane);
  PROF(1);
  fwd_smooth(T, lane);   // before the collision stage: its scratch shares LDS with the contact arrays
  PROF(2);
}
__device__ __noinline__ void project_constraint(const Dev &T, const float *JT, float *BT, float *AR, const float *RW, const int nefc, const int lane) {
  project_constraint_body(T, JT, BT, AR, RW, nefc, lane);
}
__device__ __noinline__ void fwd_constraint(const Dev &T, const float *JT, const float *BT, const float *AR, float *RW, const int nefc, const int lane,
                                             const int max_iter) {
  fwd_constraint_body(T, JT, BT, AR, RW, nefc, lane, max_iter);
}

template <bool INL>   // INL: the two constraint phases inlined (split env kernel) / called out of line (monolithic kernel)
__device__ __forceinline__ void forward_post(const Launch &P, const Dev &T, const int env, const int lane, const int ncon) {
  float *JT = P.jt + (size_t)env * 44 * MAXROW, *BT = P.bt + (size_t)env * 44 * MAXROW, *AR = P.ar + (size_t)env * MAXROW * MAXROW;
  float *RW = P.rows + (size_t)env * 5 * MAXROW;
  int nefc = 0;
  PROF(7);
  if (!(P.pad & 8)) nefc = make_constraint(T, JT, RW, ncon, lane);
  else { if (lane == 0) { S.info[1] = 0; S.info[2] = 0; } SYNC(); }
  PROF(8);
  if (!(P.pad & 4)) { if (INL) project_constraint_body(T, JT, BT, AR, RW, nefc, lane); else project_constraint(T, JT, BT, AR, RW, nefc, lane); }
  PROF(9);
  if (P.pad & 16) nefc = 0;
  if (INL) fwd_constraint_body(T, JT, BT, AR, RW, nefc, lane, (P.pad & 1) ? 0 : T.iterations);
  else fwd_constraint(T, JT, BT, AR, RW, nefc, lane, (P.pad & 1) ? 0 : T.iterations);
  PROF(10);
}
__device__ __noinline__ void forward(const Launch &P, const Dev &T, const int env, const int lane, const bool qlo) {
  forward_pre(T, lane, qlo);
  int ncon = 0;
  if (!(P.pad & 2)) ncon = collide(T, P, env, lane);
  else { if (lane == 0) { S.info[0] = 0; S.info[4] = 0; } SYNC(); }
  forward_post<false>(P, T, env, lane, ncon);
}

// [EXT] mj_integratePos from the step's start state x0q (+ the low words of its normalised root quaternion), in fp64.  `lo`: keep
// the low words of the result for the next evaluation's pose chain (RK stages 2..4); the step's final state is the rounded value.
__device__ void integrate_pos(const Dev &T, const float *vel, const double a, const bool lo, const int lane) {
  const double h = a * T.timestep_d;
  if (lane == 0) {
    for (int i = 0; i < 3; i++) {
      float hi, l;
      split_hi_lo((double)S.x0q[i] + h * (double)vel[i], hi, l);
      S.qpos[i] = hi;
      if (lo) S.u.rk.qlo[i] = l;
    }
    double w[3] = {(double)vel[3], (double)vel[4], (double)vel[5]};
    double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (n < MINVAL) { w[0] = 1; w[1] = w[2] = 0; n = 0; } else { w[0] /= n; w[1] /= n; w[2] /= n; }
    const double ang = h * n;
    double sn, cs;
    sincos(0.5 * ang, &sn, &cs);
    double qr[4] = {cs, w[0] * sn, w[1] * sn, w[2] * sn}, qo[4], qn[4];
    for (int i = 0; i < 4; i++) qo[i] = (double)S.x0q[3 + i] + (double)S.x0q_lo[i];
    dquat_normalize(qo);
    dquat_mul(qn, qo, qr);
    dquat_normalize(qn);
    for (int i = 0; i < 4; i++) {
      float hi, l;
      split_hi_lo(qn[i], hi, l);
      S.qpos[3 + i] = hi;
      if (lo) S.u.rk.qlo[3 + i] = l;
    }
  }
  if (lane >= 6 && lane < NV) {
    float hi, l;
    split_hi_lo((double)S.x0q[lane + 1] + h * (double)vel[lane], hi, l);
    S.qpos[lane + 1] = hi;
    if (lo) S.u.rk.qlo[lane + 1] = l;
  }
}

// What a step carries from one evaluation to the next besides the LDS working set (registers in the monolithic kernel, a few
// words of the env's HBM slot between the launches of the split pipeline)
struct StepCtx {
  int idx_curr, ep_len, rcnt, motion, clip_id, reason, work, stage;   // stage: RK4 stage of the evaluation about to run / in flight
  unsigned stage_ncon, stage_nefc_lo;                                  // byte i = count at RK stage i (debug)
  float ep_rew, reward;
  bool sim_err, done, after_reset;
  ClipDev clip;
};

// Loads the env's state row, applies the launch mode (action / forced state / reset state) and runs mj_checkPos / mj_checkVel.
// false: this env has nothing (more) to do in this launch.
__device__ __forceinline__ bool step_enter(const Launch &P, const Dev &T, const int env, const int lane, StepCtx &X) {
  const int mode = P.mode;
  if (mode == MODE_RESET && P.mask && !P.mask[env]) return false;
  float *st = P.state + (size_t)env * STATE;
  int *sti = (int *)st;
  X.idx_curr = sti[S_IDX]; X.ep_len = sti[S_EPLEN]; X.rcnt = sti[S_RCNT];
  X.ep_rew = st[S_EPREW];
  // TASK (DPCombinedEnv, src/combined_env.py): per-env motion 0 walk, 1 run, 2 getup, 3 to_getup; idx_curr is then the
  // unwrapped current_motion_n_steps
  const bool TASK = P.task != 0;
  X.motion = TASK ? sti[S_MOTION] : 0;
  X.motion = (X.motion < 0 || X.motion > 3) ? 0 : X.motion;
  // DPEnv task: the same slot holds the env's clip id (dmg1_set_env_clips; one DPEnv(motion=...) per worker in the reference)
  X.clip_id = TASK ? 0 : sti[S_MOTION];
  X.clip_id = (X.clip_id < 0 || X.clip_id >= DMG1_MAX_CLIPS) ? 0 : X.clip_id;
  const int NOBS_T = TASK ? NOBS_C : NOBS, NTERMS = TASK ? 8 : 5;
  if (TASK && (P.clips[0].L < 1 || P.clips[1].L < 1 || P.clips[2].L < 2)) return false;   // walk, run, getup all needed
  if (TASK) X.idx_curr = X.idx_curr < 0 ? 0 : X.idx_curr;
  if (lane < NQ) S.qpos[lane] = st[S_QPOS + lane];
  if (lane < NV) { S.qvel[lane] = st[S_QVEL + lane]; S.warm[lane] = st[S_WARM + lane]; }
  if (lane < NU) S.ctrl[lane] = st[S_CTRL + lane];
  SYNC();
  X.clip = P.clips[TASK ? (X.motion == 3 ? 2 : X.motion) : X.clip_id];
  if (X.clip.L < 1) return false;   // no clip loaded under this env's clip id
  if (mode == MODE_STEP) {
    if (lane < NU) S.ctrl[lane] = (lane < NACT) ? P.actions[(size_t)env * NACT + lane] * T.action_scale : 0.f;   // :348-351
  } else if (mode == MODE_FORCED || mode == MODE_SETSTATE) {
    if (P.in_qpos) {
      if (lane < NQ) S.qpos[lane] = P.in_qpos[(size_t)env * NQ + lane];
      if (lane < NV) S.qvel[lane] = P.in_qvel[(size_t)env * NV + lane];
    }
    if (mode == MODE_SETSTATE && P.in_warm && lane < NV) S.warm[lane] = P.in_warm[(size_t)env * NV + lane];
  } else if (mode == MODE_RESET) {
    int fi;
    if (TASK) {   // DPCombinedEnv.reset(rsi=True) (:219-227): walk with amnesty or getup, random frame; idx_init keeps the motion
      if (P.idx_init) X.idx_curr = P.idx_init[env] < 0 ? 0 : P.idx_init[env];
      else {
        X.motion = (hash32(P.seed, env, X.rcnt, 0x5EED) & 1) ? 2 : 0;
        X.clip = P.clips[X.motion];
        X.idx_curr = (int)(hash32(P.seed, env, X.rcnt, 0x5EEE) % (uint32_t)X.clip.L) + (X.motion == 0 ? P.amnesty_steps + 10 : 0);
      }
      fi = (X.motion == 3) ? 1 : X.idx_curr % X.clip.L;
    } else {
      fi = P.idx_init ? P.idx_init[env] : (int)(hash32(P.seed, env, X.rcnt, 0x5EED) % (uint32_t)X.clip.L);
      fi = fi < 0 ? 0 : (fi >= X.clip.L ? X.clip.L - 1 : fi);
      X.idx_curr = fi;
    }
    const float *rr = X.clip.reset + (size_t)fi * 88;
    if (lane < NQ) S.qpos[lane] = rr[lane];
    if (lane < NV) S.qvel[lane] = rr[44 + lane];
    X.ep_len = 0; X.ep_rew = 0; X.rcnt++;
  }
  SYNC();
  if (mode == MODE_SETSTATE && !P.run_forward) {
    if (lane < NQ) st[S_QPOS + lane] = S.qpos[lane];
    if (lane < NV) { st[S_QVEL + lane] = S.qvel[lane]; st[S_WARM + lane] = S.warm[lane]; }
    return false;
  }

  X.sim_err = false; X.done = false; X.after_reset = false;
  X.reason = 0; X.stage_ncon = 0; X.stage_nefc_lo = 0; X.work = 0; X.reward = 0; X.stage = 0;
  if (mode == MODE_STEP || mode == MODE_FORCED) {   // mj_checkPos / mj_checkVel
    const float a = (lane < NQ) ? S.qpos[lane] : 0.f, b = (lane < NV) ? S.qvel[lane] : 0.f;
    X.sim_err = __any(!(fabsf(a) <= MAXVALF) || !(fabsf(b) <= MAXVALF));
  }
  return true;
}

// MODE_STEP, after the evaluation of RK stage stage ([EXT] mj_step with mj_RungeKutta(4): A = (1/2, 1/2, 1), B = (1/6, 1/3, 1/3,
// 1/6)): accumulates, then leaves either the next stage's state in LDS (true: evaluate again) or the step's final state (false).
__device__ __forceinline__ bool rk_advance(const Launch &P, const Dev &T, const int env, const int lane, StepCtx &X) {
  const float h = T.timestep;
        const float Bw = (X.stage == 0 || X.stage == 3) ? 1.f / 6 : 1.f / 3;
        X.stage_ncon |= ((unsigned)S.info[0] & 0xFF) << (8 * X.stage); X.stage_nefc_lo |= ((unsigned)S.info[1] & 0xFF) << (8 * X.stage);
        if (P.debug) {   // 24-bit hash of the stage's contact list (geom pairs in order): the parity tests hold the index SETS bit-exact
          unsigned hsh = 0;
          for (int c = 0; c < S.info[0]; c++) hsh = (hsh * 131u + (unsigned)S.u.co.c_g1[c] * 97u + (unsigned)S.u.co.c_g2[c] + 1u) & 0xFFFFFFu;
          if (lane == 0) P.debug[(size_t)env * DMG1_DEBUG_STRIDE + 1012 + X.stage] = (float)hsh;
        }
        if (X.stage == 0) {
          const bool badv = (lane < NV) && !(fabsf(S.qacc[lane]) <= MAXVALF);   // mj_checkAcc
          X.sim_err = __any(badv);
          if (!X.sim_err) {
            if (lane < NQ) S.x0q[lane] = S.qpos[lane];
            if (lane < NV) { S.x0v[lane] = S.qvel[lane]; S.accq[lane] = 0.f; S.accv[lane] = 0.f; }
          }
        }
        if (!X.sim_err) {
          float dq = 0, dv = 0;
          if (lane < NV) {
            S.accq[lane] += Bw * S.qvel[lane]; S.accv[lane] += Bw * S.qacc[lane];
            const float a = (X.stage == 2) ? 1.f : 0.5f;
            dq = a * S.qvel[lane]; dv = a * S.qacc[lane];
          }
          SYNC();
          if (X.stage < 3) {
            if (lane < NV) S.tmp[lane] = dq;
            SYNC();
            integrate_pos(T, S.tmp, 1.0, true, lane);
            if (lane < NV) S.qvel[lane] = S.x0v[lane] + h * dv;
            SYNC();
            X.stage++;
            return true;
          }
          if (lane < NV) S.qvel[lane] = S.x0v[lane] + h * S.accv[lane];
          integrate_pos(T, S.accq, 1.0, false, lane);
          SYNC();
        }
  return false;
}

// Task layer of the last evaluation (observation, reward, termination, counters, outputs; SURVEY F6: the derived arrays are those
// of the LAST forward evaluation).  true: the env was reset inside the step (VecEnv auto-reset) and needs one more forward
// evaluation at the reset state (set_state -> sim.forward) before its observation can be written.
__device__ __forceinline__ bool task_and_finish(const Launch &P, const Dev &T, const int env, const int lane, StepCtx &X) {
  const int mode = P.mode;
  const bool TASK = P.task != 0;
  const int NOBS_T = TASK ? NOBS_C : NOBS, NTERMS = TASK ? 8 : 5;
    // ---- task layer (derived arrays are those of the LAST forward evaluation)
    const bool task_pass = !X.after_reset && (mode == MODE_STEP || mode == MODE_FORCED);
    float obs_a = 0, obs_b = 0;   // obs[lane], obs[64 + lane]
    float terms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (X.sim_err) {
      X.reward = 0; X.done = true; X.reason = 5;
      SYNC();
      if (lane < NQ) S.qpos[lane] = T.qpos0[lane];
      if (lane < NV) { S.qvel[lane] = 0; S.warm[lane] = 0; }
      if (lane < NU) S.ctrl[lane] = 0;
      SYNC();
    } else {
      const float Sc = P.vel_obs_scale;
      const int tb = T.torso_body;
      float rpy[3], tq[4] = {S.xquat[tb][0], S.xquat[tb][1], S.xquat[tb][2], S.xquat[tb][3]};
      quat_to_rpy(tq, rpy);
      const float *cv = S.cvel[tb];
      float sy, cy;
      sincosf(-rpy[2], &sy, &cy);
      const float tor[8] = {rpy[0] * Sc, rpy[1] * Sc, (cy * cv[3] - sy * cv[4]) * Sc, (sy * cv[3] + cy * cv[4]) * Sc, cv[5] * Sc,
                            cv[0] * Sc, cv[1] * Sc, cv[2] * Sc};
      float rf = 0, lf = 0;
      unsigned xc = 0;   // bit k: extra-contact geom k (foot spheres, src/config.py:19-20) touches the floor
      for (int c = 0; c < S.info[0]; c++) {
        const int g1 = S.u.co.c_g1[c], g2 = S.u.co.c_g2[c];
        const bool fl = g1 == T.floor_geom || g2 == T.floor_geom;
        if ((g1 == T.rfoot_geom || g2 == T.rfoot_geom) && fl) rf = 1;
        if ((g1 == T.lfoot_geom || g2 == T.lfoot_geom) && fl) lf = 1;
        if (fl) for (int k = 0; k < 8; k++) if (g1 == T.extra_geom[k] || g2 == T.extra_geom[k]) xc |= 1u << k;
      }
      // motion length / frame: to_getup is a 180-step pseudo clip whose target is frame 1 of getup (combined_env.py:67-99)
      const int Lm = (TASK && X.motion == 3) ? P.to_getup_len : X.clip.L;
      const int frame = TASK ? ((X.motion == 3) ? 1 : X.idx_curr % X.clip.L) : X.idx_curr;
      float ph = TASK ? (float)(X.idx_curr % Lm) / (float)Lm : (float)X.idx_curr / (float)X.clip.L;
      ph = fminf(fmaxf(ph, 0.f), 1.f);
      auto obs_at = [&](int i) -> float {
        if (i < 37) return S.qpos[7 + i];
        if (i < 74) return S.qvel[6 + i - 37] * Sc;
        if (i < 82) return tor[i - 74];
        if (!TASK) return (i == 82) ? rf : (i == 83) ? lf : ph;
        // DPCombinedEnv._get_obs (:495-505): extra contacts (:27), phase, get_player_action_obs with PAWalk
        // (heading (1,0,0) in the yaw-aligned torso frame, one-hot index 0), pa_getup_state
        if (i < 90) return (float)((xc >> (i - 82)) & 1u);
        if (i == 90) return ph;
        if (i == 91) return cy;
        if (i == 92) return sy;
        if (i == 93) return 1.f;
        if (i == 96) return (X.motion == 3) ? 1.f : 0.f;
        if (i == 97) return (X.motion == 2) ? 1.f : 0.f;
        return 0.f;
      };
      obs_a = obs_at(lane);
      if (lane < NOBS_T - 64) obs_b = obs_at(64 + lane);
      if (task_pass) {
        // ---- calc_imitation_reward, unitree_g1 branch (:193-256)
        const float *cr = X.clip.rows + (size_t)frame * CLIP_ROW;
        float e_cfg = 0, e_vel = 0, adiff = 0;
        int viol = 0;
        if (lane < NREW) {
          const int qi = T.rew_q[lane], vi = T.rew_v[lane];
          const float q = S.qpos[qi];
          e_cfg = fabsf(q - cr[lane]);
          adiff = e_cfg;
          e_vel = fabsf(cr[23 + lane] - S.qvel[vi]);
          const int k = vi;
          viol = (q <= T.d_lo[k] * 0.99f) + (q >= T.d_hi[k] * 0.99f);
        }
        e_cfg = wsum(e_cfg); e_vel = wsum(e_vel);
        const float dsum = e_cfg;
        const float nviol = wsum((float)viol);
        float rc[3], rt[3], cq[4] = {S.qpos[3], S.qpos[4], S.qpos[5], S.qpos[6]}, tqq[4] = {cr[46], cr[47], cr[48], cr[49]};
        quat_to_rpy(cq, rc);
        quat_to_rpy(tqq, rt);
        e_cfg += fabsf(rc[1] - rt[1]);
        float ee = 0;
        for (int e = 0; e < 4; e++) {
          const int g = T.ee_geom[e];
          for (int i = 0; i < 3; i++) { const float df = S.gpos[g][i] - cr[50 + 3 * e + i]; ee += df * df; }
        }
        float cc[3];
        for (int i = 0; i < 3; i++) cc[i] = wsum((lane < NB) ? T.b_mass[lane] * S.xpos[lane][i] : 0.f) * T.total_mass_inv;
        const float *tc = X.clip.com + (size_t)frame * 4;
        float ce = 0;
        for (int i = 0; i < 3; i++) { const float df = tc[i] - cc[i]; ce += df * df; }
        terms[0] = expf(-e_cfg); terms[1] = expf(-0.1f * e_vel); terms[2] = expf(-40.f * ee); terms[3] = expf(-10.f * ce);
        terms[4] = nviol / (float)NREW;
        X.reward = 0.75f * terms[0] + 0.1f * terms[1] + 0.15f * terms[2] + 0.0f * terms[3] - 0.1f * terms[4];
        // ---- termination (:418-442)
        const float zc = wsum((lane < NB) ? T.b_mass[lane] * S.xipos[lane][2] : 0.f) * T.total_mass_inv;
        if (TASK) {
          // ---- DPCombinedEnv: task reward (combined_env.py:338-354), motion state machine + termination (:393-445)
          const float ALIM = 0.2617993877991494f, MAX_ANGLE = 1.0471975511965976f;   // deg2rad(15), deg2rad(60)
          const float droll = fabsf(rc[0] - rt[0]), dpitch = fabsf(rc[1] - rt[1]);
          float imitation = X.reward, task_r = 0.f;
          if (X.motion == 0 || X.motion == 1) {   // heading + velocity error against the clip's root velocity
            const float ex = cr[62] - S.qvel[0], ey = cr[63] - S.qvel[1];
            task_r = expf(-sqrtf(ex * ex + ey * ey) * 10.f);
          }
          if (X.motion == 3) { imitation = 0.f; task_r = expf(-(dsum + dpitch + droll) / 5.f) / 3.f; }
          X.reward = imitation * 0.7f + task_r * 0.3f;
          const unsigned long long badm = __ballot(adiff > ALIM);
          const bool all_close = !__any(!(adiff < ALIM));   // lanes >= 23 carry 0
          terms[5] = imitation; terms[6] = task_r;
          terms[7] = (float)(__popcll(badm) + (dpitch > ALIM ? 1 : 0) + (droll > ALIM ? 1 : 0));   // debug_n_bad_angles
          X.done = false; X.reason = 0;
          if (X.idx_curr >= Lm - 1) {   // out of time; :396 compares PlayerAction objects by identity: getup always hands over to run
            if (X.motion == 2) { X.motion = 1; X.idx_curr = 0; }
            if (X.motion == 3) { X.motion = 2; X.idx_curr = 0; }
          }
          if (dpitch < ALIM && droll < ALIM && all_close && X.motion == 3) { X.motion = 2; X.idx_curr = 0; }
          if (X.motion == 0 || X.motion == 1) {
            const bool fallen = (zc < T.low_z) || (zc > P.high_z) || (droll > MAX_ANGLE) || (dpitch > MAX_ANGLE);
            if (fallen) {
              if (!(X.idx_curr > P.amnesty_steps)) { X.done = true; X.reason = 7; }
              X.motion = 3; X.idx_curr = 0;
            }
          }
          if (P.max_ep_length != 0 && X.ep_len >= P.max_ep_length) { X.done = true; X.reason = 3; }
          X.clip = P.clips[X.motion == 3 ? 2 : X.motion];
          X.idx_curr += 1;   // :454 (not wrapped)
        } else {
        if (!(X.clip.flags & 1)) {
          X.done = (zc < T.low_z) || (zc > P.high_z);
          X.reason = (zc < T.low_z) ? 1 : 2;
        }
        if (X.clip.flags & 4) {
          const float mx = 60.f * 3.14159265358979f / 180.f;
          if (fabsf(rc[0] - rt[0]) > mx || fabsf(rc[1] - rt[1]) > mx) { X.done = true; X.reason = 8; }
        }
        if (P.max_ep_length != 0 && X.ep_len >= P.max_ep_length) { X.done = true; X.reason = 3; }
        if ((X.clip.flags & 2) && X.idx_curr + 1 == X.clip.L) { X.done = true; X.reason = 4; }
        X.idx_curr = (X.idx_curr + 1) % X.clip.L;
        }
        X.ep_rew += X.reward;
        X.ep_len += 1;
        const bool ob = !(fabsf(obs_a) <= P.obs_bound) || !(fabsf(obs_b) <= P.obs_bound);
        if (__any(ob)) {
          obs_a = 0; obs_b = 0; X.reward = 0; X.done = true; X.reason = 6;
          for (int i = 0; i < 8; i++) terms[i] = 0;
        }
      }
    }
    if (P.debug && !X.after_reset) {
      float *dbg = P.debug + (size_t)env * DMG1_DEBUG_STRIDE;
      for (int i = lane; i < NB * 3; i += 64) dbg[i] = (&S.xpos[0][0])[i];
      if (lane < NV) { dbg[117 + lane] = S.qas[lane]; dbg[160 + lane] = S.qacc[lane]; }
      if (lane == 0) { dbg[1008] = (float)S.info[5]; dbg[1009] = (float)(S.info[6] & 0xFF); dbg[1010] = (float)((S.info[6] >> 8) & 0xFF); dbg[1011] = (float)((S.info[6] >> 16) & 0xFF);
                       dbg[203] = S.info[0]; dbg[204] = S.info[1]; dbg[205] = S.info[3]; dbg[206] = S.info[2]; dbg[207] = S.info[4];
                       for (int i = 0; i < 4; i++) { dbg[1000 + i] = (float)((X.stage_ncon >> (8 * i)) & 0xFF); dbg[1004 + i] = (float)((X.stage_nefc_lo >> (8 * i)) & 0xFF); } }
      if (lane < MAXCON) {
        float *o = dbg + 208 + 9 * lane;
        const bool on = lane < S.info[0];
        o[0] = on ? S.u.co.c_dist[lane] : 0.f; o[1] = on ? (float)S.u.co.c_g1[lane] : -1.f; o[2] = on ? (float)S.u.co.c_g2[lane] : -1.f;
        for (int i = 0; i < 3; i++) { o[3 + i] = on ? S.u.co.c_pos[lane][i] : 0.f; o[6 + i] = on ? S.u.co.c_frame[lane][i] : 0.f; }
      }
      for (int r = lane; r < MAXROW; r += 64) dbg[640 + r] = (r < S.info[1]) ? P.rows[(size_t)env * 5 * MAXROW + 2 * MAXROW + r] : 0.f;
    }
    if (task_pass) {
      if (P.rew && lane == 0) P.rew[env] = X.reward;
      if (P.done && lane == 0) P.done[env] = X.done ? 1 : 0;
      if (P.reason && lane == 0) P.reason[env] = X.reason;
      if (P.terms && lane < NTERMS) P.terms[(size_t)env * NTERMS + lane] = (lane == 0) ? terms[0] : (lane == 1) ? terms[1] : (lane == 2) ? terms[2]
                                                              : (lane == 3) ? terms[3] : (lane == 4) ? terms[4] : (lane == 5) ? terms[5] : (lane == 6) ? terms[6] : terms[7];
      if (X.done && P.auto_reset && mode == MODE_STEP) {
        if (P.terminal_obs) {
          P.terminal_obs[(size_t)env * NOBS_T + lane] = obs_a;
          if (lane < NOBS_T - 64) P.terminal_obs[(size_t)env * NOBS_T + 64 + lane] = obs_b;
        }
        int fi;
        if (TASK) {   // DPCombinedEnv.reset(rsi=True) (:219-227)
          X.motion = (hash32(P.seed, env, X.rcnt, 0x5EED) & 1) ? 2 : 0;
          X.clip = P.clips[X.motion];
          X.idx_curr = (int)(hash32(P.seed, env, X.rcnt, 0x5EEE) % (uint32_t)X.clip.L) + (X.motion == 0 ? P.amnesty_steps + 10 : 0);
          fi = X.idx_curr % X.clip.L;
        } else {
          fi = (int)(hash32(P.seed, env, X.rcnt, 0x5EED) % (uint32_t)X.clip.L);
          X.idx_curr = fi;
        }
        X.rcnt++;
        const float *rr = X.clip.reset + (size_t)fi * 88;
        SYNC();
        if (lane < NQ) S.qpos[lane] = rr[lane];
        if (lane < NV) S.qvel[lane] = rr[44 + lane];
        X.ep_len = 0; X.ep_rew = 0;
        SYNC();
        X.after_reset = true;
        X.sim_err = false;
        return true;   // one more forward evaluation at the reset state (set_state -> sim.forward)
      }
    }
    if (P.obs) {
      P.obs[(size_t)env * NOBS_T + lane] = obs_a;
      if (lane < NOBS_T - 64) P.obs[(size_t)env * NOBS_T + 64 + lane] = obs_b;
    }
  return false;
}

__device__ __forceinline__ void step_write_back(const Launch &P, const int env, const int lane, const StepCtx &X) {
  const int mode = P.mode;
  const bool TASK = P.task != 0;
  float *st = P.state + (size_t)env * STATE;
  int *sti = (int *)st;
  if (lane < NQ) st[S_QPOS + lane] = S.qpos[lane];
  if (lane < NV) { st[S_QVEL + lane] = S.qvel[lane]; st[S_WARM + lane] = S.warm[lane]; }
  if (lane < NU) st[S_CTRL + lane] = S.ctrl[lane];
  if (lane == 0) { sti[S_IDX] = X.idx_curr; sti[S_EPLEN] = X.ep_len; st[S_EPREW] = X.ep_rew; sti[S_RCNT] = X.rcnt; if (TASK) sti[S_MOTION] = X.motion; }
  if (P.cost && mode == MODE_STEP && lane == 0) P.cost[env] = X.work;
}

// The monolithic kernel: one wave runs the whole step (or the single evaluation of the other modes) of its env.
extern "C" __global__ void __launch_bounds__(64, 2) g1_step_kernel(Launch P) {
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= P.N) return;
  const int env = (P.order && P.mode == MODE_STEP) ? P.order[blockIdx.x] : (int)blockIdx.x;
  const Dev &T = *P.T;
  const int mode = P.mode;
#ifdef G1_PROFILE
  if (lane == 0) for (int i = 0; i < 16; i++) S.prof[i] = 0;
  S.prof_t = clock64();
#endif
  StepCtx X;
  if (!step_enter(P, T, env, lane, X)) return;
  for (;;) {
    if (!X.sim_err) {
      forward(P, T, env, lane, mode == MODE_STEP && !X.after_reset && X.stage > 0);   // the only call site: the evaluation is ~20 k instructions
      X.work += 4000 + S.info[1] * (8 + 6 * S.info[3]) + 3000 * ((S.info[6] >> 16) & 0xFF);   // fixed part, rows x sweeps, MPR pairs
      if (mode == MODE_STEP && !X.after_reset) {
        if (rk_advance(P, T, env, lane, X)) continue;
      } else if (mode == MODE_FORCED && !X.after_reset) {
        const bool badv = (lane < NV) && !(fabsf(S.qacc[lane]) <= MAXVALF);
        X.sim_err = __any(badv);
      }
    }
    if (task_and_finish(P, T, env, lane, X)) continue;
    break;
  }
#ifdef G1_PROFILE
  PROF(14);
  SYNC();
  if (P.debug && lane < 16) P.debug[(size_t)env * DMG1_DEBUG_STRIDE + 900 + lane] = (float)S.prof[lane];
#endif
  step_write_back(P, env, lane, X);
}

// ------------------------------------------------------------------------------------------ split pipeline (MODE_STEP)
// At 4 096 envs the monolithic launch lasts as long as its heaviest env (18 M cycles against a 4.1 M mean, 70 % of it MPR pairs run
// one after the other on that env's wave).  The split pipeline balances the collision stage across the BATCH: per evaluation
//   g1_env_kernel   (one wave per env, longest-first): [second half of the previous evaluation: contacts of the pair kernel ->
//                   constraint rows, A, PGS; RK4 bookkeeping / task layer] then [first half of the next one: poses, inertia,
//                   factorisation, smooth dynamics, broadphase]; the survivors' staged geoms go to the pair queue
//   g1_pair_kernel  (persistent waves pulling tickets, support-query pairs first): narrowphase of ONE pair per ticket, four waves
//                   per SIMD (its own register budget and 1.8 KB of LDS) instead of two
// so a step is 6 env launches + 5 pair launches (4 RK stages + the evaluation at the reset state of envs that ended).  Between
// launches an env keeps the non-union head of its LDS working set and its StepCtx in an HBM slot (14.3 KB each way: ~0.12 ms per
// step at 4 096 envs).  Arithmetic and order of operations are those of the monolithic kernel: trajectories are bit-identical
// (tests/test_g1_gpu.py); only the separating-direction cache differs (one entry per (env, pair); result-neutral either way).
constexpr int WS_HEAD = (int)offsetof(Lds, u);
static_assert(WS_HEAD % 16 == 0, "the head of the working set is copied in 16-byte words");
constexpr int WS_CTX_OFF = WS_HEAD, WS_BYTES = ((WS_HEAD + 128 + 255) / 256) * 256;
enum { WC_IDX = 0, WC_EPLEN, WC_RCNT, WC_MOTION, WC_CLIPID, WC_REASON, WC_WORK, WC_STAGE, WC_SNCON, WC_SNEFC, WC_EPREW, WC_REWARD, WC_FLAGS,
       WC_FINISHED = 15 };

__device__ __forceinline__ void ws_save(char *ws, const StepCtx &X, const int lane) {
  SYNC();
  uint4 *dst = reinterpret_cast<uint4 *>(ws);
  const uint4 *src = reinterpret_cast<const uint4 *>(&S);
  for (int i = lane; i < WS_HEAD / 16; i += 64) dst[i] = src[i];
  if (lane == 0) {
    int *w = reinterpret_cast<int *>(ws + WS_CTX_OFF);
    w[WC_IDX] = X.idx_curr; w[WC_EPLEN] = X.ep_len; w[WC_RCNT] = X.rcnt; w[WC_MOTION] = X.motion; w[WC_CLIPID] = X.clip_id;
    w[WC_REASON] = X.reason; w[WC_WORK] = X.work; w[WC_STAGE] = X.stage; w[WC_SNCON] = (int)X.stage_ncon; w[WC_SNEFC] = (int)X.stage_nefc_lo;
    w[WC_EPREW] = __float_as_int(X.ep_rew); w[WC_REWARD] = __float_as_int(X.reward);
    w[WC_FLAGS] = (X.sim_err ? 1 : 0) | (X.done ? 2 : 0) | (X.after_reset ? 4 : 0);
    w[WC_FINISHED] = 0;
  }
}
__device__ __forceinline__ void ws_load(const Launch &P, const char *ws, StepCtx &X, const int lane) {
  const uint4 *src = reinterpret_cast<const uint4 *>(ws);
  uint4 *dst = reinterpret_cast<uint4 *>(&S);
  for (int i = lane; i < WS_HEAD / 16; i += 64) dst[i] = src[i];
  const int *w = reinterpret_cast<const int *>(ws + WS_CTX_OFF);
  X.idx_curr = w[WC_IDX]; X.ep_len = w[WC_EPLEN]; X.rcnt = w[WC_RCNT]; X.motion = w[WC_MOTION]; X.clip_id = w[WC_CLIPID];
  X.reason = w[WC_REASON]; X.work = w[WC_WORK]; X.stage = w[WC_STAGE]; X.stage_ncon = (unsigned)w[WC_SNCON]; X.stage_nefc_lo = (unsigned)w[WC_SNEFC];
  X.ep_rew = __int_as_float(w[WC_EPREW]); X.reward = __int_as_float(w[WC_REWARD]);
  const int fl = w[WC_FLAGS];
  X.sim_err = fl & 1; X.done = (fl >> 1) & 1; X.after_reset = (fl >> 2) & 1;
  X.clip = P.clips[P.task ? (X.motion == 3 ? 2 : X.motion) : X.clip_id];
  SYNC();
}

// broadphase of the evaluation; the survivors (canonical order) go to the env's block of the pair queue: pair ids, staged geoms, and
// one ticket each in the list of their cost class
__device__ __forceinline__ void emit_pairs(const Dev &T, const Launch &P, const int env, const int lane) {
  int overflow = 0;
  const int nsurv = broadphase(T, PAIRCAP, overflow, lane);
  SYNC();
  PROF(3);
  const size_t base = (size_t)env * PAIRCAP;
  int32_t *ctr = P.qctr + 4 * P.round, *tickA = P.tick, *tickB = P.tick + (size_t)P.N * PAIRCAP;
  int n_an = 0, n_pm = 0, n_mpr = 0;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int b = 0; b < nsurv; b += 64) {
    const int s = b + lane;
    const bool on = s < nsurv;
    int cls = 0;
    if (on) {
      const int p = S.u.co.surv[s];
      cls = pair_class(T.g_type[T.p_g1[p]], T.g_type[T.p_g2[p]]);
      P.pq_pair[base + s] = p;
    }
    const unsigned long long mh = __ballot(on && cls != 0), ml = __ballot(on && cls == 0);
    n_mpr += __popcll(__ballot(on && cls == 2)); n_pm += __popcll(__ballot(on && cls == 1)); n_an += __popcll(ml);
    int bh = 0, bl = 0;
    if (lane == 0) {
      if (mh) bh = atomicAdd(&ctr[0], __popcll(mh));
      if (ml) bl = atomicAdd(&ctr[1], __popcll(ml));
    }
    bh = __builtin_amdgcn_readfirstlane(bh); bl = __builtin_amdgcn_readfirstlane(bl);
    if (on) {
      if (cls != 0) tickA[bh + __popcll(mh & lt)] = (env << 8) | s;
      else tickB[bl + __popcll(ml & lt)] = (env << 8) | s;
    }
  }
  for (int s = 0; s < nsurv; s++) {
    const int p = S.u.co.surv[s];
    const int g = lane < 18 ? T.p_g1[p] : T.p_g2[p];
    if (lane < 36) P.pq_geo[(base + s) * 36 + lane] = geo_entry(T, g, lane < 18 ? lane : lane - 18);
  }
  if (lane == 0) { S.info[4] = overflow; S.info[5] = nsurv; S.info[6] = n_an | (n_pm << 8) | (n_mpr << 16); }
}

// the contacts the pair kernel found, in canonical order (survivor order, then the routine's order), into the contact arrays
__device__ __forceinline__ int gather_contacts(const Dev &T, const Launch &P, const int env, const int lane) {
  const int nsurv = S.info[5];
  int overflow = S.info[4], ncon = 0;
  const size_t base = (size_t)env * PAIRCAP;
  for (int b = 0; b < nsurv; b += 64) {
    const int s = b + lane;
    const bool on = s < nsurv;
    const int n = on ? P.pq_cnt[base + s] : 0;
    int incl = n;
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    const int off = ncon + incl - n;
    if (n > 0) {
      const int p = P.pq_pair[base + s];
      const int g1 = T.p_g1[p], g2 = T.p_g2[p];
      const float mu = fmaxf(T.g_mu[g1], T.g_mu[g2]);
      for (int k = 0; k < n; k++) {
        const int c = off + k;
        if (c >= MAXCON) break;
        const float4 *r = reinterpret_cast<const float4 *>(P.pq_con + ((base + s) * 8 + k) * 16);
        const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
        S.u.co.c_dist[c] = r0.x;
        S.u.co.c_pos[c][0] = r0.y; S.u.co.c_pos[c][1] = r0.z; S.u.co.c_pos[c][2] = r0.w;
        float *fr = S.u.co.c_frame[c];
        fr[0] = r1.x; fr[1] = r1.y; fr[2] = r1.z; fr[3] = r1.w; fr[4] = r2.x; fr[5] = r2.y; fr[6] = r2.z; fr[7] = r2.w; fr[8] = r3.x;
        S.u.co.c_g1[c] = g1; S.u.co.c_g2[c] = g2; S.u.co.c_mu[c] = mu;
      }
    }
    ncon += __shfl(incl, 63);
  }
  if (ncon > MAXCON) { overflow = 1; ncon = MAXCON; }
  SYNC();
  if (lane == 0) { S.info[0] = ncon; S.info[4] = overflow; }
  SYNC();
  return ncon;
}

extern "C" __global__ void __launch_bounds__(64, 2) g1_env_kernel(Launch P) {
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= P.N) return;
  const int env = P.order ? P.order[blockIdx.x] : (int)blockIdx.x;
  const Dev &T = *P.T;
  char *ws = P.ws + (size_t)env * WS_BYTES;
  int *wctx = reinterpret_cast<int *>(ws + WS_CTX_OFF);
  StepCtx X;
#ifdef G1_PROFILE   // per-phase stamps of the split pipeline: accumulated over the rounds in the working set, written out by the last one
#define G1_PROF_OUT() do { SYNC(); if (P.debug && lane < 16) P.debug[(size_t)env * DMG1_DEBUG_STRIDE + 900 + lane] = (float)S.prof[lane]; } while (0)
  if (P.round == 0) { if (lane == 0) for (int i = 0; i < 18; i++) S.prof[i] = 0; S.prof_t = clock64(); }
#else
#define G1_PROF_OUT() do {} while (0)
#endif
  if (P.round == 0) {
    if (!step_enter(P, T, env, lane, X)) { if (lane == 0) wctx[WC_FINISHED] = 1; return; }
    if (X.sim_err) {   // mj_checkPos / mj_checkVel failed: no evaluation, the task layer reports the error (and may reset the env)
      if (!task_and_finish(P, T, env, lane, X)) { step_write_back(P, env, lane, X); if (lane == 0) wctx[WC_FINISHED] = 1; return; }
    }
  } else {
    if (wctx[WC_FINISHED]) return;
    ws_load(P, ws, X, lane);
#ifdef G1_PROFILE
    S.prof_t = clock64();      // (the time between the launches is not this env's)
#endif
    const int ncon = gather_contacts(T, P, env, lane);            // second half of the evaluation in flight
    PROF(13);                  // split pipeline: working-set reload is before the stamp; 13 = gather
    forward_post<true>(P, T, env, lane, ncon);
    // cost of this env in g1_env_kernel (the sort key of the next step's longest-first order): its pairs run in the other kernel, so
    // only the per-evaluation fixed part and rows x sweeps count — a row step costs about twice as much beyond 128 rows (A from L2)
    X.work = (X.work >> 1) + 16000 + S.info[1] * (24 + (S.info[1] > 128 ? 14 : 6) * S.info[3]);   // (later evaluations weigh more: as dm_kernels.hip)
    bool again = false;
    if (!X.after_reset) again = rk_advance(P, T, env, lane, X);
    if (!again && !task_and_finish(P, T, env, lane, X)) { PROF(14); G1_PROF_OUT(); step_write_back(P, env, lane, X); if (lane == 0) wctx[WC_FINISHED] = 1; return; }
    PROF(14);
  }
  forward_pre(T, lane, !X.after_reset && X.stage > 0);              // first half of the next evaluation
  emit_pairs(T, P, env, lane);
  PROF(4);                     // split pipeline: 3 = broadphase (inside), 4 = tickets + staged geoms
  ws_save(ws, X, lane);
}

struct PairLaunch {
  const Dev *T;
  const double *mesh_vert; const int32_t *mesh_oidx; const double *mesh_clus;
  const double *pq_geo; const int32_t *pq_pair; int32_t *pq_cnt; float *pq_con;
  const int32_t *tick; int32_t *qctr; float *sepc2;
  int32_t N, pad;
};
struct NpLds { double geo[2][18]; int32_t geoi[2][6]; double rc[8][7]; double poly[2][16][3]; double mpr_ps[4][9]; };

// Narrowphase of the split pipeline: persistent one-wave workgroups pull tickets (support-query pairs first: they cost ~10 x an
// analytic pair) until the queue is empty; one ticket = one (env, pair): its staged geoms in, its contacts out.
// (Pulling the NEXT ticket while this one's narrowphase runs — queue head and ticket word prefetched — was 11 % SLOWER, 486 against
// 436 us per launch: a wave inside a long MPR ticket then sits on a reserved ticket that an idle wave could have taken.  Filing the
// geoms' integer records with the pair to save the table look-ups gained nothing either: those tables are L1 / scalar-cache hits.)
#ifndef G1_PAIR_WAVES
#define G1_PAIR_WAVES 2   // waves per SIMD of the pair kernel (VGPR budget 256: no spills; 168 at 3 spilled 34 and measured 2 % slower)
#endif
extern "C" __global__ void __launch_bounds__(64, G1_PAIR_WAVES) g1_pair_kernel(PairLaunch Q) {
  __shared__ NpLds Wl;
  const int lane = threadIdx.x;
  const Dev &T = *Q.T;
  const NpStage W = {Wl.geo, Wl.geoi, reinterpret_cast<Con *>(&Wl.rc[0][0]), Wl.poly[0], Wl.poly[1], reinterpret_cast<Sup *>(&Wl.mpr_ps[0][0])};
  const MeshPtrs M = {Q.mesh_vert, Q.mesh_oidx, Q.mesh_clus};
  const int nA = Q.qctr[0], nB = Q.qctr[1];      // written by the env launch in front of this one
  for (;;) {   // every wave leaves as soon as the queue is empty
    int t = 0;
    if (lane == 0) t = atomicAdd(&Q.qctr[2], 1);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= nA + nB) break;
    const int tk = t < nA ? Q.tick[t] : Q.tick[(size_t)Q.N * PAIRCAP + (t - nA)];
    const int env = tk >> 8, s = tk & 255;
    const size_t slot = (size_t)env * PAIRCAP + s;
    const int p = Q.pq_pair[slot];
    if (lane < 36) Wl.geo[lane / 18][lane % 18] = Q.pq_geo[slot * 36 + lane];
    if (lane == 0) { stage_geoi(T, W, T.p_g1[p], 0); stage_geoi(T, W, T.p_g2[p], 1); }
    SYNC();
    const int n = narrow_pair<true>(M, W, Q.sepc2 + ((size_t)env * 1024 + p) * 4, true, p, 0, lane);
    SYNC();
    if (lane == 0) Q.pq_cnt[slot] = n;
    if (lane < n) {
      float dist, pos[3], fr[9];
      contact_record(W.rc[lane], dist, pos, fr);
      float4 *o = reinterpret_cast<float4 *>(Q.pq_con + (slot * 8 + lane) * 16);
      o[0] = make_float4(dist, pos[0], pos[1], pos[2]);
      o[1] = make_float4(fr[0], fr[1], fr[2], fr[3]);
      o[2] = make_float4(fr[4], fr[5], fr[6], fr[7]);
      o[3] = make_float4(fr[8], 0.f, 0.f, 0.f);
    }
    SYNC();
  }
}

// Longest-first launch order: env costs vary by an order of magnitude (a robot lying on the floor solves 100 rows for 50
// sweeps, one that was just reset 45 rows for a few), and a batch is only ~2 rounds of resident waves, so the heavy envs must
// start first.  One workgroup: 64 cost buckets (heaviest = bucket 0), histogram, prefix, scatter.  The order inside a
// bucket is arbitrary — results do not depend on the launch order (outputs and RNG are keyed by env).
extern "C" __global__ void __launch_bounds__(1024) g1_schedule_kernel(const int32_t *cost, int32_t *order, int N) {
  __shared__ int hist[64], base[64], cmax;
  const int t = threadIdx.x;
  if (t < 64) hist[t] = 0;
  if (t == 0) cmax = 1;
  __syncthreads();
  int m = 1;
  for (int i = t; i < N; i += 1024) m = max(m, cost[i]);
  atomicMax(&cmax, m);
  __syncthreads();
  const float sc = 64.f / ((float)cmax + 1.f);
  for (int i = t; i < N; i += 1024) atomicAdd(&hist[63 - min(63, (int)((float)cost[i] * sc))], 1);
  __syncthreads();
  if (t == 0) { int a = 0; for (int b = 0; b < 64; b++) { base[b] = a; a += hist[b]; } }
  __syncthreads();
  for (int i = t; i < N; i += 1024) order[atomicAdd(&base[63 - min(63, (int)((float)cost[i] * sc))], 1)] = i;
}

extern "C" __global__ void g1_gather_kernel(const float *state, int N, int off, int n, int stride, float *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * n) return;
  out[i] = state[(size_t)(i / n) * stride + off + i % n];
}
extern "C" __global__ void g1_scatter_int_kernel(float *state, int N, int off, int stride, const int32_t *in) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  ((int32_t *)state)[(size_t)i * stride + off] = in[i];
}

}  // namespace g1

// ------------------------------------------------------------------------------------------ host side (C-ABI)
struct DmG1Engine {
  DmG1Config cfg;
  int N = 0;
  std::string err;
  g1::Dev *dT = nullptr;
  double *dMesh = nullptr, *dClus = nullptr;
  int32_t *dOidx = nullptr;
  float *dState = nullptr, *dJT = nullptr, *dBT = nullptr, *dAR = nullptr, *dRowsE = nullptr, *dSepc = nullptr;
  int32_t *dOrder = nullptr, *dCost = nullptr;
  float *dRows[DMG1_MAX_CLIPS] = {}, *dReset[DMG1_MAX_CLIPS] = {}, *dCom[DMG1_MAX_CLIPS] = {}, *dDebug = nullptr;
  int L[DMG1_MAX_CLIPS] = {}, flags[DMG1_MAX_CLIPS] = {};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  // split pipeline (dmg1_step of batches >= 512 envs, or DmG1Config.pipeline = 2)
  bool split = false;
  char *dWs = nullptr;
  double *dPqGeo = nullptr;
  int32_t *dPqPair = nullptr, *dPqCnt = nullptr, *dTick = nullptr, *dQctr = nullptr;
  float *dPqCon = nullptr, *dSepc2 = nullptr;
};

static int g1_fail(DmG1Engine *e, int code, const char *msg) {
  if (e) e->err = msg;
  return code;
}

extern "C" void dmg1_default_config(DmG1Config *c) {
  memset(c, 0, sizeof *c);
  c->num_envs = 1; c->max_ep_length = 1000; c->vel_obs_scale = 0.1f; c->high_z = 2.0f; c->obs_bound = 100.0f;
  c->seed = 0; c->auto_reset = 1; c->device = 0;
  c->task = 0; c->amnesty_steps = 150; c->to_getup_len = 180;
  c->pipeline = 0;
}
extern "C" size_t dmg1_model_sizeof(void) { return sizeof(DmModelG1); }

static void g1_build_tables(const DmModelG1 &m, g1::Dev &T) {
  using namespace g1;
  memset(&T, 0, sizeof T);
  T.timestep = (float)m.timestep; T.tolerance = (float)m.tolerance;
  T.pgs_scale = (float)(1.0 / (m.meaninertia * (double)NV));
  const double tc = fmax(m.solref[0], 2 * m.timestep), dr = m.solref[1], dmax = m.solimp[1];   // refsafe
  T.K = (float)(1.0 / fmax(1e-15, dmax * dmax * tc * tc * dr * dr));
  T.B = (float)(2.0 / fmax(1e-15, dmax * tc));
  for (int i = 0; i < 5; i++) T.solimp[i] = (float)m.solimp[i];
  for (int i = 0; i < 3; i++) T.gravity[i] = (float)m.gravity[i];
  double mt = 0;
  for (int b = 1; b < NB; b++) mt += m.body_mass[b];
  T.total_mass_inv = (float)(1.0 / mt);
  T.low_z = (float)m.low_z; T.action_scale = (float)m.action_scale;
  T.iterations = m.iterations; T.npair = m.npair;
  T.torso_body = m.torso_body; T.floor_geom = m.floor_geom; T.rfoot_geom = m.rfoot_geom; T.lfoot_geom = m.lfoot_geom;
  for (int i = 0; i < 4; i++) T.ee_geom[i] = m.ee_geom[i];
  for (int i = 0; i < 8; i++) T.extra_geom[i] = m.extra_geom[i];
  for (int i = 0; i < NREW; i++) { T.rew_q[i] = m.rew_qposadr[i]; T.rew_v[i] = m.rew_dofadr[i]; T.rew_j[i] = m.rew_jnt[i]; }
  for (int i = 0; i < NQ; i++) { T.qpos0[i] = (float)m.qpos0[i]; T.qpos0_d[i] = m.qpos0[i]; }
  T.timestep_d = m.timestep;
  int maxd = 0;
  for (int b = 0; b < NB; b++) {
    T.b_parent[b] = m.body_parent[b]; T.b_depth[b] = m.body_depth[b]; T.b_dof[b] = m.body_dofadr[b];
    if (m.body_depth[b] > maxd) maxd = m.body_depth[b];
    for (int i = 0; i < 3; i++) { T.b_pos[b][i] = (float)m.body_pos[b][i]; T.b_ipos[b][i] = (float)m.body_ipos[b][i]; }
    for (int i = 0; i < 4; i++) { T.b_quat[b][i] = (float)m.body_quat[b][i]; T.b_quat_d[b][i] = m.body_quat[b][i]; }
    for (int i = 0; i < 3; i++) T.b_pos_d[b][i] = m.body_pos[b][i];
    for (int i = 0; i < 6; i++) T.b_inertia[b][i] = (float)m.body_inertia[b][i];
    T.b_mass[b] = (float)m.body_mass[b]; T.b_invw[b] = (float)m.body_invweight0[b][0];
  }
  T.maxdepth = maxd;
  for (int b = 1; b < NB; b++)
    for (int a = b; a > 0; a = m.body_parent[a]) T.b_desc[a] |= 1ull << b;
  for (int k = 0; k < NV; k++) {
    T.d_body[k] = m.dof_body[k]; T.d_madr[k] = m.dof_Madr[k]; T.d_act[k] = -1;
    int n = 0;
    for (int j = m.dof_parent[k]; j >= 0; j = m.dof_parent[j]) T.d_anc[k][n++] = (uint8_t)j;
    T.d_nanc[k] = n;
    const int j = m.dof_jnt[k];
    for (int i = 0; i < 3; i++) { T.d_axis[k][i] = (float)m.jnt_axis[j][i]; T.d_axis_d[k][i] = m.jnt_axis[j][i]; }
    T.d_arm[k] = (float)m.dof_armature[k]; T.d_damp[k] = (float)m.dof_damping[k]; T.d_invw[k] = (float)m.dof_invweight0[k];
    T.d_floss[k] = (float)m.dof_frictionloss[k];
    T.d_lo[k] = (float)m.jnt_range[j][0]; T.d_hi[k] = (float)m.jnt_range[j][1];
    if (!m.jnt_limited[j]) { T.d_lo[k] = -1e30f; T.d_hi[k] = 1e30f; }
  }
  for (int b = 1; b < NB; b++) {   // dofs that move body b: dofs of b and of its ancestors
    for (int a = b; a > 0; a = m.body_parent[a])
      for (int k = m.body_dofadr[a]; k < m.body_dofadr[a] + m.body_dofnum[a]; k++) T.b_chain[b] |= 1ull << k;
  }
  for (int a = 0; a < NU; a++) {
    const int k = m.act_dof[a];
    T.d_act[k] = a; T.d_clo[k] = (float)m.act_ctrlrange[a][0]; T.d_chi[k] = (float)m.act_ctrlrange[a][1];
  }
  for (int g = 0; g < NG; g++) {
    T.g_body[g] = m.geom_body[g]; T.g_type[g] = m.geom_type[g]; T.g_mesh[g] = m.geom_mesh[g];
    for (int i = 0; i < 3; i++) {
      T.g_pos[g][i] = (float)m.geom_pos[g][i]; T.g_size[g][i] = (float)m.geom_size[g][i];
      T.g_pos_d[g][i] = m.geom_pos[g][i]; T.g_size_d[g][i] = m.geom_size[g][i];
    }
    const double *q = m.geom_quat[g];
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double M[9] = {w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y),
                         2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
                         2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z};
    for (int i = 0; i < 9; i++) { T.g_mat[g][i] = (float)M[i]; T.g_mat_d[g][i] = M[i]; }
    T.g_rbound[g] = (float)m.geom_rbound[g]; T.g_mu[g] = (float)m.geom_friction[g][0];
    const double *zs = m.geom_size[g];
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    if (m.geom_type[g] == DM_GEOM_SPHERE) { for (int i = 0; i < 3; i++) { lo[i] = -zs[0]; hi[i] = zs[0]; } }
    else if (m.geom_type[g] == DM_GEOM_CYLINDER) { lo[0] = lo[1] = -zs[0]; hi[0] = hi[1] = zs[0]; lo[2] = -zs[1]; hi[2] = zs[1]; }
    else if (m.geom_type[g] == DM_GEOM_BOX) { for (int i = 0; i < 3; i++) { lo[i] = -zs[i]; hi[i] = zs[i]; } }
    else if (m.geom_type[g] == DM_GEOM_MESH && m.geom_mesh[g] >= 0) {
      const int me = m.geom_mesh[g];
      for (int i = 0; i < 3; i++) { lo[i] = 1e30; hi[i] = -1e30; }
      for (int v = m.mesh_vertadr[me]; v < m.mesh_vertadr[me] + m.mesh_vertnum[me]; v++)
        for (int i = 0; i < 3; i++) { lo[i] = fmin(lo[i], m.mesh_vert[v][i]); hi[i] = fmax(hi[i], m.mesh_vert[v][i]); }
    }
    for (int i = 0; i < 3; i++) { T.g_bc[g][i] = (float)(0.5 * (lo[i] + hi[i])); T.g_bh[g][i] = (float)(0.5 * (hi[i] - lo[i]) * 1.00001 + 1e-6); }
  }
  for (int g = 0; g < 96; g++) T.g_ci[g] = -1;
  int nci = 0;
  for (int p = 0; p < m.npair; p++) {
    T.p_g1[p] = (int16_t)m.pair_geom1[p]; T.p_g2[p] = (int16_t)m.pair_geom2[p];
    for (int q = 0; q < 2; q++) { const int g = q ? m.pair_geom2[p] : m.pair_geom1[p]; if (T.g_ci[g] < 0 && nci < NCG) T.g_ci[g] = nci++; }
  }
  for (int i = 0; i < DM_NMESH; i++)
    for (int k = 0; k < 3; k++) T.m_center[i][k] = m.mesh_center[i][k];   // vertex / cluster ranges: g1_build_meshes
  int p = 0;
  for (int b = 0; b < 15 && p < 128; b++)   // pairs (a <= c), c-major: index p -> (a, c)
    for (int a = 0; a <= b && p < 128; a++) { T.tri_a[p] = (uint8_t)a; T.tri_b[p] = (uint8_t)b; p++; }
  for (int k = 0; k < NV; k++) {
    const int n = T.d_nanc[k], np = n * (n + 1) / 2;
    for (int l = 0; l < 64; l++) {
      uint32_t w = 0;
      for (int hf = 0; hf < 2; hf++) {
        const int q = l + 64 * hf;
        uint32_t t = 0;
        if (q < np) { const int a = T.tri_a[q], c = T.tri_b[q]; t = (uint32_t)(T.d_madr[T.d_anc[k][a]] + (c - a)); }
        w |= t << (16 * hf);
      }
      T.f_tgt[k][l] = w;
    }
  }
}

// Reorder every hull into compact clusters: a k-d partition along the longest extent, split so that the left side is a whole
// number of 64-vertex leaves (every leaf but one per hull is full); a leaf occupies 64 vertex slots (unused slots carry the
// invalid original index 0x7fffffff) and has two bounds: an enclosing sphere (centre shrunk from the mean of the vertices by
// Badoiu-Clarkson steps) and a box about the same centre.
static void g1_kd_split(const DmModelG1 &m, int a0, std::vector<int> &idx, int lo, int hi, std::vector<std::pair<int, int>> &leaves) {
  if (hi - lo <= 64) { leaves.push_back({lo, hi}); return; }
  double mn[3] = {1e30, 1e30, 1e30}, mx[3] = {-1e30, -1e30, -1e30};
  for (int k = lo; k < hi; k++)
    for (int i = 0; i < 3; i++) { mn[i] = fmin(mn[i], m.mesh_vert[a0 + idx[k]][i]); mx[i] = fmax(mx[i], m.mesh_vert[a0 + idx[k]][i]); }
  int ax = 0;
  for (int i = 1; i < 3; i++) if (mx[i] - mn[i] > mx[ax] - mn[ax]) ax = i;
  const int half = (hi - lo) / 2, mid = lo + ((half + 63) / 64) * 64 < hi ? lo + ((half + 63) / 64) * 64 : (lo + hi) / 2;   // left side: full leaves
  std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int p, int q) {
    const double vp = m.mesh_vert[a0 + p][ax], vq = m.mesh_vert[a0 + q][ax];
    return vp < vq || (vp == vq && p < q);
  });
  g1_kd_split(m, a0, idx, lo, mid, leaves);
  g1_kd_split(m, a0, idx, mid, hi, leaves);
}
static int g1_build_meshes(const DmModelG1 &m, g1::Dev &T, std::vector<double> &verts, std::vector<int32_t> &oidx,
                            std::vector<double> &clus) {
  int vadr = 0, cadr = 0;
  for (int me = 0; me < DM_NMESH; me++) {
    const int n = m.mesh_vertnum[me], a0 = m.mesh_vertadr[me];
    T.m_vadr[me] = vadr; T.m_cadr[me] = cadr; T.m_vnum[me] = 0; T.m_cnum[me] = 0;
    if (n == 0) continue;
    std::vector<int> idx(n);
    for (int k = 0; k < n; k++) idx[k] = k;
    std::vector<std::pair<int, int>> leaves;
    g1_kd_split(m, a0, idx, 0, n, leaves);
    for (auto &lf : leaves) {
      double cc[3] = {0, 0, 0}, rad = 0;
      const int cnt = lf.second - lf.first;
      for (int k = lf.first; k < lf.second; k++) for (int i = 0; i < 3; i++) cc[i] += m.mesh_vert[a0 + idx[k]][i];
      for (int i = 0; i < 3; i++) cc[i] /= cnt;
      auto radius_at = [&](const double *c3) {
        double r = 0;
        for (int k = lf.first; k < lf.second; k++) {
          double d2 = 0;
          for (int i = 0; i < 3; i++) { const double df = m.mesh_vert[a0 + idx[k]][i] - c3[i]; d2 += df * df; }
          r = fmax(r, sqrt(d2));
        }
        return r;
      };
      rad = radius_at(cc);
      {   // a tighter enclosing sphere (Badoiu-Clarkson steps towards the farthest vertex): fewer clusters survive the bound test
        double c2[3] = {cc[0], cc[1], cc[2]};
        for (int it = 1; it <= 200; it++) {
          int far = lf.first; double best = -1;
          for (int k = lf.first; k < lf.second; k++) {
            double d2 = 0;
            for (int i = 0; i < 3; i++) { const double df = m.mesh_vert[a0 + idx[k]][i] - c2[i]; d2 += df * df; }
            if (d2 > best) { best = d2; far = k; }
          }
          for (int i = 0; i < 3; i++) c2[i] += (m.mesh_vert[a0 + idx[far]][i] - c2[i]) / (it + 1);
          const double r2 = radius_at(c2);
          if (r2 < rad) { rad = r2; for (int i = 0; i < 3; i++) cc[i] = c2[i]; }
        }
      }
      for (int k = lf.first; k < lf.second; k++) {
        for (int i = 0; i < 3; i++) verts.push_back(m.mesh_vert[a0 + idx[k]][i]);
        oidx.push_back(idx[k]);
      }
      for (int k = cnt; k < 64; k++) { verts.push_back(0); verts.push_back(0); verts.push_back(0); oidx.push_back(0x7fffffff); }
      clus.push_back(cc[0]); clus.push_back(cc[1]); clus.push_back(cc[2]); clus.push_back(rad * (1 + 1e-9) + 1e-12);
      double hx[3] = {0, 0, 0};
      for (int k = lf.first; k < lf.second; k++)
        for (int i = 0; i < 3; i++) hx[i] = fmax(hx[i], fabs(m.mesh_vert[a0 + idx[k]][i] - cc[i]));
      for (int i = 0; i < 3; i++) clus.push_back(hx[i] * (1 + 1e-9) + 1e-12);
      clus.push_back(0.0);
    }
    const int nclus = (int)leaves.size();
    if (nclus > 64 * g1::NCH) return -1;
    T.m_vnum[me] = nclus * 64; T.m_cnum[me] = nclus;
    vadr += nclus * 64; cadr += nclus;
  }
  return 0;
}

static int g1_check_model(DmG1Engine *e, const DmModelG1 &m) {
  using namespace g1;
  if (m.nq != NQ || m.nv != NV || m.nu != NU || m.nbody != NB || m.ngeom != NG || m.njnt != NJ || m.nM != NM)
    return g1_fail(e, DM_EINVAL, "model dimensions are not the Unitree G1's");
  if (m.integrator != DM_INT_RK4) return g1_fail(e, DM_EINVAL, "the G1 engine integrates with RK4 (xml :7)");
  if (m.npair < 0 || m.npair > 1024) return g1_fail(e, DM_EINVAL, "npair out of range");
  if (m.jnt_type[0] != DM_JNT_FREE || m.jnt_body[0] != 1) return g1_fail(e, DM_EINVAL, "joint 0 must be the free root");
  for (int b = 2; b < NB; b++)
    if (m.body_jntnum[b] != 1 || m.body_dofadr[b] != b + 4 || m.body_jntadr[b] != b - 1 || m.body_depth[b] > 15)
      return g1_fail(e, DM_EINVAL, "kernel assumes one hinge per body in body order");
  for (int j = 0; j < NJ; j++)
    for (int i = 0; i < 3; i++)
      if (m.jnt_pos[j][i] != 0.0) return g1_fail(e, DM_EINVAL, "kernel assumes joint anchors at the body origin");
  for (int k = 0; k < NV; k++) {
    int n = 0;
    for (int j = m.dof_parent[k]; j >= 0; j = m.dof_parent[j]) n++;
    if (n > 14) return g1_fail(e, DM_EINVAL, "dof chain deeper than 14 ancestors");
    if (m.dof_parent[k] != g1topo::PARENT[k] || m.dof_Madr[k] != g1topo::MADR[k])
      return g1_fail(e, DM_EINVAL, "dof tree differs from the compiled-in topology (regenerate csrc/dm_g1_topology.h)");
    if (k >= 6 && !(m.dof_frictionloss[k] > 0)) return g1_fail(e, DM_EINVAL, "kernel assumes friction loss on every hinge");
  }
  for (int a = 0; a < NU; a++)
    if (m.act_gear[a] != 1.0) return g1_fail(e, DM_EINVAL, "kernel assumes motor gear 1");
  {
    int seen[DM_NGEOM] = {0}, nci = 0;
    for (int p = 0; p < m.npair; p++) { seen[m.pair_geom1[p]] = 1; seen[m.pair_geom2[p]] = 1; }
    for (int g = 0; g < NG; g++) nci += seen[g];
    if (nci > NCG) return g1_fail(e, DM_EINVAL, "more colliding geoms than the kernel's LDS layout holds");
  }
  for (int g = 0; g < NG; g++) {
    if (m.geom_margin[g] != 0.0) return g1_fail(e, DM_EINVAL, "kernel assumes zero geom margins (xml has none)");
    if (m.geom_mesh[g] >= DM_NMESH) return g1_fail(e, DM_EINVAL, "mesh id out of range");
  }
  return DM_OK;
}

extern "C" int dmg1_create(const void *model, size_t model_bytes, const DmG1Config *cfg, DmG1Handle *out) {
  if (!model || !cfg || !out || cfg->num_envs < 1) return DM_EINVAL;
  if (model_bytes != sizeof(DmModelG1)) { fprintf(stderr, "dmg1_create: model struct size %zu != %zu\n", model_bytes, sizeof(DmModelG1)); return DM_EINVAL; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DM_ENODEV;
  const DmModelG1 &m = *(const DmModelG1 *)model;
  DmG1Engine *e = new DmG1Engine();
  e->cfg = *cfg;
  e->N = cfg->num_envs;
  int rc = g1_check_model(e, m);
  if (rc != DM_OK) { fprintf(stderr, "dmg1_create: %s\n", e->err.c_str()); delete e; return rc; }
  if (hipSetDevice(cfg->device) != hipSuccess) { delete e; return DM_ENODEV; }
  g1::Dev *T = new g1::Dev();
  g1_build_tables(m, *T);
  std::vector<double> verts, clus;
  std::vector<int32_t> oidx;
  if (g1_build_meshes(m, *T, verts, oidx, clus) != 0) {
    delete T; delete e; fprintf(stderr, "dmg1_create: a hull has more than %d vertex clusters\n", 64 * g1::NCH); return DM_EINVAL;
  }
  bool ok = hipMalloc(&e->dT, sizeof(g1::Dev)) == hipSuccess;
  if (ok) hipMemcpy(e->dT, T, sizeof(g1::Dev), hipMemcpyHostToDevice);
  delete T;
  ok = ok && hipMalloc(&e->dMesh, verts.size() * sizeof(double)) == hipSuccess;
  ok = ok && hipMalloc(&e->dOidx, oidx.size() * sizeof(int32_t)) == hipSuccess;
  ok = ok && hipMalloc(&e->dClus, clus.size() * sizeof(double)) == hipSuccess;
  if (ok) {
    hipMemcpy(e->dMesh, verts.data(), verts.size() * sizeof(double), hipMemcpyHostToDevice);
    hipMemcpy(e->dOidx, oidx.data(), oidx.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    hipMemcpy(e->dClus, clus.data(), clus.size() * sizeof(double), hipMemcpyHostToDevice);
  }
  const size_t N = (size_t)e->N;
  ok = ok && hipMalloc(&e->dState, N * g1::STATE * sizeof(float)) == hipSuccess;
  ok = ok && hipMalloc(&e->dJT, N * 44 * g1::MAXROW * sizeof(float)) == hipSuccess;
  ok = ok && hipMalloc(&e->dBT, N * 44 * g1::MAXROW * sizeof(float)) == hipSuccess;
  ok = ok && hipMalloc(&e->dAR, N * g1::MAXROW * g1::MAXROW * sizeof(float)) == hipSuccess;
  ok = ok && hipMalloc(&e->dRowsE, N * 5 * g1::MAXROW * sizeof(float)) == hipSuccess;
  ok = ok && hipMalloc(&e->dSepc, N * (4 * g1::SEPC + 4) * sizeof(float)) == hipSuccess;
  if (ok) hipMemset(e->dSepc, 0xFF, N * (4 * g1::SEPC + 4) * sizeof(float));   // pair id -1 everywhere
  ok = ok && hipMalloc(&e->dOrder, N * sizeof(int32_t)) == hipSuccess && hipMalloc(&e->dCost, N * sizeof(int32_t)) == hipSuccess;
  if (ok) hipMemset(e->dCost, 0, N * sizeof(int32_t));
  e->split = cfg->pipeline == 2 || (cfg->pipeline == 0 && e->N >= 512);
#ifdef G1_PROFILE
  if (const char *pl = getenv("DMG1_PIPELINE")) e->split = atoi(pl) == 2;
#endif
  if (e->split) {   // ~135 KB per env: 0.55 GB at 4 096 envs
    const size_t NP = N * g1::PAIRCAP;
    ok = ok && hipMalloc(&e->dWs, N * g1::WS_BYTES) == hipSuccess;
    ok = ok && hipMalloc(&e->dPqGeo, NP * 36 * sizeof(double)) == hipSuccess;
    ok = ok && hipMalloc(&e->dPqPair, NP * sizeof(int32_t)) == hipSuccess && hipMalloc(&e->dPqCnt, NP * sizeof(int32_t)) == hipSuccess;
    ok = ok && hipMalloc(&e->dPqCon, NP * 8 * 16 * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc(&e->dTick, 2 * NP * sizeof(int32_t)) == hipSuccess && hipMalloc(&e->dQctr, 8 * 4 * sizeof(int32_t)) == hipSuccess;
    ok = ok && hipMalloc(&e->dSepc2, N * 1024 * 4 * sizeof(float)) == hipSuccess;
    if (ok) hipMemset(e->dSepc2, 0xFF, N * 1024 * 4 * sizeof(float));   // pair id -1 everywhere
  }
  if (!ok) { dmg1_destroy(e); return DM_ENOMEM; }
  std::vector<float> init(N * g1::STATE, 0.f);
  for (size_t i = 0; i < N; i++)
    for (int k = 0; k < g1::NQ; k++) init[i * g1::STATE + g1::S_QPOS + k] = (float)m.qpos0[k];
  hipMemcpy(e->dState, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice);
  hipEventCreate(&e->ev0); hipEventCreate(&e->ev1);
  *out = e;
  return DM_OK;
}

extern "C" int dmg1_destroy(DmG1Handle e) {
  if (!e) return DM_EINVAL;
  hipFree(e->dWs); hipFree(e->dPqGeo); hipFree(e->dPqPair); hipFree(e->dPqCnt); hipFree(e->dPqCon); hipFree(e->dTick); hipFree(e->dQctr); hipFree(e->dSepc2);
  hipFree(e->dT); hipFree(e->dMesh); hipFree(e->dOidx); hipFree(e->dClus); hipFree(e->dState); hipFree(e->dJT); hipFree(e->dBT); hipFree(e->dAR); hipFree(e->dRowsE); hipFree(e->dSepc); hipFree(e->dOrder); hipFree(e->dCost);
  for (int c = 0; c < DMG1_MAX_CLIPS; c++) { hipFree(e->dRows[c]); hipFree(e->dReset[c]); hipFree(e->dCom[c]); }
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  delete e;
  return DM_OK;
}
extern "C" const char *dmg1_last_error(DmG1Handle e) { return e ? e->err.c_str() : "null handle"; }

extern "C" int dmg1_load_clip(DmG1Handle e, int clip_id, int L, const double *q, const double *v, const double *bx, const double *gx, int flags) {
  using namespace g1;
  if (!e || L < 1 || !q || !v || !bx || !gx || clip_id < 0 || clip_id >= DMG1_MAX_CLIPS) return DM_EINVAL;
  g1::Dev *T = new g1::Dev();
  hipMemcpy(T, e->dT, sizeof(g1::Dev), hipMemcpyDeviceToHost);
  std::vector<float> rows((size_t)L * CLIP_ROW, 0.f), reset((size_t)L * 88, 0.f), com((size_t)L * 4, 0.f);
  double mt = 0;
  for (int b = 0; b < NB; b++) mt += T->b_mass[b];
  for (int f = 0; f < L; f++) {
    float *r = &rows[(size_t)f * CLIP_ROW];
    for (int i = 0; i < NREW; i++) { r[i] = (float)q[(size_t)f * NQ + T->rew_q[i]]; r[23 + i] = (float)v[(size_t)f * NV + T->rew_v[i]]; }
    for (int i = 0; i < 4; i++) r[46 + i] = (float)q[(size_t)f * NQ + 3 + i];
    r[62] = (float)v[(size_t)f * NV]; r[63] = (float)v[(size_t)f * NV + 1];
    for (int e4 = 0; e4 < 4; e4++)
      for (int i = 0; i < 3; i++) r[50 + 3 * e4 + i] = (float)gx[((size_t)f * NG + T->ee_geom[e4]) * 3 + i];
    for (int i = 0; i < NQ; i++) reset[(size_t)f * 88 + i] = (float)q[(size_t)f * NQ + i];
    for (int i = 0; i < NV; i++) reset[(size_t)f * 88 + 44 + i] = (float)v[(size_t)f * NV + i];
    for (int i = 0; i < 3; i++) {
      double s = 0;
      for (int b = 0; b < NB; b++) s += (double)T->b_mass[b] * bx[((size_t)f * NB + b) * 3 + i];
      com[(size_t)f * 4 + i] = (float)(s / mt);
    }
  }
  delete T;
  const int c = clip_id;
  if (hipSetDevice(e->cfg.device) != hipSuccess) return g1_fail(e, DM_EHIP, "hipSetDevice failed");
  hipFree(e->dRows[c]); hipFree(e->dReset[c]); hipFree(e->dCom[c]);
  e->dRows[c] = e->dReset[c] = e->dCom[c] = nullptr;
  if (hipMalloc(&e->dRows[c], rows.size() * 4) != hipSuccess || hipMalloc(&e->dReset[c], reset.size() * 4) != hipSuccess ||
      hipMalloc(&e->dCom[c], com.size() * 4) != hipSuccess)
    return g1_fail(e, DM_ENOMEM, "clip tables");
  hipMemcpy(e->dRows[c], rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(e->dReset[c], reset.data(), reset.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(e->dCom[c], com.data(), com.size() * 4, hipMemcpyHostToDevice);
  e->L[c] = L; e->flags[c] = flags;
  return DM_OK;
}

static int g1_launch(DmG1Engine *e, g1::Launch &P, void *stream, bool time_it) {
  P.T = e->dT; P.mesh_vert = e->dMesh; P.mesh_oidx = e->dOidx; P.mesh_clus = e->dClus; P.state = e->dState; P.jt = e->dJT; P.bt = e->dBT; P.ar = e->dAR; P.rows = e->dRowsE; P.sepc = e->dSepc;
  for (int c = 0; c < DMG1_MAX_CLIPS; c++) { P.clips[c].rows = e->dRows[c]; P.clips[c].reset = e->dReset[c]; P.clips[c].com = e->dCom[c]; P.clips[c].L = e->L[c]; P.clips[c].flags = e->flags[c]; }
  P.task = e->cfg.task; P.amnesty_steps = e->cfg.amnesty_steps; P.to_getup_len = e->cfg.to_getup_len;
  P.N = e->N; P.auto_reset = e->cfg.auto_reset; P.max_ep_length = e->cfg.max_ep_length;
  P.vel_obs_scale = e->cfg.vel_obs_scale; P.high_z = e->cfg.high_z; P.obs_bound = e->cfg.obs_bound; P.seed = e->cfg.seed;
  P.debug = e->dDebug;
  if (!e->L[0]) return g1_fail(e, DM_EINVAL, "no clip loaded (dmg1_load_clip clip 0 first)");
  if (e->cfg.task && (e->L[1] < 1 || e->L[2] < 2)) return g1_fail(e, DM_EINVAL, "combined task needs clips 0, 1, 2 = walk, run, getup");
  if (hipSetDevice(e->cfg.device) != hipSuccess) return g1_fail(e, DM_EHIP, "hipSetDevice failed");
  bool lpt = true;
#ifdef G1_PROFILE   // phase switches of the diagnostic build only: the shipped library never reads the environment on a launch
  if (const char *sk = getenv("DMG1_SKIP")) P.pad = atoi(sk);   // bit 0 no PGS sweeps, 1 no collision, 2 no A matrix, 3 no rows, 4 no constraint solve
  lpt = !getenv("DMG1_NO_LPT");
#endif
  hipStream_t s = (hipStream_t)stream;
  if (time_it) hipEventRecord(e->ev0, s);
  if (P.mode == g1::MODE_STEP && e->N >= 512 && lpt) {   // longest-first order from the previous step's costs
    hipLaunchKernelGGL(g1::g1_schedule_kernel, dim3(1), dim3(1024), 0, s, e->dCost, e->dOrder, e->N);
    P.order = e->dOrder;
  }
  if (P.mode == g1::MODE_STEP) P.cost = e->dCost;
  if (P.mode == g1::MODE_STEP && e->split) {
    // split pipeline: 6 env launches around 5 pair launches (4 RK stages + the evaluation at the reset state of envs that ended)
    P.ws = e->dWs; P.pq_geo = e->dPqGeo; P.pq_pair = e->dPqPair; P.pq_cnt = e->dPqCnt; P.pq_con = e->dPqCon; P.tick = e->dTick;
    P.qctr = e->dQctr; P.sepc2 = e->dSepc2;
    hipMemsetAsync(e->dQctr, 0, 8 * 4 * sizeof(int32_t), s);
    g1::PairLaunch Q;
    memset(&Q, 0, sizeof Q);
    Q.T = e->dT; Q.mesh_vert = e->dMesh; Q.mesh_oidx = e->dOidx; Q.mesh_clus = e->dClus; Q.pq_geo = e->dPqGeo; Q.pq_pair = e->dPqPair;
    Q.pq_cnt = e->dPqCnt; Q.pq_con = e->dPqCon; Q.tick = e->dTick; Q.sepc2 = e->dSepc2; Q.N = e->N;
    const int pair_waves = std::min(e->N * 8, 256 * 4 * G1_PAIR_WAVES);   // persistent: every wave pulls tickets until the queue is empty
    for (int r = 0; r < 6; r++) {
      P.round = r;
      hipLaunchKernelGGL(g1::g1_env_kernel, dim3(e->N), dim3(64), 0, s, P);
      if (r < 5) {
        Q.qctr = e->dQctr + 4 * r;
        hipLaunchKernelGGL(g1::g1_pair_kernel, dim3(pair_waves), dim3(64), 0, s, Q);
      }
    }
  } else {
    hipLaunchKernelGGL(g1::g1_step_kernel, dim3(e->N), dim3(64), 0, s, P);
  }
  if (time_it) { hipEventRecord(e->ev1, s); e->timed = true; }
  return hipGetLastError() == hipSuccess ? DM_OK : g1_fail(e, DM_EHIP, "kernel launch failed");
}

extern "C" int dmg1_reset(DmG1Handle e, const uint8_t *mask, const int32_t *idx_init, float *obs_out, void *stream) {
  if (!e || !e->L[0]) return e ? g1_fail(e, DM_EINVAL, "no clip loaded") : DM_EINVAL;
  g1::Launch P;
  memset(&P, 0, sizeof P);
  P.mode = g1::MODE_RESET; P.mask = mask; P.idx_init = idx_init; P.obs = obs_out;
  return g1_launch(e, P, stream, false);
}
extern "C" int dmg1_step(DmG1Handle e, const float *actions, float *obs, float *rew, uint8_t *done, float *terms, int32_t *reason,
                         float *terminal_obs, void *stream) {
  if (!e || !actions) return DM_EINVAL;
  if (!e->L[0]) return g1_fail(e, DM_EINVAL, "no clip loaded");
  g1::Launch P;
  memset(&P, 0, sizeof P);
  P.mode = g1::MODE_STEP; P.actions = actions; P.obs = obs; P.rew = rew; P.done = done; P.terms = terms; P.reason = reason;
  P.terminal_obs = terminal_obs;
  return g1_launch(e, P, stream, true);
}
extern "C" int dmg1_step_forced(DmG1Handle e, const float *qpos, const float *qvel, float *obs, float *rew, uint8_t *done,
                                float *terms, int32_t *reason, void *stream) {
  if (!e || !qpos || !qvel) return DM_EINVAL;
  if (!e->L[0]) return g1_fail(e, DM_EINVAL, "no clip loaded");
  g1::Launch P;
  memset(&P, 0, sizeof P);
  P.mode = g1::MODE_FORCED; P.in_qpos = qpos; P.in_qvel = qvel; P.obs = obs; P.rew = rew; P.done = done; P.terms = terms; P.reason = reason;
  return g1_launch(e, P, stream, false);
}
extern "C" int dmg1_set_state(DmG1Handle e, const float *qpos, const float *qvel, const float *warm, int run_forward, void *stream) {
  if (!e || !qpos || !qvel) return DM_EINVAL;
  g1::Launch P;
  memset(&P, 0, sizeof P);
  P.mode = g1::MODE_SETSTATE; P.in_qpos = qpos; P.in_qvel = qvel; P.in_warm = warm; P.run_forward = run_forward;
  return g1_launch(e, P, stream, false);
}
static void g1_gather(DmG1Engine *e, int off, int n, float *out, void *stream) {
  const int tot = e->N * n;
  hipLaunchKernelGGL(g1::g1_gather_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->dState, e->N, off, n, g1::STATE, out);
}
extern "C" int dmg1_get_state(DmG1Handle e, float *qpos, float *qvel, float *warm, void *stream) {
  if (!e) return DM_EINVAL;
  if (qpos) g1_gather(e, g1::S_QPOS, g1::NQ, qpos, stream);
  if (qvel) g1_gather(e, g1::S_QVEL, g1::NV, qvel, stream);
  if (warm) g1_gather(e, g1::S_WARM, g1::NV, warm, stream);
  return DM_OK;
}
extern "C" int dmg1_get_counters(DmG1Handle e, int32_t *idx, int32_t *eplen, float *eprew, void *stream) {
  if (!e) return DM_EINVAL;
  if (idx) g1_gather(e, g1::S_IDX, 1, (float *)idx, stream);
  if (eplen) g1_gather(e, g1::S_EPLEN, 1, (float *)eplen, stream);
  if (eprew) g1_gather(e, g1::S_EPREW, 1, eprew, stream);
  return DM_OK;
}
extern "C" int dmg1_set_counters(DmG1Handle e, const int32_t *idx, const int32_t *eplen, void *stream) {
  if (!e) return DM_EINVAL;
  const int nb = (e->N + 255) / 256;
  if (idx) hipLaunchKernelGGL(g1::g1_scatter_int_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, e->dState, e->N, g1::S_IDX, g1::STATE, idx);
  if (eplen) hipLaunchKernelGGL(g1::g1_scatter_int_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, e->dState, e->N, g1::S_EPLEN, g1::STATE, eplen);
  return DM_OK;
}
extern "C" int dmg1_queue_counters(DmG1Handle e, int32_t *host_out24) {
  if (!e || !host_out24) return DM_EINVAL;
  memset(host_out24, 0, 24 * sizeof(int32_t));
  if (!e->split) return DM_OK;
  if (hipSetDevice(e->cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return DM_EHIP;
  return hipMemcpy(host_out24, e->dQctr, 24 * sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess ? DM_OK : DM_EHIP;
}
#ifdef G1_PAIRSTATS
extern "C" int dmg1_pairstats(unsigned long long *host_out8, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host_out8, HIP_SYMBOL(g1::g_pairstats), 8 * sizeof(unsigned long long));
  if (reset) { unsigned long long z[8] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g1::g_pairstats), z, sizeof z); }
  return 0;
}
#endif
extern "C" int dmg1_set_seed(DmG1Handle e, uint64_t seed) {
  if (!e) return DM_EINVAL;
  e->cfg.seed = seed;
  return DM_OK;
}
extern "C" int dmg1_obs_dim(DmG1Handle e) { return e ? (e->cfg.task ? DMG1_NOBS_COMBINED : DMG1_NOBS) : DM_EINVAL; }
extern "C" int dmg1_get_motion(DmG1Handle e, int32_t *motion, void *stream) {
  if (!e || !motion) return DM_EINVAL;
  g1_gather(e, g1::S_MOTION, 1, (float *)motion, stream);
  return DM_OK;
}
extern "C" int dmg1_set_motion(DmG1Handle e, const int32_t *motion, void *stream) {
  if (!e || !motion) return DM_EINVAL;
  hipLaunchKernelGGL(g1::g1_scatter_int_kernel, dim3((e->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->dState, e->N, g1::S_MOTION, g1::STATE, motion);
  return DM_OK;
}
extern "C" int dmg1_set_env_clips(DmG1Handle e, const int32_t *clip_ids, void *stream) { return dmg1_set_motion(e, clip_ids, stream); }
extern "C" int dmg1_get_env_clips(DmG1Handle e, int32_t *clip_ids, void *stream) { return dmg1_get_motion(e, clip_ids, stream); }
extern "C" int dmg1_set_debug(DmG1Handle e, float *debug) {
  if (!e) return DM_EINVAL;
  e->dDebug = debug;
  return DM_OK;
}
extern "C" float dmg1_last_kernel_ms(DmG1Handle e) {
  if (!e || !e->timed) return -1.f;
  float ms = -1.f;
  if (hipEventSynchronize(e->ev1) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&ms, e->ev0, e->ev1) != hipSuccess) return -1.f;
  return ms;
}
