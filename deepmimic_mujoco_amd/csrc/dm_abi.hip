// dm_abi.hip — host side of libdeepmimic_hip.so: the C-ABI of include/deepmimic_hip.h.
// Builds the fp32 device tables from DmModel, owns the per-env HBM state rows and
// the clip tables, and launches the fused step kernel (dm_kernels.hip).
#include "../../include/deepmimic_hip.h"
#include "dm_device.h"
#include "dm_topology.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "dm_kernels.hip"
#include "dm_ppo.hip"
#include "dm_policy.hip"
#include "dm_ppo_mlp.hip"
#include "dm_ppo_wide.hip"

struct DmEngine {
  DmConfig cfg;
  DmModel model;
  int N = 0;
  DmDev *dT = nullptr;
  float *dState = nullptr;
  float *dArScratch = nullptr;
  float *dClipRows[DM_MAX_CLIPS] = {nullptr};
  float *dClipReset[DM_MAX_CLIPS] = {nullptr};
  int clipL[DM_MAX_CLIPS] = {0};
  int clipFlags[DM_MAX_CLIPS] = {0};
  float *debug = nullptr;
  int32_t *dOrder = nullptr;   // slot -> env permutation for dm_step (longest-first)
  int32_t *dCost = nullptr;    // per-env work estimate written by dm_step
  static constexpr int NEV = 512;              // ring of event pairs: one per dm_step while timing is on
  hipEvent_t ev0[NEV] = {}, ev1[NEV] = {};
  long nrec = 0;                               // launches recorded since dm_enable_timing(1)
  long nlaunch = 0;                            // launches since dm_enable_timing
  int stride = 1;                              // every stride-th launch carries an event pair
  int waves = 0;                               // 0: kernel variant from the batch size; 2 / 3: forced (env DM_WAVES, experiments)
  bool timing = false;
  float last_ms = 0;
  std::string err;
};

static int fail(DmEngine *e, int code, const char *what, hipError_t he = hipSuccess) {
  if (e) {
    e->err = what;
    if (he != hipSuccess) { e->err += ": "; e->err += hipGetErrorString(he); }
  }
  return code;
}
#define HIPCHK(e, call)                                   \
  do {                                                    \
    hipError_t _r = (call);                               \
    if (_r != hipSuccess) return fail(e, DM_EHIP, #call, _r); \
  } while (0)

extern "C" void dm_default_config(DmConfig *c) {
  memset(c, 0, sizeof(*c));
  c->num_envs = 1;
  c->max_ep_length = 1000;   // src/deepmimic_env.py:260
  c->vel_obs_scale = 0.1f;   // :261
  c->low_z = 0.7f;           // src/config.py:13
  c->high_z = 2.0f;          // src/deepmimic_env.py:423
  c->w_pose = 0.75f; c->w_vel = 0.1f; c->w_end_eff = 0.15f; c->w_com = 0.0f; c->w_joint_limit = -0.1f;  // :400-404
  c->obs_bound = 100.0f;     // :465
  c->seed = 1234;
  c->auto_reset = 1;
  c->device = 0;
  c->lpt_schedule = 1;
  c->task = DM_TASK_DPENV;
  c->amnesty_steps = 150;    // src/combined_env.py:34
  c->to_getup_len = 180;     // :97
  c->integrator = DM_CFG_INT_MODEL;   // the XML's (RK4, xml :9)
}

static void build_tables(const DmModel &m, DmDev &T) {
  memset(&T, 0, sizeof(T));
  T.timestep = (float)m.timestep;
  T.tolerance = (float)m.tolerance;
  T.pgs_scale = (float)(1.0 / (m.meaninertia * (DM_NV > 1 ? DM_NV : 1)));
  for (int i = 0; i < 3; i++) T.gravity[i] = (float)m.gravity[i];
  double tc = fmax(m.solref[0], 2 * m.timestep), dr = m.solref[1], dmax = m.solimp[1];
  T.K = (float)(1.0 / fmax(1e-15, dmax * dmax * tc * tc * dr * dr));
  T.B = (float)(2.0 / fmax(1e-15, dmax * tc));
  for (int i = 0; i < 5; i++) T.solimp[i] = (float)m.solimp[i];
  double mt = 0;
  for (int b = 0; b < DM_NBODY; b++) mt += m.body_mass[b];
  T.total_mass_inv = (float)(1.0 / mt);
  for (int i = 0; i < DM_NQ; i++) T.qpos0[i] = (float)m.qpos0[i];
  T.iterations = m.iterations;
  T.npair = m.npair;
  T.torso_body = m.torso_body; T.rfoot_geom = m.rfoot_geom; T.lfoot_geom = m.lfoot_geom; T.floor_geom = m.floor_geom;
  for (int i = 0; i < 4; i++) T.ee_geom[i] = m.ee_geom[i];
  for (int b = 0; b < DM_NBODY; b++) {
    T.b_parent[b] = m.body_parent[b]; T.b_depth[b] = m.body_depth[b];
    T.b_dofadr[b] = m.body_dofadr[b] < 0 ? 0 : m.body_dofadr[b]; T.b_dofnum[b] = m.body_dofnum[b];
    for (int i = 0; i < 3; i++) { T.b_pos[b][i] = (float)m.body_pos[b][i]; T.b_ipos[b][i] = (float)m.body_ipos[b][i]; }
    for (int i = 0; i < 6; i++) T.b_inertia[b][i] = (float)m.body_inertia[b][i];
    T.b_mass[b] = (float)m.body_mass[b];
    T.b_invw[b] = (float)m.body_invweight0[b][0];
    uint32_t sub = 0;
    for (int c = 1; c < DM_NBODY; c++) {
      int a = c;
      while (a > 0 && a != b) a = m.body_parent[a];
      if (a == b && b > 0) sub |= 1u << c;
    }
    T.b_subtree[b] = sub;
    uint64_t chain = 0;
    for (int a = b; a > 0; a = m.body_parent[a])
      for (int k = 0; k < m.body_dofnum[a]; k++) chain |= 1ull << (m.body_dofadr[a] + k);
    T.b_chain[b] = chain;
    uint32_t cb = 0;
    int ids[8], nc = 0;
    for (int a = b; a > 0; a = m.body_parent[a]) ids[nc++] = a;
    for (int c = 0; c < nc && c < 4; c++) cb |= (uint32_t)ids[nc - 1 - c] << (8 * c);   // root first
    T.b_chainb[b] = cb;
  }
  for (int k = 0; k < DM_NV; k++) {
    int j = m.dof_jnt[k];
    T.d_body[k] = m.dof_body[k];
    T.d_act[k] = -1;
    T.d_limited[k] = (m.jnt_type[j] == DM_JNT_HINGE) ? m.jnt_limited[j] : 0;
    for (int i = 0; i < 3; i++) T.d_axis[k][i] = (float)m.jnt_axis[j][i];
    T.d_arm[k] = (float)m.dof_armature[k]; T.d_damp[k] = (float)m.dof_damping[k];
    T.d_invw[k] = (float)m.dof_invweight0[k];
    T.d_lo[k] = (float)m.jnt_range[j][0]; T.d_hi[k] = (float)m.jnt_range[j][1];
    int n = 0;
    for (int a = m.dof_parent[k]; a >= 0 && n < DMK_MAXANC; a = m.dof_parent[a]) T.d_anc[k][n++] = (uint8_t)a;
    T.d_nanc[k] = n;
    for (int d = 0; d < n; d++) T.d_ancabs[k][d] = T.d_anc[k][n - 1 - d];   // absolute depth d
    T.d_pbody[k] = m.body_parent[m.dof_body[k]];
  }
  for (int k = 0; k < DM_NV; k++) {
    for (int a = 0; a < T.d_nanc[k]; a++) T.d_desc[T.d_anc[k][a]] |= 1ull << k;
    T.nanc_pack[k >> 4] |= (uint64_t)T.d_nanc[k] << (4 * (k & 15));
    for (int a = 0; a < T.d_nanc[k]; a++) T.d_ancm[k] |= 1ull << T.d_anc[k][a];
    T.d_madr[k] = m.dof_Madr[k];
  }
  for (int a = 0; a < DM_NU; a++) {
    int k = m.act_dof[a];
    T.d_act[k] = a; T.d_gear[k] = (float)m.act_gear[a];
    T.d_clo[k] = (float)m.act_ctrlrange[a][0]; T.d_chi[k] = (float)m.act_ctrlrange[a][1];
  }
  for (int g = 0; g < DM_NGEOM; g++) {
    T.g_body[g] = m.geom_body[g]; T.g_type[g] = m.geom_type[g]; T.g_condim[g] = m.geom_condim[g];
    for (int i = 0; i < 3; i++) { T.g_pos[g][i] = (float)m.geom_pos[g][i]; T.g_size[g][i] = (float)m.geom_size[g][i]; }
    const double *q = m.geom_quat[g];
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double M[9] = {w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y),
                   2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
                   2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z};
    for (int i = 0; i < 9; i++) T.g_mat[g][i] = (float)M[i];
    T.g_rbound[g] = (float)m.geom_rbound[g]; T.g_margin[g] = (float)m.geom_margin[g];
    T.g_mu[g] = (float)m.geom_friction[g][0];
  }
  for (int p = 0; p < DM_MAXPAIR; p++) { T.p_g1[p] = (int16_t)m.pair_geom1[p]; T.p_g2[p] = (int16_t)m.pair_geom2[p]; }
  for (int p = 0; p < m.npair; p++) {
    int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p];
    DmPairDev &pr = T.pairs[p];
    pr.g1 = (int16_t)g1; pr.g2 = (int16_t)g2;
    pr.t1 = (int8_t)m.geom_type[g1]; pr.t2 = (int8_t)m.geom_type[g2];
    pr.margin = (float)fmax(m.geom_margin[g1], m.geom_margin[g2]);
    pr.rbsum = (float)((m.geom_type[g1] == DM_GEOM_PLANE ? 0.0 : m.geom_rbound[g1]) + m.geom_rbound[g2]);
    for (int i = 0; i < 3; i++) { pr.z1[i] = (float)m.geom_size[g1][i]; pr.z2[i] = (float)m.geom_size[g2][i]; }
  }
  int p = 0;
  for (int b = 0; b < DMK_MAXANC; b++)
    for (int a = 0; a <= b; a++) { T.tri_a[p] = (uint8_t)a; T.tri_b[p] = (uint8_t)b; p++; }
}

static int check_model(DmEngine *e, const DmModel &m) {
  if (m.nq != DM_NQ || m.nv != DM_NV || m.nu != DM_NU || m.nbody != DM_NBODY || m.ngeom != DM_NGEOM)
    return fail(e, DM_EINVAL, "model dimensions do not match the compiled-in humanoid3d dimensions");
  if (m.integrator != DM_INT_RK4 && m.integrator != DM_INT_EULER) return fail(e, DM_EINVAL, "integrator must be RK4 or Euler");
  if (m.npair < 0 || m.npair > DM_MAXPAIR) return fail(e, DM_EINVAL, "npair out of range");
  for (int b = 1; b < DM_NBODY; b++) {
    if (m.body_quat[b][0] != 1.0) return fail(e, DM_EINVAL, "kernels assume identity body quaternions");
    if (m.body_dofnum[b] != 1 && m.body_dofnum[b] != 3 && m.body_dofnum[b] != 6)
      return fail(e, DM_EINVAL, "kernels assume 1- or 3-hinge bodies under a free root");
    if (m.body_depth[b] < 1 || m.body_depth[b] > 4) return fail(e, DM_EINVAL, "tree depth > 4");
  }
  for (int j = 0; j < DM_NJNT; j++)
    for (int i = 0; i < 3; i++)
      if (m.jnt_pos[j][i] != 0.0) return fail(e, DM_EINVAL, "kernels assume joint anchors at the body origin");
  if (m.jnt_type[0] != DM_JNT_FREE || m.jnt_body[0] != 1) return fail(e, DM_EINVAL, "joint 0 must be the free root");
  for (int k = 0; k < DM_NV; k++)
    if (m.dof_parent[k] != topo::PARENT[k] || m.dof_Madr[k] != topo::MADR[k])
      return fail(e, DM_EINVAL, "dof tree differs from the compiled-in topology (regenerate csrc/dm_topology.h)");
  for (int b = 0; b < DM_NBODY; b++)
    if (m.body_parent[b] != topo::BPARENT[b])
      return fail(e, DM_EINVAL, "body tree differs from the compiled-in topology (regenerate csrc/dm_topology.h)");
  return DM_OK;
}

extern "C" int dm_create(const DmModel *model, const DmConfig *cfg, DmHandle *out) {
  if (!model || !cfg || !out || cfg->num_envs < 1) return DM_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DM_ENODEV;
  DmEngine *e = new DmEngine();
  e->cfg = *cfg;
  e->model = *model;
  e->N = cfg->num_envs;
  if (const char *w = getenv("DM_WAVES")) e->waves = atoi(w);
  int rc = check_model(e, *model);
  if (rc != DM_OK) { fprintf(stderr, "dm_create: %s\n", e->err.c_str()); delete e; return rc; }
  if (hipSetDevice(cfg->device) != hipSuccess) { delete e; return DM_ENODEV; }
  DmDev T;
  build_tables(*model, T);
  if (hipMalloc(&e->dT, sizeof(DmDev)) != hipSuccess) { delete e; return DM_ENOMEM; }
  hipMemcpy(e->dT, &T, sizeof(DmDev), hipMemcpyHostToDevice);
  size_t sb = (size_t)e->N * DMK_STATE_STRIDE * sizeof(float);
  if (hipMalloc(&e->dState, sb) != hipSuccess) { hipFree(e->dT); delete e; return DM_ENOMEM; }
  std::vector<float> init((size_t)e->N * DMK_STATE_STRIDE, 0.f);
  for (int i = 0; i < e->N; i++)
    for (int k = 0; k < DM_NQ; k++) init[(size_t)i * DMK_STATE_STRIDE + DMS_QPOS + k] = (float)model->qpos0[k];
  hipMemcpy(e->dState, init.data(), sb, hipMemcpyHostToDevice);
  if (hipMalloc(&e->dArScratch, (size_t)e->N * DMK_MAXROW * DMK_MAXROW * sizeof(float)) != hipSuccess) {
    hipFree(e->dState); hipFree(e->dT); delete e; return DM_ENOMEM;
  }
  if (hipMalloc(&e->dOrder, e->N * sizeof(int32_t)) != hipSuccess || hipMalloc(&e->dCost, e->N * sizeof(int32_t)) != hipSuccess) {
    hipFree(e->dArScratch); hipFree(e->dState); hipFree(e->dT); delete e; return DM_ENOMEM;
  }
  hipMemset(e->dCost, 0, e->N * sizeof(int32_t));
  { std::vector<int32_t> ident(e->N);
    for (int i = 0; i < e->N; i++) ident[i] = i;
    hipMemcpy(e->dOrder, ident.data(), e->N * sizeof(int32_t), hipMemcpyHostToDevice); }   // until the first dm_step schedules
  for (int i = 0; i < DmEngine::NEV; i++) { hipEventCreate(&e->ev0[i]); hipEventCreate(&e->ev1[i]); }
  *out = e;
  return DM_OK;
}

extern "C" int dm_destroy(DmHandle e) {
  if (!e) return DM_EINVAL;
  hipSetDevice(e->cfg.device);
  hipDeviceSynchronize();
  for (int i = 0; i < DM_MAX_CLIPS; i++) { if (e->dClipRows[i]) hipFree(e->dClipRows[i]); if (e->dClipReset[i]) hipFree(e->dClipReset[i]); }
  if (e->dState) hipFree(e->dState);
  if (e->dArScratch) hipFree(e->dArScratch);
  if (e->dOrder) hipFree(e->dOrder);
  if (e->dCost) hipFree(e->dCost);
  if (e->dT) hipFree(e->dT);
  for (int i = 0; i < DmEngine::NEV; i++) { if (e->ev0[i]) hipEventDestroy(e->ev0[i]); if (e->ev1[i]) hipEventDestroy(e->ev1[i]); }
  delete e;
  return DM_OK;
}

extern "C" const char *dm_last_error(DmHandle e) { return e ? e->err.c_str() : "null handle"; }
extern "C" int dm_num_envs(DmHandle e) { return e ? e->N : DM_EINVAL; }
extern "C" int dm_obs_dim(DmHandle e) { return e ? (e->cfg.task == DM_TASK_COMBINED ? DM_NOBS_COMBINED : DM_NOBS) : DM_EINVAL; }
extern "C" int dm_terms_dim(DmHandle e) { return e ? (e->cfg.task == DM_TASK_COMBINED ? 8 : 5) : DM_EINVAL; }

extern "C" int dm_load_clip(DmHandle e, int clip_id, int L, const double *qpos, const double *qvel,
                            const double *body_xpos, const double *geom_xpos) {
  if (!e || clip_id < 0 || clip_id >= DM_MAX_CLIPS || L < 1 || !qpos || !qvel || !body_xpos || !geom_xpos)
    return fail(e, DM_EINVAL, "dm_load_clip: bad argument");
  HIPCHK(e, hipSetDevice(e->cfg.device));
  const DmModel &m = e->model;
  std::vector<float> rows((size_t)L * DMK_CLIP_ROW, 0.f), reset((size_t)L * DMK_RESET_ROW, 0.f);
  double mt = 0;
  for (int b = 0; b < DM_NBODY; b++) mt += m.body_mass[b];
  for (int f = 0; f < L; f++) {
    float *r = &rows[(size_t)f * DMK_CLIP_ROW];
    const double *q = qpos + (size_t)f * DM_NQ, *v = qvel + (size_t)f * DM_NV;
    for (int i = 0; i < 28; i++) { r[i] = (float)q[7 + i]; r[28 + i] = (float)v[6 + i]; }
    for (int i = 0; i < 4; i++) r[56 + i] = (float)q[3 + i];
    for (int ee = 0; ee < 4; ee++)
      for (int i = 0; i < 3; i++) r[60 + 3 * ee + i] = (float)geom_xpos[((size_t)f * DM_NGEOM + m.ee_geom[ee]) * 3 + i];
    for (int i = 0; i < 3; i++) {
      double c = 0;
      for (int b = 0; b < DM_NBODY; b++) c += body_xpos[((size_t)f * DM_NBODY + b) * 3 + i] * m.body_mass[b];
      r[72 + i] = (float)(c / mt);
      r[75 + i] = (float)q[i];
    }
    r[78] = (float)v[0]; r[79] = (float)v[1];   // root linear velocity xy (DPCombinedEnv task reward)
    float *rr = &reset[(size_t)f * DMK_RESET_ROW];
    for (int i = 0; i < DM_NQ; i++) rr[i] = (float)q[i];
    for (int i = 0; i < DM_NV; i++) rr[35 + i] = (float)v[i];
  }
  if (e->dClipRows[clip_id]) { hipFree(e->dClipRows[clip_id]); hipFree(e->dClipReset[clip_id]); }
  HIPCHK(e, hipMalloc(&e->dClipRows[clip_id], rows.size() * sizeof(float)));
  HIPCHK(e, hipMalloc(&e->dClipReset[clip_id], reset.size() * sizeof(float)));
  HIPCHK(e, hipMemcpy(e->dClipRows[clip_id], rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(e->dClipReset[clip_id], reset.data(), reset.size() * sizeof(float), hipMemcpyHostToDevice));
  e->clipL[clip_id] = L;
  return DM_OK;
}

__global__ void dm_schedule_kernel(const int32_t *cost, int32_t *order, int n);

__global__ void dm_set_clip_kernel(float *state, const int32_t *ids, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) reinterpret_cast<int *>(state + (size_t)i * DMK_STATE_STRIDE)[DMS_CLIP] = ids ? ids[i] : 0;
}
__global__ void dm_get_clip_kernel(const float *state, int32_t *ids, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ids[i] = reinterpret_cast<const int *>(state + (size_t)i * DMK_STATE_STRIDE)[DMS_CLIP];
}
__global__ void dm_counters_kernel(float *state, int n, int32_t *idx, int32_t *len, float *rew, const int32_t *sidx,
                                   const int32_t *slen) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int *si = reinterpret_cast<int *>(state + (size_t)i * DMK_STATE_STRIDE);
  if (sidx) si[DMS_IDX] = sidx[i];
  if (slen) si[DMS_EPLEN] = slen[i];
  if (idx) idx[i] = si[DMS_IDX];
  if (len) len[i] = si[DMS_EPLEN];
  if (rew) rew[i] = state[(size_t)i * DMK_STATE_STRIDE + DMS_EPREW];
}
__global__ void dm_get_state_kernel(const float *state, const int32_t *ids, int n, int N, float *qpos, float *qvel,
                                    float *warm, float *ctrl) {
  int slot = blockIdx.x, lane = threadIdx.x;
  if (slot >= n) return;
  int env = ids ? ids[slot] : slot;
  if (env < 0 || env >= N) return;
  const float *st = state + (size_t)env * DMK_STATE_STRIDE;
  if (qpos && lane < DM_NQ) qpos[(size_t)slot * DM_NQ + lane] = st[DMS_QPOS + lane];
  if (qvel && lane < DM_NV) qvel[(size_t)slot * DM_NV + lane] = st[DMS_QVEL + lane];
  if (warm && lane < DM_NV) warm[(size_t)slot * DM_NV + lane] = st[DMS_WARM + lane];
  if (ctrl && lane < DM_NU) ctrl[(size_t)slot * DM_NU + lane] = st[DMS_CTRL + lane];
}

extern "C" int dm_get_env_clips(DmHandle e, int32_t *clip_ids, void *stream) {
  if (!e || !clip_ids) return DM_EINVAL;
  HIPCHK(e, hipSetDevice(e->cfg.device));
  hipLaunchKernelGGL(dm_get_clip_kernel, dim3((e->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->dState, clip_ids, e->N);
  HIPCHK(e, hipGetLastError());
  return DM_OK;
}

extern "C" int dm_set_clip_flags(DmHandle e, int clip_id, int flags) {
  if (!e || clip_id < 0 || clip_id >= DM_MAX_CLIPS) return DM_EINVAL;
  e->clipFlags[clip_id] = flags;
  return DM_OK;
}

extern "C" int dm_set_env_clips(DmHandle e, const int32_t *clip_ids, void *stream) {
  if (!e) return DM_EINVAL;
  HIPCHK(e, hipSetDevice(e->cfg.device));
  hipLaunchKernelGGL(dm_set_clip_kernel, dim3((e->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->dState, clip_ids, e->N);
  HIPCHK(e, hipGetLastError());
  return DM_OK;
}

static void fill_launch(DmEngine *e, DmLaunch &P, int mode) {
  memset(&P, 0, sizeof(P));
  P.T = e->dT;
  P.state = e->dState;
  P.ar_scratch = e->dArScratch;
  for (int i = 0; i < DM_MAX_CLIPS; i++) { P.clips[i].rows = e->dClipRows[i]; P.clips[i].reset = e->dClipReset[i]; P.clips[i].L = e->clipL[i]; P.clips[i].flags = e->clipFlags[i]; }
  P.N = e->N; P.mode = mode; P.auto_reset = e->cfg.auto_reset; P.max_ep_length = e->cfg.max_ep_length;
  P.vel_obs_scale = e->cfg.vel_obs_scale; P.low_z = e->cfg.low_z; P.high_z = e->cfg.high_z; P.obs_bound = e->cfg.obs_bound;
  P.w_pose = e->cfg.w_pose; P.w_vel = e->cfg.w_vel; P.w_ee = e->cfg.w_end_eff; P.w_com = e->cfg.w_com; P.w_jl = e->cfg.w_joint_limit;
  P.seed = e->cfg.seed;
  P.amnesty_steps = e->cfg.amnesty_steps; P.to_getup_len = e->cfg.to_getup_len;
  P.integrator = e->cfg.integrator == DM_CFG_INT_EULER ? DM_INT_EULER : e->cfg.integrator == DM_CFG_INT_RK4 ? DM_INT_RK4 : e->model.integrator;
  P.f8 = e->cfg.stale_contact_slots ? 1 : 0;
  P.debug = e->debug;
}

static int launch(DmEngine *e, DmLaunch &P, int nslots, void *stream) {
  P.nslots = nslots;
  if (e->clipL[0] < 1) return fail(e, DM_EINVAL, "no clip loaded (dm_load_clip clip 0 first)");
  if (e->cfg.task == DM_TASK_COMBINED && (e->clipL[1] < 1 || e->clipL[2] < 2))
    return fail(e, DM_EINVAL, "combined task needs clips 0,1,2 = walk, run, getup");
  HIPCHK(e, hipSetDevice(e->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  const int evi = (int)(e->nrec % DmEngine::NEV);
  const bool rec = e->timing && (e->nlaunch++ % e->stride) == 0;
  if (rec) hipEventRecord(e->ev0[evi], s);
  const dim3 grid((P.nslots + DMK_ENVS_PER_BLOCK - 1) / DMK_ENVS_PER_BLOCK), block(64 * DMK_ENVS_PER_BLOCK);
  const bool three_waves = e->waves == 3 || (e->waves == 0 && P.nslots >= 3072);
  if (e->cfg.task == DM_TASK_COMBINED) {
    if (three_waves) hipLaunchKernelGGL(dm_step_combined_kernel_w3, grid, block, 0, s, P);
    else hipLaunchKernelGGL(dm_step_combined_kernel, grid, block, 0, s, P);
#ifdef DM_EXPERIMENT_W4
  } else if (e->waves == 4) {
    hipLaunchKernelGGL(dm_step_kernel_w4, grid, block, 0, s, P);
#endif
  } else if (three_waves) {
    // three waves per SIMD (168 VGPRs): faster than the two-wave build from 3 072 envs up since r2 (per-stage laundering of
    // the lane id: no hoisted lane-compare masks spilled across the stage loop): 13.4 vs 12.9 M at 4 096 envs, 21.4 vs
    // 17.5 M at 65 536
    hipLaunchKernelGGL(dm_step_kernel_w3, grid, block, 0, s, P);
  } else {
    hipLaunchKernelGGL(dm_step_kernel, grid, block, 0, s, P);
  }
  if (rec) { hipEventRecord(e->ev1[evi], s); e->nrec++; }
  HIPCHK(e, hipGetLastError());
  return DM_OK;
}

extern "C" int dm_reset(DmHandle e, const uint8_t *mask, const int32_t *idx_init, float *obs_out, void *stream) {
  if (!e) return DM_EINVAL;
  DmLaunch P;
  fill_launch(e, P, DMK_MODE_RESET);
  P.mask = mask; P.idx_init = idx_init; P.obs = obs_out;
  return launch(e, P, e->N, stream);
}

extern "C" int dm_step(DmHandle e, const float *actions, float *obs, float *rew, uint8_t *done, float *terms,
                       int32_t *reason, float *terminal_obs, void *stream) {
  if (!e || !actions || !obs || !rew || !done) return fail(e, DM_EINVAL, "dm_step: null buffer");
  DmLaunch P;
  fill_launch(e, P, DMK_MODE_STEP);
  P.actions = actions; P.obs = obs; P.rew = rew; P.done = done; P.terms = terms; P.reason = reason; P.terminal_obs = terminal_obs;
  P.cost = e->dCost;
  if (e->cfg.lpt_schedule) {
    HIPCHK(e, hipSetDevice(e->cfg.device));
    hipLaunchKernelGGL(dm_schedule_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, e->dCost, e->dOrder, e->N);
    P.env_ids = e->dOrder;
  }
  return launch(e, P, e->N, stream);
}

extern "C" int dm_physics_step(DmHandle e, const float *actions, void *stream) {
  if (!e || !actions) return fail(e, DM_EINVAL, "dm_physics_step: null buffer");
  DmLaunch P;
  fill_launch(e, P, DMK_MODE_PHYSICS);
  P.actions = actions;
  P.auto_reset = 0;
  if (e->cfg.lpt_schedule) {   // same launch order as a dm_step from this state: the work estimates of the last dm_step
    HIPCHK(e, hipSetDevice(e->cfg.device));
    hipLaunchKernelGGL(dm_schedule_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, e->dCost, e->dOrder, e->N);
    P.env_ids = e->dOrder;
  }
  return launch(e, P, e->N, stream);
}

extern "C" int dm_step_forced(DmHandle e, const float *qpos, const float *qvel, float *obs, float *rew, uint8_t *done,
                              float *terms, int32_t *reason, void *stream) {
  if (!e || !qpos || !qvel || !obs || !rew || !done) return fail(e, DM_EINVAL, "dm_step_forced: null buffer");
  DmLaunch P;
  fill_launch(e, P, DMK_MODE_FORCED);
  P.in_qpos = qpos; P.in_qvel = qvel; P.obs = obs; P.rew = rew; P.done = done; P.terms = terms; P.reason = reason;
  P.auto_reset = 0;
  return launch(e, P, e->N, stream);
}

extern "C" int dm_set_state(DmHandle e, const int32_t *env_ids, int n, const float *qpos, const float *qvel,
                            const float *warm, const float *ctrl, int run_forward, void *stream) {
  if (!e || !qpos || !qvel || n < 1 || n > e->N) return fail(e, DM_EINVAL, "dm_set_state: bad argument");
  DmLaunch P;
  fill_launch(e, P, DMK_MODE_SETSTATE);
  P.env_ids = env_ids; P.in_qpos = qpos; P.in_qvel = qvel; P.in_warm = warm; P.in_ctrl = ctrl; P.run_forward = run_forward;
  return launch(e, P, n, stream);
}

extern "C" int dm_forward(DmHandle e, const int32_t *env_ids, int n, void *stream) {
  if (!e || n < 1 || n > e->N) return fail(e, DM_EINVAL, "dm_forward: bad argument");
  DmLaunch P;
  fill_launch(e, P, DMK_MODE_SETSTATE);
  P.env_ids = env_ids; P.run_forward = 1;     // in_qpos == null: state is kept
  return launch(e, P, n, stream);
}

extern "C" int dm_get_state(DmHandle e, const int32_t *env_ids, int n, float *qpos, float *qvel, float *warm,
                            float *ctrl, void *stream) {
  if (!e || n < 1 || n > e->N) return fail(e, DM_EINVAL, "dm_get_state: bad argument");
  HIPCHK(e, hipSetDevice(e->cfg.device));
  hipLaunchKernelGGL(dm_get_state_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, e->dState, env_ids, n, e->N, qpos, qvel, warm, ctrl);
  HIPCHK(e, hipGetLastError());
  return DM_OK;
}

extern "C" int dm_get_counters(DmHandle e, int32_t *idx, int32_t *len, float *rew, void *stream) {
  if (!e) return DM_EINVAL;
  HIPCHK(e, hipSetDevice(e->cfg.device));
  hipLaunchKernelGGL(dm_counters_kernel, dim3((e->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->dState, e->N, idx, len, rew,
                     (const int32_t *)nullptr, (const int32_t *)nullptr);
  HIPCHK(e, hipGetLastError());
  return DM_OK;
}
extern "C" int dm_set_counters(DmHandle e, const int32_t *idx, const int32_t *len, void *stream) {
  if (!e) return DM_EINVAL;
  HIPCHK(e, hipSetDevice(e->cfg.device));
  hipLaunchKernelGGL(dm_counters_kernel, dim3((e->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->dState, e->N,
                     (int32_t *)nullptr, (int32_t *)nullptr, (float *)nullptr, idx, len);
  HIPCHK(e, hipGetLastError());
  return DM_OK;
}

// Bucket sort of the per-env work estimates, heaviest first (one 1024-thread block; N <= 65536 in practice).
__global__ void dm_schedule_kernel(const int32_t *cost, int32_t *order, int n) {
  __shared__ int hist[256];
  __shared__ int base[256];
  const int tid = threadIdx.x;
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    int b = cost[i] >> 5;
    b = 255 - (b > 255 ? 255 : (b < 0 ? 0 : b));
    atomicAdd(&hist[b], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int b = 0; b < 256; b++) { base[b] = acc; acc += hist[b]; }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    int b = cost[i] >> 5;
    b = 255 - (b > 255 ? 255 : (b < 0 ? 0 : b));
    order[atomicAdd(&base[b], 1)] = i;
  }
}

extern "C" int dm_get_work(DmHandle e, int32_t *work_out, void *stream) {
  if (!e || !work_out) return DM_EINVAL;
  HIPCHK(e, hipSetDevice(e->cfg.device));
  HIPCHK(e, hipMemcpyAsync(work_out, e->dCost, e->N * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return DM_OK;
}

extern "C" int dm_set_seed(DmHandle e, uint64_t seed) {
  if (!e) return DM_EINVAL;
  e->cfg.seed = seed;
  return DM_OK;
}

extern "C" int dm_set_debug(DmHandle e, float *buf) {
  if (!e) return DM_EINVAL;
  e->debug = buf;
  return DM_OK;
}

extern "C" int dm_fill_random_actions(DmHandle e, float *actions, uint32_t step_index, void *stream) {
  if (!e || !actions) return DM_EINVAL;
  HIPCHK(e, hipSetDevice(e->cfg.device));
  int n = e->N * DM_NU;
  hipLaunchKernelGGL(dm_fill_actions_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, actions, e->N, e->cfg.seed, step_index);
  HIPCHK(e, hipGetLastError());
  return DM_OK;
}

extern "C" int dm_enable_timing(DmHandle e, int enable) {
  if (!e) return DM_EINVAL;
  e->timing = enable != 0;
  e->stride = enable > 1 ? enable : 1;
  e->nrec = 0;
  e->nlaunch = 0;
  return DM_OK;
}
extern "C" int dm_last_step_ms(DmHandle e, float *ms) {
  if (!e || !ms || !e->timing || e->nrec < 1) return DM_EINVAL;
  const int i = (int)((e->nrec - 1) % DmEngine::NEV);
  HIPCHK(e, hipEventSynchronize(e->ev1[i]));
  HIPCHK(e, hipEventElapsedTime(ms, e->ev0[i], e->ev1[i]));
  return DM_OK;
}
extern "C" int dm_mean_step_ms(DmHandle e, float *ms, int32_t *count) {
  if (!e || !ms || !e->timing || e->nrec < 1) return DM_EINVAL;
  const long n = e->nrec < DmEngine::NEV ? e->nrec : DmEngine::NEV;
  double acc = 0;
  for (long k = 0; k < n; k++) {
    const int i = (int)((e->nrec - 1 - k) % DmEngine::NEV);
    float t = 0;
    HIPCHK(e, hipEventSynchronize(e->ev1[i]));
    HIPCHK(e, hipEventElapsedTime(&t, e->ev0[i], e->ev1[i]));
    acc += t;
  }
  *ms = (float)(acc / (double)n);
  if (count) *count = (int32_t)n;
  return DM_OK;
}
