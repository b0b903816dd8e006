/*
 * dm_oracle.c — CPU oracle (TEST INFRASTRUCTURE, not product code).
 * See dm_oracle.h for scope and parity status ("physics parity unpinned").
 *
 * Layout follows the reference call stack:
 *   DPEnv.step                      src/deepmimic_env.py:335-484
 *     do_simulation -> mj_step      [EXT] MuJoCo "Computation" chapter
 *     get_obs                       src/deepmimic_env.py:33-143
 *     calc_imitation_reward         src/deepmimic_env.py:193-256
 *     termination / counters        src/deepmimic_env.py:418-476
 */
#include "dm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NQ DM_NQ
#define NV DM_NV
#define NU DM_NU
#define NB DM_NBODY
#define NG DM_NGEOM
#define MINVAL 1e-15
#define MAXVAL 1e10

/* ------------------------------------------------------------------ sensitivity-study switches
 * Defaults = the MuJoCo behaviour restated from SURVEY App. B.  tests/sensitivity_extracted_policy.py flips ONE of
 * them at a time to see how the reference's MuJoCo-trained policy reacts.  Nothing else sets them. */
static struct {
  double refsafe;        /* 1: solref time constant clamped to >= 2h (mjDSBL_REFSAFE off) */
  double redge;          /* pyramid edge regulariser Rpy = redge * mu^2 * R(first edge); MuJoCo: 2 */
  double warmstart;      /* 0: MuJoCo rule (keep qacc_warmstart forces if their dual cost < 0); 1: always zero; 2: always warm */
  double pgs_early_exit; /* 1: stop when scaled improvement < tolerance */
  double planebox_all;   /* 1: plane-box also keeps corners with ldist > 0 (MuJoCo skips them) */
  double diag_scale;     /* multiplies diagApprox (invweight0) of every row */
  double solref_limit;   /* > 0: time constant of joint-limit rows only (solreflimit) */
  double mu_scale;       /* multiplies the contact friction coefficient */
  double stale_ws;       /* 1: RK stages 2..4 warm-start from what the step found (later MuJoCo saves qacc_warmstart in mj_advance) */
  double support_tie;    /* support mappings: candidates within this distance of the maximum count as tied and the lowest vertex index
                          * (box / cylinder: the positive side) wins.  0 = libccd / MuJoCo literally: the first strict maximum, i.e.
                          * a tie is decided by the last bit of the dot products.  See dm_convex.h "ties". */
} TW = {1, 2, 0, 1, 0, 1, 0, 1, 0, 1e-12};

int dmo_set_tweak(const char *name, double v) {
  if (!strcmp(name, "reset")) { TW.refsafe = 1; TW.redge = 2; TW.warmstart = 0; TW.pgs_early_exit = 1; TW.planebox_all = 0;
                                TW.diag_scale = 1; TW.solref_limit = 0; TW.mu_scale = 1; TW.stale_ws = 0; TW.support_tie = 1e-12; return 0; }
#define TWK(nm) if (!strcmp(name, #nm)) { TW.nm = v; return 0; }
  TWK(refsafe) TWK(redge) TWK(warmstart) TWK(pgs_early_exit) TWK(planebox_all) TWK(diag_scale) TWK(solref_limit)
  TWK(mu_scale) TWK(stale_ws) TWK(support_tie)
#undef TWK
  return -1;
}

/* ------------------------------------------------------------------ small math */
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double *r, const double *a, const double *b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double norm3(const double *a) { return sqrt(dot3(a, a)); }
static double normalize3(double *a) {
  double n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; }
  else { a[0] /= n; a[1] /= n; a[2] /= n; }
  return n;
}
static void mul_quat(double *r, const double *a, const double *b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void normalize4(double *q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static void quat2mat(double *m, const double *q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void rot_vec(double *r, const double *m, const double *v) { /* r = M v */
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void rot_vec_t(double *r, const double *m, const double *v) { /* r = M^T v */
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  double y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  double z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void rot_vec_quat(double *r, const double *v, const double *q) {
  double m[9];
  quat2mat(m, q);
  rot_vec(r, m, v);
}
static void axis_angle_quat(double *q, const double *axis, double ang) {
  double s = sin(0.5 * ang);
  q[0] = cos(0.5 * ang); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static void mul_mat3(double *r, const double *a, const double *b) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(r, t, sizeof t);
}

/* spatial algebra in the COM-based frame, vectors are [angular(3), linear(3)] */
static void mul_inert_vec(double *r, const double *I, const double *v) {
  /* I = [xx yy zz xy xz yz, hx hy hz, m], h = m * offset */
  r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
  r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] - I[6] * v[5] + I[8] * v[3];
  r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
  r[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
  r[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
  r[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}
static void cross_motion(double *r, const double *vel, const double *v) {
  double a[3], b[3], c[3];
  cross3(a, vel, v);
  cross3(b, vel, v + 3);
  cross3(c, vel + 3, v);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void cross_force(double *r, const double *vel, const double *f) {
  double a[3], b[3], c[3];
  cross3(a, vel, f);
  cross3(b, vel + 3, f + 3);
  cross3(c, vel, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
static double dot6(const double *a, const double *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

/* ------------------------------------------------------------------ data */
DmoData *dmo_data_new(const DmModel *m) {
  DmoData *d = (DmoData *)calloc(1, sizeof(DmoData));
  d->maxcon = DMO_MAXCON;
  d->maxrow = DMO_MAXROW;
  d->efc_J = (double *)calloc((size_t)DMO_MAXROW * NV, sizeof(double));
  d->efc_AR = (double *)calloc((size_t)DMO_MAXROW * DMO_MAXROW, sizeof(double));
  dmo_data_reset(m, d);
  return d;
}
void dmo_data_free(DmoData *d) {
  if (!d) return;
  free(d->efc_J);
  free(d->efc_AR);
  free(d);
}
void dmo_data_reset(const DmModel *m, DmoData *d) {
  memcpy(d->qpos, m->qpos0, sizeof d->qpos);
  memset(d->qvel, 0, sizeof d->qvel);
  memset(d->ctrl, 0, sizeof d->ctrl);
  memset(d->qacc_warmstart, 0, sizeof d->qacc_warmstart);
  d->time = 0;
  d->ncon = d->nefc = 0;
  memset(d->contact, 0, sizeof d->contact); /* mj_resetData clears the contact array: what F8's stale slots can hold */
}

/* ------------------------------------------------------------------ position stage */
static void kinematics(const DmModel *m, DmoData *d) { /* [EXT] mj_kinematics */
  memset(d->xpos[0], 0, sizeof d->xpos[0]);
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  quat2mat(d->xmat[0], d->xquat[0]);
  memset(d->xipos[0], 0, sizeof d->xipos[0]);
  for (int b = 1; b < NB; b++) {
    int p = m->body_parent[b], ja = m->body_jntadr[b], jn = m->body_jntnum[b];
    double pos[3], q[4];
    if (jn == 1 && m->jnt_type[ja] == DM_JNT_FREE) {
      int qa = m->jnt_qposadr[ja];
      normalize4(d->qpos + qa + 3); /* MuJoCo normalises the stored quaternion in place */
      memcpy(pos, d->qpos + qa, sizeof pos);
      memcpy(q, d->qpos + qa + 3, sizeof q);
      memcpy(d->xanchor[ja], pos, sizeof pos);
      rot_vec_quat(d->xaxis[ja], m->jnt_axis[ja], q);
    } else {
      double t[3];
      rot_vec(t, d->xmat[p], m->body_pos[b]);
      for (int i = 0; i < 3; i++) pos[i] = d->xpos[p][i] + t[i];
      mul_quat(q, d->xquat[p], m->body_quat[b]);
      for (int j = ja; j < ja + jn; j++) {
        double v[3], ql[4];
        rot_vec_quat(v, m->jnt_pos[j], q);
        for (int i = 0; i < 3; i++) d->xanchor[j][i] = v[i] + pos[i];
        rot_vec_quat(d->xaxis[j], m->jnt_axis[j], q);
        int qa = m->jnt_qposadr[j];
        axis_angle_quat(ql, m->jnt_axis[j], d->qpos[qa] - m->qpos0[qa]);
        mul_quat(q, q, ql);
        rot_vec_quat(v, m->jnt_pos[j], q);
        for (int i = 0; i < 3; i++) pos[i] = d->xanchor[j][i] - v[i];
      }
    }
    normalize4(q);
    memcpy(d->xpos[b], pos, sizeof pos);
    memcpy(d->xquat[b], q, sizeof q);
    quat2mat(d->xmat[b], q);
    double t[3];
    rot_vec(t, d->xmat[b], m->body_ipos[b]);
    for (int i = 0; i < 3; i++) d->xipos[b][i] = pos[i] + t[i];
  }
  for (int g = 0; g < NG; g++) {
    int b = m->geom_body[g];
    double t[3], gm[9];
    rot_vec(t, d->xmat[b], m->geom_pos[g]);
    for (int i = 0; i < 3; i++) d->geom_xpos[g][i] = d->xpos[b][i] + t[i];
    quat2mat(gm, m->geom_quat[g]);
    mul_mat3(d->geom_xmat[g], d->xmat[b], gm);
  }
}

static void com_pos(const DmModel *m, DmoData *d) { /* [EXT] mj_comPos */
  double mt = 0, c[3] = {0, 0, 0};
  for (int b = 1; b < NB; b++) {
    mt += m->body_mass[b];
    for (int i = 0; i < 3; i++) c[i] += m->body_mass[b] * d->xipos[b][i];
  }
  for (int i = 0; i < 3; i++) d->subtree_com[i] = c[i] / mt;
  memset(d->cinert[0], 0, sizeof d->cinert[0]);
  for (int b = 1; b < NB; b++) {
    /* inertia about subtree_com in world axes: R I R^T + m(|o|^2 1 - o o^T) */
    const double *I = m->body_inertia[b];
    double Ib[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]};
    double Rt[9], T[9], W[9];
    const double *R = d->xmat[b];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Rt[3 * i + j] = R[3 * j + i];
    mul_mat3(T, R, Ib);
    mul_mat3(W, T, Rt);
    double o[3], mass = m->body_mass[b];
    for (int i = 0; i < 3; i++) o[i] = d->xipos[b][i] - d->subtree_com[i];
    double oo = dot3(o, o);
    double *ci = d->cinert[b];
    ci[0] = W[0] + mass * (oo - o[0] * o[0]);
    ci[1] = W[4] + mass * (oo - o[1] * o[1]);
    ci[2] = W[8] + mass * (oo - o[2] * o[2]);
    ci[3] = W[1] - mass * o[0] * o[1];
    ci[4] = W[2] - mass * o[0] * o[2];
    ci[5] = W[5] - mass * o[1] * o[2];
    ci[6] = mass * o[0]; ci[7] = mass * o[1]; ci[8] = mass * o[2];
    ci[9] = mass;
  }
  for (int j = 0; j < DM_NJNT; j++) {
    int b = m->jnt_body[j], da = m->jnt_dofadr[j];
    double off[3];
    for (int i = 0; i < 3; i++) off[i] = d->subtree_com[i] - d->xanchor[j][i];
    if (m->jnt_type[j] == DM_JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        memset(d->cdof[da + k], 0, sizeof d->cdof[0]);
        d->cdof[da + k][3 + k] = 1;
      }
      for (int k = 0; k < 3; k++) {
        double ax[3] = {d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k]};
        memcpy(d->cdof[da + 3 + k], ax, sizeof ax);
        cross3(d->cdof[da + 3 + k] + 3, ax, off);
      }
    } else {
      memcpy(d->cdof[da], d->xaxis[j], 3 * sizeof(double));
      cross3(d->cdof[da] + 3, d->xaxis[j], off);
    }
  }
}

static void crb(const DmModel *m, DmoData *d) { /* [EXT] mj_crb */
  memcpy(d->crb, d->cinert, sizeof d->crb);
  for (int b = NB - 1; b > 0; b--) {
    int p = m->body_parent[b];
    if (p > 0)
      for (int i = 0; i < 10; i++) d->crb[p][i] += d->crb[b][i];
  }
  memset(d->qM, 0, sizeof d->qM);
  for (int i = 0; i < NV; i++) {
    double buf[6];
    int adr = m->dof_Madr[i];
    mul_inert_vec(buf, d->crb[m->dof_body[i]], d->cdof[i]);
    d->qM[adr] += m->dof_armature[i];
    for (int j = i; j >= 0; j = m->dof_parent[j]) d->qM[adr++] += dot6(d->cdof[j], buf);
  }
}

static void factor_m(const DmModel *m, DmoData *d) { /* [EXT] mj_factorM: M = L^T D L */
  double *L = d->qLD;
  memcpy(L, d->qM, sizeof d->qM);
  for (int k = NV - 1; k >= 0; k--) {
    int kk = m->dof_Madr[k], ki = kk + 1;
    for (int i = m->dof_parent[k]; i >= 0; i = m->dof_parent[i], ki++) {
      double tmp = L[ki] / L[kk];
      int ij = m->dof_Madr[i], kj = ki;
      for (int j = i; j >= 0; j = m->dof_parent[j]) L[ij++] -= L[kj++] * tmp;
      L[ki] = tmp;
    }
  }
  for (int i = 0; i < NV; i++) {
    d->qLDiagInv[i] = 1.0 / L[m->dof_Madr[i]];
    d->qLDiagSqrtInv[i] = 1.0 / sqrt(L[m->dof_Madr[i]]);
  }
}
static void solve_lt(const DmModel *m, const DmoData *d, double *x) { /* x <- L^-T x */
  for (int i = NV - 1; i >= 0; i--) {
    int a = m->dof_Madr[i] + 1;
    for (int j = m->dof_parent[i]; j >= 0; j = m->dof_parent[j]) x[j] -= d->qLD[a++] * x[i];
  }
}
static void solve_l(const DmModel *m, const DmoData *d, double *x) { /* x <- L^-1 x */
  for (int i = 0; i < NV; i++) {
    int a = m->dof_Madr[i] + 1;
    for (int j = m->dof_parent[i]; j >= 0; j = m->dof_parent[j]) x[i] -= d->qLD[a++] * x[j];
  }
}
static void solve_m(const DmModel *m, const DmoData *d, double *x) { /* x <- M^-1 x */
  solve_lt(m, d, x);
  for (int i = 0; i < NV; i++) x[i] *= d->qLDiagInv[i];
  solve_l(m, d, x);
}
static void solve_m2(const DmModel *m, const DmoData *d, double *x) { /* x <- D^-1/2 L^-T x */
  solve_lt(m, d, x);
  for (int i = 0; i < NV; i++) x[i] *= d->qLDiagSqrtInv[i];
}

/* ------------------------------------------------------------------ narrowphase
 * Every routine returns the number of contacts written; frame[0:3] is the
 * normal pointing from geom1 to geom2, pos the midpoint between the surfaces. */
typedef struct { double dist, pos[3], normal[3], tangent[3]; } RawCon;

static int c_plane_sphere(RawCon *c, double margin, const double *ppos, const double *pmat,
                          const double *spos, double r) {
  double n[3] = {pmat[2], pmat[5], pmat[8]}, df[3];
  for (int i = 0; i < 3; i++) df[i] = spos[i] - ppos[i];
  double dist = dot3(df, n) - r;
  if (dist > margin) return 0;
  c->dist = dist;
  memcpy(c->normal, n, sizeof n);
  memset(c->tangent, 0, sizeof c->tangent);
  for (int i = 0; i < 3; i++) c->pos[i] = spos[i] - n[i] * (r + 0.5 * dist);
  return 1;
}
static int c_plane_capsule(RawCon *c, double margin, const double *ppos, const double *pmat,
                           const double *cpos, const double *cmat, const double *size) {
  double ax[3] = {cmat[2], cmat[5], cmat[8]}, e[3];
  int n = 0;
  for (int s = 1; s >= -1; s -= 2) { /* +segment end first, then -segment [EXT] */
    for (int i = 0; i < 3; i++) e[i] = cpos[i] + s * ax[i] * size[1];
    int k = c_plane_sphere(c + n, margin, ppos, pmat, e, size[0]);
    if (k) memcpy(c[n].tangent, ax, sizeof ax); /* align contact frame with capsule axis */
    n += k;
  }
  return n;
}
static int c_plane_box(RawCon *c, double margin, const double *ppos, const double *pmat,
                       const double *bpos, const double *bmat, const double *size) {
  double n[3] = {pmat[2], pmat[5], pmat[8]}, df[3];
  for (int i = 0; i < 3; i++) df[i] = bpos[i] - ppos[i];
  double dist = dot3(df, n);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double v[3] = {size[0] * ((i & 1) ? 1 : -1), size[1] * ((i & 2) ? 1 : -1), size[2] * ((i & 4) ? 1 : -1)};
    double corner[3];
    rot_vec(corner, bmat, v);
    double ld = dot3(n, corner);
    if (dist + ld > margin || (ld > 0 && !TW.planebox_all)) continue;
    c[cnt].dist = dist + ld;
    memcpy(c[cnt].normal, n, sizeof n);
    memset(c[cnt].tangent, 0, sizeof c[cnt].tangent);
    for (int k = 0; k < 3; k++) c[cnt].pos[k] = corner[k] + bpos[k] - n[k] * 0.5 * c[cnt].dist;
    if (++cnt >= 4) return 4;
  }
  return cnt;
}
static int c_sphere_sphere(RawCon *c, double margin, const double *p1, double r1, const double *p2, double r2) {
  double df[3];
  for (int i = 0; i < 3; i++) df[i] = p2[i] - p1[i];
  double cd = norm3(df), dist = cd - r1 - r2;
  if (dist > margin) return 0;
  c->dist = dist;
  if (cd < MINVAL) { c->normal[0] = 1; c->normal[1] = c->normal[2] = 0; }
  else for (int i = 0; i < 3; i++) c->normal[i] = df[i] / cd;
  memset(c->tangent, 0, sizeof c->tangent);
  for (int i = 0; i < 3; i++) c->pos[i] = p1[i] + c->normal[i] * (r1 + 0.5 * dist);
  return 1;
}
static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

static int c_sphere_capsule(RawCon *c, double margin, const double *spos, double r,
                            const double *cpos, const double *cmat, const double *size) {
  double ax[3] = {cmat[2], cmat[5], cmat[8]}, v[3], pt[3];
  for (int i = 0; i < 3; i++) v[i] = spos[i] - cpos[i];
  double x = clampd(dot3(ax, v), -size[1], size[1]);
  for (int i = 0; i < 3; i++) pt[i] = cpos[i] + ax[i] * x;
  return c_sphere_sphere(c, margin, spos, r, pt, size[0]);
}
static int c_capsule_capsule(RawCon *c, double margin, const double *p1, const double *m1, const double *s1,
                             const double *p2, const double *m2, const double *s2) {
  double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]}, df[3];
  for (int i = 0; i < 3; i++) df[i] = p1[i] - p2[i];
  double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2);
  double u = -dot3(a1, df), v = dot3(a2, df);
  double det = ma * mc - mb * mb;
  double v1[3], v2[3];
  if (fabs(det) >= MINVAL) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > s1[1]) { x1 = s1[1]; x2 = (v - mb * s1[1]) / mc; }
    else if (x1 < -s1[1]) { x1 = -s1[1]; x2 = (v + mb * s1[1]) / mc; }
    if (x2 > s2[1]) { x2 = s2[1]; x1 = clampd((u - mb * s2[1]) / ma, -s1[1], s1[1]); }
    else if (x2 < -s2[1]) { x2 = -s2[1]; x1 = clampd((u + mb * s2[1]) / ma, -s1[1], s1[1]); }
    for (int i = 0; i < 3; i++) { v1[i] = p1[i] + a1[i] * x1; v2[i] = p2[i] + a2[i] * x2; }
    return c_sphere_sphere(c, margin, v1, s1[0], v2, s2[0]);
  }
  /* parallel axes: test the two ends of capsule 1, then of capsule 2, keep at most 2 */
  int n = 0;
  for (int s = 1; s >= -1 && n < 2; s -= 2) {
    for (int i = 0; i < 3; i++) v1[i] = p1[i] + s * a1[i] * s1[1];
    double d2[3];
    for (int i = 0; i < 3; i++) d2[i] = v1[i] - p2[i];
    double x2 = clampd(dot3(d2, a2), -s2[1], s2[1]);
    for (int i = 0; i < 3; i++) v2[i] = p2[i] + a2[i] * x2;
    n += c_sphere_sphere(c + n, margin, v1, s1[0], v2, s2[0]);
  }
  for (int s = 1; s >= -1 && n < 2; s -= 2) {
    for (int i = 0; i < 3; i++) v2[i] = p2[i] + s * a2[i] * s2[1];
    double d1[3];
    for (int i = 0; i < 3; i++) d1[i] = v2[i] - p1[i];
    double x1 = clampd(dot3(d1, a1), -s1[1], s1[1]);
    for (int i = 0; i < 3; i++) v1[i] = p1[i] + a1[i] * x1;
    n += c_sphere_sphere(c + n, margin, v1, s1[0], v2, s2[0]);
  }
  return n;
}
static int c_sphere_box(RawCon *c, double margin, const double *spos, double r,
                        const double *bpos, const double *bmat, const double *size) {
  double t[3], ctr[3], cl[3], nl[3];
  for (int i = 0; i < 3; i++) t[i] = spos[i] - bpos[i];
  rot_vec_t(ctr, bmat, t);
  for (int i = 0; i < 3; i++) cl[i] = clampd(ctr[i], -size[i], size[i]);
  for (int i = 0; i < 3; i++) nl[i] = cl[i] - ctr[i];
  double dd = norm3(nl), dist, pl[3];
  if (dd - r > margin) return 0;
  if (dd <= MINVAL) { /* centre inside the box: nearest face */
    double closest = 2 * (size[0] + size[1] + size[2]);
    int k = 0;
    for (int i = 0; i < 6; i++) {
      double test = size[i / 2] - ((i % 2) ? -1.0 : 1.0) * ctr[i / 2];
      if (test < closest) { closest = test; k = i; }
    }
    nl[0] = nl[1] = nl[2] = 0;
    nl[k / 2] = (k % 2) ? 1.0 : -1.0; /* from the face towards the interior */
    dist = -closest - r;
  } else {
    for (int i = 0; i < 3; i++) nl[i] /= dd;
    dist = dd - r;
  }
  for (int i = 0; i < 3; i++) pl[i] = ctr[i] + nl[i] * (r + 0.5 * dist);
  c->dist = dist;
  rot_vec(c->normal, bmat, nl);
  rot_vec(t, bmat, pl);
  for (int i = 0; i < 3; i++) c->pos[i] = t[i] + bpos[i];
  memset(c->tangent, 0, sizeof c->tangent);
  return 1;
}
/* derivative (w.r.t. t) of 0.5*dist^2 from point p + a t to the box, box frame */
static double cb_grad(const double *p, const double *a, const double *size, double t) {
  double g = 0;
  for (int i = 0; i < 3; i++) {
    double x = p[i] + a[i] * t;
    double ex = x - clampd(x, -size[i], size[i]);
    g += a[i] * ex;
  }
  return g;
}
/* Capsule-box.  MuJoCo's routine is not restated line by line:
 * closest point of the capsule segment to the box (root of the monotone
 * derivative, fixed 48 bisection steps) -> sphere-box there; second contact =
 * sphere-box at the segment end farther from that point, if within margin. */
static int c_capsule_box(RawCon *c, double margin, const double *cpos, const double *cmat, const double *csize,
                         const double *bpos, const double *bmat, const double *bsize) {
  double axw[3] = {cmat[2], cmat[5], cmat[8]}, t[3], p[3], a[3];
  double hl = csize[1], r = csize[0];
  for (int i = 0; i < 3; i++) t[i] = cpos[i] - bpos[i];
  rot_vec_t(p, bmat, t);
  rot_vec_t(a, bmat, axw);
  double lo = -hl, hi = hl, ts;
  if (cb_grad(p, a, bsize, lo) >= 0) ts = lo;
  else if (cb_grad(p, a, bsize, hi) <= 0) ts = hi;
  else {
    for (int it = 0; it < 48; it++) {
      double mid = 0.5 * (lo + hi);
      if (cb_grad(p, a, bsize, mid) < 0) lo = mid; else hi = mid;
    }
    ts = 0.5 * (lo + hi);
  }
  double pt[3];
  for (int i = 0; i < 3; i++) pt[i] = cpos[i] + axw[i] * ts;
  int n = c_sphere_box(c, margin, pt, r, bpos, bmat, bsize);
  double te = (ts > 0) ? -hl : hl;
  if (fabs(te - ts) > 1e-3 * hl) {
    for (int i = 0; i < 3; i++) pt[i] = cpos[i] + axw[i] * te;
    n += c_sphere_box(c + n, margin, pt, r, bpos, bmat, bsize);
  }
  return n;
}
/* Box-box: separating-axis test over the 15 axes, then reference-face clipping
 * (face case) or closest points of the two edges (edge case).  Restated from the
 * classic SAT+clipping construction, not from MuJoCo's routine. */
static int c_box_box(RawCon *c, double margin, const double *p1, const double *R1, const double *s1,
                     const double *p2, const double *R2, const double *s2) {
  double R[9], AR[9], t[3], tw[3];
  for (int i = 0; i < 3; i++) tw[i] = p2[i] - p1[i];
  rot_vec_t(t, R1, tw);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      R[3 * i + j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];
      AR[3 * i + j] = fabs(R[3 * i + j]) + 1e-9;
    }
  double best = -1e30, bn[3] = {0, 0, 0};
  int code = -1;
  for (int i = 0; i < 3; i++) { /* faces of box 1 */
    double s = fabs(t[i]) - (s1[i] + s2[0] * AR[3 * i] + s2[1] * AR[3 * i + 1] + s2[2] * AR[3 * i + 2]);
    if (s > margin) return 0;
    if (s > best) { best = s; code = i; }
  }
  for (int j = 0; j < 3; j++) { /* faces of box 2 */
    double tj = t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j];
    double s = fabs(tj) - (s2[j] + s1[0] * AR[j] + s1[1] * AR[3 + j] + s1[2] * AR[6 + j]);
    if (s > margin) return 0;
    if (s > best) { best = s; code = 3 + j; }
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { /* edge i of box1 x edge j of box2, in box-1 frame */
      double ei[3] = {0, 0, 0}, ej[3] = {R[j], R[3 + j], R[6 + j]}, ax[3];
      ei[i] = 1;
      cross3(ax, ei, ej);
      double l = norm3(ax);
      if (l < 1e-6) continue;
      for (int k = 0; k < 3; k++) ax[k] /= l;
      double ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) ra += s1[k] * fabs(ax[k]);
      for (int k = 0; k < 3; k++) {
        double ek[3] = {R[k], R[3 + k], R[6 + k]};
        rb += s2[k] * fabs(dot3(ax, ek));
      }
      double s = fabs(dot3(t, ax)) - (ra + rb);
      if (s > margin) return 0;
      if (s > best + 0.05 * fabs(best) + 1e-6) { /* face axes win ties (bias towards face contacts) */
        best = s; code = 6 + 3 * i + j;
        memcpy(bn, ax, sizeof ax);
      }
    }
  if (code < 0) return 0;
  if (code >= 6) { /* edge-edge: one contact at the closest points of the two edges */
    int i = (code - 6) / 3, j = (code - 6) % 3;
    double n1[3] = {bn[0], bn[1], bn[2]};
    if (dot3(n1, t) < 0) for (int k = 0; k < 3; k++) n1[k] = -n1[k];
    /* support point on box 1 edge (box-1 frame), on box 2 edge */
    double pa[3], pb[3];
    for (int k = 0; k < 3; k++) pa[k] = (k == i) ? 0 : ((n1[k] > 0) ? s1[k] : -s1[k]);
    for (int k = 0; k < 3; k++) pb[k] = t[k];
    for (int k = 0; k < 3; k++) {
      if (k == j) continue;
      double ek[3] = {R[k], R[3 + k], R[6 + k]};
      double sg = (dot3(n1, ek) > 0) ? -s2[k] : s2[k];
      for (int q = 0; q < 3; q++) pb[q] += sg * ek[q];
    }
    double ua[3] = {0, 0, 0}, ub[3] = {R[j], R[3 + j], R[6 + j]}, w[3];
    ua[i] = 1;
    for (int k = 0; k < 3; k++) w[k] = pb[k] - pa[k];
    double uaub = dot3(ua, ub), q1 = dot3(ua, w), q2 = -dot3(ub, w), dd = 1 - uaub * uaub;
    double alpha = 0, beta = 0;
    if (dd > 1e-12) { alpha = (q1 + uaub * q2) / dd; beta = (uaub * q1 + q2) / dd; }
    alpha = clampd(alpha, -s1[i], s1[i]);
    beta = clampd(beta, -s2[j], s2[j]);
    double mid[3];
    for (int k = 0; k < 3; k++) mid[k] = 0.5 * ((pa[k] + ua[k] * alpha) + (pb[k] + ub[k] * beta));
    double mw[3];
    rot_vec(mw, R1, mid);
    rot_vec(c->normal, R1, n1);
    for (int k = 0; k < 3; k++) c->pos[k] = mw[k] + p1[k];
    c->dist = best;
    memset(c->tangent, 0, sizeof c->tangent);
    return 1;
  }
  /* face case: reference box = owner of the best face axis */
  const double *Ra, *Rb, *sa, *sb, *pa, *pb;
  int ax, flip;
  if (code < 3) { Ra = R1; Rb = R2; sa = s1; sb = s2; pa = p1; pb = p2; ax = code; flip = 0; }
  else { Ra = R2; Rb = R1; sa = s2; sb = s1; pa = p2; pb = p1; ax = code - 3; flip = 1; }
  double nrm[3] = {Ra[ax], Ra[3 + ax], Ra[6 + ax]}, dab[3];
  for (int k = 0; k < 3; k++) dab[k] = pb[k] - pa[k];
  if (dot3(nrm, dab) < 0) for (int k = 0; k < 3; k++) nrm[k] = -nrm[k]; /* from ref towards incident */
  /* incident face: axis of b most anti-parallel to nrm */
  int ib = 0;
  double bestd = -1;
  double nb[3];
  rot_vec_t(nb, Rb, nrm);
  for (int k = 0; k < 3; k++) if (fabs(nb[k]) > bestd) { bestd = fabs(nb[k]); ib = k; }
  double sgn = (nb[ib] > 0) ? -1.0 : 1.0; /* face whose outward normal opposes nrm */
  int u = (ib + 1) % 3, v = (ib + 2) % 3;
  double poly[16][3], tmp[16][3];
  int np = 4;
  for (int q = 0; q < 4; q++) {
    double su = (q == 0 || q == 3) ? -sb[u] : sb[u], sv = (q < 2) ? -sb[v] : sb[v];
    double loc[3];
    loc[ib] = sgn * sb[ib]; loc[u] = su; loc[v] = sv;
    double w[3], rel[3];
    rot_vec(w, Rb, loc);
    for (int k = 0; k < 3; k++) rel[k] = w[k] + pb[k] - pa[k];
    rot_vec_t(poly[q], Ra, rel); /* in reference-box frame */
  }
  int ru = (ax + 1) % 3, rv = (ax + 2) % 3;
  int axes[2] = {ru, rv};
  for (int e = 0; e < 2; e++)
    for (int sd = -1; sd <= 1; sd += 2) { /* clip against sd*x[axis] <= sa[axis] */
      int a = axes[e], nn = 0;
      for (int q = 0; q < np; q++) {
        double *P = poly[q], *Q = poly[(q + 1) % np];
        double dp = sd * P[a] - sa[a], dq = sd * Q[a] - sa[a];
        if (dp <= 0) { memcpy(tmp[nn++], P, 3 * sizeof(double)); }
        if ((dp < 0 && dq > 0) || (dp > 0 && dq < 0)) {
          double f = dp / (dp - dq);
          for (int k = 0; k < 3; k++) tmp[nn][k] = P[k] + f * (Q[k] - P[k]);
          nn++;
        }
        if (nn >= 15) break;
      }
      np = nn;
      memcpy(poly, tmp, sizeof(double) * 3 * np);
      if (np == 0) return 0;
    }
  double nl[3];
  rot_vec_t(nl, Ra, nrm); /* +-e_ax in ref frame */
  int cnt = 0;
  for (int q = 0; q < np && cnt < 8; q++) {
    double depth = dot3(nl, poly[q]) - sa[ax];
    if (depth > margin) continue;
    double pl[3], pw[3];
    for (int k = 0; k < 3; k++) pl[k] = poly[q][k] - nl[k] * 0.5 * depth;
    rot_vec(pw, Ra, pl);
    for (int k = 0; k < 3; k++) c[cnt].pos[k] = pw[k] + pa[k];
    for (int k = 0; k < 3; k++) c[cnt].normal[k] = flip ? -nrm[k] : nrm[k];
    c[cnt].dist = depth;
    memset(c[cnt].tangent, 0, sizeof c[cnt].tangent);
    cnt++;
  }
  while (cnt > 4) { /* keep the 4 deepest, preserving polygon order (one lane holds 4 slots on the GPU) */
    int w = 0;
    for (int q = 1; q < cnt; q++) if (c[q].dist >= c[w].dist) w = q;
    for (int q = w; q < cnt - 1; q++) c[q] = c[q + 1];
    cnt--;
  }
  return cnt;
}

#include "dm_convex.h"

static void make_frame(double *f) { /* [EXT] mju_makeFrame */
  normalize3(f);
  if (norm3(f + 3) < 0.5) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double t = dot3(f, f + 3);
  for (int i = 0; i < 3; i++) f[3 + i] -= t * f[i];
  normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}

/* analytic pair routines of MuJoCo's collision table [EXT mjCOLLISIONFUNC]; -1 = none (mjc_Convex / plane-mesh) */
static int analytic_pair(RawCon *rc, double margin, int t1, const double *x1, const double *M1, const double *z1, int t2,
                         const double *x2, const double *M2, const double *z2) {
  if (t1 == DM_GEOM_PLANE && t2 == DM_GEOM_SPHERE) return c_plane_sphere(rc, margin, x1, M1, x2, z2[0]);
  if (t1 == DM_GEOM_PLANE && t2 == DM_GEOM_CAPSULE) return c_plane_capsule(rc, margin, x1, M1, x2, M2, z2);
  if (t1 == DM_GEOM_PLANE && t2 == DM_GEOM_CYLINDER) return c_plane_cylinder(rc, margin, x1, M1, x2, M2, z2);
  if (t1 == DM_GEOM_PLANE && t2 == DM_GEOM_BOX) return c_plane_box(rc, margin, x1, M1, x2, M2, z2);
  if (t1 == DM_GEOM_SPHERE && t2 == DM_GEOM_SPHERE) return c_sphere_sphere(rc, margin, x1, z1[0], x2, z2[0]);
  if (t1 == DM_GEOM_SPHERE && t2 == DM_GEOM_CAPSULE) return c_sphere_capsule(rc, margin, x1, z1[0], x2, M2, z2);
  if (t1 == DM_GEOM_SPHERE && t2 == DM_GEOM_BOX) return c_sphere_box(rc, margin, x1, z1[0], x2, M2, z2);
  if (t1 == DM_GEOM_CAPSULE && t2 == DM_GEOM_CAPSULE) return c_capsule_capsule(rc, margin, x1, M1, z1, x2, M2, z2);
  if (t1 == DM_GEOM_CAPSULE && t2 == DM_GEOM_BOX) return c_capsule_box(rc, margin, x1, M1, z1, x2, M2, z2);
  if (t1 == DM_GEOM_BOX && t2 == DM_GEOM_BOX) return c_box_box(rc, margin, x1, M1, z1, x2, M2, z2);
  return -1;
}

#ifdef DM_ROBOT_G1
static void cvx_from_model(const DmModel *m, const DmoData *d, int g, CvxGeom *c) {
  c->type = m->geom_type[g];
  c->pos = d->geom_xpos[g]; c->mat = d->geom_xmat[g]; c->size = m->geom_size[g];
  c->vert = NULL; c->nvert = 0;
  memcpy(c->center, d->geom_xpos[g], sizeof c->center);
  if (c->type == DM_GEOM_MESH) {
    int me = m->geom_mesh[g];
    c->vert = m->mesh_vert[m->mesh_vertadr[me]];
    c->nvert = m->mesh_vertnum[me];
    double t[3];
    rot_vec(t, c->mat, m->mesh_center[me]);
    for (int i = 0; i < 3; i++) c->center[i] += t[i];
  }
}
#endif

/* Test hook: the MPR routine on one pair (vert1 / vert2: hull vertices for DM_GEOM_MESH, else NULL).  out: dist, pos3,
 * normal3; returns the number of contacts (0 / 1). */
int dmo_mpr(int t1, const double *x1, const double *M1, const double *z1, const double *vert1, int n1, int t2,
            const double *x2, const double *M2, const double *z2, const double *vert2, int n2, double *out) {
  CvxGeom a = {t1, x1, M1, z1, vert1, n1, {x1[0], x1[1], x1[2]}}, b = {t2, x2, M2, z2, vert2, n2, {x2[0], x2[1], x2[2]}};
  RawCon rc;
  int n = c_convex(&rc, &a, &b);
  if (n) { out[0] = rc.dist; memcpy(out + 1, rc.pos, 3 * sizeof(double)); memcpy(out + 4, rc.normal, 3 * sizeof(double)); }
  return n;
}

/* Test hook: plane against a hull (c_plane_mesh).  out: n x 7 (dist, pos3, normal3) */
int dmo_plane_mesh(const double *ppos, const double *pmat, const double *gpos, const double *gmat, const double *vert,
                   int nvert, double margin, double *out) {
  RawCon rc[4];
  int n = c_plane_mesh(rc, margin, ppos, pmat, gpos, gmat, vert, nvert);
  for (int k = 0; k < n; k++) {
    out[7 * k] = rc[k].dist; memcpy(out + 7 * k + 1, rc[k].pos, 3 * sizeof(double));
    memcpy(out + 7 * k + 4, rc[k].normal, 3 * sizeof(double));
  }
  return n;
}

/* Test hook: one primitive pair through the same dispatch as collision().  out: n x 10 doubles (dist, pos3, normal3,
 * tangent3); returns n, or -1 for an unsupported type pair. */
int dmo_narrowphase(int t1, const double *x1, const double *M1, const double *z1, int t2, const double *x2,
                    const double *M2, const double *z2, double margin, double *out) {
  RawCon rc[8];
  int n = analytic_pair(rc, margin, t1, x1, M1, z1, t2, x2, M2, z2);
  for (int k = 0; k < n; k++) {
    out[10 * k] = rc[k].dist;
    memcpy(out + 10 * k + 1, rc[k].pos, 3 * sizeof(double));
    memcpy(out + 10 * k + 4, rc[k].normal, 3 * sizeof(double));
    memcpy(out + 10 * k + 7, rc[k].tangent, 3 * sizeof(double));
  }
  return n;
}

static void collision(const DmModel *m, DmoData *d) { /* [EXT] mj_collision */
  d->ncon = 0;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
    const double *x1 = d->geom_xpos[g1], *x2 = d->geom_xpos[g2];
    const double *M1 = d->geom_xmat[g1], *M2 = d->geom_xmat[g2];
    const double *z1 = m->geom_size[g1], *z2 = m->geom_size[g2];
    /* bounding-sphere filter (result-neutral) */
    {
      double df[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
      if (t1 != DM_GEOM_PLANE) {
        if (norm3(df) > m->geom_rbound[g1] + m->geom_rbound[g2] + margin) continue;
      } else if (m->geom_rbound[g2] > 0) {
        double nrm[3] = {M1[2], M1[5], M1[8]};
        if (dot3(df, nrm) > m->geom_rbound[g2] + margin) continue;
      }
    }
    RawCon rc[8];
    int n = analytic_pair(rc, margin, t1, x1, M1, z1, t2, x2, M2, z2);
    if (n < 0) { /* no analytic routine: plane-mesh, or libccd MPR through mjc_Convex [EXT] */
      n = 0;
#ifdef DM_ROBOT_G1
      if (t1 == DM_GEOM_PLANE && t2 == DM_GEOM_MESH) {
        int me = m->geom_mesh[g2];
        n = c_plane_mesh(rc, margin, x1, M1, x2, M2, m->mesh_vert[m->mesh_vertadr[me]], m->mesh_vertnum[me]);
      } else if (t1 != DM_GEOM_PLANE) {
        CvxGeom a, b;
        cvx_from_model(m, d, g1, &a);
        cvx_from_model(m, d, g2, &b);
        n = c_convex(rc, &a, &b);
      }
#endif
    }
    for (int k = 0; k < n; k++) {
      if (d->ncon >= d->maxcon) { d->overflow_con++; continue; }
      DmoContact *c = &d->contact[d->ncon++];
      c->dist = rc[k].dist;
      memcpy(c->pos, rc[k].pos, sizeof c->pos);
      memcpy(c->frame, rc[k].normal, 3 * sizeof(double));
      memcpy(c->frame + 3, rc[k].tangent, 3 * sizeof(double));
      make_frame(c->frame);
      c->geom1 = g1; c->geom2 = g2; c->pair = p;
      c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
      c->mu = TW.mu_scale * fmax(m->geom_friction[g1][0], m->geom_friction[g2][0]);
      c->includemargin = margin; /* gap = 0 */
    }
  }
}

/* ------------------------------------------------------------------ constraints */
static void jac_point(const DmModel *m, const DmoData *d, double *jp /*3 x nv*/, const double *pt, int body) {
  memset(jp, 0, sizeof(double) * 3 * NV);
  double off[3];
  for (int i = 0; i < 3; i++) off[i] = pt[i] - d->subtree_com[i];
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parent[body];
  if (body <= 0) return;
  for (int k = m->body_dofadr[body] + m->body_dofnum[body] - 1; k >= 0; k = m->dof_parent[k]) {
    double t[3];
    cross3(t, d->cdof[k], off);
    for (int i = 0; i < 3; i++) jp[i * NV + k] = d->cdof[k][3 + i] + t[i];
  }
}

static double impedance(const double *solimp, double pos, double margin) { /* [EXT] getimpedance */
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (dmin == dmax || width <= MINVAL) return 0.5 * (dmin + dmax);
  double x = fabs(pos - margin) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  double y;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}

static int add_row(DmoData *d, int type, int id, double pos, double margin, double diag) {
  if (d->nefc >= d->maxrow) { d->overflow_row++; return -1; }
  int r = d->nefc++;
  d->efc_type[r] = type; d->efc_id[r] = id;
  d->efc_pos[r] = pos; d->efc_margin[r] = margin; d->efc_diagApprox[r] = diag;
  memset(d->efc_J + (size_t)r * NV, 0, sizeof(double) * NV);
  return r;
}

static void make_constraint(const DmModel *m, DmoData *d) { /* [EXT] mj_makeConstraint */
  d->nefc = 0;
#ifdef DM_ROBOT_G1
  /* dof friction loss rows come first [EXT mj_instantiateFriction]: J = e_dof, pos = margin = 0 */
  for (int k = 0; k < NV; k++) {
    if (m->dof_frictionloss[k] <= 0) continue;
    int r = add_row(d, 3, k, 0.0, 0.0, m->dof_invweight0[k]);
    if (r >= 0) { d->efc_J[(size_t)r * NV + k] = 1.0; d->efc_frictionloss[r] = m->dof_frictionloss[k]; }
  }
#endif
  d->nfriction = d->nefc;
  /* joint limits, in joint order (jnt margin = 0) */
  for (int j = 0; j < DM_NJNT; j++) {
    if (!m->jnt_limited[j] || m->jnt_type[j] != DM_JNT_HINGE) continue;
    double q = d->qpos[m->jnt_qposadr[j]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[j][(side + 1) / 2] - q);
      if (dist < 0) {
        int r = add_row(d, 0, j, dist, 0.0, m->dof_invweight0[m->jnt_dofadr[j]]);
        if (r >= 0) d->efc_J[(size_t)r * NV + m->jnt_dofadr[j]] = -side;
      }
    }
  }
  d->nlimit = d->nefc - d->nfriction;
  /* contacts, in contact order; pyramidal cones */
  double j1[3 * NV], j2[3 * NV], jd[3 * NV];
  for (int ci = 0; ci < d->ncon; ci++) {
    DmoContact *c = &d->contact[ci];
    int b1 = m->geom_body[c->geom1], b2 = m->geom_body[c->geom2];
    c->efc_address = d->nefc;
    jac_point(m, d, j1, c->pos, b1);
    jac_point(m, d, j2, c->pos, b2);
    for (int i = 0; i < 3 * NV; i++) jd[i] = j2[i] - j1[i];
    double jf[3 * NV]; /* rotated into the contact frame */
    for (int r = 0; r < 3; r++)
      for (int k = 0; k < NV; k++)
        jf[r * NV + k] = c->frame[3 * r] * jd[k] + c->frame[3 * r + 1] * jd[NV + k] + c->frame[3 * r + 2] * jd[2 * NV + k];
    double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
    if (c->dim == 1) {
      int r = add_row(d, 1, ci, c->dist, c->includemargin, tran);
      if (r >= 0) memcpy(d->efc_J + (size_t)r * NV, jf, sizeof(double) * NV);
    } else {
      for (int k = 1; k < c->dim; k++) {
        double fri = c->mu; /* friction[0] == friction[1] (sliding) */
        double diag = tran + fri * fri * tran;
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
          int r = add_row(d, 2, ci, c->dist, c->includemargin, diag);
          if (r >= 0)
            for (int q = 0; q < NV; q++) d->efc_J[(size_t)r * NV + q] = jf[q] + sgn * fri * jf[k * NV + q];
        }
      }
    }
  }
  /* impedance, regulariser, reference acceleration [EXT mj_makeImpedance, mj_referenceConstraint] */
  double dr = m->solref[1], dmax = m->solimp[1];
  for (int r = 0; r < d->nefc; r++) {
    double tc = (d->efc_type[r] == 0 && TW.solref_limit > 0) ? TW.solref_limit : m->solref[0];
    if (TW.refsafe) tc = fmax(tc, 2 * m->timestep); /* refsafe */
    double K = 1.0 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
    double B = 2.0 / fmax(MINVAL, dmax * tc);
    double imp = impedance(m->solimp, d->efc_pos[r], d->efc_margin[r]);
    d->efc_R[r] = fmax(MINVAL, (1 - imp) * TW.diag_scale * d->efc_diagApprox[r] / imp);
    double vel = 0;
    for (int k = 0; k < NV; k++) vel += d->efc_J[(size_t)r * NV + k] * d->qvel[k];
    d->efc_vel[r] = vel;
    if (d->efc_type[r] == 3) K = 0; /* friction rows carry no position term [EXT mj_makeImpedance] */
    d->efc_aref[r] = -B * vel - K * imp * (d->efc_pos[r] - d->efc_margin[r]);
  }
  /* pyramidal contacts: all edges share Rpy = 2 mu^2 R(first edge), impratio = 1 */
  for (int ci = 0; ci < d->ncon; ci++) {
    DmoContact *c = &d->contact[ci];
    if (c->dim <= 1) continue;
    int a = c->efc_address, n = 2 * (c->dim - 1);
    if (a + n > d->nefc) continue; /* dropped by the row cap */
    double Rpy = TW.redge * c->mu * c->mu * d->efc_R[a];
    for (int k = 0; k < n; k++) d->efc_R[a + k] = Rpy;
  }
  for (int r = 0; r < d->nefc; r++) d->efc_D[r] = 1.0 / d->efc_R[r];
}

static void project_constraint(const DmModel *m, DmoData *d) { /* AR = J M^-1 J^T + R */
  int n = d->nefc;
  if (!n) return;
  double *B = (double *)malloc(sizeof(double) * (size_t)n * NV);
  memcpy(B, d->efc_J, sizeof(double) * (size_t)n * NV);
  for (int r = 0; r < n; r++) solve_m2(m, d, B + (size_t)r * NV);
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = 0;
      for (int k = 0; k < NV; k++) s += B[(size_t)i * NV + k] * B[(size_t)j * NV + k];
      d->efc_AR[(size_t)i * n + j] = d->efc_AR[(size_t)j * n + i] = s;
    }
  for (int i = 0; i < n; i++) d->efc_AR[(size_t)i * n + i] += d->efc_R[i];
  free(B);
}

/* ------------------------------------------------------------------ velocity / acceleration stages */
static void com_vel(const DmModel *m, DmoData *d) { /* [EXT] mj_comVel */
  memset(d->cvel[0], 0, sizeof d->cvel[0]);
  for (int b = 1; b < NB; b++) {
    double cv[6];
    memcpy(cv, d->cvel[m->body_parent[b]], sizeof cv);
    int da = m->body_dofadr[b];
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      if (m->jnt_type[j] == DM_JNT_FREE) {
        for (int k = 0; k < 3; k++) memset(d->cdof_dot[da + k], 0, sizeof d->cdof_dot[0]);
        for (int k = 0; k < 3; k++)
          for (int i = 0; i < 6; i++) cv[i] += d->cdof[da + k][i] * d->qvel[da + k];
        for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot[da + k], cv, d->cdof[da + k]);
        for (int k = 3; k < 6; k++)
          for (int i = 0; i < 6; i++) cv[i] += d->cdof[da + k][i] * d->qvel[da + k];
        da += 6;
      } else {
        cross_motion(d->cdof_dot[da], cv, d->cdof[da]);
        for (int i = 0; i < 6; i++) cv[i] += d->cdof[da][i] * d->qvel[da];
        da += 1;
      }
    }
    memcpy(d->cvel[b], cv, sizeof cv);
  }
}

static void rne_bias(const DmModel *m, DmoData *d) { /* [EXT] mj_rne(flg_acc = 0) */
  double cacc[NB][6], cfrc[NB][6];
  memset(cacc, 0, sizeof cacc);
  memset(cfrc, 0, sizeof cfrc);
  for (int i = 0; i < 3; i++) cacc[0][3 + i] = -m->gravity[i];
  for (int b = 1; b < NB; b++) {
    memcpy(cacc[b], cacc[m->body_parent[b]], sizeof cacc[0]);
    for (int k = m->body_dofadr[b]; k < m->body_dofadr[b] + m->body_dofnum[b]; k++)
      for (int i = 0; i < 6; i++) cacc[b][i] += d->cdof_dot[k][i] * d->qvel[k];
    double t[6], t1[6];
    mul_inert_vec(cfrc[b], d->cinert[b], cacc[b]);
    mul_inert_vec(t, d->cinert[b], d->cvel[b]);
    cross_force(t1, d->cvel[b], t);
    for (int i = 0; i < 6; i++) cfrc[b][i] += t1[i];
  }
  for (int b = NB - 1; b > 0; b--) {
    int p = m->body_parent[b];
    if (p > 0)
      for (int i = 0; i < 6; i++) cfrc[p][i] += cfrc[b][i];
  }
  for (int k = 0; k < NV; k++) d->qfrc_bias[k] = dot6(d->cdof[k], cfrc[m->dof_body[k]]);
}

static void fwd_smooth(const DmModel *m, DmoData *d) {
  com_vel(m, d);
  for (int k = 0; k < NV; k++) d->qfrc_passive[k] = -m->dof_damping[k] * d->qvel[k];
  rne_bias(m, d);
  memset(d->qfrc_actuator, 0, sizeof d->qfrc_actuator);
  for (int a = 0; a < NU; a++) { /* [EXT] mj_fwdActuation: clamp ctrl, motor gain 1, joint transmission */
    double c = clampd(d->ctrl[a], m->act_ctrlrange[a][0], m->act_ctrlrange[a][1]);
    d->qfrc_actuator[m->act_dof[a]] += m->act_gear[a] * c;
  }
  for (int k = 0; k < NV; k++) {
    d->qfrc_smooth[k] = d->qfrc_passive[k] - d->qfrc_bias[k] + d->qfrc_actuator[k];
    d->qacc_smooth[k] = d->qfrc_smooth[k];
  }
  solve_m(m, d, d->qacc_smooth);
}

static void fwd_constraint(const DmModel *m, DmoData *d) { /* [EXT] mj_fwdConstraint + mj_solPGS */
  int n = d->nefc;
  d->solver_iter = 0;
  if (!n) {
    memcpy(d->qacc, d->qacc_smooth, sizeof d->qacc);
    memcpy(d->qacc_warmstart, d->qacc_smooth, sizeof d->qacc);
    memset(d->qfrc_constraint, 0, sizeof d->qfrc_constraint);
    return;
  }
  const double *J = d->efc_J, *AR = d->efc_AR;
  double *f = d->efc_force, *b = d->efc_b;
  for (int r = 0; r < n; r++) {
    double s = 0;
    for (int k = 0; k < NV; k++) s += J[(size_t)r * NV + k] * d->qacc_smooth[k];
    b[r] = s - d->efc_aref[r];
  }
  /* warm start: forces implied by qacc_warmstart, kept only if their dual cost is negative */
  double cost = 0;
  for (int r = 0; r < n; r++) {
    double jar = -d->efc_aref[r];
    for (int k = 0; k < NV; k++) jar += J[(size_t)r * NV + k] * d->qacc_warmstart[k];
    if (d->efc_type[r] == 3) { /* friction loss: quadratic inside +-R*floss, saturated outside [EXT mj_constraintUpdate] */
      double fl = d->efc_frictionloss[r], rf = d->efc_R[r] * fl;
      f[r] = jar <= -rf ? fl : (jar >= rf ? -fl : -d->efc_D[r] * jar);
    } else {
      f[r] = jar < 0 ? -d->efc_D[r] * jar : 0.0;
    }
  }
  for (int r = 0; r < n; r++) {
    double s = 0;
    for (int c = 0; c < n; c++) s += AR[(size_t)r * n + c] * f[c];
    cost += f[r] * (0.5 * s + b[r]);
  }
  if ((cost > 0 && TW.warmstart != 2) || TW.warmstart == 1) memset(f, 0, sizeof(double) * n);
  /* projected Gauss-Seidel on the dual, all rows scalar and unilateral */
  double scale = 1.0 / (m->meaninertia * (NV > 1 ? NV : 1));
  int iter = 0;
  while (iter < m->iterations) {
    double improvement = 0;
    for (int i = 0; i < n; i++) {
      double res = b[i];
      for (int c = 0; c < n; c++) res += AR[(size_t)i * n + c] * f[c];
      double old = f[i], aii = AR[(size_t)i * n + i];
      f[i] -= res / aii;
      if (d->efc_type[i] == 3) { /* box constraint of a friction-loss row */
        double fl = d->efc_frictionloss[i];
        if (f[i] < -fl) f[i] = -fl; else if (f[i] > fl) f[i] = fl;
      } else if (f[i] < 0) f[i] = 0;
      double dl = f[i] - old;
      improvement -= 0.5 * dl * dl * aii + dl * res;
    }
    iter++;
    if (TW.pgs_early_exit && improvement * scale < m->tolerance) break;
  }
  d->solver_iter = iter;
  for (int k = 0; k < NV; k++) {
    double s = 0;
    for (int r = 0; r < n; r++) s += J[(size_t)r * NV + k] * f[r];
    d->qfrc_constraint[k] = s;
    d->qacc[k] = s;
  }
  solve_m(m, d, d->qacc);
  for (int k = 0; k < NV; k++) d->qacc[k] += d->qacc_smooth[k];
  memcpy(d->qacc_warmstart, d->qacc, sizeof d->qacc);
}

static int bad(const double *x, int n) {
  for (int i = 0; i < n; i++)
    if (isnan(x[i]) || x[i] > MAXVAL || x[i] < -MAXVAL) return 1;
  return 0;
}

static void forward_nocheck(const DmModel *m, DmoData *d) {
  kinematics(m, d);
  com_pos(m, d);
  crb(m, d);
  factor_m(m, d);
  collision(m, d);
  make_constraint(m, d);
  project_constraint(m, d);
  fwd_smooth(m, d);
  fwd_constraint(m, d);
}

int dmo_forward(const DmModel *m, DmoData *d) {
  if (bad(d->qpos, NQ) || bad(d->qvel, NV)) return 1;
  forward_nocheck(m, d);
  return bad(d->qacc, NV);
}

static void integrate_pos(const DmModel *m, double *qpos, const double *qvel, double h) { /* [EXT] mj_integratePos */
  for (int j = 0; j < DM_NJNT; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == DM_JNT_FREE) {
      for (int i = 0; i < 3; i++) qpos[qa + i] += h * qvel[da + i];
      double w[3] = {qvel[da + 3], qvel[da + 4], qvel[da + 5]};
      double ang = h * normalize3(w), qr[4], qn[4];
      axis_angle_quat(qr, w, ang);
      normalize4(qpos + qa + 3);
      mul_quat(qn, qpos + qa + 3, qr);
      normalize4(qn);
      memcpy(qpos + qa + 3, qn, sizeof qn);
    } else {
      qpos[qa] += h * qvel[da];
    }
  }
}

static int32_t contact_hash(const DmoData *d) { /* test diagnostics: the contact index list of an evaluation, in order */
  uint32_t h = 0;
  for (int c = 0; c < d->ncon; c++) h = (h * 131u + (uint32_t)d->contact[c].geom1 * 97u + (uint32_t)d->contact[c].geom2 + 1u) & 0xFFFFFFu;
  return (int32_t)h;
}

int dmo_step(const DmModel *m, DmoData *d) { /* [EXT] mj_step with mj_RungeKutta(4) */
  if (bad(d->qpos, NQ) || bad(d->qvel, NV)) return 1;
  double ws0[NV];
  memcpy(ws0, d->qacc_warmstart, sizeof ws0); /* warm start as the step found it (TW.stale_ws only) */
  forward_nocheck(m, d);
  if (bad(d->qacc, NV)) return 1;
  d->stage_ncon[0] = d->ncon; d->stage_nefc[0] = d->nefc; d->stage_chash[0] = contact_hash(d);
  double h = m->timestep;
  if (m->integrator == DM_INT_RK4) {
    static const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1}};
    static const double Bw[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
    static const double Ct[3] = {0.5, 0.5, 1.0};
    double X0q[NQ], X0v[NV], Xv[4][NV], F[4][NV], t0 = d->time;
    memcpy(X0q, d->qpos, sizeof X0q);
    memcpy(X0v, d->qvel, sizeof X0v);
    memcpy(Xv[0], d->qvel, sizeof X0v);
    memcpy(F[0], d->qacc, sizeof X0v);
    for (int i = 1; i < 4; i++) {
      double dq[NV], dv[NV];
      for (int k = 0; k < NV; k++) {
        dq[k] = dv[k] = 0;
        for (int j = 0; j < i; j++) { dq[k] += A[i - 1][j] * Xv[j][k]; dv[k] += A[i - 1][j] * F[j][k]; }
      }
      memcpy(d->qpos, X0q, sizeof X0q);
      integrate_pos(m, d->qpos, dq, h);
      for (int k = 0; k < NV; k++) d->qvel[k] = X0v[k] + h * dv[k];
      memcpy(Xv[i], d->qvel, sizeof X0v);
      d->time = t0 + Ct[i - 1] * h;
      if (TW.stale_ws) memcpy(d->qacc_warmstart, ws0, sizeof ws0);
      forward_nocheck(m, d);
      d->stage_ncon[i] = d->ncon; d->stage_nefc[i] = d->nefc; d->stage_chash[i] = contact_hash(d);
      memcpy(F[i], d->qacc, sizeof X0v);
    }
    double dq[NV], dv[NV];
    for (int k = 0; k < NV; k++) {
      dq[k] = dv[k] = 0;
      for (int j = 0; j < 4; j++) { dq[k] += Bw[j] * Xv[j][k]; dv[k] += Bw[j] * F[j][k]; }
    }
    memcpy(d->qpos, X0q, sizeof X0q);
    for (int k = 0; k < NV; k++) d->qvel[k] = X0v[k] + h * dv[k];
    integrate_pos(m, d->qpos, dq, h);
    d->time = t0 + h;
  } else { /* [EXT] mj_Euler: semi-implicit Euler; joint damping integrated implicitly when any dof has damping:
            * (M + h B) qacc' = qfrc_smooth + qfrc_constraint, then qvel += h qacc', qpos (+)= h qvel(new) */
    double qa[NV];
    int damped = 0;
    for (int k = 0; k < NV; k++) damped |= m->dof_damping[k] > 0;
    if (!damped) {
      memcpy(qa, d->qacc, sizeof qa);
    } else {
      double qM0[DM_NM];
      memcpy(qM0, d->qM, sizeof qM0);
      for (int k = 0; k < NV; k++) d->qM[m->dof_Madr[k]] += h * m->dof_damping[k];
      factor_m(m, d);
      for (int k = 0; k < NV; k++) qa[k] = d->qfrc_smooth[k] + d->qfrc_constraint[k];
      solve_m(m, d, qa);
      memcpy(d->qM, qM0, sizeof qM0); /* MuJoCo factorises a copy: qM / qLD stay those of M */
      factor_m(m, d);
    }
    for (int k = 0; k < NV; k++) d->qvel[k] += h * qa[k];
    integrate_pos(m, d->qpos, d->qvel, h);
    d->time += h;
  }
  return 0;
}

int dmo_set_state(const DmModel *m, DmoData *d, const double *qpos, const double *qvel) {
  memcpy(d->qpos, qpos, sizeof d->qpos);
  memcpy(d->qvel, qvel, sizeof d->qvel);
  return dmo_forward(m, d);
}

/* ------------------------------------------------------------------ DPEnv semantics */
void dmo_quat_to_rpy(const double *q, double *rpy) {
  /* py3dtf.Quaternion(x,y,z,w).to_rpy() [EXT]: standard ZYX, no normalisation */
  double w = q[0], x = q[1], y = q[2], z = q[3];
  rpy[0] = atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
  double s = 2 * (w * y - z * x);
  rpy[1] = asin(s > 1 ? 1 : (s < -1 ? -1 : s));
  rpy[2] = atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z));
}

void dmo_get_obs(const DmModel *m, const DmoData *d, int idx_curr, int L, double *obs) {
  const double S = 0.1; /* VEL_OBS_SCALE, deepmimic_env.py:261 */
  const int NP = NQ - 7, NW = NV - 6, T0 = NP + NW;               /* humanoid3d 28 + 28, G1 37 + 37 */
  for (int i = 0; i < NP; i++) obs[i] = d->qpos[7 + i];          /* :34 */
  for (int i = 0; i < NW; i++) obs[NP + i] = d->qvel[6 + i] * S; /* :35-37 */
  int b = m->torso_body;                                         /* :47-76 */
  double rpy[3];
  dmo_quat_to_rpy(d->xquat[b], rpy);
  const double *cv = d->cvel[b];
  double cy = cos(-rpy[2]), sy = sin(-rpy[2]);
  double vx = cy * cv[3] - sy * cv[4], vy = sy * cv[3] + cy * cv[4], vz = cv[5];
  obs[T0] = rpy[0] * S; obs[T0 + 1] = rpy[1] * S;
  obs[T0 + 2] = vx * S; obs[T0 + 3] = vy * S; obs[T0 + 4] = vz * S;
  obs[T0 + 5] = cv[0] * S; obs[T0 + 6] = cv[1] * S; obs[T0 + 7] = cv[2] * S;
  double rf = 0, lf = 0;                                         /* :78-105; F8: the reference scans every slot of mjdata.contact */
  const int nslot = d->stale_contact_slots ? d->maxcon : d->ncon; /* (slots >= ncon keep what earlier evaluations wrote there) */
  for (int c = 0; c < nslot; c++) {
    int g1 = d->contact[c].geom1, g2 = d->contact[c].geom2;
    int floor = (g1 == m->floor_geom || g2 == m->floor_geom);
    if ((g1 == m->rfoot_geom || g2 == m->rfoot_geom) && floor) rf = 1;
    if ((g1 == m->lfoot_geom || g2 == m->lfoot_geom) && floor) lf = 1;
  }
  obs[T0 + 8] = rf; obs[T0 + 9] = lf; /* G1: both geoms are visual spheres (contype 0): always 0 */
  double ph = (double)idx_curr / (double)L;                      /* :139-143 */
  obs[T0 + 10] = ph < 0 ? 0 : (ph > 1 ? 1 : ph);
}

double dmo_reward(const DmModel *m, const DmoData *d, const DmoClip *clip, int idx, double *terms) {
  const double *tq = clip->qpos + (size_t)idx * NQ, *tv = clip->qvel + (size_t)idx * NV;
  double err = 0;
#ifdef DM_ROBOT_G1
  for (int k = 0; k < DM_NREWJ; k++) { int i = m->rew_qposadr[k]; err += fabs(d->qpos[i] - tq[i]); } /* :204-211 */
#else
  for (int i = 7; i < NQ; i++) err += fabs(d->qpos[i] - tq[i]);   /* :213-214 */
#endif
  double rc[3], rt[3];
  dmo_quat_to_rpy(d->qpos + 3, rc);                               /* :216-221 */
  dmo_quat_to_rpy(tq + 3, rt);
  err += fabs(rc[1] - rt[1]);
  double r_cfg = exp(-err);
  double ev = 0;
#ifdef DM_ROBOT_G1
  for (int k = 0; k < DM_NREWJ; k++) { int i = m->rew_dofadr[k]; ev += fabs(tv[i] - d->qvel[i]); }
#else
  for (int i = 6; i < NV; i++) ev += fabs(tv[i] - d->qvel[i]);    /* :225-226 */
#endif
  double r_vel = exp(-0.1 * ev);
  double ee = 0;                                                  /* :228-233 */
  for (int e = 0; e < DM_NEE; e++) {
    int g = m->ee_geom[e];
    const double *t = clip->geom_xpos + ((size_t)idx * NG + g) * 3;
    for (int i = 0; i < 3; i++) { double df = d->geom_xpos[g][i] - t[i]; ee += df * df; }
  }
  double r_ee = exp(-40 * ee);
  double mt = 0, ct[3] = {0, 0, 0}, cc[3] = {0, 0, 0};            /* :235-240 (frame origins x mass) */
  for (int b = 0; b < NB; b++) {
    mt += m->body_mass[b];
    const double *t = clip->body_xpos + ((size_t)idx * NB + b) * 3;
    for (int i = 0; i < 3; i++) { ct[i] += t[i] * m->body_mass[b]; cc[i] += d->xpos[b][i] * m->body_mass[b]; }
  }
  double ce = 0;
  for (int i = 0; i < 3; i++) { double df = (ct[i] - cc[i]) / mt; ce += df * df; }
  double r_com = exp(-10 * ce);
  int viol = 0;                                                   /* :242-247 */
#ifdef DM_ROBOT_G1
  for (int k = 0; k < DM_NREWJ; k++) {                            /* :244-246 */
    int j = m->rew_jnt[k];
    double q = d->qpos[m->jnt_qposadr[j]];
    viol += (q <= m->jnt_range[j][0] * 0.99) + (q >= m->jnt_range[j][1] * 0.99);
  }
  double qlim = (double)viol / (double)DM_NREWJ;
#else
  for (int j = 1; j < DM_NJNT; j++) {
    double q = d->qpos[m->jnt_qposadr[j]];
    viol += (q <= m->jnt_range[j][0] * 0.99) + (q >= m->jnt_range[j][1] * 0.99);
  }
  double qlim = (double)viol / 28.0;
#endif
  terms[0] = r_cfg; terms[1] = r_vel; terms[2] = r_ee; terms[3] = r_com; terms[4] = qlim;
  return 0.75 * r_cfg + 0.1 * r_vel + 0.15 * r_ee + 0.0 * r_com + (-0.1) * qlim; /* :400-404,249 */
}

int dmo_env_step(const DmModel *m, DmoData *d, DmoEnv *e, const DmoClip *clip, const double *action,
                 const double *fq, const double *fv, double *obs, double *reward, double *terms, int32_t *reason) {
  int err = 0;
  *reason = DMO_REASON_NONE;
  if (fq && fv) {
    err = dmo_set_state(m, d, fq, fv);                            /* :355-357 */
  } else {
#ifdef DM_ROBOT_G1
    for (int a = 0; a < NU; a++) d->ctrl[a] = a < m->n_policy_action ? action[a] * m->action_scale : 0.0; /* :348-351 */
#else
    for (int a = 0; a < NU; a++) d->ctrl[a] = action[a] * 1.0;    /* :347, do_simulation sets ctrl */
#endif
    err = dmo_step(m, d);                                         /* :362 */
  }
  if (err) {                                                      /* :366-378 */
    dmo_data_reset(m, d); /* MuJoCo resets mjData when it raises the warning [EXT] */
    memset(obs, 0, sizeof(double) * DM_NOBS);
    memset(terms, 0, sizeof(double) * 5);
    *reward = 0;
    *reason = DMO_REASON_SIM_ERROR;
    return 1;
  }
  dmo_get_obs(m, d, e->idx_curr, clip->L, obs);                   /* :389 */
  *reward = dmo_reward(m, d, clip, e->idx_curr, terms);           /* :405-408 */
  int done = 0;
  double mt = 0, zc = 0;                                          /* :420-424 */
  for (int b = 0; b < NB; b++) { mt += m->body_mass[b]; zc += m->body_mass[b] * d->xipos[b][2]; }
  zc /= mt;
#ifdef DM_ROBOT_G1
  const double low_z = m->low_z;                                  /* src/config.py:22 */
#else
  const double low_z = 0.7;                                       /* src/config.py:13 */
#endif
  if (!(clip->flags & 1)) {                                       /* :420 not a floor motion */
    done = (zc < low_z) || (zc > 2.0);
    *reason = (zc < low_z) ? DMO_REASON_LOW_Z : DMO_REASON_HIGH_Z; /* written every step (:424) */
  }
#ifdef DM_ROBOT_G1
  if (clip->flags & 4) {                                          /* :426-433 G1 "run": roll / pitch deviation > 60 deg */
    double rc[3], rt[3];
    dmo_quat_to_rpy(d->qpos + 3, rc);
    dmo_quat_to_rpy(clip->qpos + (size_t)e->idx_curr * NQ + 3, rt);
    const double max_angle = 60.0 * 3.14159265358979323846 / 180.0;
    if (fabs(rc[0] - rt[0]) > max_angle || fabs(rc[1] - rt[1]) > max_angle) { done = 1; *reason = DMO_REASON_RUN_ANGLE; }
  }
#endif
  if (e->episode_length >= 1000) { done = 1; *reason = DMO_REASON_MAX_EP_LEN; } /* :435-438 */
  if ((clip->flags & 2) && e->idx_curr + 1 == clip->L) { done = 1; *reason = DMO_REASON_ACYCLIC_END; } /* :440-442 */
  e->idx_curr = (e->idx_curr + 1) % clip->L;                      /* :452 */
  e->episode_reward += *reward;
  e->episode_length += 1;
  for (int i = 0; i < DM_NOBS; i++)                               /* :465-476 */
    if (obs[i] > 100.0 || obs[i] < -100.0) {
      memset(obs, 0, sizeof(double) * DM_NOBS);
      memset(terms, 0, sizeof(double) * 5);
      *reward = 0;
      *reason = DMO_REASON_OBS_BOUNDS;
      return 1;
    }
  return done;
}

int dmo_env_reset(const DmModel *m, DmoData *d, DmoEnv *e, const DmoClip *clip, int idx_init, double *obs) {
  e->episode_reward = 0;                                          /* :497-499 */
  e->episode_length = 0;
  e->idx_curr = idx_init;                                         /* :312-316 */
  int err = dmo_set_state(m, d, clip->qpos + (size_t)idx_init * NQ, clip->qvel + (size_t)idx_init * NV); /* :506-508 */
  dmo_get_obs(m, d, e->idx_curr, clip->L, obs);
  return err;
}

/* ------------------------------------------------------------------ DPCombinedEnv semantics (src/combined_env.py)
 * humanoid3d build: the class's logic on the 34-DoF model (no action scale, no extra-contact geoms, low_z 0.7).
 * G1 build: the class as the reference runs it (:164-178): walk / run / getup_facedown_towalk, ACT_SCALE 20 with 14 padded
 * hand torques (:251-254), ADD_EXTRA_CONTACT_OBS (:27) -> obs 98, imitation terms on the 23-joint subset, low_z 0.4. */
#define COMB_AMNESTY_STEPS 150   /* DPCombinedEnvConfig.AMNESTY_STEPS :34 */
#define COMB_MAX_EP_LENGTH 2000  /* :22 */
#define COMB_TO_GETUP_LEN 180    /* MTToGetup.length :97 */

static int comb_len(const DmoCombEnv *e, const DmoClip *clips) { /* current_motion_mocap.get_length() */
  return e->motion == DMO_MOTION_TO_GETUP ? COMB_TO_GETUP_LEN : clips[e->motion].L;
}
static const DmoClip *comb_clip(const DmoCombEnv *e, const DmoClip *clips, int *frame) {
  /* MotionTransition getters always return frame 1 of the target clip (:72-79) */
  if (e->motion == DMO_MOTION_TO_GETUP) { *frame = 1; return &clips[DMO_MOTION_GETUP]; }
  *frame = e->n_steps % clips[e->motion].L;
  return &clips[e->motion];
}

void dmo_combined_obs(const DmModel *m, const DmoData *d, const DmoCombEnv *e, const DmoClip *clips, double *obs) {
  double base[DM_NOBS];
  int L = comb_len(e, clips);
  const int T8 = (NQ - 7) + (NV - 6) + 8;                         /* qpos[7:], qvel[6:], torso: 64 / 82 */
  dmo_get_obs(m, d, e->n_steps % L, L, base);                    /* shared get_obs (deepmimic_env.py:33-45) */
  for (int i = 0; i < T8; i++) obs[i] = base[i];                  /* ADD_FOOT_CONTACT_OBS False (:25) */
  int o = T8;
#ifdef DM_ROBOT_G1
  for (int k = 0; k < 8; k++) obs[o + k] = 0;                     /* get_extra_contact_obs (deepmimic_env.py:107-121) */
  for (int c = 0; c < d->ncon; c++) {
    int g1 = d->contact[c].geom1, g2 = d->contact[c].geom2;
    if (g1 != m->floor_geom && g2 != m->floor_geom) continue;
    for (int k = 0; k < 8; k++) if (g1 == m->extra_geom[k] || g2 == m->extra_geom[k]) obs[o + k] = 1;
  }
  o += 8;
#endif
  obs[o] = base[T8 + 2];                                          /* phase */
  /* get_player_action_obs (deepmimic_env.py:145-173) with PAWalk: heading (1,0,0), onehot index 0 */
  double rpy[3];
  dmo_quat_to_rpy(d->xquat[m->torso_body], rpy);
  const double hwx = 1.0, hwy = 0.0;
  obs[o + 1] = hwx * cos(-rpy[2]) - hwy * sin(-rpy[2]);
  obs[o + 2] = hwx * sin(-rpy[2]) + hwy * cos(-rpy[2]);
  obs[o + 3] = 1; obs[o + 4] = 0; obs[o + 5] = 0;
  obs[o + 6] = (e->motion == DMO_MOTION_TO_GETUP) ? 1 : 0;        /* pa_getup_state (:499-504) */
  obs[o + 7] = (e->motion == DMO_MOTION_GETUP) ? 1 : 0;
}

static void comb_change(DmoCombEnv *e, int motion) { e->motion = motion; e->n_steps = 0; } /* :529-533 */

int dmo_combined_step(const DmModel *m, DmoData *d, DmoCombEnv *e, const DmoClip *clips, const double *action,
                      const double *fq, const double *fv, double *obs, double *reward, double *terms, int32_t *reason) {
  int err = 0;
  *reason = DMO_REASON_NONE;
  if (fq && fv) {
    err = dmo_set_state(m, d, fq, fv);                            /* :260-262 */
  } else {
#ifdef DM_ROBOT_G1
    for (int a = 0; a < NU; a++) d->ctrl[a] = a < m->n_policy_action ? action[a] * m->action_scale : 0.0; /* :251-254 */
#else
    for (int a = 0; a < NU; a++) d->ctrl[a] = action[a] * 1.0;    /* :251 (ACT_SCALE applies to unitree_g1 only) */
#endif
    err = dmo_step(m, d);                                         /* :267 */
  }
  if (err) {                                                      /* :271-284 */
    dmo_data_reset(m, d);
    memset(obs, 0, sizeof(double) * DMO_NOBS_COMBINED);
    memset(terms, 0, sizeof(double) * 8);
    *reward = 0;
    *reason = DMO_REASON_SIM_ERROR;
    return 1;
  }
  dmo_combined_obs(m, d, e, clips, obs);                          /* :313 (before any motion change) */
  /* ---- reward (:325-358) */
  int frame;
  const DmoClip *clip = comb_clip(e, clips, &frame);
  double imitation = dmo_reward(m, d, clip, frame, terms);
  const double *tq = clip->qpos + (size_t)frame * NQ, *tv = clip->qvel + (size_t)frame * NV;
  double rc[3], rt[3];
  dmo_quat_to_rpy(d->qpos + 3, rc);
  dmo_quat_to_rpy(tq + 3, rt);
  double dsum = 0, dmax = 0;
  int nbad = 0;
  const double PI = 3.14159265358979323846;
  const double ALIM = 15.0 * (PI / 180.0), MAX_ANGLE = 60.0 * (PI / 180.0); /* np.deg2rad(15), np.deg2rad(60) */
#ifdef DM_ROBOT_G1
  for (int k = 0; k < DM_NREWJ; k++) {                            /* config_angle_diffs of the 23-joint subset */
    const int i = m->rew_qposadr[k];
#else
  for (int i = 7; i < NQ; i++) {
#endif
    double a = fabs(d->qpos[i] - tq[i]);
    dsum += a;
    if (a > dmax) dmax = a;
    nbad += a > ALIM;
  }
  const double droll = fabs(rc[0] - rt[0]), dpitch = fabs(rc[1] - rt[1]);
  nbad += (dpitch > ALIM) + (droll > ALIM);                       /* debug_n_bad_angles :411 */
  double task = 0;
  if (e->motion == DMO_MOTION_WALK || e->motion == DMO_MOTION_RUN) {   /* :340-346 */
    double ex = tv[0] - d->qvel[0], ey = tv[1] - d->qvel[1];
    task = exp(-sqrt(ex * ex + ey * ey) * 10.0);
  }
  if (e->motion == DMO_MOTION_TO_GETUP) {                         /* :347-351 */
    imitation = 0;
    task = exp(-(dsum + dpitch + droll) / 5.0) / 3.0;
  }
  *reward = imitation * 0.7 + task * 0.3;                         /* :352-354 */
  terms[5] = imitation; terms[6] = task; terms[7] = nbad;
  /* ---- termination / motion state machine (:393-445) */
  int done = 0;
  const int out_of_time = e->n_steps >= comb_len(e, clips) - 1;   /* :394 */
  if (out_of_time) {
    /* :396 `current_player_action == PAWalk()` compares two distinct objects -> always False -> run */
    if (e->motion == DMO_MOTION_GETUP) comb_change(e, DMO_MOTION_RUN);
    if (e->motion == DMO_MOTION_TO_GETUP) comb_change(e, DMO_MOTION_GETUP);
  }
  /* is_player_action_change is hard-wired False (:300) */
  const int successful = (dpitch < ALIM) && (droll < ALIM) && (dmax < ALIM); /* :406-410 */
  if (successful && e->motion == DMO_MOTION_TO_GETUP) comb_change(e, DMO_MOTION_GETUP);
  if (e->motion == DMO_MOTION_WALK || e->motion == DMO_MOTION_RUN) { /* :416-440 (motion as changed above) */
    double mt = 0, zc = 0;
    for (int b = 0; b < NB; b++) { mt += m->body_mass[b]; zc += m->body_mass[b] * d->xipos[b][2]; }
    zc /= mt;
#ifdef DM_ROBOT_G1
    int fallen = (zc < m->low_z) || (zc > 2.0);                   /* robot_config.low_z (:419) */
#else
    int fallen = (zc < 0.7) || (zc > 2.0);
#endif
    if (droll > MAX_ANGLE) fallen = 1;
    if (dpitch > MAX_ANGLE) fallen = 1;
    if (fallen) {
      if (!(e->n_steps > COMB_AMNESTY_STEPS)) { done = 1; *reason = DMO_REASON_FALLEN_NO_AMNESTY; }
      comb_change(e, DMO_MOTION_TO_GETUP);
    }
  }
  if (e->episode_length >= COMB_MAX_EP_LENGTH) { done = 1; *reason = DMO_REASON_MAX_EP_LEN; } /* :442-445 */
  /* ---- post-step (:454-460) */
  e->n_steps += 1;
  e->episode_reward += *reward;
  e->episode_length += 1;
  for (int i = 0; i < DMO_NOBS_COMBINED; i++)                     /* :472-484 */
    if (obs[i] > 100.0 || obs[i] < -100.0) {
      memset(obs, 0, sizeof(double) * DMO_NOBS_COMBINED);
      memset(terms, 0, sizeof(double) * 8);
      *reward = 0;
      *reason = DMO_REASON_OBS_BOUNDS;
      return 1;
    }
  return done;
}

int dmo_combined_reset(const DmModel *m, DmoData *d, DmoCombEnv *e, const DmoClip *clips, int motion, int n_steps,
                       double *obs) {
  e->motion = motion;                                             /* :219-227 */
  e->n_steps = n_steps;
  e->episode_reward = 0;                                          /* :233-235 */
  e->episode_length = 0;
  int frame;
  const DmoClip *clip = comb_clip(e, clips, &frame);              /* get_current_motion_state :199-203 */
  int err = dmo_set_state(m, d, clip->qpos + (size_t)frame * NQ, clip->qvel + (size_t)frame * NV);
  dmo_combined_obs(m, d, e, clips, obs);
  return err;
}

/* ------------------------------------------------------------------ CPU baseline driver */

static uint32_t hash32(uint64_t seed, uint32_t env, uint32_t step, uint32_t j) {
  /* counter-based generator shared with the HIP bench path (csrc/dm_kernels.hip: dm_hash32) */
  uint64_t x = seed ^ ((uint64_t)env * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)step * 0xBF58476D1CE4E5B9ull) ^
               ((uint64_t)j * 0x94D049BB133111EBull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

double dmo_bench_steps(const DmModel *m, const DmoClip *clip, int nenv, int nsteps, uint64_t seed) {
  double acc = 0;
  DmoData *d = dmo_data_new(m);
  for (int e = 0; e < nenv; e++) {
    DmoEnv env;
    double obs[DM_NOBS], rew, terms[5], act[NU];
    int32_t reason;
    dmo_data_reset(m, d);
    dmo_env_reset(m, d, &env, clip, e % clip->L, obs);
    for (int s = 0; s < nsteps; s++) {
      for (int a = 0; a < NU; a++) act[a] = -2.0 + 4.0 * (hash32(seed, e, s, a) >> 8) * (1.0 / 16777216.0);
      int done = dmo_env_step(m, d, &env, clip, act, NULL, NULL, obs, &rew, terms, &reason);
      acc += rew;
      if (done) dmo_env_reset(m, d, &env, clip, (e + s) % clip->L, obs);
    }
  }
  dmo_data_free(d);
  return acc;
}

/* ------------------------------------------------------------------ reflection for the Python test wrapper */
#define FIELD(nm, ptr, cnt) if (!strcmp(name, nm)) { src = (const double *)(ptr); n = (cnt); }
int dmo_get(const DmoData *d, const char *name, double *out, int cap) {
  const double *src = NULL;
  int n = 0;
  FIELD("qpos", d->qpos, NQ) FIELD("qvel", d->qvel, NV) FIELD("ctrl", d->ctrl, NU)
  FIELD("qacc_warmstart", d->qacc_warmstart, NV) FIELD("qacc", d->qacc, NV)
  FIELD("qacc_smooth", d->qacc_smooth, NV) FIELD("xpos", d->xpos, NB * 3) FIELD("xquat", d->xquat, NB * 4)
  FIELD("xmat", d->xmat, NB * 9) FIELD("xipos", d->xipos, NB * 3) FIELD("geom_xpos", d->geom_xpos, NG * 3)
  FIELD("geom_xmat", d->geom_xmat, NG * 9) FIELD("subtree_com", d->subtree_com, 3)
  FIELD("cvel", d->cvel, NB * 6) FIELD("cdof", d->cdof, NV * 6) FIELD("cdof_dot", d->cdof_dot, NV * 6)
  FIELD("cinert", d->cinert, NB * 10) FIELD("qM", d->qM, DM_NM) FIELD("qLD", d->qLD, DM_NM)
  FIELD("qfrc_bias", d->qfrc_bias, NV) FIELD("qfrc_passive", d->qfrc_passive, NV)
  FIELD("qfrc_actuator", d->qfrc_actuator, NV) FIELD("qfrc_smooth", d->qfrc_smooth, NV)
  FIELD("qfrc_constraint", d->qfrc_constraint, NV)
  FIELD("efc_J", d->efc_J, d->nefc * NV) FIELD("efc_AR", d->efc_AR, d->nefc * d->nefc)
  FIELD("efc_R", d->efc_R, d->nefc) FIELD("efc_D", d->efc_D, d->nefc) FIELD("efc_aref", d->efc_aref, d->nefc)
  FIELD("efc_b", d->efc_b, d->nefc) FIELD("efc_force", d->efc_force, d->nefc) FIELD("efc_pos", d->efc_pos, d->nefc)
  FIELD("efc_vel", d->efc_vel, d->nefc) FIELD("time", &d->time, 1)
  FIELD("efc_frictionloss", d->efc_frictionloss, d->nefc)
  if (src) {
    if (n > cap) return -n;
    memcpy(out, src, sizeof(double) * n);
    return n;
  }
  if (!strcmp(name, "contact")) { /* per contact: dist, pos3, frame9, geom1, geom2, dim, pair = 17 */
    n = d->ncon * 17;
    if (n > cap) return -n;
    for (int c = 0; c < d->ncon; c++) {
      const DmoContact *k = &d->contact[c];
      double *o = out + 17 * c;
      o[0] = k->dist;
      memcpy(o + 1, k->pos, 3 * sizeof(double));
      memcpy(o + 4, k->frame, 9 * sizeof(double));
      o[13] = k->geom1; o[14] = k->geom2; o[15] = k->dim; o[16] = k->pair;
    }
    return n;
  }
  return -1000000;
}
int dmo_set(DmoData *d, const char *name, const double *in, int n) {
  double *dst = NULL;
  int cnt = 0;
  if (!strcmp(name, "qpos")) { dst = d->qpos; cnt = NQ; }
  if (!strcmp(name, "qvel")) { dst = d->qvel; cnt = NV; }
  if (!strcmp(name, "ctrl")) { dst = d->ctrl; cnt = NU; }
  if (!strcmp(name, "qacc_warmstart")) { dst = d->qacc_warmstart; cnt = NV; }
  if (!strcmp(name, "time")) { dst = &d->time; cnt = 1; }
  if (!dst || n != cnt) return -1;
  memcpy(dst, in, sizeof(double) * n);
  return 0;
}
int dmo_get_int(const DmoData *d, const char *name) {
  if (!strcmp(name, "ncon")) return d->ncon;
  if (!strcmp(name, "nefc")) return d->nefc;
  if (!strcmp(name, "nlimit")) return d->nlimit;
  if (!strcmp(name, "nfriction")) return d->nfriction;
  if (!strcmp(name, "solver_iter")) return d->solver_iter;
  if (!strcmp(name, "overflow_con")) return d->overflow_con;
  if (!strcmp(name, "overflow_row")) return d->overflow_row;
  if (!strcmp(name, "maxcon")) return d->maxcon;
  if (!strcmp(name, "maxrow")) return d->maxrow;
  if (!strncmp(name, "stage_ncon", 10)) return d->stage_ncon[name[10] - '0'];
  if (!strncmp(name, "stage_nefc", 10)) return d->stage_nefc[name[10] - '0'];
  if (!strncmp(name, "stage_chash", 11)) return d->stage_chash[name[11] - '0'];
  return -1;
}
int dmo_model_sizeof(void) { return (int)sizeof(DmModel); }
int dmo_set_flag(DmoData *d, const char *name, int v) {
  if (!strcmp(name, "stale_contact_slots")) { d->stale_contact_slots = v; return 0; }
  return -1;
}

int dmo_set_caps(DmoData *d, int maxcon, int maxrow) {
  if (maxcon < 1 || maxcon > DMO_MAXCON || maxrow < 1 || maxrow > DMO_MAXROW) return -1;
  d->maxcon = maxcon; d->maxrow = maxrow;
  return 0;
}
