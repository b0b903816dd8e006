/*
 * dm_convex.h — general convex narrowphase of the CPU oracle (TEST INFRASTRUCTURE, included by dm_oracle.c).
 *
 * The Unitree G1 model (deepmimic_unitree_g1.xml) collides cylinders, boxes, spheres and 32 convex meshes; MuJoCo 2.x
 * sends every pair without an analytic routine to mjc_Convex, which calls libccd's Minkowski Portal Refinement
 * (ccdMPRPenetration, D. Fiser's libccd 2.x src/mpr.c — a dependency of MuJoCo that is not in /root/reference) [EXT].
 * This file restates that published algorithm step for step (discoverPortal / refinePortal / findPenetr / findPos,
 * tolerance 1e-6, 50 iterations = MuJoCo's mpr_tolerance / mpr_iterations defaults), MuJoCo's support mappings,
 * and the two plane routines the G1 model needs (mjc_PlaneCylinder, mjc_PlaneConvex for a mesh).
 * PARITY UNPINNED: nothing in the reference tree holds a golden contact; tests check it against the analytic
 * primitive routines of dm_oracle.c on penetrating pairs and against geometric invariants.
 *
 * Ties (the one deliberate deviation from the literal routine, TW.support_tie = 1e-12; 0 restores the literal scan).
 * MPR queries a hull in the normal direction of its current portal.  Whenever two portal vertices a_i - b, a_j - b share
 * the witness b on one shape (every edge-edge and face-vertex contact reaches that state), a_i and a_j have EXACTLY the
 * same support value in that direction by construction, so libccd's "first strict maximum" is decided by the last bit of
 * two dot products — and the two choices end in different final portals, whose closest points to the origin differ in
 * depth by percents and in direction in the second digit.  Measured on this oracle alone (tests/test_g1_oracle.py): the
 * contact normal of a hand-on-hip pose of the walk clip flips between two values under 1e-15 perturbations of qpos, in
 * half of the trials.  Real MuJoCo makes that choice by the rounding of its own build, so neither value is "MuJoCo's";
 * a parity check against ANY second implementation needs a rule that does not depend on the last bit: support values
 * within 1e-12 m of the maximum are tied and the lowest vertex index wins (box corners / cylinder caps: the positive
 * side).  The HIP engine applies the same rule (csrc/dm_g1.hip: scan4_pick / wave_pick / support_local).
 */
#ifndef DM_CONVEX_H
#define DM_CONVEX_H

#define CCD_EPS 2.220446049250313e-16
#define MPR_TOLERANCE 1e-6
#define MPR_ITERATIONS 50

typedef struct CvxGeom {
  int type;
  const double *pos, *mat, *size; /* world position, row-major rotation, size */
  const double *vert;             /* mesh: hull vertices in the geom frame */
  int nvert;
  double center[3];               /* world interior point (geom centre; hull centroid for a mesh) */
} CvxGeom;

/* support vertex of a hull in the local direction dl: the first strict maximum ([EXT] mjc_MeshSupport without the hill-climb
 * graph), then — see "ties" above — the lowest index among the vertices within TW.support_tie of it */
static int mesh_support_index(const double *vert, int nvert, const double *dl) {
  int best = 0;
  double bd = -1e300;
  double sv[nvert > 0 ? nvert : 1]; /* the scan's values, kept for the tie pass (<= 5 365 vertices: 43 KB of stack) */
  for (int k = 0; k < nvert; k++) {
    double s = dot3(vert + 3 * k, dl);
    sv[k] = s;
    if (s > bd) { bd = s; best = k; }
  }
  if (TW.support_tie > 0)
    for (int k = 0; k < best; k++)
      if (sv[k] >= bd - TW.support_tie) return k;
  return best;
}

static void cvx_support(const CvxGeom *g, const double *dir, double *out) { /* [EXT] mjccd_support */
  double dl[3], p[3] = {0, 0, 0};
  rot_vec_t(dl, g->mat, dir);
  switch (g->type) {
    case DM_GEOM_SPHERE: {
      double n = norm3(dl);
      if (n > 0) for (int i = 0; i < 3; i++) p[i] = dl[i] * g->size[0] / n;
      break;
    }
    case DM_GEOM_CAPSULE: {
      double n = norm3(dl);
      if (n > 0) for (int i = 0; i < 3; i++) p[i] = dl[i] * g->size[0] / n;
      p[2] += dl[2] >= 0 ? g->size[1] : -g->size[1];
      break;
    }
    case DM_GEOM_CYLINDER: {
      double n = sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
      if (n > MINVAL) { p[0] = dl[0] * g->size[0] / n; p[1] = dl[1] * g->size[0] / n; }
      p[2] = dl[2] * g->size[1] >= -0.5 * TW.support_tie ? g->size[1] : -g->size[1];   /* the two caps differ by 2 dl_z size */
      break;
    }
    case DM_GEOM_BOX:
      for (int i = 0; i < 3; i++) p[i] = dl[i] * g->size[i] >= -0.5 * TW.support_tie ? g->size[i] : -g->size[i];
      break;
    case DM_GEOM_MESH: {
      int best = mesh_support_index(g->vert, g->nvert, dl);
      memcpy(p, g->vert + 3 * best, sizeof p);
      break;
    }
    default: break;
  }
  rot_vec(out, g->mat, p);
  for (int i = 0; i < 3; i++) out[i] += g->pos[i];
}

/* ------------------------------------------------------------------ libccd MPR, restated */
typedef struct { double v[3], v1[3], v2[3]; } MprSup; /* point of A - B and its witnesses on A and B */
typedef struct { MprSup ps[4]; int size; } MprSimplex;

static int ccd_is_zero(double x) { return fabs(x) < CCD_EPS; }
static int ccd_eq(double a, double b) {
  double ab = fabs(a - b);
  if (ab < CCD_EPS) return 1;
  a = fabs(a); b = fabs(b);
  return b > a ? ab < CCD_EPS * b : ab < CCD_EPS * a;
}
static void sub3(double *r, const double *a, const double *b) { for (int i = 0; i < 3; i++) r[i] = a[i] - b[i]; }

static void mpr_support(const CvxGeom *a, const CvxGeom *b, const double *dir, MprSup *s) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  cvx_support(a, dir, s->v1);
  cvx_support(b, nd, s->v2);
  sub3(s->v, s->v1, s->v2);
}

static void mpr_portal_dir(const MprSimplex *p, double *dir) {
  double a[3], b[3];
  sub3(a, p->ps[2].v, p->ps[1].v);
  sub3(b, p->ps[3].v, p->ps[1].v);
  cross3(dir, a, b);
  normalize3(dir);
}

static int mpr_reach_tolerance(const MprSimplex *p, const MprSup *v4, const double *dir) {
  double dv4 = dot3(v4->v, dir);
  double d1 = dv4 - dot3(p->ps[1].v, dir), d2 = dv4 - dot3(p->ps[2].v, dir), d3 = dv4 - dot3(p->ps[3].v, dir);
  double m = d1 < d2 ? d1 : d2;
  m = m < d3 ? m : d3;
  return ccd_eq(m, MPR_TOLERANCE) || m < MPR_TOLERANCE;
}

static void mpr_expand_portal(MprSimplex *p, const MprSup *v4) {
  double v4v0[3];
  cross3(v4v0, v4->v, p->ps[0].v);
  if (dot3(p->ps[1].v, v4v0) > 0) {
    if (dot3(p->ps[2].v, v4v0) > 0) p->ps[1] = *v4; else p->ps[3] = *v4;
  } else {
    if (dot3(p->ps[3].v, v4v0) > 0) p->ps[2] = *v4; else p->ps[1] = *v4;
  }
}

/* 0: portal found, 1: touching contact, 2: origin on the segment v0-v1, -1: no intersection */
static int mpr_discover_portal(const CvxGeom *a, const CvxGeom *b, MprSimplex *p) {
  double dir[3], va[3], vb[3], dot;
  memcpy(p->ps[0].v1, a->center, sizeof a->center);
  memcpy(p->ps[0].v2, b->center, sizeof b->center);
  sub3(p->ps[0].v, a->center, b->center);
  p->size = 1;
  if (ccd_is_zero(p->ps[0].v[0]) && ccd_is_zero(p->ps[0].v[1]) && ccd_is_zero(p->ps[0].v[2])) p->ps[0].v[0] += CCD_EPS * 10;
  for (int i = 0; i < 3; i++) dir[i] = -p->ps[0].v[i];
  normalize3(dir);
  mpr_support(a, b, dir, &p->ps[1]);
  p->size = 2;
  dot = dot3(p->ps[1].v, dir);
  if (ccd_is_zero(dot) || dot < 0) return -1;
  cross3(dir, p->ps[0].v, p->ps[1].v);
  if (ccd_is_zero(dot3(dir, dir))) {
    if (ccd_is_zero(p->ps[1].v[0]) && ccd_is_zero(p->ps[1].v[1]) && ccd_is_zero(p->ps[1].v[2])) return 1;
    return 2;
  }
  normalize3(dir);
  mpr_support(a, b, dir, &p->ps[2]);
  dot = dot3(p->ps[2].v, dir);
  if (ccd_is_zero(dot) || dot < 0) return -1;
  p->size = 3;
  sub3(va, p->ps[1].v, p->ps[0].v);
  sub3(vb, p->ps[2].v, p->ps[0].v);
  cross3(dir, va, vb);
  normalize3(dir);
  if (dot3(dir, p->ps[0].v) > 0) {
    MprSup t = p->ps[1]; p->ps[1] = p->ps[2]; p->ps[2] = t;
    for (int i = 0; i < 3; i++) dir[i] = -dir[i];
  }
  int guard = 0;
  while (p->size < 4) {
    if (++guard > 1000) return -1; /* libccd loops here without a bound; degenerate inputs only */
    mpr_support(a, b, dir, &p->ps[3]);
    dot = dot3(p->ps[3].v, dir);
    if (ccd_is_zero(dot) || dot < 0) return -1;
    int cont = 0;
    cross3(va, p->ps[1].v, p->ps[3].v);
    dot = dot3(va, p->ps[0].v);
    if (dot < 0 && !ccd_is_zero(dot)) { p->ps[2] = p->ps[3]; cont = 1; }
    if (!cont) {
      cross3(va, p->ps[3].v, p->ps[2].v);
      dot = dot3(va, p->ps[0].v);
      if (dot < 0 && !ccd_is_zero(dot)) { p->ps[1] = p->ps[3]; cont = 1; }
    }
    if (cont) {
      sub3(va, p->ps[1].v, p->ps[0].v);
      sub3(vb, p->ps[2].v, p->ps[0].v);
      cross3(dir, va, vb);
      normalize3(dir);
    } else {
      p->size = 4;
    }
  }
  return 0;
}

static int mpr_refine_portal(const CvxGeom *a, const CvxGeom *b, MprSimplex *p) {
  double dir[3];
  MprSup v4;
  for (int guard = 0; guard < 10000; guard++) {
    mpr_portal_dir(p, dir);
    double dot = dot3(dir, p->ps[1].v);
    if (ccd_is_zero(dot) || dot > 0) return 0; /* portal encapsules the origin */
    mpr_support(a, b, dir, &v4);
    dot = dot3(v4.v, dir);
    if (!(ccd_is_zero(dot) || dot > 0) || mpr_reach_tolerance(p, &v4, dir)) return -1;
    mpr_expand_portal(p, &v4);
  }
  return -1;
}

/* closest point of the triangle (a, b, c) to the origin (what ccdVec3PointTriDist2 returns as witness) */
static double tri_closest_origin(const double *a, const double *b, const double *c, double *w) {
  double ab[3], ac[3], ap[3] = {-a[0], -a[1], -a[2]};
  sub3(ab, b, a); sub3(ac, c, a);
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { memcpy(w, a, 3 * sizeof(double)); return dot3(w, w); }
  double bp[3] = {-b[0], -b[1], -b[2]};
  double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { memcpy(w, b, 3 * sizeof(double)); return dot3(w, w); }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) {
    double v = d1 / (d1 - d3);
    for (int i = 0; i < 3; i++) w[i] = a[i] + v * ab[i];
    return dot3(w, w);
  }
  double cp[3] = {-c[0], -c[1], -c[2]};
  double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { memcpy(w, c, 3 * sizeof(double)); return dot3(w, w); }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) {
    double v = d2 / (d2 - d6);
    for (int i = 0; i < 3; i++) w[i] = a[i] + v * ac[i];
    return dot3(w, w);
  }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    double v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for (int i = 0; i < 3; i++) w[i] = b[i] + v * (c[i] - b[i]);
    return dot3(w, w);
  }
  double den = 1.0 / (va + vb + vc), v = vb * den, u = vc * den;
  for (int i = 0; i < 3; i++) w[i] = a[i] + ab[i] * v + ac[i] * u;
  return dot3(w, w);
}

static void mpr_find_pos(const MprSimplex *p, double *pos) {
  double dir[3], b[4], t[3], sum;
  mpr_portal_dir(p, dir);
  cross3(t, p->ps[1].v, p->ps[2].v); b[0] = dot3(t, p->ps[3].v);
  cross3(t, p->ps[3].v, p->ps[2].v); b[1] = dot3(t, p->ps[0].v);
  cross3(t, p->ps[0].v, p->ps[1].v); b[2] = dot3(t, p->ps[3].v);
  cross3(t, p->ps[2].v, p->ps[1].v); b[3] = dot3(t, p->ps[0].v);
  sum = b[0] + b[1] + b[2] + b[3];
  if (ccd_is_zero(sum) || sum < 0) {
    b[0] = 0;
    cross3(t, p->ps[2].v, p->ps[3].v); b[1] = dot3(t, dir);
    cross3(t, p->ps[3].v, p->ps[1].v); b[2] = dot3(t, dir);
    cross3(t, p->ps[1].v, p->ps[2].v); b[3] = dot3(t, dir);
    sum = b[1] + b[2] + b[3];
  }
  double inv = 1.0 / sum, p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
  for (int k = 0; k < 4; k++)
    for (int i = 0; i < 3; i++) { p1[i] += b[k] * p->ps[k].v1[i]; p2[i] += b[k] * p->ps[k].v2[i]; }
  for (int i = 0; i < 3; i++) pos[i] = 0.5 * inv * (p1[i] + p2[i]);
}

/* ccdMPRPenetration: 0 = penetration found (depth, dir from a to b, pos), -1 = none */
static int mpr_penetration(const CvxGeom *a, const CvxGeom *b, double *depth, double *dir, double *pos) {
  MprSimplex p;
  int res = mpr_discover_portal(a, b, &p);
  if (res < 0) return -1;
  if (res == 1) { /* touching: depth 0, direction undefined */
    *depth = 0; dir[0] = dir[1] = dir[2] = 0;
    for (int i = 0; i < 3; i++) pos[i] = 0.5 * (p.ps[1].v1[i] + p.ps[1].v2[i]);
    return 0;
  }
  if (res == 2) { /* origin on the segment v0-v1 */
    for (int i = 0; i < 3; i++) pos[i] = 0.5 * (p.ps[1].v1[i] + p.ps[1].v2[i]);
    memcpy(dir, p.ps[1].v, 3 * sizeof(double));
    *depth = normalize3(dir);
    return 0;
  }
  if (mpr_refine_portal(a, b, &p) < 0) return -1;
  MprSup v4;
  double d[3];
  for (int it = 0;; it++) {
    mpr_portal_dir(&p, d);
    mpr_support(a, b, d, &v4);
    if (mpr_reach_tolerance(&p, &v4, d) || it > MPR_ITERATIONS) {
      *depth = sqrt(tri_closest_origin(p.ps[1].v, p.ps[2].v, p.ps[3].v, dir));
      if (ccd_is_zero(*depth)) dir[0] = dir[1] = dir[2] = 0; else normalize3(dir);
      mpr_find_pos(&p, pos);
      return 0;
    }
    mpr_expand_portal(&p, &v4);
  }
}

/* [EXT] mjc_Convex (margin 0): one contact, normal from geom1 to geom2; spheres get their analytic normal (mjc_fixNormal) */
static int c_convex(RawCon *c, const CvxGeom *a, const CvxGeom *b) {
  double depth, dir[3], pos[3];
  if (mpr_penetration(a, b, &depth, dir, pos) != 0) return 0;
  if (dir[0] == 0 && dir[1] == 0 && dir[2] == 0) return 0; /* contact found but normal undefined */
  c->dist = -depth;
  memcpy(c->pos, pos, sizeof pos);
  memcpy(c->normal, dir, sizeof dir);
  double n1[3], n2[3];
  int h1 = 0, h2 = 0;
  if (a->type == DM_GEOM_SPHERE) { sub3(n1, pos, a->pos); h1 = normalize3(n1) > MINVAL; }
  if (b->type == DM_GEOM_SPHERE) { sub3(n2, b->pos, pos); h2 = normalize3(n2) > MINVAL; }
  if (h1 && h2) { for (int i = 0; i < 3; i++) c->normal[i] = n1[i] + n2[i]; normalize3(c->normal); }
  else if (h1) memcpy(c->normal, n1, sizeof n1);
  else if (h2) memcpy(c->normal, n2, sizeof n2);
  c->tangent[0] = c->tangent[1] = c->tangent[2] = 0;
  return 1;
}

/* ------------------------------------------------------------------ plane routines */
/* [EXT] mjc_PlaneCylinder: deepest rim point, the other end of that generator, and two rim points at +-120 degrees */
static int c_plane_cylinder(RawCon *c, double margin, const double *ppos, const double *pmat, const double *cpos,
                            const double *cmat, const double *size) {
  double normal[3] = {pmat[2], pmat[5], pmat[8]}, axis[3] = {cmat[2], cmat[5], cmat[8]};
  double prjaxis = dot3(normal, axis);
  if (prjaxis > 0) { for (int i = 0; i < 3; i++) axis[i] = -axis[i]; prjaxis = -prjaxis; }
  double dif[3], vec[3];
  sub3(dif, cpos, ppos);
  double dist0 = dot3(dif, normal);
  for (int i = 0; i < 3; i++) vec[i] = axis[i] * prjaxis - normal[i];
  double len2 = dot3(vec, vec);
  if (len2 >= MINVAL * MINVAL) { double s = size[0] / sqrt(len2); for (int i = 0; i < 3; i++) vec[i] *= s; }
  else for (int i = 0; i < 3; i++) vec[i] = cmat[3 * i] * size[0];
  double prjvec = dot3(vec, normal);
  for (int i = 0; i < 3; i++) axis[i] *= size[1];
  prjaxis *= size[1];
  int n = 0;
  if (dist0 + prjaxis + prjvec > margin) return 0;
#define PC_ADD(dd, EXPR) do { c[n].dist = (dd); for (int i = 0; i < 3; i++) c[n].pos[i] = cpos[i] + (EXPR) - normal[i] * (dd) * 0.5; \
    memcpy(c[n].normal, normal, sizeof normal); c[n].tangent[0] = c[n].tangent[1] = c[n].tangent[2] = 0; n++; } while (0)
  PC_ADD(dist0 + prjaxis + prjvec, vec[i] + axis[i]);
  if (dist0 - prjaxis + prjvec <= margin) PC_ADD(dist0 - prjaxis + prjvec, vec[i] - axis[i]);
  double prjvec1 = -0.5 * prjvec;
  if (dist0 + prjaxis + prjvec1 <= margin) {
    double vec1[3];
    cross3(vec1, vec, axis);
    normalize3(vec1);
    for (int i = 0; i < 3; i++) vec1[i] *= size[0] * sqrt(3.0) * 0.5;
    PC_ADD(dist0 + prjaxis + prjvec1, vec1[i] + axis[i] - 0.5 * vec[i]);
    PC_ADD(dist0 + prjaxis + prjvec1, -vec1[i] + axis[i] - 0.5 * vec[i]);
  }
#undef PC_ADD
  return n;
}

/* [EXT] mjc_PlaneConvex for a mesh: the support vertex towards the plane (the documented contact), then the support vertices
 * of three directions tilted by 0.3 around the plane normal (120 degrees apart), kept when they are other vertices and within
 * the margin: four contacts at most.  LOW-CONFIDENCE restatement of MuJoCo's multi-contact rule for plane-mesh pairs. */
static int c_plane_mesh(RawCon *c, double margin, const double *ppos, const double *pmat, const double *gpos,
                        const double *gmat, const double *vert, int nvert) {
  double normal[3] = {pmat[2], pmat[5], pmat[8]}, t1[3] = {pmat[0], pmat[3], pmat[6]}, t2[3] = {pmat[1], pmat[4], pmat[7]};
  int used[4], n = 0;
  for (int k = 0; k < 4; k++) {
    double dw[3], dl[3];
    if (k == 0) for (int i = 0; i < 3; i++) dw[i] = -normal[i];
    else {
      double ang = 2.0 * 3.14159265358979323846 * (k - 1) / 3.0, ca = 0.3 * cos(ang), sa = 0.3 * sin(ang);
      for (int i = 0; i < 3; i++) dw[i] = -normal[i] + ca * t1[i] + sa * t2[i];
    }
    rot_vec_t(dl, gmat, dw);
    int vi = mesh_support_index(vert, nvert, dl), dup = 0;
    for (int j = 0; j < n; j++) dup |= used[j] == vi;
    if (dup) continue;
    double v[3], dif[3];
    rot_vec(v, gmat, vert + 3 * vi);
    for (int i = 0; i < 3; i++) v[i] += gpos[i];
    sub3(dif, v, ppos);
    double dist = dot3(dif, normal);
    if (dist > margin) { if (k == 0) return 0; continue; }
    used[n] = vi;
    c[n].dist = dist;
    for (int i = 0; i < 3; i++) c[n].pos[i] = v[i] - 0.5 * dist * normal[i];
    memcpy(c[n].normal, normal, sizeof normal);
    c[n].tangent[0] = c[n].tangent[1] = c[n].tangent[2] = 0;
    n++;
  }
  return n;
}

#endif /* DM_CONVEX_H */
