// dm_kernels.hip — fused DPEnv.step() kernel for gfx950 (CDNA4), one wavefront per environment.
//
// Reference path being replaced:
//   DPEnv.step                  src/deepmimic_env.py:335-484
//     do_simulation -> mj_step  [EXT] MuJoCo RK4 x (kinematics, inertia, collision, PGS)
//     get_obs                   src/deepmimic_env.py:33-143
//     calc_imitation_reward     src/deepmimic_env.py:193-256
//     termination, counters     src/deepmimic_env.py:418-476
//   VecEnv worker auto-reset    [EXT] SubprocVecEnv (src/sb3_ppo.py:275)
//
// Lane roles per phase are described in dm_device.h / DESIGN.md §3.  The physics
// algorithm is the one restated in SURVEY.md Appendix B; the fp64 oracle
// (oracle/dm_oracle.c) is an independent scalar implementation of the same spec.
#include "../../include/deepmimic_hip.h"
#include "dm_device.h"
#include "dm_topology.h"

#include <math.h>

#include <type_traits>

// One wavefront owns one environment, so LDS hand-offs between phases only need ordering inside the
// wave: DS operations of a wave execute in order; the fences stop the compiler moving LDS accesses.
#define SYNC()                                              \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)
#define DMK_ENVS_PER_BLOCK 1
#ifndef DMK_WAVES_PER_SIMD
#define DMK_WAVES_PER_SIMD 2
#endif
// Diagnostic build only (-DDM_PROFILE): per-phase cycle stamps, written to the debug buffer
// [352:368).  The shipped library never executes a stamp.
#ifdef DM_PROFILE
#define PROF_DECL do { if ((threadIdx.x & 63) == 0) { for (int _i = 0; _i < 16; _i++) g_S.prof[_i] = 0; for (int _i = 0; _i < 4; _i++) g_S.prof_stage[_i] = 0; g_S.prof_t = g_S.prof_t0 = __builtin_amdgcn_s_memtime(); } } while (0)
// ticks from kernel entry to the end of RK stage i (scripts/sched_predict.py: how well stage 0 predicts the step)
#define PROF_STAGE(i) do { if ((threadIdx.x & 63) == 0 && (i) < 4) g_S.prof_stage[i] = (unsigned)(__builtin_amdgcn_s_memtime() - g_S.prof_t0); } while (0)
#define PROF(i) do { if ((threadIdx.x & 63) == 0) { unsigned long long _t = __builtin_amdgcn_s_memtime(); g_S.prof[i] += (unsigned)(_t - g_S.prof_t); g_S.prof_t = _t; } } while (0)
#if DM_PROFILE == 2
#define PROF2(i) PROF(i)
#else
#define PROF2(i) do {} while (0)
#endif
#else
#define PROF2(i) do {} while (0)
#define PROF_STAGE(i) do {} while (0)
#define PROF_DECL do {} while (0)
#define PROF(i) do {} while (0)
#endif
typedef __attribute__((address_space(3))) const float *lds_cfloat_p;
typedef float mfma_f16v __attribute__((ext_vector_type(16)));
#ifndef DMK_MFMA_MIN_ROWS
#define DMK_MFMA_MIN_ROWS 8   // below this the 17-MFMA product (1088 cycles) loses to nefc x (34 v_readlane + 34 FMA)
#endif
typedef const __attribute__((address_space(1))) DmDev GDev;          // model tables: global address space
typedef const __attribute__((address_space(1))) DmPairDev GPair;
#define MINVALF 1e-15f
#define MAXVALF 1e10f

namespace {

// Compile-time loop with early exit: f(integral_constant<int, I>) returns false to stop.  Guarantees
// static register indices for the per-lane A-matrix row (a rolled loop would put it in scratch).
template <int I, int N>
struct StaticFor {
  template <class F>
  static __device__ __forceinline__ void run(F &&f) {
    if constexpr (I < N) {
      if (f(std::integral_constant<int, I>{})) StaticFor<I + 1, N>::run(f);
    }
  }
};

__device__ __forceinline__ float rl(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// a where the lane's bit is set in the compile-time lane mask, b elsewhere: one v_cndmask with the mask in SGPRs
// (a per-lane bit test would cost three VALU instructions)
template <unsigned long long MASK>
__device__ __forceinline__ float lane_sel(float a, float b) {
  float out;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(out) : "v"(b), "v"(a), "s"(MASK));
  return out;
}
template <unsigned long long MASK>
__device__ __forceinline__ int lane_sel_i(int a, int b) {
  int out;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(out) : "v"(b), "v"(a), "s"(MASK));
  return out;
}
// Column elimination step for dof K: walk the static ancestor chain I = parent(K), parent(parent(K)), ...
template <int K, int I>
struct ElimAnc {
  static __device__ __forceinline__ void run(float (&C)[DMK_NV], const float Cs) {
    if constexpr (I >= 0) {
      C[I] = fmaf(-rl(Cs, I), C[K], C[I]);
      ElimAnc<K, (I >= 0 ? topo::PARENT[I >= 0 ? I : 0] : -1)>::run(C, Cs);
    }
  }
};
// Triangular solves with the published sparse factor (lane = dof, rows UNSCALED: L[i][j] = M[i][j] * dinv[i]).
// The broadcast value comes from a static lane, the factor entry from a per-lane base plus a static offset, and the
// set of lanes a step touches is a compile-time lane mask: 5 instructions per step.
// x <- L^-T x (then scaled by the caller).  negdep = -(depth of this lane's dof).
__device__ __forceinline__ float solve_LT(float x, const float dv, const int negdep, const float *M) {
  StaticFor<0, DMK_NV - 1>::run([&](auto ic) {
    constexpr int i = DMK_NV - 1 - decltype(ic)::value;   // 33 .. 1
    const float xi = rl(x * dv, i);
    const float l = lane_sel<topo::anc_mask(i)>(M[topo::MADR[i] + topo::NANC[i] + negdep], 0.f);
    x = fmaf(-l, xi, x);
    return true;
  });
  return x;
}
// z <- z - M[:, j] x_j with x = z * dinv (D^-1 and L^-1 fused); mrow = MADR[lane] + depth(lane).
__device__ __forceinline__ float solve_L(float x, const float dv, const int mrow, const float *M) {
  StaticFor<0, DMK_NV - 1>::run([&](auto jc) {
    constexpr int j = decltype(jc)::value;                 // 0 .. 32
    const float xj = rl(x * dv, j);
    const float l = lane_sel<topo::desc_mask(j)>(M[mrow - topo::NANC[j]], 0.f);
    x = fmaf(-l, xj, x);
    return true;
  });
  return x;
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 64 lanes, result uniform in every lane
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
}
__device__ __forceinline__ unsigned long long lanemask_lt(int lane) {
  return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}
__device__ __forceinline__ int prefix_count(bool flag, int lane, int *total) {
  unsigned long long m = __ballot(flag);
  *total = __popcll(m);
  return __popcll(m & lanemask_lt(lane));
}

__device__ __forceinline__ void cross3(float *r, const float *a, const float *b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ float dot3(const float *a, const float *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
__device__ __forceinline__ void quat_mul(float *r, const float *a, const float *b) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
__device__ __forceinline__ void quat_rot(float *r, const float *q, const float *v) {
  float t[3], u[3];
  cross3(t, q + 1, v);
  t[0] *= 2.f; t[1] *= 2.f; t[2] *= 2.f;
  cross3(u, q + 1, t);
  r[0] = v[0] + q[0] * t[0] + u[0];
  r[1] = v[1] + q[0] * t[1] + u[1];
  r[2] = v[2] + q[0] * t[2] + u[2];
}
__device__ __forceinline__ void quat2mat(float *m, const float *q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
__device__ __forceinline__ void quat_normalize(float *q) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVALF) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { float i = 1.0f / n; q[0] *= i; q[1] *= i; q[2] *= i; q[3] *= i; }
}
__device__ __forceinline__ void mat_vec(float *r, const float *m, const float *v) {
  float x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  float y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  float z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void mat_t_vec(float *r, const float *m, const float *v) {
  float x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  float y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  float z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void mul_inert_vec(float *r, const float *I, const float *v) {
  r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
  r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] - I[6] * v[5] + I[8] * v[3];
  r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
  r[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
  r[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
  r[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}
__device__ __forceinline__ void cross_motion(float *r, const float *vel, const float *v) {
  float a[3], b[3], c[3];
  cross3(a, vel, v);
  cross3(b, vel, v + 3);
  cross3(c, vel + 3, v);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
__device__ __forceinline__ void cross_force(float *r, const float *vel, const float *f) {
  float a[3], b[3], c[3];
  cross3(a, vel, f);
  cross3(b, vel + 3, f + 3);
  cross3(c, vel, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// py3dtf.Quaternion(x,y,z,w).to_rpy() restated: standard ZYX, no normalisation
__device__ __forceinline__ void quat_to_rpy(const float *q, float *rpy) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  rpy[0] = atan2f(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
  rpy[1] = asinf(clampf(2 * (w * y - z * x), -1.f, 1.f));
  rpy[2] = atan2f(2 * (w * z + x * y), 1 - 2 * (y * y + z * z));
}

// counter-based generator shared with oracle/dm_oracle.c (hash32)
__device__ __host__ __forceinline__ uint32_t dm_hash32(uint64_t seed, uint32_t env, uint32_t step, uint32_t j) {
  uint64_t x = seed ^ ((uint64_t)env * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)step * 0xBF58476D1CE4E5B9ull) ^
               ((uint64_t)j * 0x94D049BB133111EBull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

// ---------------------------------------------------------------- narrowphase (per lane)
// up to 4 candidate contacts per lane in registers (static slots)
struct Cand {
  float d[4], p[4][3], n[4][3], t[3];
  int valid;  // bit s: slot s holds a contact
};

template <int S>
__device__ __forceinline__ void np_plane_sphere(Cand &c, float margin, const float *ppos, const float *pn,
                                                const float *spos, float r) {
  float df[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
  float dist = dot3(df, pn) - r;
  if (dist > margin) return;
  c.valid |= 1 << S;
  c.d[S] = dist;
  for (int i = 0; i < 3; i++) { c.n[S][i] = pn[i]; c.p[S][i] = spos[i] - pn[i] * (r + 0.5f * dist); }
}
template <int S>
__device__ __forceinline__ void np_sphere_sphere(Cand &c, float margin, const float *p1, float r1,
                                                 const float *p2, float r2) {
  float df[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  float cd = sqrtf(dot3(df, df)), dist = cd - r1 - r2;
  if (dist > margin) return;
  c.valid |= 1 << S;
  c.d[S] = dist;
  float inv = (cd < MINVALF) ? 0.f : 1.0f / cd;
  c.n[S][0] = (cd < MINVALF) ? 1.f : df[0] * inv;
  c.n[S][1] = df[1] * inv;
  c.n[S][2] = df[2] * inv;
  for (int i = 0; i < 3; i++) c.p[S][i] = p1[i] + c.n[S][i] * (r1 + 0.5f * dist);
}
template <int S>
__device__ __forceinline__ void np_sphere_box(Cand &c, float margin, const float *spos, float r,
                                              const float *bpos, const float *bmat, const float *size) {
  float t[3] = {spos[0] - bpos[0], spos[1] - bpos[1], spos[2] - bpos[2]}, ctr[3], cl[3], nl[3];
  mat_t_vec(ctr, bmat, t);
  for (int i = 0; i < 3; i++) { cl[i] = clampf(ctr[i], -size[i], size[i]); nl[i] = cl[i] - ctr[i]; }
  float dd = sqrtf(dot3(nl, nl)), dist;
  if (dd - r > margin) return;
  if (dd <= MINVALF) {
    float closest = 2 * (size[0] + size[1] + size[2]);
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      float test = size[i / 2] - ((i % 2) ? -1.0f : 1.0f) * ctr[i / 2];
      if (test < closest) { closest = test; k = i; }
    }
    nl[0] = nl[1] = nl[2] = 0;
    float sgn = (k % 2) ? 1.0f : -1.0f;
    if (k / 2 == 0) nl[0] = sgn; else if (k / 2 == 1) nl[1] = sgn; else nl[2] = sgn;
    dist = -closest - r;
  } else {
    float inv = 1.0f / dd;
    nl[0] *= inv; nl[1] *= inv; nl[2] *= inv;
    dist = dd - r;
  }
  float pl[3] = {ctr[0] + nl[0] * (r + 0.5f * dist), ctr[1] + nl[1] * (r + 0.5f * dist),
                 ctr[2] + nl[2] * (r + 0.5f * dist)};
  c.valid |= 1 << S;
  c.d[S] = dist;
  mat_vec(c.n[S], bmat, nl);
  mat_vec(t, bmat, pl);
  for (int i = 0; i < 3; i++) c.p[S][i] = t[i] + bpos[i];
}
__device__ __forceinline__ float cb_grad(const float *p, const float *a, const float *size, float t) {
  float g = 0;
  for (int i = 0; i < 3; i++) {
    float x = p[i] + a[i] * t;
    g += a[i] * (x - clampf(x, -size[i], size[i]));
  }
  return g;
}

__device__ __forceinline__ void make_frame(float *f) {  // [EXT] mju_makeFrame
  float n = sqrtf(dot3(f, f));
  if (n < MINVALF) { f[0] = 1; f[1] = 0; f[2] = 0; }
  else { float i = 1.f / n; f[0] *= i; f[1] *= i; f[2] *= i; }
  if (sqrtf(dot3(f + 3, f + 3)) < 0.5f) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5f && f[1] > -0.5f) f[4] = 1; else f[5] = 1;
  }
  float t = dot3(f, f + 3);
  f[3] -= t * f[0]; f[4] -= t * f[1]; f[5] -= t * f[2];
  n = sqrtf(dot3(f + 3, f + 3));
  if (n < MINVALF) { f[3] = 1; f[4] = 0; f[5] = 0; }
  else { float i = 1.f / n; f[3] *= i; f[4] *= i; f[5] *= i; }
  cross3(f + 6, f, f + 3);
}

// Box-box for the single foot-foot pair: SAT over 15 axes, then reference-face clipping
// (polygon scratch in LDS) or closest points of two edges.  Same construction as the oracle.
__device__ __noinline__ void np_box_box(EnvLds &S, GPair *pr) {
  // operands are fetched here (LDS poses, pair record) so the caller needs no stack arrays for them
  const float margin = pr->margin;
  float p1[3], p2[3], R1[9], R2[9], s1[3], s2[3];
  for (int i = 0; i < 3; i++) { p1[i] = S.gpos[pr->g1][i]; p2[i] = S.gpos[pr->g2][i]; s1[i] = pr->z1[i]; s2[i] = pr->z2[i]; }
  for (int i = 0; i < 9; i++) { R1[i] = S.gmat[pr->g1][i]; R2[i] = S.gmat[pr->g2][i]; }
  S.u.bb.ncand = 0;
  float R[9], AR[9], t[3], tw[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  mat_t_vec(t, R1, tw);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      R[3 * i + j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];
      AR[3 * i + j] = fabsf(R[3 * i + j]) + 1e-9f;
    }
  float best = -1e30f, bn[3] = {0, 0, 0};
  int code = -1;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    float s = fabsf(t[i]) - (s1[i] + s2[0] * AR[3 * i] + s2[1] * AR[3 * i + 1] + s2[2] * AR[3 * i + 2]);
    if (s > margin) return;
    if (s > best) { best = s; code = i; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    float tj = t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j];
    float s = fabsf(tj) - (s2[j] + s1[0] * AR[j] + s1[1] * AR[3 + j] + s1[2] * AR[6 + j]);
    if (s > margin) return;
    if (s > best) { best = s; code = 3 + j; }
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float ei[3] = {0, 0, 0}, ej[3] = {R[j], R[3 + j], R[6 + j]}, ax[3];
      ei[i] = 1;
      cross3(ax, ei, ej);
      float l = sqrtf(dot3(ax, ax));
      if (l < 1e-6f) continue;
      ax[0] /= l; ax[1] /= l; ax[2] /= l;
      float ra = s1[0] * fabsf(ax[0]) + s1[1] * fabsf(ax[1]) + s1[2] * fabsf(ax[2]), rb = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        float ek[3] = {R[k], R[3 + k], R[6 + k]};
        rb += s2[k] * fabsf(dot3(ax, ek));
      }
      float s = fabsf(dot3(t, ax)) - (ra + rb);
      if (s > margin) return;
      if (s > best + 0.05f * fabsf(best) + 1e-6f) { best = s; code = 6 + 3 * i + j; bn[0] = ax[0]; bn[1] = ax[1]; bn[2] = ax[2]; }
    }
  if (code < 0) return;
  float (*cand)[8] = S.u.bb.cand;
  if (code >= 6) {
    int i = (code - 6) / 3, j = (code - 6) % 3;
    float n1[3] = {bn[0], bn[1], bn[2]};
    if (dot3(n1, t) < 0) { n1[0] = -n1[0]; n1[1] = -n1[1]; n1[2] = -n1[2]; }
    float pa[3], pb[3] = {t[0], t[1], t[2]};
    for (int k = 0; k < 3; k++) pa[k] = (k == i) ? 0.f : ((n1[k] > 0) ? s1[k] : -s1[k]);
    for (int k = 0; k < 3; k++) {
      if (k == j) continue;
      float ek[3] = {R[k], R[3 + k], R[6 + k]};
      float sg = (dot3(n1, ek) > 0) ? -s2[k] : s2[k];
      pb[0] += sg * ek[0]; pb[1] += sg * ek[1]; pb[2] += sg * ek[2];
    }
    float ua[3] = {0, 0, 0}, ub[3] = {R[j], R[3 + j], R[6 + j]}, w[3];
    ua[i] = 1;
    for (int k = 0; k < 3; k++) w[k] = pb[k] - pa[k];
    float uaub = dot3(ua, ub), q1 = dot3(ua, w), q2 = -dot3(ub, w), dd = 1 - uaub * uaub;
    float alpha = 0, beta = 0;
    if (dd > 1e-12f) { alpha = (q1 + uaub * q2) / dd; beta = (uaub * q1 + q2) / dd; }
    alpha = clampf(alpha, -s1[i], s1[i]);
    beta = clampf(beta, -s2[j], s2[j]);
    float mid[3], mw[3], nw[3];
    for (int k = 0; k < 3; k++) mid[k] = 0.5f * ((pa[k] + ua[k] * alpha) + (pb[k] + ub[k] * beta));
    mat_vec(mw, R1, mid);
    mat_vec(nw, R1, n1);
    cand[0][0] = best;
    for (int k = 0; k < 3; k++) { cand[0][1 + k] = mw[k] + p1[k]; cand[0][4 + k] = nw[k]; }
    S.u.bb.ncand = 1;
    return;
  }
  const float *Ra, *Rb, *sa, *sb, *pa, *pb;
  int ax, flip;
  if (code < 3) { Ra = R1; Rb = R2; sa = s1; sb = s2; pa = p1; pb = p2; ax = code; flip = 0; }
  else { Ra = R2; Rb = R1; sa = s2; sb = s1; pa = p2; pb = p1; ax = code - 3; flip = 1; }
  float nrm[3] = {Ra[ax], Ra[3 + ax], Ra[6 + ax]}, dab[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  if (dot3(nrm, dab) < 0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
  int ib = 0;
  float bestd = -1, nb[3];
  mat_t_vec(nb, Rb, nrm);
  for (int k = 0; k < 3; k++) if (fabsf(nb[k]) > bestd) { bestd = fabsf(nb[k]); ib = k; }
  float sgn = (nb[ib] > 0) ? -1.0f : 1.0f;
  int u = (ib + 1) % 3, v = (ib + 2) % 3;
  float (*poly)[3] = S.u.bb.poly[0];
  float (*tmp)[3] = S.u.bb.poly[1];
  int np = 4;
  for (int q = 0; q < 4; q++) {
    float su = (q == 0 || q == 3) ? -sb[u] : sb[u], sv = (q < 2) ? -sb[v] : sb[v];
    float loc[3];
    loc[ib] = sgn * sb[ib]; loc[u] = su; loc[v] = sv;
    float w[3], rel[3], pq[3];
    mat_vec(w, Rb, loc);
    for (int k = 0; k < 3; k++) rel[k] = w[k] + pb[k] - pa[k];
    mat_t_vec(pq, Ra, rel);
    poly[q][0] = pq[0]; poly[q][1] = pq[1]; poly[q][2] = pq[2];
  }
  int axes[2] = {(ax + 1) % 3, (ax + 2) % 3};
  for (int e = 0; e < 2; e++)
    for (int sd = -1; sd <= 1; sd += 2) {
      int a = axes[e], nn = 0;
      for (int q = 0; q < np; q++) {
        float *Pq = poly[q], *Qq = poly[(q + 1) % np];
        float dp = sd * Pq[a] - sa[a], dq = sd * Qq[a] - sa[a];
        if (dp <= 0) { tmp[nn][0] = Pq[0]; tmp[nn][1] = Pq[1]; tmp[nn][2] = Pq[2]; nn++; }
        if ((dp < 0 && dq > 0) || (dp > 0 && dq < 0)) {
          float f = dp / (dp - dq);
          for (int k = 0; k < 3; k++) tmp[nn][k] = Pq[k] + f * (Qq[k] - Pq[k]);
          nn++;
        }
        if (nn >= 15) break;
      }
      np = nn;
      for (int q = 0; q < np; q++) { poly[q][0] = tmp[q][0]; poly[q][1] = tmp[q][1]; poly[q][2] = tmp[q][2]; }
      if (np == 0) return;
    }
  float nl[3];
  mat_t_vec(nl, Ra, nrm);
  int cnt = 0;
  for (int q = 0; q < np && cnt < 8; q++) {
    float pq[3] = {poly[q][0], poly[q][1], poly[q][2]};
    float depth = dot3(nl, pq) - sa[ax];
    if (depth > margin) continue;
    float pl[3], pw[3];
    for (int k = 0; k < 3; k++) pl[k] = pq[k] - nl[k] * 0.5f * depth;
    mat_vec(pw, Ra, pl);
    cand[cnt][0] = depth;
    for (int k = 0; k < 3; k++) { cand[cnt][1 + k] = pw[k] + pa[k]; cand[cnt][4 + k] = flip ? -nrm[k] : nrm[k]; }
    cnt++;
  }
  // keep the 4 deepest, preserving polygon order
  while (cnt > 4) {
    int w = 0;
    for (int q = 1; q < cnt; q++) if (cand[q][0] >= cand[w][0]) w = q;
    for (int q = w; q < cnt - 1; q++)
      for (int k = 0; k < 7; k++) cand[q][k] = cand[q + 1][k];
    cnt--;
  }
  S.u.bb.ncand = cnt;
}

// general-power branch of the impedance: never taken with MuJoCo's default solimp (power 2); out of line so that powf's
// register footprint does not weigh on the constraint-row phase
__device__ __noinline__ float impedance_pow(float x, float mid, float power) {
  return (x <= mid) ? powf(x, power) / powf(mid, power - 1) : 1 - powf(1 - x, power) / powf(1 - mid, power - 1);
}
__device__ __forceinline__ float impedance(const float *solimp, float pos, float margin) {
  float dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (dmin == dmax || width <= MINVALF) return 0.5f * (dmin + dmax);
  float x = fabsf(pos - margin) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  float y;
  if (power == 1) y = x;
  else if (power == 2) y = (x <= mid) ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);   // MuJoCo default
  else y = impedance_pow(x, mid, power);
  return dmin + y * (dmax - dmin);
}

}  // namespace

__shared__ EnvLds g_S;  // one environment per 64-thread block

#define bp (T.b_parent[lb])
#define bdofadr (T.b_dofadr[lb])
#define bdofnum (T.b_dofnum[lb])
#define bpos (T.b_pos[lb])
#define bipos (T.b_ipos[lb])
#define binert (T.b_inertia[lb])
#define bsub (T.b_subtree[lb])
#define baxis(j) (T.d_axis[(bdofadr + (j)) < DMK_NV ? (bdofadr + (j)) : 0])
#define dbody (T.d_body[lk])
#define dnanc (T.d_nanc[lk])
#define dact (T.d_act[lk])
#define dlimited (T.d_limited[lk])
#define darm (T.d_arm[lk])
#define ddamp (T.d_damp[lk])
#define dlo (T.d_lo[lk])
#define dhi (T.d_hi[lk])
#define dgear (T.d_gear[lk])
#define dclo (T.d_clo[lk])
#define dchi (T.d_chi[lk])
#define gbody (T.g_body[lg])
#define gposl (T.g_pos[lg])
#define gmatl (T.g_mat[lg])

// ---- forward evaluation, part 1: kinematics, inertia, factorisation, bias forces, qacc_smooth
// Every pass is lane-parallel: bodies walk their (<= 4 deep) ancestor chain instead of waiting for a
// level-by-level sweep, and the L^T D L factorisation keeps row i of M in the registers of lane i.
// DAMP2 (Euler integrator only): re-factorise M + h diag(damping) at the same configuration (cdof / cinert of the
// evaluation just done are still in LDS) and return (M + h B)^-1 rhs — MuJoCo's Euler integrates joint damping
// implicitly [EXT mj_Euler]; kinematics, velocity stage and bias forces are skipped.
template <bool DAMP2 = false>
__device__ __forceinline__ float fwd_smooth(GDev &T, const int lane, const float rhs = 0.f) {
  EnvLds &S = g_S;
  const int lb = lane < DMK_NB ? lane : 0;       // lane as body
  const int lk = lane < DMK_NV ? lane : 0;       // lane as dof
  const int lg = lane < DMK_NG ? lane : 0;       // lane as geom
  const bool isbody = lane >= 1 && lane < DMK_NB, isdof = lane < DMK_NV;
  // role constants of this phase (function-local registers)
  const float bmass = (lane < DMK_NB) ? T.b_mass[lb] : 0.f;
  const int b_dofadr = T.b_dofadr[lb], b_dofnum = T.b_dofnum[lb];
  const uint32_t b_chain4 = T.b_chainb[lb];      // ancestor bodies root..self, one byte each (0 = none)
  const unsigned b_sub = T.b_subtree[lb];
  const int d_body = T.d_body[lk], d_nanc = T.d_nanc[lk], d_pbody = T.d_pbody[lk];
  const float mtot_inv = T.total_mass_inv;

  if constexpr (!DAMP2) {
  // ---- P0: normalise the free-joint quaternion in place (mj_kinematics does); half-angle sin/cos
  {
    float q[4] = {S.qpos[3], S.qpos[4], S.qpos[5], S.qpos[6]};
    quat_normalize(q);
    SYNC();
    if (lane < 4) S.qpos[3 + lane] = (lane == 0) ? q[0] : (lane == 1) ? q[1] : (lane == 2) ? q[2] : q[3];
    if (lane >= 6 && lane < DMK_NV) {
      float sn, cs;
      sincosf(0.5f * S.qpos[lane + 1], &sn, &cs);
      S.cs[lane][0] = cs; S.cs[lane][1] = sn;
    }
    SYNC();
  }
  PROF2(12);
  // ---- P1 (lane = body): rotation of the body relative to its parent (product of its hinge
  // quaternions) and each hinge axis expressed in the parent-body frame
  if (isbody) {
    float q[4] = {1.f, 0.f, 0.f, 0.f};
    if (lb == 1) {
      for (int i = 0; i < 4; i++) q[i] = S.qpos[3 + i];
    } else {
      {  // first hinge: q is still the identity (spelled out: without fast-math hipcc keeps the 0 * x products)
        const int k = b_dofadr;
        const float al[3] = {T.d_axis[k][0], T.d_axis[k][1], T.d_axis[k][2]};
        S.xaxis[k][0] = al[0]; S.xaxis[k][1] = al[1]; S.xaxis[k][2] = al[2];       // parent-frame axis for now
        const float c = S.cs[k][0], sn = S.cs[k][1];
        q[0] = c; q[1] = al[0] * sn; q[2] = al[1] * sn; q[3] = al[2] * sn;
      }
#pragma unroll
      for (int j = 1; j < 3; j++) {
        if (j < b_dofnum) {
          const int k = b_dofadr + j;
          const float al[3] = {T.d_axis[k][0], T.d_axis[k][1], T.d_axis[k][2]};
          float ax[3], ql[4], qn[4];
          quat_rot(ax, q, al);
          S.xaxis[k][0] = ax[0]; S.xaxis[k][1] = ax[1]; S.xaxis[k][2] = ax[2];   // parent-frame axis for now
          const float c = S.cs[k][0], sn = S.cs[k][1];
          ql[0] = c; ql[1] = al[0] * sn; ql[2] = al[1] * sn; ql[3] = al[2] * sn;
          quat_mul(qn, q, ql);
          q[0] = qn[0]; q[1] = qn[1]; q[2] = qn[2]; q[3] = qn[3];
        }
      }
    }
    for (int i = 0; i < 4; i++) S.u.v.qloc[lb][i] = q[i];
    for (int i = 0; i < 3; i++) S.u.v.qloc[lb][4 + i] = (lb == 1) ? S.qpos[i] : T.b_pos[lb][i];
  }
  SYNC();
  // ---- P2 (lane = body): compose along the ancestor chain root -> self
  if (isbody) {
    // the chain always starts at the root body (byte 0 of b_chainb = 1): start from its local pose
    float q[4], pos[3];
    for (int i = 0; i < 4; i++) q[i] = S.u.v.qloc[1][i];
    for (int i = 0; i < 3; i++) pos[i] = S.u.v.qloc[1][4 + i];
#pragma unroll
    for (int c = 1; c < 4; c++) {
      const int cb = (b_chain4 >> (8 * c)) & 0xFF;
      if (cb != 0) {
        float ql[4], pl[3], t[3], qn[4];
        for (int i = 0; i < 4; i++) ql[i] = S.u.v.qloc[cb][i];
        for (int i = 0; i < 3; i++) pl[i] = S.u.v.qloc[cb][4 + i];
        quat_rot(t, q, pl);
        pos[0] += t[0]; pos[1] += t[1]; pos[2] += t[2];
        quat_mul(qn, q, ql);
        q[0] = qn[0]; q[1] = qn[1]; q[2] = qn[2]; q[3] = qn[3];
      }
    }
    quat_normalize(q);
    float m9[9], tv[3];
    quat2mat(m9, q);
    const float ip[3] = {T.b_ipos[lb][0], T.b_ipos[lb][1], T.b_ipos[lb][2]};
    mat_vec(tv, m9, ip);
    for (int i = 0; i < 3; i++) { S.xpos[lb][i] = pos[i]; S.xipos[lb][i] = pos[i] + tv[i]; }
    for (int i = 0; i < 4; i++) S.xquat[lb][i] = q[i];
    for (int i = 0; i < 9; i++) S.xmat[lb][i] = m9[i];
  }
  SYNC();
  PROF2(13);
  // ---- P3: world joint axes (lane = dof), geom poses (lane = geom), whole-body COM
  float com[3];
  {
    float ax[3] = {0.f, 0.f, 0.f};
    if (lane >= 6 && isdof) {
      const float pq[4] = {S.xquat[d_pbody][0], S.xquat[d_pbody][1], S.xquat[d_pbody][2], S.xquat[d_pbody][3]};
      const float al[3] = {S.xaxis[lk][0], S.xaxis[lk][1], S.xaxis[lk][2]};
      quat_rot(ax, pq, al);
    } else if (lane >= 3 && lane < 6) {
      ax[0] = S.xmat[1][lane - 3]; ax[1] = S.xmat[1][3 + lane - 3]; ax[2] = S.xmat[1][6 + lane - 3];
    }
    if (lane < DMK_NG) {
      const int gb = T.g_body[lg];
      float bm[9], tv[3], gl[3] = {T.g_pos[lg][0], T.g_pos[lg][1], T.g_pos[lg][2]}, gm[9];
      for (int i = 0; i < 9; i++) { bm[i] = S.xmat[gb][i]; gm[i] = T.g_mat[lg][i]; }
      mat_vec(tv, bm, gl);
      for (int i = 0; i < 3; i++) S.gpos[lg][i] = S.xpos[gb][i] + tv[i];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
          S.gmat[lg][3 * i + j] = bm[3 * i] * gm[j] + bm[3 * i + 1] * gm[3 + j] + bm[3 * i + 2] * gm[6 + j];
    }
    const float xi[3] = {S.xipos[lb][0], S.xipos[lb][1], S.xipos[lb][2]};
    for (int i = 0; i < 3; i++) com[i] = wave_sum(bmass * xi[i]) * mtot_inv;
    if (lane == 0) { S.com[0] = com[0]; S.com[1] = com[1]; S.com[2] = com[2]; }
    SYNC();  // every lane has read the parent-frame axes
    // cdof (lane = dof), COM-based spatial frame
    if (isdof) {
      float cd[6] = {0, 0, 0, 0, 0, 0};
      if (lane < 3) {
        cd[3 + lane] = 1.f;
      } else {
        const float off[3] = {com[0] - S.xpos[d_body][0], com[1] - S.xpos[d_body][1], com[2] - S.xpos[d_body][2]};
        cd[0] = ax[0]; cd[1] = ax[1]; cd[2] = ax[2];
        cross3(cd + 3, ax, off);
        S.xaxis[lk][0] = ax[0]; S.xaxis[lk][1] = ax[1]; S.xaxis[lk][2] = ax[2];
      }
      for (int i = 0; i < 6; i++) S.cdof[lk][i] = cd[i];
      S.cdof[lk][6] = 0; S.cdof[lk][7] = 0;
    }
  }
  PROF2(14);
  // ---- cinert (lane = body)
  if (isbody) {
    float R[9], Tm[9], W[6], bi[6];
    for (int i = 0; i < 9; i++) R[i] = S.xmat[lb][i];
    for (int i = 0; i < 6; i++) bi[i] = T.b_inertia[lb][i];
    const float Ib[9] = {bi[0], bi[3], bi[4], bi[3], bi[1], bi[5], bi[4], bi[5], bi[2]};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Tm[3 * i + j] = R[3 * i] * Ib[j] + R[3 * i + 1] * Ib[3 + j] + R[3 * i + 2] * Ib[6 + j];
    W[0] = Tm[0] * R[0] + Tm[1] * R[1] + Tm[2] * R[2];
    W[1] = Tm[3] * R[3] + Tm[4] * R[4] + Tm[5] * R[5];
    W[2] = Tm[6] * R[6] + Tm[7] * R[7] + Tm[8] * R[8];
    W[3] = Tm[0] * R[3] + Tm[1] * R[4] + Tm[2] * R[5];
    W[4] = Tm[0] * R[6] + Tm[1] * R[7] + Tm[2] * R[8];
    W[5] = Tm[3] * R[6] + Tm[4] * R[7] + Tm[5] * R[8];
    const float o[3] = {S.xipos[lb][0] - com[0], S.xipos[lb][1] - com[1], S.xipos[lb][2] - com[2]};
    const float oo = dot3(o, o);
    float *ci = S.cinert[lb];
    ci[0] = W[0] + bmass * (oo - o[0] * o[0]);
    ci[1] = W[1] + bmass * (oo - o[1] * o[1]);
    ci[2] = W[2] + bmass * (oo - o[2] * o[2]);
    ci[3] = W[3] - bmass * o[0] * o[1];
    ci[4] = W[4] - bmass * o[0] * o[2];
    ci[5] = W[5] - bmass * o[1] * o[2];
    ci[6] = bmass * o[0]; ci[7] = bmass * o[1]; ci[8] = bmass * o[2];
    ci[9] = bmass;
  }
  SYNC();
  PROF(1);
  }  // !DAMP2
  // ---- composite inertia per body via the subtree mask (lane = body)
  {
    // body c adds its cinert to c and to every ancestor of c: a compile-time lane set (dm_topology.h)
    float acc[10];
    for (int i = 0; i < 10; i++) acc[i] = 0;
    StaticFor<1, DMK_NB>::run([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      const float w = lane_sel<topo::body_ancself_mask(c)>(1.f, 0.f);
#pragma unroll
      for (int i = 0; i < 10; i++) acc[i] = fmaf(w, g_S.cinert[c][i], acc[i]);
      return true;
    });
    if (isbody)
      for (int i = 0; i < 10; i++) S.u.v.crb[lb][i] = acc[i];
  }
  SYNC();
  // ---- M by COLUMNS: lane j keeps C[k] = M[k][j] for every dof k of its subtree (0 elsewhere).  Lane k publishes
  // buf_k = I_crb(body(k)) cdof_k; every lane dots its own cdof with each buf_k (uniform LDS reads), the static lane
  // set "ancestor-or-self of k" masks the result, the armature goes on the diagonal.  All register indices are static.
  float C[DMK_NV];
  {
    float cd[6] = {0, 0, 0, 0, 0, 0};
    if (isdof) {
      float buf[6], I[10];
      for (int i = 0; i < 10; i++) I[i] = S.u.v.crb[d_body][i];
      for (int i = 0; i < 6; i++) cd[i] = S.cdof[lk][i];
      mul_inert_vec(buf, I, cd);
      for (int i = 0; i < 6; i++) S.u.v.mbuf[lk][i] = buf[i];
    }
    SYNC();
    const float armv = isdof ? (DAMP2 ? fmaf(T.timestep, T.d_damp[lk], T.d_arm[lk]) : T.d_arm[lk]) : 0.f;
    // dots[k] = cdof_lane . buf_k.  k < 32 on the matrix pipe (32 x 32 x 6: operand A = buf from LDS, operand B = the
    // lanes' cdof with the odd components swapped into the upper half-wave; dof lanes 32, 33 only own rows 32, 33);
    // k = 32, 33 with FMAs.
    float dots[DMK_NV];
    {
      const float *bufp = &S.u.v.mbuf[lane & 31][lane >> 5];
      mfma_f16v acc;
#pragma unroll
      for (int v = 0; v < 16; v++) acc[v] = 0.f;
#pragma unroll
      for (int p = 0; p < 3; p++) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(cd[2 * p]), __float_as_uint(cd[2 * p + 1]), false, false);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bufp[2 * p], __uint_as_float(sw[0]), acc, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < 16; v++) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[v]), __float_as_uint(acc[v]), false, false);
        dots[(v / 4) * 8 + (v % 4)] = acc[v];
        dots[(v / 4) * 8 + (v % 4) + 4] = __uint_as_float(sw[1]);
      }
#pragma unroll
      for (int k = 32; k < DMK_NV; k++) {
        float d = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) d = fmaf(cd[i], S.u.v.mbuf[k][i], d);
        dots[k] = d;
      }
    }
    StaticFor<0, DMK_NV>::run([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const float off = lane_sel<topo::anc_mask(k)>(dots[k], 0.f);             // strict ancestors of k keep M[k][j]
      C[k] = lane_sel<(1ull << k)>(dots[k] + armv, off);                       // lane k: diagonal + armature
      return true;
    });
  }
  PROF(2);
  // ---- L^T D L on the columns: eliminate dof k = 33..1.  M[k][k] and the multipliers M[k][i] / M[k][k] of the
  // ancestors i of k are broadcast from static lanes with v_readlane; every lane applies
  // C[i] -= C[k] * M[k][i] / M[k][k] to the registers of the ancestors i — lanes outside the ancestor set of k hold
  // C[k] = 0, so no predicate is needed.  2 VALU per (k, ancestor) pair, 276 pairs.
  StaticFor<0, DMK_NV - 1>::run([&](auto ic) {
    constexpr int k = DMK_NV - 1 - decltype(ic)::value;   // 33 .. 1
    const float rk = __builtin_amdgcn_rcpf(rl(C[k], k));
    const float Cs = C[k] * rk;
    ElimAnc<k, topo::PARENT[k]>::run(C, Cs);
    return true;
  });
  // publish the factor in the sparse MuJoCo layout (row k: M(k,k), M(k,parent), ...), rows UNSCALED:
  // L[i][j] = M[i][j] * dinv[i] is applied by the users.  Lane j owns M[k][j] at S.M[MADR[k] + NANC[k] - depth_j];
  // lanes outside the ancestor set of k write to a spare slot.
  const int d_nanc_s = isdof ? d_nanc : 0;
  {
    StaticFor<0, DMK_NV>::run([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const int real = topo::MADR[k] + topo::NANC[k] - d_nanc_s;
      const int idx = lane_sel_i<(topo::anc_mask(k) | (1ull << k))>(real, DM_NM + 1);
      g_S.M[idx] = C[k];
      return true;
    });
  }
  SYNC();
  float dv = 0.f;
  const int mrow = T.d_madr[lk] + d_nanc;          // S.M[mrow - depth(j)] = M[lane][j]
  const uint64_t ancm = T.d_ancm[lk];              // bit j: dof j is a strict ancestor of this lane's dof
  if (isdof) {
    const float Md = S.M[T.d_madr[lk]];
    dv = 1.0f / Md;
    S.dinv[lk] = dv;
    S.dsqrtinv[lk] = 1.0f / sqrtf(Md);
  }
  SYNC();
  PROF(3);
  float xs = 0;
  if constexpr (DAMP2) {
    xs = isdof ? rhs : 0.f;
  } else {
  // ---- velocity stage, lane-parallel.  Pass A (lane = body): w_b = sum over the body's dofs of cdof * qvel
  if (isbody) {
    float w[6] = {0, 0, 0, 0, 0, 0};
    const int nd = (lb == 1) ? 6 : b_dofnum;
    for (int j = 0; j < 6; j++)
      if (j < nd) {
        const int k = b_dofadr + j;
        const float qv = S.qvel[k];
        for (int i = 0; i < 6; i++) w[i] += S.cdof[k][i] * qv;
      }
    for (int i = 0; i < 6; i++) S.u.v.cacc[lb][i] = w[i];       // (scratch: w_b)
  }
  SYNC();
  // Pass B (lane = body): velocity of the parent, then the body's dofs in order: cdof_dot, u_b = sum cdof_dot * qvel
  if (isbody) {
    float cv[6] = {0, 0, 0, 0, 0, 0}, u[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; c++) {       // ancestors only (the chain has at most 3 proper ancestors)
      const int cb = (b_chain4 >> (8 * c)) & 0xFF;
      if (cb != 0 && cb != lb)
        for (int i = 0; i < 6; i++) cv[i] += S.u.v.cacc[cb][i];
    }
    if (lb == 1) {
      for (int k = 0; k < 3; k++) {
        const float qv = S.qvel[k];
        for (int i = 0; i < 6; i++) cv[i] += S.cdof[k][i] * qv;
      }
      float dd[3][6];
#pragma unroll
      for (int k = 3; k < 6; k++) {
        float cd[6];
        for (int i = 0; i < 6; i++) cd[i] = S.cdof[k][i];
        cross_motion(dd[k - 3], cv, cd);
      }
#pragma unroll
      for (int k = 3; k < 6; k++) {
        const float qv = S.qvel[k];
        for (int i = 0; i < 6; i++) { u[i] += dd[k - 3][i] * qv; cv[i] += S.cdof[k][i] * qv; }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 3; j++) {
        if (j < b_dofnum) {
          const int k = b_dofadr + j;
          float cd[6], dd[6];
          const float qv = S.qvel[k];
          for (int i = 0; i < 6; i++) cd[i] = S.cdof[k][i];
          cross_motion(dd, cv, cd);
          for (int i = 0; i < 6; i++) { u[i] += dd[i] * qv; cv[i] += cd[i] * qv; }
        }
      }
    }
    for (int i = 0; i < 6; i++) { S.cvel[lb][i] = cv[i]; S.u.v.cfrcsub[lb][i] = u[i]; }   // (scratch: u_b)
  }
  SYNC();
  PROF2(15);
  // Pass C (lane = body): cacc = -gravity + sum over the chain of u, cfrc = I cacc + cvel x* (I cvel)
  if (isbody) {
    float ca[6] = {0, 0, 0, -T.gravity[0], -T.gravity[1], -T.gravity[2]}, cv[6], I[10], f0[6], t0[6], t1[6];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int cb = (b_chain4 >> (8 * c)) & 0xFF;
      if (cb != 0)
        for (int i = 0; i < 6; i++) ca[i] += S.u.v.cfrcsub[cb][i];
    }
    for (int i = 0; i < 6; i++) cv[i] = S.cvel[lb][i];
    for (int i = 0; i < 10; i++) I[i] = S.cinert[lb][i];
    mul_inert_vec(f0, I, ca);
    mul_inert_vec(t0, I, cv);
    cross_force(t1, cv, t0);
    for (int i = 0; i < 6; i++) S.u.v.cfrc[lb][i] = f0[i] + t1[i];
  }
  SYNC();
  // Pass D (lane = body): subtree sums of cfrc
  {
    float acc[6] = {0, 0, 0, 0, 0, 0};
    StaticFor<1, DMK_NB>::run([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      const float w = lane_sel<topo::body_ancself_mask(c)>(1.f, 0.f);
#pragma unroll
      for (int i = 0; i < 6; i++) acc[i] = fmaf(w, g_S.u.v.cfrc[c][i], acc[i]);
      return true;
    });
    if (isbody)
      for (int i = 0; i < 6; i++) S.u.v.cacc[lb][i] = acc[i];      // (scratch reuse: subtree force)
  }
  SYNC();
  PROF2(10);
  // ---- smooth forces and qacc_smooth = M^-1 (passive - bias + actuation)   (lane = dof)
  if (isdof) {
    float bias = 0;
    for (int i = 0; i < 6; i++) bias += S.cdof[lk][i] * S.u.v.cacc[d_body][i];
    float act = 0;
    const int da = T.d_act[lk];
    if (da >= 0) act = T.d_gear[lk] * clampf(S.ctrl[da], T.d_clo[lk], T.d_chi[lk]);
    xs = -T.d_damp[lk] * S.qvel[lk] - bias + act;
  }
  }  // !DAMP2
  xs = solve_LT(xs, dv, -d_nanc_s, g_S.M);
  xs = solve_L(xs, dv, isdof ? mrow : DMK_MAXANC, g_S.M);
  xs *= dv;
  if constexpr (!DAMP2) {
    if (isdof) S.qacc_smooth[lk] = xs;
  }
  SYNC();
  PROF(4);
  return xs;
}

// Euler integrator only: out of line, so that the RK4 path's register allocation is the one it had without the option
__device__ __noinline__ float euler_damped_solve(GDev &T, const int lane, const float rhs) {
  return fwd_smooth<true>(T, lane, rhs);
}

// ---- forward evaluation, part 2: collision detection -> contact list in LDS; returns ncon | overflow << 8
__device__ __forceinline__ int fwd_collide(GDev &T, const int lane) {
  EnvLds &S = g_S;
  int ncon = 0, overflow = 0;
  const int lb = lane < DMK_NB ? lane : 0;       // lane as body
  const int lk = lane < DMK_NV ? lane : 0;       // lane as dof
  const int lg = lane < DMK_NG ? lane : 0;       // lane as geom
  const int bdep = (lane < DMK_NB) ? T.b_depth[lb] : -1;
  const float bmass = (lane < DMK_NB) ? T.b_mass[lb] : 0.f;
  const float mtot_inv = T.total_mass_inv;
  (void)lg; (void)bdep; (void)bmass; (void)mtot_inv; (void)lb; (void)lk;
    // ---- collision: lane = candidate pair, two rounds in canonical order
    int base = 0;
    overflow = 0;
    for (int round = 0; round < 2; round++) {
      const int p = round * 64 + lane;
      Cand c;
      c.valid = 0;
      c.t[0] = c.t[1] = c.t[2] = 0;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        c.d[s] = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) { c.p[s][i] = 0; c.n[s][i] = 0; }
      }
      int g1 = 0, g2 = 0;
      bool isbb = false;
      float margin = 0;
      // per-pair constants come as one contiguous 12-word record (host-precomputed)
      bool act = false;
      int t1 = 0, t2 = 0;
      float x1[3], x2[3], z1[3], z2[3];
      if (p < T.npair) {
        GPair &pr = T.pairs[p];
        g1 = pr.g1; g2 = pr.g2; t1 = pr.t1; t2 = pr.t2;
        margin = pr.margin;
        for (int i = 0; i < 3; i++) { x1[i] = S.gpos[g1][i]; x2[i] = S.gpos[g2][i]; z1[i] = pr.z1[i]; z2[i] = pr.z2[i]; }
        const float df[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
        if (t1 == DM_GEOM_PLANE) {
          const float pn[3] = {S.gmat[g1][2], S.gmat[g1][5], S.gmat[g1][8]};
          act = dot3(df, pn) - pr.rbsum <= margin;
        } else {
          act = !(dot3(df, df) > (pr.rbsum + margin) * (pr.rbsum + margin));
        }
      }
      if (!__any(act)) continue;   // nothing near in this round (wave-uniform): no narrowphase, no compaction
      if (act) {
        float M1[9], M2[9];
        for (int i = 0; i < 9; i++) { M1[i] = S.gmat[g1][i]; M2[i] = S.gmat[g2][i]; }
        if (act) {
          if (t1 == DM_GEOM_PLANE) {
            float pn[3] = {M1[2], M1[5], M1[8]};
            if (t2 == DM_GEOM_SPHERE) {
              np_plane_sphere<0>(c, margin, x1, pn, x2, z2[0]);
            } else if (t2 == DM_GEOM_CAPSULE) {
              float ax[3] = {M2[2], M2[5], M2[8]}, e[3];
              for (int i = 0; i < 3; i++) e[i] = x2[i] + ax[i] * z2[1];
              np_plane_sphere<0>(c, margin, x1, pn, e, z2[0]);
              for (int i = 0; i < 3; i++) e[i] = x2[i] - ax[i] * z2[1];
              np_plane_sphere<1>(c, margin, x1, pn, e, z2[0]);
              c.t[0] = ax[0]; c.t[1] = ax[1]; c.t[2] = ax[2];
            } else {  // box: the (at most 4) corners below the centre and within margin, in corner order
              // corner i = x2 + s0 A0 + s1 A1 + s2 A2 (A_j = half-size_j * column j of the box frame, s_j = +-1 from
              // bit j of i), so its height over the centre is +-a0 +-a1 +-a2 with a_j = pn . A_j: the eight tests cost
              // two additions each and only the (at most four) accepted corners are built
              const float df[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
              const float dist = dot3(df, pn);
              float A[3][3], a[3];
#pragma unroll
              for (int j = 0; j < 3; j++) {
                A[j][0] = M2[j] * z2[j]; A[j][1] = M2[3 + j] * z2[j]; A[j][2] = M2[6 + j] * z2[j];
                a[j] = pn[0] * A[j][0] + pn[1] * A[j][1] + pn[2] * A[j][2];
              }
              unsigned okm = 0;
#pragma unroll
              for (int i = 0; i < 8; i++) {
                const float ld = ((i & 1) ? a[0] : -a[0]) + ((i & 2) ? a[1] : -a[1]) + ((i & 4) ? a[2] : -a[2]);
                okm |= (!(dist + ld > margin || ld > 0)) ? (1u << i) : 0u;
              }
#pragma unroll
              for (int s = 0; s < 4; s++) {
                if (okm) {
                  const int i = __ffs(okm) - 1;
                  okm &= okm - 1;
                  const float s0 = (i & 1) ? 1.f : -1.f, s1 = (i & 2) ? 1.f : -1.f, s2 = (i & 4) ? 1.f : -1.f;
                  const float corner[3] = {s0 * A[0][0] + s1 * A[1][0] + s2 * A[2][0], s0 * A[0][1] + s1 * A[1][1] + s2 * A[2][1],
                                           s0 * A[0][2] + s1 * A[1][2] + s2 * A[2][2]};
                  const float cdist = dist + dot3(pn, corner);
                  c.d[s] = cdist;
                  for (int q = 0; q < 3; q++) { c.p[s][q] = corner[q] + x2[q] - pn[q] * 0.5f * cdist; c.n[s][q] = pn[q]; }
                  c.valid |= 1 << s;
                }
              }
            }
          } else if (t2 != DM_GEOM_BOX) {  // sphere/capsule vs sphere/capsule
            float r1 = z1[0], r2 = z2[0];
            if (t1 == DM_GEOM_SPHERE && t2 == DM_GEOM_SPHERE) {
              np_sphere_sphere<0>(c, margin, x1, r1, x2, r2);
            } else if (t1 == DM_GEOM_SPHERE) {  // sphere - capsule
              float ax[3] = {M2[2], M2[5], M2[8]}, v[3] = {x1[0] - x2[0], x1[1] - x2[1], x1[2] - x2[2]}, pt[3];
              float x = clampf(dot3(ax, v), -z2[1], z2[1]);
              for (int i = 0; i < 3; i++) pt[i] = x2[i] + ax[i] * x;
              np_sphere_sphere<0>(c, margin, x1, r1, pt, r2);
            } else {  // capsule - capsule
              float a1[3] = {M1[2], M1[5], M1[8]}, a2[3] = {M2[2], M2[5], M2[8]};
              float df[3] = {x1[0] - x2[0], x1[1] - x2[1], x1[2] - x2[2]};
              float ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2);
              float u = -dot3(a1, df), v = dot3(a2, df);
              float det = ma * mc - mb * mb;
              float v1[3], v2[3];
              if (fabsf(det) >= MINVALF) {
                float xx1 = (mc * u - mb * v) / det, xx2 = (ma * v - mb * u) / det;
                if (xx1 > z1[1]) { xx1 = z1[1]; xx2 = (v - mb * z1[1]) / mc; }
                else if (xx1 < -z1[1]) { xx1 = -z1[1]; xx2 = (v + mb * z1[1]) / mc; }
                if (xx2 > z2[1]) { xx2 = z2[1]; xx1 = clampf((u - mb * z2[1]) / ma, -z1[1], z1[1]); }
                else if (xx2 < -z2[1]) { xx2 = -z2[1]; xx1 = clampf((u + mb * z2[1]) / ma, -z1[1], z1[1]); }
                for (int i = 0; i < 3; i++) { v1[i] = x1[i] + a1[i] * xx1; v2[i] = x2[i] + a2[i] * xx2; }
                np_sphere_sphere<0>(c, margin, v1, r1, v2, r2);
              } else {  // parallel axes: ends of capsule 1, then of capsule 2, at most 2 contacts
                Cand e;
                e.valid = 0;
                for (int i = 0; i < 3; i++) v1[i] = x1[i] + a1[i] * z1[1];
                float d2[3] = {v1[0] - x2[0], v1[1] - x2[1], v1[2] - x2[2]};
                float xx2 = clampf(dot3(d2, a2), -z2[1], z2[1]);
                for (int i = 0; i < 3; i++) v2[i] = x2[i] + a2[i] * xx2;
                np_sphere_sphere<0>(e, margin, v1, r1, v2, r2);
                for (int i = 0; i < 3; i++) v1[i] = x1[i] - a1[i] * z1[1];
                for (int i = 0; i < 3; i++) d2[i] = v1[i] - x2[i];
                xx2 = clampf(dot3(d2, a2), -z2[1], z2[1]);
                for (int i = 0; i < 3; i++) v2[i] = x2[i] + a2[i] * xx2;
                np_sphere_sphere<1>(e, margin, v1, r1, v2, r2);
                for (int i = 0; i < 3; i++) v2[i] = x2[i] + a2[i] * z2[1];
                float d1[3] = {v2[0] - x1[0], v2[1] - x1[1], v2[2] - x1[2]};
                float xx1 = clampf(dot3(d1, a1), -z1[1], z1[1]);
                for (int i = 0; i < 3; i++) v1[i] = x1[i] + a1[i] * xx1;
                np_sphere_sphere<2>(e, margin, v1, r1, v2, r2);
                for (int i = 0; i < 3; i++) v2[i] = x2[i] - a2[i] * z2[1];
                for (int i = 0; i < 3; i++) d1[i] = v2[i] - x1[i];
                xx1 = clampf(dot3(d1, a1), -z1[1], z1[1]);
                for (int i = 0; i < 3; i++) v1[i] = x1[i] + a1[i] * xx1;
                np_sphere_sphere<3>(e, margin, v1, r1, v2, r2);
                int cnt = 0;  // keep the first two, in test order
#pragma unroll
                for (int s = 0; s < 4; s++) {
                  bool ok = ((e.valid >> s) & 1) && cnt < 2;
#pragma unroll
                  for (int o = 0; o < 2; o++)
                    if (ok && cnt == o) {
                      c.d[o] = e.d[s];
                      for (int q = 0; q < 3; q++) { c.p[o][q] = e.p[s][q]; c.n[o][q] = e.n[s][q]; }
                      c.valid |= 1 << o;
                    }
                  cnt += ok ? 1 : 0;
                }
              }
            }
          } else if (t1 == DM_GEOM_SPHERE) {
            np_sphere_box<0>(c, margin, x1, z1[0], x2, M2, z2);
          } else if (t1 == DM_GEOM_CAPSULE) {  // capsule - box (see oracle c_capsule_box)
            float axw[3] = {M1[2], M1[5], M1[8]}, tt[3] = {x1[0] - x2[0], x1[1] - x2[1], x1[2] - x2[2]}, pp[3], aa[3];
            float hl = z1[1], r = z1[0];
            mat_t_vec(pp, M2, tt);
            mat_t_vec(aa, M2, axw);
            float lo = -hl, hi = hl, ts;
            if (cb_grad(pp, aa, z2, lo) >= 0) ts = lo;
            else if (cb_grad(pp, aa, z2, hi) <= 0) ts = hi;
            else {
              for (int q = 0; q < 26; q++) {
                float mid = 0.5f * (lo + hi);
                if (cb_grad(pp, aa, z2, mid) < 0) lo = mid; else hi = mid;
              }
              ts = 0.5f * (lo + hi);
            }
            float pt[3];
            for (int i = 0; i < 3; i++) pt[i] = x1[i] + axw[i] * ts;
            np_sphere_box<0>(c, margin, pt, r, x2, M2, z2);
            float te = (ts > 0) ? -hl : hl;
            if (fabsf(te - ts) > 1e-3f * hl) {
              for (int i = 0; i < 3; i++) pt[i] = x1[i] + axw[i] * te;
              np_sphere_box<1>(c, margin, pt, r, x2, M2, z2);
            }
          } else {
            isbb = true;
          }
        }
        if (isbb) {
          np_box_box(S, &T.pairs[p]);
          int nc = S.u.bb.ncand;
#pragma unroll
          for (int s = 0; s < 4; s++)
            if (s < nc) {
              c.valid |= 1 << s;
              c.d[s] = S.u.bb.cand[s][0];
              for (int q = 0; q < 3; q++) { c.p[s][q] = S.u.bb.cand[s][1 + q]; c.n[s][q] = S.u.bb.cand[s][4 + q]; }
            }
        }
      }
      // compaction in canonical order: exclusive prefix of the per-lane contact counts
      int cnt = __popc(c.valid);
      unsigned long long lt = lanemask_lt(lane);
      unsigned long long b0 = __ballot(cnt & 1), b1 = __ballot(cnt & 2), b2 = __ballot(cnt & 4);
      int off = __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt);
      int total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
      if (total == 0) continue;
      int w = base + off;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        if ((c.valid >> s) & 1) {
          if (w < DMK_MAXCON) {
            float fr[9] = {c.n[s][0], c.n[s][1], c.n[s][2], c.t[0], c.t[1], c.t[2], 0, 0, 0};
            make_frame(fr);
            S.c_dist[w] = c.d[s];
            for (int q = 0; q < 3; q++) S.c_pos[w][q] = c.p[s][q];
            for (int q = 0; q < 9; q++) S.c_frame[w][q] = fr[q];
            S.c_g1[w] = g1; S.c_g2[w] = g2;
          }
          w++;
        }
      }
      base += total;
    }
    if (base > DMK_MAXCON) { overflow |= 1; base = DMK_MAXCON; }
    ncon = base;
    SYNC();

    PROF(5);
  return ncon | (overflow << 8);
}

// ---- one constraint row (limit or pyramid edge / frictionless contact) for the calling lane:
// Jacobian row -> reference acceleration, regulariser -> B row = D^-1/2 L^-T J^T (returned in J)
template <bool MF>   // MF: all rows live in lanes 0..31 (nefc <= 32): the dof-by-row products run on the matrix pipe
__device__ __forceinline__ void build_row(GDev &T, EnvLds &S, const int r, const int nefc, const float (&com)[3],
                                          float (&J)[DMK_NV], float &R, float &Dd, float &aref, float &bb, float &jw) {
      float rpos = 0, rmargin = 0, rdiag = 1, mu = 0;
      int rtype = -1, full = 0;
      float wl[3] = {0, 0, 0}, wa[3] = {0, 0, 0};
      unsigned long long cm1 = 0, cm2 = 0;
      int ldof = -1;
      float lsign = 0;
      if (r < nefc) {
        int info = S.rowinfo[r];
        if (info < 0) {
          int v = -info - 1;
          ldof = v >> 1;
          lsign = (v & 1) ? -1.f : 1.f;
          rtype = 0;
          float q = S.qpos[ldof + 1];
          rpos = (v & 1) ? (T.d_hi[ldof] - q) : (q - T.d_lo[ldof]);
          rmargin = 0;
          rdiag = T.d_invw[ldof];
        } else {
          int ci = (info >> 3) & 0x3F, e = info & 7;
          full = (info & 0x4000) ? 1 : 0;
          int g1 = S.c_g1[ci], g2 = S.c_g2[ci];
          int b1 = T.g_body[g1], b2 = T.g_body[g2];
          int cd1 = T.g_condim[g1], cd2 = T.g_condim[g2];
          int dim = cd1 > cd2 ? cd1 : cd2;
          mu = fmaxf(T.g_mu[g1], T.g_mu[g2]);
          float fr[9];
          for (int i = 0; i < 9; i++) fr[i] = S.c_frame[ci][i];
          float tran = T.b_invw[b1] + T.b_invw[b2];
          if (dim < 3) {
            rtype = 1;
            wl[0] = fr[0]; wl[1] = fr[1]; wl[2] = fr[2];
            rdiag = tran;
          } else {
            rtype = 2;
            const bool second = (e >> 1) != 0;   // tangent 2 for edges 2,3
            const float tx = second ? fr[6] : fr[3], ty = second ? fr[7] : fr[4], tz = second ? fr[8] : fr[5];
            float sg = (e & 1) ? -mu : mu;
            wl[0] = fr[0] + sg * tx; wl[1] = fr[1] + sg * ty; wl[2] = fr[2] + sg * tz;
            rdiag = tran + mu * mu * tran;
          }
          float off[3] = {S.c_pos[ci][0] - com[0], S.c_pos[ci][1] - com[1], S.c_pos[ci][2] - com[2]};
          cross3(wa, off, wl);
          cm1 = T.b_chain[b1]; cm2 = T.b_chain[b2];
          rpos = S.c_dist[ci];
          rmargin = fmaxf(T.g_margin[g1], T.g_margin[g2]);
        }
      }
      float vel = 0, jqs = 0;
      jw = 0;
      float vals[32];
      if (MF) {
        // val[k][row] = cdof_k . (wa, wl) for 32 dofs x 32 rows: three v_mfma_f32_32x32x2f32 (K = 6).  Operand A: lane l
        // reads cdof[l % 32][k0 + l / 32] from LDS; operand B: this lane's weights with the k0 + 1 component of the
        // rows swapped into the upper half-wave.  The result fragment is completed with one swap per register.
        const float wv[6] = {wa[0], wa[1], wa[2], wl[0], wl[1], wl[2]};
        const int lane_ = (int)(threadIdx.x & 63);
        const float *cdp = &S.cdof[lane_ & 31][lane_ >> 5];
        mfma_f16v acc;
#pragma unroll
        for (int v = 0; v < 16; v++) acc[v] = 0.f;
#pragma unroll
        for (int p = 0; p < 3; p++) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(wv[2 * p]), __float_as_uint(wv[2 * p + 1]), false, false);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cdp[2 * p], __uint_as_float(sw[0]), acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 16; v++) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[v]), __float_as_uint(acc[v]), false, false);
          vals[(v / 4) * 8 + (v % 4)] = acc[v];
          vals[(v / 4) * 8 + (v % 4) + 4] = __uint_as_float(sw[1]);
        }
      }
#pragma unroll
      for (int k = 0; k < DMK_NV; k++) {
        float val;
        if (MF && k < 32) {
          val = vals[k];
        } else {
          const float4 ca = *reinterpret_cast<const float4 *>(&S.cdof[k][0]);
          const float2 cb = *reinterpret_cast<const float2 *>(&S.cdof[k][4]);
          val = ca.x * wa[0] + ca.y * wa[1] + ca.z * wa[2] + ca.w * wl[0] + cb.x * wl[1] + cb.y * wl[2];
        }
        float sg = (float)((int)((cm2 >> k) & 1ull) - (int)((cm1 >> k) & 1ull));   // 0 for limit / unused rows (no chains)
        float j = (k == ldof) ? lsign : sg * val;
        J[k] = j;
        vel += j * S.qvel[k];
        jqs += j * S.qacc_smooth[k];
        jw += j * S.warm[k];
      }
      R = 1; Dd = 0; aref = 0; bb = 0;
      if (r < nefc) {
        float sol[5] = {T.solimp[0], T.solimp[1], T.solimp[2], T.solimp[3], T.solimp[4]};
        float imp = impedance(sol, rpos, rmargin);
        R = fmaxf(MINVALF, (1 - imp) * rdiag / imp);
        if (rtype == 2 && full) R = 2 * mu * mu * R;
        aref = -T.B * vel - T.K * imp * (rpos - rmargin);
        bb = jqs - aref;
        Dd = 1.0f / R;
      }
      // ---- B row = D^-1/2 L^-T J^T, in place (lane = row)
      {
        // x <- L^-T x on the row held by this lane: only the 276 ancestor pairs of the dof tree, all
        // register and LDS indices static (dm_topology.h); factor entries are wave-uniform broadcasts.
        // Step i's loads are tied to step i+1's result so they are not all hoisted ahead of the FMAs.
        lds_cfloat_p Mp = (lds_cfloat_p)S.M;
#pragma unroll
        for (int i = DMK_NV - 1; i >= 1; i--) {
          const float xi = J[i] * S.dinv[i];
          asm volatile("" : "+v"(Mp), "+v"(J[0]));
#pragma unroll
          for (int j = 0; j < i; j++)
            if (topo::is_anc(j, i)) J[j] -= Mp[topo::midx(i, j)] * xi;
        }
      }
#pragma unroll
      for (int k = 0; k < DMK_NV; k++) J[k] *= S.dsqrtinv[k];
}

// ---- wide constraint path, 64 < nefc <= 128 (lying / getting-up poses: ~20 floor contacts x 4 pyramid rows).
// Every lane carries two rows (lane and lane + 64); the whole A matrix streams through a per-env global
// scratch (column i at ar[i * 128 + row]), one column prefetched ahead of the Gauss-Seidel update.  Rare and
// out of line so that it costs the common path neither registers nor instruction cache.
__device__ __noinline__ float fwd_constraint_wide(GDev &T, const int lane, const int nefc, const float xs, const float dv,
                                                  float *ar) {
  EnvLds &S = g_S;
  const int lk = lane < DMK_NV ? lane : 0;
  const float com[3] = {S.com[0], S.com[1], S.com[2]};
  float Ja[DMK_NV], Jb[DMK_NV];
  float Ra, Da, arefa, bba, jwa, Rb, Db, arefb, bbb, jwb;
  build_row<false>(T, S, lane, nefc, com, Ja, Ra, Da, arefa, bba, jwa);
  build_row<false>(T, S, lane + 64, nefc, com, Jb, Rb, Db, arefb, bbb, jwb);
  float ARda = Ra, ARdb = Rb;
#pragma unroll
  for (int k = 0; k < DMK_NV; k++) { ARda += Ja[k] * Ja[k]; ARdb += Jb[k] * Jb[k]; }
  const bool va = lane < nefc, vb = lane + 64 < nefc;
#pragma unroll 1
  for (int i = 0; i < nefc; i++) {
    const int src = i & 63;
    float acca = 0, accb = 0;
    if (i < 64) {
#pragma unroll
      for (int k = 0; k < DMK_NV; k++) { const float bi = rl(Ja[k], src); acca = fmaf(Ja[k], bi, acca); accb = fmaf(Jb[k], bi, accb); }
      if (lane == src) acca += Ra;
    } else {
#pragma unroll
      for (int k = 0; k < DMK_NV; k++) { const float bi = rl(Jb[k], src); acca = fmaf(Ja[k], bi, acca); accb = fmaf(Jb[k], bi, accb); }
      if (lane == src) accb += Rb;
    }
    ar[i * 128 + lane] = acca;
    ar[i * 128 + 64 + lane] = accb;
  }
  const float ARinva = va ? 1.0f / ARda : 0.f, ARinvb = vb ? 1.0f / ARdb : 0.f;
  // warm start
  float fa = 0, fb = 0, ra = bba, rb = bbb;
  {
    const float jara = jwa - arefa, jarb = jwb - arefb;
    const float fwa = (va && jara < 0) ? -Da * jara : 0.f, fwb = (vb && jarb < 0) ? -Db * jarb : 0.f;
    float rwa = bba, rwb = bbb;
#pragma unroll 1
    for (int i = 0; i < nefc; i++) {
      const float fi = (i < 64) ? rl(fwa, i & 63) : rl(fwb, i & 63);
      rwa = fmaf(ar[i * 128 + lane], fi, rwa);
      rwb = fmaf(ar[i * 128 + 64 + lane], fi, rwb);
    }
    const float cost = wave_sum(fwa * (0.5f * (rwa - bba) + bba) + fwb * (0.5f * (rwb - bbb) + bbb));
    if (!(cost > 0)) { fa = fwa; fb = fwb; ra = rwa; rb = rwb; }
  }
  const float scale = T.pgs_scale, tol = T.tolerance;
  const int max_iter = T.iterations;
  int iter = 0;
  while (iter < max_iter) {  // same sweep as the one-row-per-lane path: residuals only, forces committed per sweep
    float na = ar[lane], nb = ar[64 + lane];
    const float nfa = -fa, nfb = -fb;
    float rsa = ra, rsb = rb;
#pragma unroll 1
    for (int i = 0; i < 64; i++) {
      const float ca = na, cb = nb;
      na = ar[(i + 1) * 128 + lane]; nb = ar[(i + 1) * 128 + 64 + lane];   // nefc > 64: column i + 1 exists
      const float dli = rl(fmaxf(nfa, -ra * ARinva), i);
      rsa = (lane == i) ? ra : rsa;
      ra = fmaf(ca, dli, ra);
      rb = fmaf(cb, dli, rb);
    }
#pragma unroll 1
    for (int i = 64; i < nefc; i++) {
      const float ca = na, cb = nb;
      if (i + 1 < nefc) { na = ar[(i + 1) * 128 + lane]; nb = ar[(i + 1) * 128 + 64 + lane]; }
      const float dli = rl(fmaxf(nfb, -rb * ARinvb), i - 64);
      rsb = (lane == i - 64) ? rb : rsb;
      ra = fmaf(ca, dli, ra);
      rb = fmaf(cb, dli, rb);
    }
    const float dla = fmaxf(nfa, -rsa * ARinva), dlb = fmaxf(nfb, -rsb * ARinvb);
    const float impv = -wave_sum(dla * fmaf(0.5f * dla, ARda, rsa) + dlb * fmaf(0.5f * dlb, ARdb, rsb));
    fa += dla;
    fb += dlb;
    iter++;
    if (impv * scale < tol) break;
  }
  if (lane == 0) S.info[3] = iter;
  // qacc = qacc_smooth + L^-1 D^-1/2 sum_r f_r B_r
  float v = 0;
#pragma unroll
  for (int k = 0; k < DMK_NV; k++) {
    const float sm = wave_sum(fa * Ja[k] + fb * Jb[k]);
    if (lane == k) v = sm;
  }
  const int mrow = T.d_madr[lk] + T.d_nanc[lk];
  const uint64_t ancm = T.d_ancm[lk];
  v *= S.dsqrtinv[lk] * S.M[T.d_madr[lk]];               // z = D (D^-1/2 v)
  v = solve_L(v, dv, (lane < DMK_NV) ? mrow : DMK_MAXANC, g_S.M);   // x = L^-1 (.) with x = z * dinv
  return xs + v * dv;
}

// ---- forward evaluation, part 3: constraint rows, A = J M^-1 J^T + R, PGS, qacc; returns qacc[lane]
__device__ __forceinline__ float fwd_constraint(GDev &T, const int lane, const int ncon, int overflow, float *dbg_force,
                                                float *ar_scratch) {
  EnvLds &S = g_S;
  int nefc = 0, nlimit = 0, solver_iter = 0;
  const int lb = lane < DMK_NB ? lane : 0;       // lane as body
  const int lk = lane < DMK_NV ? lane : 0;       // lane as dof
  const int lg = lane < DMK_NG ? lane : 0;       // lane as geom
  const int bdep = (lane < DMK_NB) ? T.b_depth[lb] : -1;
  const float bmass = (lane < DMK_NB) ? T.b_mass[lb] : 0.f;
  const float mtot_inv = T.total_mass_inv;
  (void)lg; (void)bdep; (void)bmass; (void)mtot_inv; (void)lb; (void)lk;
  const float xs = (lane < DMK_NV) ? S.qacc_smooth[lk] : 0.f;
  const float dv = (lane < DMK_NV) ? S.dinv[lk] : 0.f;
  const float com[3] = {S.com[0], S.com[1], S.com[2]};
  float qacc_out;
    // ---- constraint rows: limits in joint order, then contacts in contact order
    {
      bool lim = false;
      float ldist = 0;
      int lneg = 0;
      if (lane >= 6 && lane < DMK_NV && dlimited) {
        float q = S.qpos[lane + 1];
        if (q < dlo) { lim = true; ldist = q - dlo; lneg = 0; }
        else if (q > dhi) { lim = true; ldist = dhi - q; lneg = 1; }
      }
      int myrow = prefix_count(lim, lane, &nlimit);
      if (lim) { S.rowinfo[myrow] = (int16_t)(-(((lane << 1) | lneg) + 1)); }
      int dim3 = 0, have = lane < ncon;
      if (have) {
        int cd1 = T.g_condim[S.c_g1[lane]], cd2 = T.g_condim[S.c_g2[lane]];
        dim3 = (cd1 > cd2 ? cd1 : cd2) >= 3;
      }
      unsigned long long lt = lanemask_lt(lane);
      unsigned long long m1 = __ballot(have && !dim3), m4 = __ballot(have && dim3);
      int roff = nlimit + __popcll(m1 & lt) + 4 * __popcll(m4 & lt);
      int rtot = nlimit + __popcll(m1) + 4 * __popcll(m4);
      if (have) {
        int nr = dim3 ? 4 : 1;
        for (int e = 0; e < nr; e++)
          if (roff + e < DMK_MAXROW) S.rowinfo[roff + e] = (int16_t)((lane << 3) | e | (((roff + nr) <= DMK_MAXROW) ? 0x4000 : 0));
      }
      if (rtot > DMK_MAXROW) { overflow |= 2; rtot = DMK_MAXROW; }
      nefc = rtot;
    }
    SYNC();

    qacc_out = xs;  // qacc = qacc_smooth when there is no constraint
    solver_iter = 0;
    if (nefc > DMK_LANEROW) {
      qacc_out = fwd_constraint_wide(T, lane, nefc, xs, dv, ar_scratch);
      solver_iter = S.info[3];
    } else if (nefc > 0) {
      // ---- Jacobian row, reference acceleration, regulariser, B row (lane = row)
      float J[DMK_NV];
      float R, Dd, aref, bb, jw;
      if (nefc <= DMK_REGROW) build_row<true>(T, S, lane, nefc, com, J, R, Dd, aref, bb, jw);
      else build_row<false>(T, S, lane, nefc, com, J, R, Dd, aref, bb, jw);
      PROF(6);
      // ---- A row: AR[i] = B_lane . B_i (+ R on the diagonal).  Columns 0..31 live in registers; the
      // rare columns 32..63 (nefc > 32: p99 of the benchmark workload is 16) go to a per-env global
      // scratch, written and later re-read by the same lane.
      float AR[DMK_REGROW];
      float *arx = ar_scratch + lane;                      // column i >= 32 at arx[(i - 32) * 64]
      float ARd = R;
#pragma unroll
      for (int k = 0; k < DMK_NV; k++) ARd += J[k] * J[k];
      if (nefc <= DMK_REGROW && nefc >= DMK_MFMA_MIN_ROWS) {
        // All rows sit in lanes 0..31: A = B B^T is one 32 x 32 x 34 product on the matrix pipe.  v_mfma_f32_32x32x2f32
        // wants lane l to supply B[l % 32][k0 + l / 32] for both operands: v_permlane32_swap puts the rows' k0 + 1
        // entries into the upper half-wave.  17 MFMAs + 17 swaps instead of nefc x (34 v_readlane + 34 FMA).
        mfma_f16v acc;
#pragma unroll
        for (int v = 0; v < 16; v++) acc[v] = 0.f;
        StaticFor<0, DMK_NV / 2>::run([&](auto kc) {
          constexpr int k0 = decltype(kc)::value * 2;
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(J[k0]), __float_as_uint(J[k0 + 1]), false, false);
          const float x = __uint_as_float(sw[0]);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, acc, 0, 0, 0);
          return true;
        });
        // result fragment: lane (c, h) holds A[i][c] = A[c][i] for i = 8 (v / 4) + 4 h + v % 4; the upper half-wave's
        // sixteen values move down with one more swap each, so that lane c < 32 owns its whole row
        StaticFor<0, 16>::run([&](auto vc) {
          constexpr int v = decltype(vc)::value;
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[v]), __float_as_uint(acc[v]), false, false);
          constexpr int i0 = (v / 4) * 8 + (v % 4);
          AR[i0] = lane_sel<(1ull << i0)>(acc[v] + R, acc[v]);
          const float up = __uint_as_float(sw[1]);
          AR[i0 + 4] = lane_sel<(1ull << (i0 + 4))>(up + R, up);
          return true;
        });
      } else
      StaticFor<0, DMK_REGROW>::run([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (i >= nefc) {  // PGS walks the columns in blocks of four: zero the tail of the last block, stop after it
          if ((i & 3) == 0) return false;
          AR[i] = 0.f;
          return true;
        }
        float acc0 = 0, acc1 = 0;
#pragma unroll
        for (int k = 0; k < DMK_NV; k += 2) {
          acc0 = fmaf(J[k], rl(J[k], i), acc0);
          acc1 = fmaf(J[k + 1], rl(J[k + 1], i), acc1);
        }
        float acc = acc0 + acc1;
        if (lane == i) acc += R;
        AR[i] = acc;
        return true;
      });
#pragma unroll 1
      for (int i = DMK_REGROW; i < nefc; i++) {
        float acc0 = 0, acc1 = 0;
#pragma unroll
        for (int k = 0; k < DMK_NV; k += 2) {
          acc0 = fmaf(J[k], rl(J[k], i), acc0);
          acc1 = fmaf(J[k + 1], rl(J[k + 1], i), acc1);
        }
        float acc = acc0 + acc1;
        if (lane == i) acc += R;
        arx[(i - DMK_REGROW) * 64] = acc;
      }
      const float ARinv = (lane < nefc) ? 1.0f / ARd : 0.f;
      PROF(7);
      // ---- warm start (mj_fwdConstraint): forces implied by qacc_warmstart if their dual cost < 0
      float f = 0, r = bb;
      {
        float jar = jw - aref;
        float fw = (lane < nefc && jar < 0) ? -Dd * jar : 0.f;
        float rw = bb;
        StaticFor<0, DMK_REGROW>::run([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if (i >= nefc) return false;
          rw += AR[i] * rl(fw, i);
          return true;
        });
#pragma unroll 1
        for (int i = DMK_REGROW; i < nefc; i++) rw += arx[(i - DMK_REGROW) * 64] * rl(fw, i);
        float cost = wave_sum(fw * (0.5f * (rw - bb) + bb));
        if (!(cost > 0)) { f = fw; r = rw; }
      }
      // ---- projected Gauss-Seidel.  Lane j's force changes only at step j of a sweep, so a sweep carries the
      // residual alone: step i broadcasts dl_i = max(-f_i, -r_i / A_ii) (= projected force minus force) and every
      // lane updates its residual; lane i records the residual it saw.  The force and the cost decrease
      // (mj_solPGS "improvement") are committed once per sweep from the recorded residual.  The row bound is tested
      // once per block of four (a row >= nefc has f = 0, 1/A_ii = 0 => dl = 0, and its column was zeroed above).
      // The residual is carried scaled, g = -r / A_ii, with the columns pre-multiplied by -1 / A_ii of the lane's
      // own row: the step is then dl_i = max(-f_i, g_i) and g += As[i] dl_i — five VALU per row.
      const float scale = T.pgs_scale, tol = T.tolerance;
      const int max_iter = T.iterations;
      const float nAinv = -ARinv;
      float g = r * nAinv;
      StaticFor<0, DMK_REGROW / 4>::run([&](auto bc) {
        constexpr int b = decltype(bc)::value * 4;
        if (b >= nefc) return false;
        AR[b] *= nAinv; AR[b + 1] *= nAinv; AR[b + 2] *= nAinv; AR[b + 3] *= nAinv;
        return true;
      });
      int iter = 0;
      while (iter < max_iter) {
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));  // keeps the per-row lane compares inside the sweep (else 64 SGPRs of masks spill)
        const float nf = -f;
        float gs = g;
        StaticFor<0, DMK_REGROW / 4>::run([&](auto bc) {
          constexpr int b = decltype(bc)::value * 4;
          if (b >= nefc) return false;
          auto step = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const float dli = rl(fmaxf(nf, g), i);
            gs = (lane_s == i) ? g : gs;
            g = fmaf(AR[i], dli, g);
          };
          step(std::integral_constant<int, b>{});
          step(std::integral_constant<int, b + 1>{});
          step(std::integral_constant<int, b + 2>{});
          step(std::integral_constant<int, b + 3>{});
          return true;
        });
        if (nefc > DMK_REGROW) {                            // overflow columns, streamed one row ahead
          float anext = arx[0];
#pragma unroll 1
          for (int i = DMK_REGROW; i < nefc; i++) {
            const float acur = anext * nAinv;
            if (i + 1 < nefc) anext = arx[(i + 1 - DMK_REGROW) * 64];
            const float dli = rl(fmaxf(nf, g), i);
            gs = (lane_s == i) ? g : gs;
            g = fmaf(acur, dli, g);
          }
        }
        const float dl = fmaxf(nf, gs);
        const float impv = -wave_sum(dl * ARd * (0.5f * dl - gs));   // dl (dl A_ii / 2 + r) with r = -g A_ii
        f += dl;
        iter++;
        if (impv * scale < tol) break;
      }
      solver_iter = iter;
      PROF(8);
      // ---- qacc = qacc_smooth + L^-1 D^-1/2 sum_r f_r B_r
      // v[k] = sum_r f_r B_r[k]: rows live in the lanes of their constraints, the result is wanted in the
      // lanes of the dofs.  Few rows carry force, so transpose only those through LDS (34 writes per
      // active row-lane, one read per active row per dof-lane); fall back to 34 wave reductions otherwise.
      float v = 0;
      {
        const bool actv = (lane < nefc) && (f != 0.f);
        int nact;
        const int slot = prefix_count(actv, lane, &nact);
        if (nact <= 20) {
          if (actv) {
#pragma unroll
            for (int k = 0; k < DMK_NV; k++) S.u.fin.tr[slot][k] = f * J[k];
          }
          SYNC();
          if (lane < DMK_NV)
            for (int a = 0; a < nact; a++) v += S.u.fin.tr[a][lk];
          SYNC();
        } else {
#pragma unroll
          for (int k = 0; k < DMK_NV; k++) {
            float s = wave_sum(f * J[k]);
            if (lane == k) v = s;
          }
        }
      }
      const int mrow = T.d_madr[lk] + T.d_nanc[lk];
      const uint64_t ancm = T.d_ancm[lk];
      v *= S.dsqrtinv[lk] * S.M[T.d_madr[lk]];               // z = D (D^-1/2 v)
      v = solve_L(v, dv, (lane < DMK_NV) ? mrow : DMK_MAXANC, g_S.M);   // x = L^-1 (.) with x = z * dinv
      qacc_out = xs + v * dv;
#ifndef DM_PROFILE
      if (dbg_force && lane < DMK_LANEROW) dbg_force[lane] = (lane < nefc) ? f : 0.f;
#endif
    }
    SYNC();
    if (lane < DMK_NV) { S.qacc[lk] = qacc_out; S.warm[lk] = qacc_out; }
    SYNC();
    PROF(9);
  if (lane == 0) { S.info[0] = ncon; S.info[1] = nefc; S.info[2] = nlimit; S.info[3] = solver_iter; S.info[4] = overflow; }
  return qacc_out;
}

// ======================================================================================
// TASK 0: DPEnv (src/deepmimic_env.py).  TASK 1: DPCombinedEnv state machine (src/combined_env.py) on the same
// physics; the state row's clip slot holds the motion id (0 walk, 1 run, 2 getup, 3 to_getup) and the idx slot n_steps.
template <int TASK>
__device__ __forceinline__ void step_body(const DmLaunch &P) {
  constexpr int NOBS_T = TASK ? DM_NOBS_COMBINED : DM_NOBS;   // obs row stride
  constexpr int NOBS_B = NOBS_T - 64;                          // obs entries held by obs_b lanes
  constexpr int NTERMS = TASK ? 8 : 5;
  // Model tables are read from global memory at their use sites: 6 KB shared by every wave on the CU,
  // so they sit in the vector L1 / scalar cache; keeping them out of LDS leaves room for more envs.
  GDev &T = *(GDev *)P.T;
  const int lane = threadIdx.x & 63;
  EnvLds &S = g_S;
  const int slot = blockIdx.x;
  if (slot >= P.nslots) return;
  const int env = P.env_ids ? P.env_ids[slot] : slot;
  if (env < 0 || env >= P.N) return;
  const int mode = P.mode;
  if (mode == DMK_MODE_RESET && P.mask && !P.mask[env]) return;
  float *st = P.state + (size_t)env * DMK_STATE_STRIDE;
  int *sti = reinterpret_cast<int *>(st);

  // ---------------------------------------------------------------- lane roles (values are read
  // from the LDS-resident table at their use sites so they do not pin registers)
  const int lb = lane < DMK_NB ? lane : 0;       // lane as body
  const int lk = lane < DMK_NV ? lane : 0;       // lane as dof
  const int lg = lane < DMK_NG ? lane : 0;       // lane as geom
  const float h = T.timestep;
  const float mtot_inv = T.total_mass_inv;

  PROF_DECL;
  // ---------------------------------------------------------------- LDS init
  if (lane == 0) {
    S.xpos[0][0] = S.xpos[0][1] = S.xpos[0][2] = 0;
    S.xquat[0][0] = 1; S.xquat[0][1] = S.xquat[0][2] = S.xquat[0][3] = 0;
    for (int i = 0; i < 9; i++) S.xmat[0][i] = (i % 4 == 0) ? 1.f : 0.f;
    S.xipos[0][0] = S.xipos[0][1] = S.xipos[0][2] = 0;
    for (int i = 0; i < 6; i++) S.cvel[0][i] = 0;
    for (int i = 0; i < 10; i++) S.cinert[0][i] = 0;
  }

  // ---------------------------------------------------------------- load state
  const int clip_id = sti[DMS_CLIP];
  int motion = TASK ? ((clip_id >= 0 && clip_id < 4) ? clip_id : 0) : 0;
  DmClipDev clip = P.clips[TASK ? (motion == 3 ? 2 : motion) : ((clip_id >= 0 && clip_id < 8) ? clip_id : 0)];
  if (clip.L < 1 || clip.rows == nullptr) return;  // no clip loaded for this env's clip id
  if (TASK && (P.clips[0].L < 1 || P.clips[1].L < 1 || P.clips[2].L < 2)) return;  // walk, run, getup all needed
  int idx_curr = sti[DMS_IDX], ep_len = sti[DMS_EPLEN], rcnt = sti[DMS_RCNT];
  if (TASK) idx_curr = idx_curr < 0 ? 0 : idx_curr;  // n_steps: not wrapped (:454)
  else idx_curr = idx_curr < 0 ? 0 : (idx_curr >= clip.L ? clip.L - 1 : idx_curr);
  float ep_rew = st[DMS_EPREW];
  // F8 option (DmConfig.stale_contact_slots, src/deepmimic_env.py:88): mujoco-py hands the WHOLE contact array to the foot-contact
  // scan, so slots >= ncon still show what an earlier evaluation wrote.  One bit per slot and foot, kept in the state row.
  unsigned f8r = P.f8 ? (unsigned)sti[DMS_F8R] : 0u, f8l = P.f8 ? (unsigned)sti[DMS_F8L] : 0u;
  if (lane < DMK_NQ) S.qpos[lane] = st[DMS_QPOS + lane];
  if (lane < DMK_NV) { S.qvel[lane] = st[DMS_QVEL + lane]; S.warm[lane] = st[DMS_WARM + lane]; }
  if (lane < DMK_NU) S.ctrl[lane] = st[DMS_CTRL + lane];
  SYNC();
  if (mode == DMK_MODE_STEP || mode == DMK_MODE_PHYSICS) {
    if (lane < DMK_NU) S.ctrl[lane] = P.actions[(size_t)env * DMK_NU + lane];  // ctrl = action * 1.0 (:347)
  } else if (mode == DMK_MODE_FORCED || mode == DMK_MODE_SETSTATE) {
    if (P.in_qpos) {   // null in SETSTATE mode = dm_forward: keep the stored state, only re-run the forward evaluation
      if (lane < DMK_NQ) S.qpos[lane] = P.in_qpos[(size_t)slot * DMK_NQ + lane];
      if (lane < DMK_NV) S.qvel[lane] = P.in_qvel[(size_t)slot * DMK_NV + lane];
    }
    if (mode == DMK_MODE_SETSTATE) {
      if (P.in_warm && lane < DMK_NV) S.warm[lane] = P.in_warm[(size_t)slot * DMK_NV + lane];
      if (P.in_ctrl && lane < DMK_NU) S.ctrl[lane] = P.in_ctrl[(size_t)slot * DMK_NU + lane];
    }
  } else if (mode == DMK_MODE_RESET) {
    int fi;
    if (TASK) {
      // DPCombinedEnv.reset(rsi=True) (:219-227): walk with amnesty or getup, random frame; explicit idx_init
      // keeps the env's motion id and sets n_steps
      if (P.idx_init) { idx_curr = P.idx_init[env] < 0 ? 0 : P.idx_init[env]; }
      else {
        motion = (dm_hash32(P.seed, env, rcnt, 0x5EED) & 1) ? 2 : 0;
        clip = P.clips[motion];
        idx_curr = (int)(dm_hash32(P.seed, env, rcnt, 0x5EEE) % (uint32_t)clip.L) + (motion == 0 ? P.amnesty_steps + 10 : 0);
      }
      fi = (motion == 3) ? 1 : idx_curr % clip.L;
    } else {
      fi = P.idx_init ? P.idx_init[env] : (int)(dm_hash32(P.seed, env, rcnt, 0x5EED) % (uint32_t)clip.L);
      fi = fi < 0 ? 0 : (fi >= clip.L ? clip.L - 1 : fi);
      idx_curr = fi;
    }
    const float *rr = clip.reset + (size_t)fi * DMK_RESET_ROW;
    if (lane < DMK_NQ) S.qpos[lane] = rr[lane];
    if (lane < DMK_NV) S.qvel[lane] = rr[35 + lane];
    ep_len = 0; ep_rew = 0; rcnt++;
  }
  SYNC();

  // RK4 bookkeeping, lane = dof (k<3 root translation, 3..5 root rotation, >=6 hinges): X0 (position, velocity), the
  // weighted sums of the stage derivatives and the current stage velocity live in LDS (S.rk / S.rkq), not in VGPRs —
  // they are touched once per stage and would otherwise occupy nine registers across every forward evaluation
  int it = (mode == DMK_MODE_STEP || mode == DMK_MODE_PHYSICS) ? 0 : 4;
  bool after_reset = false, done = false, sim_err = false;
  int reason = DM_REASON_NONE;
  float reward = 0;
  int ncon = 0, nefc = 0, nlimit = 0, solver_iter = 0, overflow = 0;
  float com[3] = {0, 0, 0};
  unsigned stage_ncon = 0, stage_nefc = 0;  // byte i = count at RK stage i (debug)
  int work = 0;                              // work estimate of this step (for longest-first scheduling)

  if (mode == DMK_MODE_SETSTATE && !P.run_forward) {  // store only
    if (lane < DMK_NQ) st[DMS_QPOS + lane] = S.qpos[lane];
    if (lane < DMK_NV) { st[DMS_QVEL + lane] = S.qvel[lane]; st[DMS_WARM + lane] = S.warm[lane]; }
    if (lane < DMK_NU) st[DMS_CTRL + lane] = S.ctrl[lane];
    return;
  }
  if (mode == DMK_MODE_STEP || mode == DMK_MODE_FORCED || mode == DMK_MODE_PHYSICS) {  // mj_checkPos / mj_checkVel
    float a = (lane < DMK_NQ) ? S.qpos[lane] : 0.f, b = (lane < DMK_NV) ? S.qvel[lk] : 0.f;
    sim_err = __any(!(fabsf(a) <= MAXVALF) || !(fabsf(b) <= MAXVALF));
  }

  PROF(0);
  for (;;) {
   float qacc_out = 0;
   if (!sim_err) {
    // ============================================================== forward evaluation
    // launder the table pointer once per stage: keeps hipcc from hoisting table loads out of the
    // stage loop into registers that would then live (and spill) across the whole kernel
    GDev *Tp = (GDev *)P.T;
    asm volatile("" : "+s"(Tp));
    GDev &Ts = *Tp;
#ifndef DM_NO_LANE_LAUNDER
    // the lane id is laundered per stage too: per-lane compares (lane == k, lane < n) are then recomputed inside the stage
    // (one v_cmp) instead of being hoisted out of the stage loop into SGPRs that spill to VGPR lanes (v_writelane / two
    // v_readlane per use)
    int lane_st = lane;
    asm volatile("" : "+v"(lane_st));
#else
    const int lane_st = lane;
#endif
    fwd_smooth(Ts, lane_st);
    if (it == 0) {  // capture X0 (after normalisation)
      if (lane < DMK_NV) {
        const float v0 = S.qvel[lk];
        S.rk[0][lk] = (lane < 3) ? S.qpos[lane] : ((lane >= 6) ? S.qpos[lane + 1] : 0.f);
        S.rk[1][lk] = v0; S.rk[2][lk] = 0.f; S.rk[3][lk] = 0.f; S.rk[4][lk] = v0;
      }
      if (lane < 4) S.rkq[lane] = S.qpos[3 + lane];
    }
    {
      const int cr = fwd_collide(Ts, lane_st);
      ncon = cr & 0xFF;
      qacc_out = fwd_constraint(Ts, lane_st, ncon, cr >> 8, P.debug ? P.debug + (size_t)env * DM_DEBUG_STRIDE + 352 : nullptr,
                                P.ar_scratch + (size_t)env * DMK_MAXROW * DMK_MAXROW);
      nefc = S.info[1]; nlimit = S.info[2]; solver_iter = S.info[3]; overflow = S.info[4];
      if (it < 4) { stage_ncon |= (unsigned)(ncon & 0xFF) << (8 * it); stage_nefc |= (unsigned)(nefc & 0xFF) << (8 * it); }
      // the sort key of the next step's longest-first order: the later evaluations of the step weigh more (1/8, 1/4, 1/2, 1, and the
      // evaluation at the reset state last of all) — the key goes stale within a step (the plain sum: 13.56 M, this: 13.73 M; last only: 13.65 M)
      // (rows weigh 16: measured, building the rows costs twice the sweeps on average; 4 .. 32 and a wide-path surcharge: +-0.3 %)
      work = (work >> 1) + 64 + (nefc > 0 ? 48 + 16 * nefc : 0) + nefc * solver_iter;
      if (P.f8) {   // every evaluation rewrites slots [0, ncon) of the contact array and leaves the rest
        bool rfs = false, lfs = false;
        if (lane < ncon) {
          const int g1 = S.c_g1[lane], g2 = S.c_g2[lane];
          const bool fl = (g1 == Ts.floor_geom || g2 == Ts.floor_geom);
          rfs = fl && (g1 == Ts.rfoot_geom || g2 == Ts.rfoot_geom);
          lfs = fl && (g1 == Ts.lfoot_geom || g2 == Ts.lfoot_geom);
        }
        const unsigned keep = ncon >= 32 ? 0u : (~0u << ncon);
        f8r = (f8r & keep) | (unsigned)__ballot(rfs);
        f8l = (f8l & keep) | (unsigned)__ballot(lfs);
      }
    }
    // ============================================================== end of forward evaluation
    if (it == 0 || (it == 4 && mode == DMK_MODE_FORCED && !after_reset)) {  // mj_checkAcc
      bool badv = (lane < DMK_NV) && !(fabsf(qacc_out) <= MAXVALF);
      sim_err = __any(badv);
    }
   }  // !sim_err

    const bool euler = (P.integrator == DM_INT_EULER) && (mode == DMK_MODE_STEP || mode == DMK_MODE_PHYSICS);
    float x0q = 0, x0v = 0, accq = 0, accv = 0, curv = 0;
    float q0[4] = {1, 0, 0, 0};
    if (!sim_err && it < 4) {
      const bool dl = lane < DMK_NV;
      x0q = dl ? S.rk[0][lk] : 0.f; x0v = dl ? S.rk[1][lk] : 0.f; accq = dl ? S.rk[2][lk] : 0.f;
      accv = dl ? S.rk[3][lk] : 0.f; curv = dl ? S.rk[4][lk] : 0.f;
      for (int i = 0; i < 4; i++) q0[i] = S.rkq[i];
    }
    if (!sim_err && it < 3 && !euler) {  // RK4 intermediate stages (tableau A = diag(1/2, 1/2, 1), B = 1/6 1/3 1/3 1/6)
      const float Bw = (it == 0) ? (1.f / 6.f) : (1.f / 3.f);
      const float Aw = (it == 2) ? 1.f : 0.5f;
      accq += Bw * curv;
      accv += Bw * qacc_out;
      float dq = Aw * curv, dv = Aw * qacc_out;
      curv = x0v + h * dv;
      float w3[3] = {rl(dq, 3), rl(dq, 4), rl(dq, 5)};
      float wn = sqrtf(dot3(w3, w3));
      float qr[4] = {1, 0, 0, 0}, qn[4];
      if (wn >= MINVALF) {
        float s, c;
        sincosf(0.5f * h * wn, &s, &c);
        float inv = s / wn;
        qr[0] = c; qr[1] = w3[0] * inv; qr[2] = w3[1] * inv; qr[3] = w3[2] * inv;
      }
      quat_mul(qn, q0, qr);
      quat_normalize(qn);
      if (lane < DMK_NV) { S.qvel[lk] = curv; S.rk[2][lk] = accq; S.rk[3][lk] = accv; S.rk[4][lk] = curv; }
      if (lane < 3) S.qpos[lane] = x0q + h * dq;
      if (lane >= 6 && lane < DMK_NV) S.qpos[lane + 1] = x0q + h * dq;
      if (lane < 4) S.qpos[3 + lane] = (lane == 0) ? qn[0] : (lane == 1) ? qn[1] : (lane == 2) ? qn[2] : qn[3];
      SYNC();
      PROF_STAGE(it);
      it++;
      continue;
    }
    if (!sim_err && (it == 3 || (it == 0 && euler))) {  // final combination (mj_advance)
      if (euler) {
        // [EXT mj_Euler] semi-implicit Euler, joint damping integrated implicitly: (M + h B) a' = M a, i.e.
        // a' = a - (M + h B)^-1 h B a; velocity first, then the position with the NEW velocity
        const float hb = (lane < DMK_NV) ? h * T.d_damp[lk] : 0.f;
        float a2 = qacc_out;
        if (__any(hb != 0.f)) a2 = qacc_out - euler_damped_solve(T, lane, hb * qacc_out);
        accv = a2;
        accq = x0v + h * a2;
        it = 3;
      } else {
        accq += (1.f / 6.f) * curv;
        accv += (1.f / 6.f) * qacc_out;
      }
      float w3[3] = {rl(accq, 3), rl(accq, 4), rl(accq, 5)};
      float wn = sqrtf(dot3(w3, w3));
      float qr[4] = {1, 0, 0, 0}, qn[4];
      if (wn >= MINVALF) {
        float s, c;
        sincosf(0.5f * h * wn, &s, &c);
        float inv = s / wn;
        qr[0] = c; qr[1] = w3[0] * inv; qr[2] = w3[1] * inv; qr[3] = w3[2] * inv;
      }
      quat_mul(qn, q0, qr);
      quat_normalize(qn);
      if (lane < DMK_NV) S.qvel[lk] = x0v + h * accv;
      if (lane < 3) S.qpos[lane] = x0q + h * accq;
      if (lane >= 6 && lane < DMK_NV) S.qpos[lane + 1] = x0q + h * accq;
      if (lane < 4) S.qpos[3 + lane] = (lane == 0) ? qn[0] : (lane == 1) ? qn[1] : (lane == 2) ? qn[2] : qn[3];
      SYNC();
    }

    PROF(10);
    PROF_STAGE(it);
    if (mode == DMK_MODE_PHYSICS) {   // dm_physics_step: sim.step() alone; an instability resets the data as MuJoCo does
      if (sim_err) {
        f8r = f8l = 0;
        SYNC();
        if (lane < DMK_NQ) S.qpos[lane] = T.qpos0[lane];
        if (lane < DMK_NV) { S.qvel[lane] = 0; S.warm[lane] = 0; }
        if (lane < DMK_NU) S.ctrl[lane] = 0;
        SYNC();
      }
      break;
    }
    // ============================================================== task layer: obs, reward, done
    // (derived arrays are those of the LAST forward evaluation: SURVEY F6)
    const bool task_pass = !after_reset && (mode == DMK_MODE_STEP || mode == DMK_MODE_FORCED);
    float obs_a = 0, obs_b = 0;  // obs[lane], obs[64 + lane]
    float terms[5] = {0, 0, 0, 0, 0};
    float extra[3] = {0, 0, 0};  // TASK 1: imitation_reward, task_reward, n_bad_angles
    if (sim_err) {
      // MujocoException path (:366-378): zero obs, zero reward, done, empty info; MuJoCo resets mjData
      reward = 0; done = true; reason = DM_REASON_SIM_ERROR;
      f8r = f8l = 0;   // mj_resetData clears the contact array
      SYNC();
      if (lane < DMK_NQ) S.qpos[lane] = T.qpos0[lane];
      if (lane < DMK_NV) { S.qvel[lane] = 0; S.warm[lane] = 0; }
      if (lane < DMK_NU) S.ctrl[lane] = 0;
      SYNC();
    } else {
      // ---- get_obs (:33-45): qpos[7:], 0.1*qvel[6:], torso(8), foot contacts(2), phase(1)
      const float Sc = P.vel_obs_scale;
      if (lane < 28) obs_a = S.qpos[7 + lane];
      else if (lane < 56) obs_a = S.qvel[6 + lane - 28] * Sc;
      const int tb = T.torso_body;
      float rpy[3], tq[4] = {S.xquat[tb][0], S.xquat[tb][1], S.xquat[tb][2], S.xquat[tb][3]};
      quat_to_rpy(tq, rpy);
      float cv[6];
      for (int i = 0; i < 6; i++) cv[i] = S.cvel[tb][i];
      float sy, cy;
      sincosf(-rpy[2], &sy, &cy);
      const float tor[8] = {rpy[0] * Sc, rpy[1] * Sc, (cy * cv[3] - sy * cv[4]) * Sc, (sy * cv[3] + cy * cv[4]) * Sc,
                            cv[5] * Sc, cv[0] * Sc, cv[1] * Sc, cv[2] * Sc};
#pragma unroll
      for (int i = 0; i < 8; i++)
        if (lane == 56 + i) obs_a = tor[i];
      bool rf = false, lf = false;
      if (lane < ncon) {
        int g1 = S.c_g1[lane], g2 = S.c_g2[lane];
        bool fl = (g1 == T.floor_geom || g2 == T.floor_geom);
        rf = fl && (g1 == T.rfoot_geom || g2 == T.rfoot_geom);
        lf = fl && (g1 == T.lfoot_geom || g2 == T.lfoot_geom);
      }
      const float rff = (P.f8 ? f8r != 0 : __any(rf)) ? 1.f : 0.f, lff = (P.f8 ? f8l != 0 : __any(lf)) ? 1.f : 0.f;
      // TASK 1: motion length / frame of the current motion; to_getup is a 180-step pseudo clip whose target is
      // frame 1 of getup (MotionTransition getters, combined_env.py:67-99)
      const int Lm = (TASK && motion == 3) ? P.to_getup_len : clip.L;
      const int frame = TASK ? ((motion == 3) ? 1 : idx_curr % clip.L) : idx_curr;
      if (TASK) {
        // _get_obs (:495-505): no foot-contact obs (:25), phase, then get_player_action_obs (deepmimic_env.py:145-173)
        // with PAWalk (heading (1,0,0), one-hot index 0) and pa_getup_state
        const float ph = clampf((float)(idx_curr % Lm) / (float)Lm, 0.f, 1.f);
        if (lane == 0) obs_b = ph;
        if (lane == 1) obs_b = cy;            // hx = cos(-yaw)
        if (lane == 2) obs_b = sy;            // hy = sin(-yaw)
        if (lane == 3) obs_b = 1.f;
        if (lane == 6) obs_b = (motion == 3) ? 1.f : 0.f;
        if (lane == 7) obs_b = (motion == 2) ? 1.f : 0.f;
      } else {
        const float ph = clampf((float)idx_curr / (float)clip.L, 0.f, 1.f);
        if (lane == 0) obs_b = rff;
        if (lane == 1) obs_b = lff;
        if (lane == 2) obs_b = ph;
      }

      if (task_pass) {
        // ---- calc_imitation_reward (:193-256) against clip row idx_curr
        // row: [0:28) qpos[7:] | [28:56) qvel[6:] | [56:60) root quat | [60:72) ee xpos | [72:75) com
        const float *row = clip.rows + (size_t)frame * DMK_CLIP_ROW;
        const float ra = row[lane];
        const float adiff = (lane < 28) ? fabsf(S.qpos[7 + (lane < 28 ? lane : 0)] - ra) : 0.f;  // config_angle_diffs
        float ecfg = wave_sum(adiff);
        const float dsum = ecfg;
        const float evel = wave_sum((lane >= 28 && lane < 56) ? fabsf(ra - S.qvel[6 + ((lane >= 28 && lane < 56) ? lane - 28 : 0)]) : 0.f);
        const float tquat[4] = {rl(ra, 56), rl(ra, 57), rl(ra, 58), rl(ra, 59)};
        const float cq[4] = {S.qpos[3], S.qpos[4], S.qpos[5], S.qpos[6]};
        float rc[3], rt[3];
        quat_to_rpy(cq, rc);
        quat_to_rpy(tquat, rt);
        ecfg += fabsf(rc[1] - rt[1]);
        const float r_cfg = expf(-ecfg);
        const float r_vel = expf(-0.1f * evel);
        float df = 0;
        if (lane < 12) {
          int e = lane / 3, cc = lane % 3;
          df = S.gpos[T.ee_geom[e]][cc] - row[60 + lane];
        }
        const float r_ee = expf(-40.f * wave_sum(df * df));
        float ce = 0;
        const float bmass_t = (lane < DMK_NB) ? T.b_mass[lb] : 0.f;     // read here: not a register held across the forward evaluations
        for (int i = 0; i < 3; i++) {
          float d2 = row[72 + i] - wave_sum(bmass_t * S.xpos[lb][i]) * mtot_inv;
          ce += d2 * d2;
        }
        const float r_com = expf(-10.f * ce);
        float viol = 0;
        if (lane >= 6 && lane < DMK_NV) {
          float q = S.qpos[lane + 1];
          viol = ((q <= dlo * 0.99f) ? 1.f : 0.f) + ((q >= dhi * 0.99f) ? 1.f : 0.f);
        }
        const float qlim = wave_sum(viol) / 28.0f;
        terms[0] = r_cfg; terms[1] = r_vel; terms[2] = r_ee; terms[3] = r_com; terms[4] = qlim;
        reward = P.w_pose * r_cfg + P.w_vel * r_vel + P.w_ee * r_ee + P.w_com * r_com + P.w_jl * qlim;
        const float zc = S.com[2];
        done = false;
        reason = DM_REASON_NONE;
        if (TASK) {
          // ---- DPCombinedEnv: task reward (combined_env.py:338-354)
          const float ALIM = 0.2617993877991494f, MAX_ANGLE = 1.0471975511965976f;  // deg2rad(15), deg2rad(60)
          const float droll = fabsf(rc[0] - rt[0]), dpitch = fabsf(rc[1] - rt[1]);
          float imitation = reward, task_r = 0.f;
          if (motion == 0 || motion == 1) {                       // heading + velocity error vs the clip's root velocity
            const float ex = row[78] - S.qvel[0], ey = row[79] - S.qvel[1];
            task_r = expf(-sqrtf(ex * ex + ey * ey) * 10.f);
          }
          if (motion == 3) { imitation = 0.f; task_r = expf(-(dsum + dpitch + droll) / 5.f) / 3.f; }
          reward = imitation * 0.7f + task_r * 0.3f;
          const unsigned long long badm = __ballot(adiff > ALIM);
          const bool all_close = !__any(!(adiff < ALIM));          // lanes >= 28 carry 0
          extra[0] = imitation; extra[1] = task_r;
          extra[2] = (float)(__popcll(badm) + (dpitch > ALIM ? 1 : 0) + (droll > ALIM ? 1 : 0));  // debug_n_bad_angles
          // ---- motion state machine + termination (:393-445)
          if (idx_curr >= Lm - 1) {                               // out of time
            // :396 compares PlayerAction objects by identity -> getup always hands over to run
            if (motion == 2) { motion = 1; idx_curr = 0; }
            if (motion == 3) { motion = 2; idx_curr = 0; }
          }
          if (dpitch < ALIM && droll < ALIM && all_close && motion == 3) { motion = 2; idx_curr = 0; }
          if (motion == 0 || motion == 1) {
            const bool fallen = (zc < P.low_z) || (zc > P.high_z) || (droll > MAX_ANGLE) || (dpitch > MAX_ANGLE);
            if (fallen) {
              if (!(idx_curr > P.amnesty_steps)) { done = true; reason = DM_REASON_FALLEN_NO_AMNESTY; }
              motion = 3; idx_curr = 0;
            }
          }
          if (P.max_ep_length != 0 && ep_len >= P.max_ep_length) { done = true; reason = DM_REASON_MAX_EP_LEN; }
          clip = P.clips[motion == 3 ? 2 : motion];
          idx_curr += 1;                                          // :454 (not wrapped)
        } else {
          // ---- termination (:418-442)
          if (!(clip.flags & DM_CLIP_FLOOR)) {                      // :420-424 (written every step, even when not done)
            done = (zc < P.low_z) || (zc > P.high_z);
            reason = (zc < P.low_z) ? DM_REASON_LOW_Z : DM_REASON_HIGH_Z;
          }
          if (P.max_ep_length != 0 && ep_len >= P.max_ep_length) { done = true; reason = DM_REASON_MAX_EP_LEN; }
          if ((clip.flags & DM_CLIP_ACYCLIC) && idx_curr + 1 == clip.L) { done = true; reason = DM_REASON_ACYCLIC_END; }  // :440-442
          // ---- post-step counters (:452-455)
          idx_curr = (idx_curr + 1) % clip.L;
        }
        ep_rew += reward;
        ep_len += 1;
        // ---- observation guard (:465-476)
        bool ob = !(fabsf(obs_a) <= P.obs_bound) || !(fabsf(obs_b) <= P.obs_bound);
        if (__any(ob)) {
          obs_a = 0; obs_b = 0; reward = 0; done = true; reason = DM_REASON_OBS_BOUNDS;
          for (int i = 0; i < 5; i++) terms[i] = 0;
          extra[0] = extra[1] = extra[2] = 0;
        }
      }
    }
    if (P.debug && !after_reset) {
      float *dbg = P.debug + (size_t)env * DM_DEBUG_STRIDE;
      if (lane < 42) dbg[lane] = (&S.xpos[0][0])[lane];
      if (lane < 48) dbg[42 + lane] = (&S.gpos[0][0])[lane];
      for (int i = lane; i < 84; i += 64) dbg[90 + i] = (&S.cvel[0][0])[i];
      if (lane < DMK_NV) { dbg[174 + lane] = S.qacc[lane]; dbg[208 + lane] = S.qacc_smooth[lane]; }
      if (lane == 0) { dbg[242] = ncon; dbg[243] = nefc; dbg[244] = solver_iter; dbg[245] = nlimit; dbg[246] = overflow;
                       dbg[247] = __int_as_float((int)stage_ncon); dbg[248] = __int_as_float((int)stage_nefc); }
      if (lane < DMK_MAXCON) {
        dbg[256 + 3 * lane] = (lane < ncon) ? (float)S.c_g1[lane] : -1.f;
        dbg[257 + 3 * lane] = (lane < ncon) ? (float)S.c_g2[lane] : -1.f;
        dbg[258 + 3 * lane] = (lane < ncon) ? S.c_dist[lane] : 0.f;
      }
    }
    if (task_pass) {
      if (P.rew && lane == 0) P.rew[env] = reward;
      if (P.done && lane == 0) P.done[env] = done ? 1 : 0;
      if (P.reason && lane == 0) P.reason[env] = reason;
      if (P.terms && lane < NTERMS)
        P.terms[(size_t)env * NTERMS + lane] = (lane == 0) ? terms[0] : (lane == 1) ? terms[1] : (lane == 2) ? terms[2] : (lane == 3) ? terms[3]
                                             : (lane == 4) ? terms[4] : (lane == 5) ? extra[0] : (lane == 6) ? extra[1] : extra[2];
      if (done && P.auto_reset && mode == DMK_MODE_STEP) {
        // VecEnv worker: info["terminal_observation"] = obs; obs = env.reset()
        if (P.terminal_obs) {
          P.terminal_obs[(size_t)env * NOBS_T + lane] = obs_a;
          if (lane < NOBS_B) P.terminal_obs[(size_t)env * NOBS_T + 64 + lane] = obs_b;
        }
        int fi;
        if (TASK) {  // DPCombinedEnv.reset(rsi=True) (:219-227)
          motion = (dm_hash32(P.seed, env, rcnt, 0x5EED) & 1) ? 2 : 0;
          clip = P.clips[motion];
          fi = (int)(dm_hash32(P.seed, env, rcnt, 0x5EEE) % (uint32_t)clip.L);
          idx_curr = fi + (motion == 0 ? P.amnesty_steps + 10 : 0);
          fi = idx_curr % clip.L;
        } else {
          fi = (int)(dm_hash32(P.seed, env, rcnt, 0x5EED) % (uint32_t)clip.L);
          idx_curr = fi;
        }
        rcnt++;
        const float *rr = clip.reset + (size_t)fi * DMK_RESET_ROW;
        SYNC();
        if (lane < DMK_NQ) S.qpos[lane] = rr[lane];
        if (lane < DMK_NV) S.qvel[lane] = rr[35 + lane];
        ep_len = 0; ep_rew = 0;
        SYNC();
        after_reset = true;
        sim_err = false;
        it = 4;
        continue;  // one more forward evaluation at the reset state (set_state -> sim.forward)
      }
    }
    if (P.obs) {
      P.obs[(size_t)env * NOBS_T + lane] = obs_a;
      if (lane < NOBS_B) P.obs[(size_t)env * NOBS_T + 64 + lane] = obs_b;
    }
    break;
  }

  PROF(11);
#ifdef DM_PROFILE
  if (P.debug && lane < 16) { unsigned v = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) if (lane == i) v = g_S.prof[i];
    P.debug[(size_t)env * DM_DEBUG_STRIDE + 352 + lane] = (float)v;
    if (lane < 4) P.debug[(size_t)env * DM_DEBUG_STRIDE + 368 + lane] = (float)g_S.prof_stage[lane]; }
#endif
  // ---------------------------------------------------------------- state write-back
  if (lane < DMK_NQ) st[DMS_QPOS + lane] = S.qpos[lane];
  if (lane < DMK_NV) { st[DMS_QVEL + lane] = S.qvel[lane]; st[DMS_WARM + lane] = S.warm[lane]; }
  if (lane < DMK_NU) st[DMS_CTRL + lane] = S.ctrl[lane];
  if (lane == 0) { sti[DMS_IDX] = idx_curr; sti[DMS_EPLEN] = ep_len; st[DMS_EPREW] = ep_rew; sti[DMS_RCNT] = rcnt; }
  if (TASK && lane == 0) sti[DMS_CLIP] = motion;
  if (P.f8 && lane == 0) { sti[DMS_F8R] = (int)f8r; sti[DMS_F8L] = (int)f8l; }
  if (P.cost && lane == 0) P.cost[env] = work;
}

extern "C" __global__ void __launch_bounds__(64 * DMK_ENVS_PER_BLOCK, DMK_WAVES_PER_SIMD) dm_step_kernel(DmLaunch P) {
  step_body<0>(P);
}
extern "C" __global__ void __launch_bounds__(64 * DMK_ENVS_PER_BLOCK, DMK_WAVES_PER_SIMD) dm_step_combined_kernel(DmLaunch P) {
  step_body<1>(P);
}
// Same body compiled for three waves per SIMD (168 VGPRs, 93 spilled): dm_step launches it from 3 072 envs up
// (dm_abi.hip: launch()); below that the batch does not fill two waves per SIMD and the 256-VGPR build is faster.
extern "C" __global__ void __launch_bounds__(64 * DMK_ENVS_PER_BLOCK, 3) dm_step_kernel_w3(DmLaunch P) {
  step_body<0>(P);
}
extern "C" __global__ void __launch_bounds__(64 * DMK_ENVS_PER_BLOCK, 3) dm_step_combined_kernel_w3(DmLaunch P) {
  step_body<1>(P);
}
#ifdef DM_EXPERIMENT_W4
// Experiment: four waves per SIMD = 128 VGPRs.  Never selected by dm_step; DM_WAVES=4 forces it.
extern "C" __global__ void __launch_bounds__(64 * DMK_ENVS_PER_BLOCK, 4) dm_step_kernel_w4(DmLaunch P) {
  step_body<0>(P);
}
#endif

// Uniform random actions in [-2, 2) for bench.py config 2 (same generator as the oracle driver).
extern "C" __global__ void dm_fill_actions_kernel(float *actions, int n, uint64_t seed, uint32_t step) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * DMK_NU) return;
  int env = i / DMK_NU, j = i % DMK_NU;
  actions[i] = -2.0f + 4.0f * (float)(dm_hash32(seed, env, step, j) >> 8) * (1.0f / 16777216.0f);
}
