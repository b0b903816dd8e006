// Policy side of a rollout step as ONE launch (gfx950): both trunks of the actor-critic MLP, the Gaussian sampling
// head and the rollout-buffer writes of the policy's outputs.
//
// Reference: src/sb3_ppo.py:307-313 -> [EXT] SB3 collect_rollouts: per env step `policy.forward(obs)` =
//   mean  = action_net(tanh(L2(tanh(L1(obs)))))      (pi trunk,  D -> H1 -> H2 -> A)
//   value = value_net (tanh(L2'(tanh(L1'(obs)))))    (vf trunk,  D -> H1 -> H2 -> 1)
//   act = mean + exp(log_std) * N(0,1); logp; clipped action -> env.step; store (obs, act, value, logp).
// With the library GEMMs this is 10 small kernels + the sampling and store kernels per env step, ~0.15 ms of launch-
// bound work next to the 0.31 ms physics step.  Here a workgroup of four waves owns 32 batch rows of one trunk and
// carries them through all three layers: activations stay in LDS (fp32), weights stream from L2 in MFMA operand
// order (dm_policy_pack, re-run only when the weights change), every product is v_mfma_f32_32x32x2f32 (fp32 in,
// fp32 accumulate — same arithmetic type as the library path).  Layer 2 is produced in chunks of 128 columns which
// layer 3 consumes immediately (split-K over the four waves), so [1024,512] fits the 160 KB LDS as well.
//
// Operand layout (wave64, 32x32x2): A lane (r = lane & 31, h = lane >> 5) supplies X[row r][k = h], B lane supplies
// W[neuron r][k = h]; acc[j] is (row (j >> 2) * 8 + h * 4 + (j & 3), neuron r).  Any permutation of k is a valid
// contraction order as long as A and B agree, so each lane fetches FOUR consecutive k (one 16-byte LDS read, one
// 16-byte global read) and the four MFMAs of a k-block of 8 use k = 4h + c, c = 0..3.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int POL_R = 32;          // batch rows per workgroup
constexpr int POL_THREADS = 256;   // four waves
constexpr int POL_CHUNK = 128;     // layer-2 columns per chunk (one 32-column tile per wave)
constexpr int POL_PAD = 4;         // LDS row padding in floats: keeps 16-byte alignment and spreads rows over the banks

typedef float pol_f16v __attribute__((ext_vector_type(16)));

struct PolArgs {
  const float *obs;                 // [N, D]
  int N, D, Dp, H1, H2, A;
  const float4 *pk[2];              // packed weights of the pi / vf trunk (dm_policy_pack)
  const float *b1[2], *b2[2], *b3[2];
  const float *log_std, *lo, *hi;
  unsigned long long seed;
  const unsigned *counter;
  unsigned draw_offset;
  int deterministic;
  float *mean_out, *act, *act_env, *logp, *val, *obs_copy;
};

__device__ __forceinline__ unsigned pol_hash32(unsigned long long seed, unsigned a, unsigned b, unsigned c) {
  unsigned long long x = seed ^ ((unsigned long long)a * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)b * 0xBF58476D1CE4E5B9ull) ^
                         ((unsigned long long)c * 0x94D049BB133111EBull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (unsigned)(x >> 32);
}

// acc += X[32 x 8 (kb1 - kb0)] W^T for one 32-neuron tile; xs = LDS activations (row stride sx), P = the tile's packed
// weights (64 float4 per k-block of 8).
template <int U>
__device__ __forceinline__ void pol_load(const float4 *p, const float *xrow, int kb, float4 (&w)[U], float4 (&a)[U]) {
#pragma unroll
  for (int u = 0; u < U; u++) w[u] = p[(size_t)(kb + u) * 64];
#pragma unroll
  for (int u = 0; u < U; u++) a[u] = *reinterpret_cast<const float4 *>(xrow + (kb + u) * 8);
}
template <int U>
__device__ __forceinline__ void pol_mfma(const float4 (&w)[U], const float4 (&a)[U], pol_f16v &acc) {
#pragma unroll
  for (int u = 0; u < U; u++) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, w[u].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, w[u].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, w[u].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, w[u].w, acc, 0, 0, 0);
  }
}
// first weight batch of a tile, issued early (before the barrier that publishes the tile's activations: the weight
// stream does not depend on them), consumed by pol_tile_pre
template <int U>
__device__ __forceinline__ void pol_prefetch(const float4 *P, int lane, int kb0, float4 (&w)[U]) {
#pragma unroll
  for (int u = 0; u < U; u++) w[u] = P[(size_t)(kb0 + u) * 64 + lane];
}
template <int U, bool PRE>
__device__ __forceinline__ void pol_tile_impl(const float *xs, int sx, const float4 *P, int lane, int kb0, int kb1, pol_f16v &acc,
                                              float4 (&wA)[U]) {
  const float *xrow = xs + (lane & 31) * sx + 4 * (lane >> 5);
  const float4 *p = P + lane;
  const int nb = (kb1 - kb0) / U;
  // ping-pong over two register sets (no copies): the loads of batch it + 1 are in flight under the 4 U MFMAs of batch
  // it, and the MFMAs wait only for their own batch (s_waitcnt vmcnt(U))
  float4 aA[U], wB[U], aB[U];
  if (nb > 0) {
    if (PRE) {
#pragma unroll
      for (int u = 0; u < U; u++) aA[u] = *reinterpret_cast<const float4 *>(xrow + (kb0 + u) * 8);
    } else {
      pol_load<U>(p, xrow, kb0, wA, aA);
    }
  }
  int it = 0;
  for (; it + 2 <= nb; it += 2) {
    pol_load<U>(p, xrow, kb0 + (it + 1) * U, wB, aB);
    pol_mfma<U>(wA, aA, acc);
    if (it + 2 < nb) pol_load<U>(p, xrow, kb0 + (it + 2) * U, wA, aA);
    pol_mfma<U>(wB, aB, acc);
  }
  if (it < nb) pol_mfma<U>(wA, aA, acc);
  for (int kb = kb0 + nb * U; kb < kb1; kb++) {
    const float4 w = p[(size_t)kb * 64];
    const float4 a = *reinterpret_cast<const float4 *>(xrow + kb * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc, 0, 0, 0);
  }
}
template <int U>
__device__ __forceinline__ void pol_tile(const float *xs, int sx, const float4 *P, int lane, int kb0, int kb1, pol_f16v &acc) {
  float4 wA[U];
  pol_tile_impl<U, false>(xs, sx, P, lane, kb0, kb1, acc, wA);
}
// the first batch's weights (kb0 .. kb0 + U - 1; requires kb1 - kb0 >= U) were fetched by pol_prefetch<U>
template <int U>
__device__ __forceinline__ void pol_tile_pre(const float *xs, int sx, const float4 *P, int lane, int kb0, int kb1, pol_f16v &acc,
                                             float4 (&wpre)[U]) {
  pol_tile_impl<U, true>(xs, sx, P, lane, kb0, kb1, acc, wpre);
}

// hs[row][o0 + r] = tanh(acc + bias)
__device__ __forceinline__ void pol_store_tanh(const pol_f16v &acc, const float *bias, int o0, float *hs, int sh, int col0, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const float b = bias[o0 + r];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int row = (j >> 2) * 8 + h * 4 + (j & 3);
    hs[row * sh + col0 + r] = tanhf(acc[j] + b);
  }
}

__global__ void __launch_bounds__(POL_THREADS) pol_forward_kernel(PolArgs a) {
  extern __shared__ __align__(16) float pol_lds[];
  const int trunk = blockIdx.y, b0 = blockIdx.x * POL_R;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int sx = a.Dp + POL_PAD, s1 = a.H1 + POL_PAD, sc = POL_CHUNK + POL_PAD;
  const int region0 = POL_R * (sx > sc ? sx : sc);
  float *xs = pol_lds;                 // observations; dead after layer 1
  float *h2c = pol_lds;                // layer-2 chunk (aliases xs)
  float *red = pol_lds;                // layer-3 partial sums of the four waves (aliases h2c after the last chunk)
  float *h1 = pol_lds + region0;

  // ---- observations -> LDS (zero-padded to Dp columns / 32 rows); the pi workgroup also files them in the rollout buffer
  for (int i = tid; i < POL_R * a.D; i += POL_THREADS) {
    const int row = i / a.D, c = i - row * a.D;
    const bool ok = (b0 + row) < a.N;
    const float v = ok ? a.obs[(size_t)b0 * a.D + i] : 0.f;
    xs[row * sx + c] = v;
  }
  for (int i = tid; i < POL_R * (a.Dp - a.D); i += POL_THREADS) {
    const int row = i / (a.Dp - a.D), c = a.D + i - row * (a.Dp - a.D);
    xs[row * sx + c] = 0.f;
  }
  __syncthreads();

  const int KB1 = a.Dp >> 3, T1 = a.H1 >> 5, KB2 = a.H1 >> 3, T2 = a.H2 >> 5, KB3 = a.H2 >> 3;
  const float4 *P1 = a.pk[trunk];
  const float4 *P2 = P1 + (size_t)T1 * KB1 * 64;
  const float4 *P3 = P2 + (size_t)T2 * KB2 * 64;

  // ---- layer 1: tiles round-robin over the waves
  for (int to = wave; to < T1; to += 4) {
    pol_f16v acc;
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    pol_tile<3>(xs, sx, P1 + (size_t)to * KB1 * 64, lane, 0, KB1, acc);
    pol_store_tanh(acc, a.b1[trunk], to * 32, h1, s1, to * 32, lane);
  }
  __syncthreads();

  // ---- layer 2 in chunks of four tiles (one per wave), layer 3 split-K over the waves on each chunk
  pol_f16v acc3;
#pragma unroll
  for (int j = 0; j < 16; j++) acc3[j] = 0.f;
  for (int c0 = 0; c0 < T2; c0 += 4) {
    const int nt = (T2 - c0) < 4 ? (T2 - c0) : 4;
    if (wave < nt) {
      pol_f16v acc;
#pragma unroll
      for (int j = 0; j < 16; j++) acc[j] = 0.f;
      pol_tile<8>(h1, s1, P2 + (size_t)(c0 + wave) * KB2 * 64, lane, 0, KB2, acc);
      pol_store_tanh(acc, a.b2[trunk], (c0 + wave) * 32, h2c, sc, wave * 32, lane);
    }
    __syncthreads();
    // the chunk holds 4 nt k-blocks of layer 3; wave w takes nt of them
    {
      const float *xrow = h2c + (lane & 31) * sc + 4 * (lane >> 5);
      const float4 *p = P3 + (size_t)(c0 * 4) * 64 + lane;
      for (int kb = wave * nt; kb < (wave + 1) * nt; kb++) {
        const float4 w = p[(size_t)kb * 64];
        const float4 x = *reinterpret_cast<const float4 *>(xrow + kb * 8);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, w.x, acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, w.y, acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, w.z, acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, w.w, acc3, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int row = (j >> 2) * 8 + h * 4 + (j & 3);
      red[wave * 1024 + row * 32 + r] = acc3[j];
    }
  }
  __syncthreads();

  // ---- heads
  if (trunk == 1) {
    if (tid < POL_R && (b0 + tid) < a.N)
      a.val[b0 + tid] = red[tid * 32] + red[1024 + tid * 32] + red[2048 + tid * 32] + red[3072 + tid * 32] + a.b3[1][0];
    // the observation copy for the rollout buffer, at the very end of the (lighter) value workgroup: a global store followed by
    // loads makes the compiler wait for the write acknowledgement (possible alias), which at the top of the kernel would
    // sit in front of the first weight loads
    if (a.obs_copy)
      for (int i = tid; i < POL_R * a.D; i += POL_THREADS)
        if ((b0 + i / a.D) < a.N) a.obs_copy[(size_t)b0 * a.D + i] = a.obs[(size_t)b0 * a.D + i];
    return;
  }
  // eight threads per row, two action pairs each (A <= 32): same draws as ppo_sample_kernel (seed, env, counter, index)
  const int row = tid >> 3, q = tid & 7, e = b0 + row;
  const bool ok = e < a.N;
  const unsigned ctr = a.counter[0] + a.draw_offset;
  float lp = 0.f;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int j = 2 * (q + 8 * s);
    if (j < a.A) {
      float eps[2] = {0.f, 0.f};
      if (!a.deterministic) {
        const float u1 = ((float)(pol_hash32(a.seed, (unsigned)e, ctr, (unsigned)j) >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
        const float u2 = (float)(pol_hash32(a.seed, (unsigned)e, ctr, (unsigned)j + 1u) >> 8) * (1.0f / 16777216.0f);
        const float rad = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        eps[0] = rad * cs;
        eps[1] = rad * sn;
      }
      for (int t = 0; t < 2 && j + t < a.A; t++) {
        const int c = j + t;
        const float m = red[row * 32 + c] + red[1024 + row * 32 + c] + red[2048 + row * 32 + c] + red[3072 + row * 32 + c] + a.b3[0][c];
        const float ls = a.log_std[c];
        const float v = m + expf(ls) * eps[t];
        if (ok) {
          if (a.mean_out) a.mean_out[(size_t)e * a.A + c] = m;
          a.act[(size_t)e * a.A + c] = v;
          a.act_env[(size_t)e * a.A + c] = fminf(fmaxf(v, a.lo[c]), a.hi[c]);
        }
        lp += -0.5f * eps[t] * eps[t] - ls - 0.9189385332046727f;
      }
    }
  }
  lp += __shfl_xor(lp, 1);
  lp += __shfl_xor(lp, 2);
  lp += __shfl_xor(lp, 4);
  if (q == 0 && ok) a.logp[e] = lp;
}

// W (element (o, k) at W[o so + k sk]; nn.Linear [O x K] row-major is so = K, sk = 1, its transpose so = 1, sk = O) ->
// P[tile][k-block][lane] float4 = W(32 tile + (lane & 31), 8 kb + 4 (lane >> 5) + 0..3), zero outside O x K
__device__ __forceinline__ void pol_pack_one(const float *W, int O, int K, int so, int sk, int tiles, int KB, float4 *P, int i) {
  if (i >= tiles * KB * 64) return;
  const int lane = i & 63, kb = (i >> 6) % KB, to = (i >> 6) / KB;
  const int o = to * 32 + (lane & 31), k = kb * 8 + 4 * (lane >> 5);
  float v[4];
#pragma unroll
  for (int c = 0; c < 4; c++) v[c] = (o < O && k + c < K) ? W[(size_t)o * so + (size_t)(k + c) * sk] : 0.f;
  P[i] = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ void pol_pack_kernel(const float *W, int O, int K, int tiles, int KB, float4 *P) {
  pol_pack_one(W, O, K, K, 1, tiles, KB, P, blockIdx.x * blockDim.x + threadIdx.x);
}

inline int pol_dp(int D) { return (D + 7) & ~7; }
inline bool pol_dims_ok(int D, int H1, int H2, int A) {
  return D >= 1 && H1 >= 32 && H2 >= 32 && (H1 % 32) == 0 && (H2 % 32) == 0 && A >= 1 && A <= 32;
}
inline size_t pol_lds_bytes(int D, int H1) {
  const int sx = pol_dp(D) + POL_PAD, sc = POL_CHUNK + POL_PAD;
  return (size_t)(POL_R * (sx > sc ? sx : sc) + POL_R * (H1 + POL_PAD)) * sizeof(float);
}

}  // namespace

extern "C" long long dm_policy_packed_floats(int D, int H1, int H2, int A) {
  if (!pol_dims_ok(D, H1, H2, A)) return -22;
  return 256ll * ((long long)(H1 / 32) * (pol_dp(D) / 8) + (long long)(H2 / 32) * (H1 / 8) + (long long)(H2 / 8));
}

extern "C" int dm_policy_pack(const float *W1, const float *W2, const float *W3, int D, int H1, int H2, int A, float *packed,
                              void *stream) {
  if (!W1 || !W2 || !W3 || !packed || !pol_dims_ok(D, H1, H2, A)) return -22;
  if (pol_lds_bytes(D, H1) > 160 * 1024) return -22;
  hipStream_t s = (hipStream_t)stream;
  float4 *P1 = reinterpret_cast<float4 *>(packed);
  const int T1 = H1 / 32, KB1 = pol_dp(D) / 8, T2 = H2 / 32, KB2 = H1 / 8, KB3 = H2 / 8;
  float4 *P2 = P1 + (size_t)T1 * KB1 * 64, *P3 = P2 + (size_t)T2 * KB2 * 64;
  hipLaunchKernelGGL(pol_pack_kernel, dim3((T1 * KB1 * 64 + 255) / 256), dim3(256), 0, s, W1, H1, D, T1, KB1, P1);
  hipLaunchKernelGGL(pol_pack_kernel, dim3((T2 * KB2 * 64 + 255) / 256), dim3(256), 0, s, W2, H2, H1, T2, KB2, P2);
  hipLaunchKernelGGL(pol_pack_kernel, dim3((KB3 * 64 + 255) / 256), dim3(256), 0, s, W3, A, H2, 1, KB3, P3);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int dm_policy_forward(const float *obs, int N, int D, int H1, int H2, int A, const float *pi_packed, const float *pi_b1,
                                 const float *pi_b2, const float *pi_b3, const float *vf_packed, const float *vf_b1,
                                 const float *vf_b2, const float *vf_b3, const float *log_std, unsigned long long seed,
                                 const unsigned *counter, unsigned draw_offset, int deterministic, const float *lo, const float *hi,
                                 float *mean_out, float *act, float *act_env, float *logp, float *val, float *obs_copy, void *stream) {
  if (!obs || N < 1 || !pol_dims_ok(D, H1, H2, A) || !pi_packed || !pi_b1 || !pi_b2 || !pi_b3 || !vf_packed || !vf_b1 || !vf_b2 ||
      !vf_b3 || !log_std || !counter || !lo || !hi || !act || !act_env || !logp || !val)
    return -22;
  if ((reinterpret_cast<uintptr_t>(pi_packed) | reinterpret_cast<uintptr_t>(vf_packed)) & 15) return -22;
  const size_t lds = pol_lds_bytes(D, H1);
  if (lds > 160 * 1024) return -22;
  // hipFuncSetAttribute applies to the CURRENT device: remember the raised limit per device ordinal
  static size_t lds_allowed[64];
  int dev_id = 0;
  if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) return -5;
  const size_t allowed = lds_allowed[dev_id] ? lds_allowed[dev_id] : (size_t)64 * 1024;
  if (lds > allowed) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(pol_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return -5;
    lds_allowed[dev_id] = lds;
  }
  PolArgs a;
  a.obs = obs; a.N = N; a.D = D; a.Dp = pol_dp(D); a.H1 = H1; a.H2 = H2; a.A = A;
  a.pk[0] = reinterpret_cast<const float4 *>(pi_packed); a.pk[1] = reinterpret_cast<const float4 *>(vf_packed);
  a.b1[0] = pi_b1; a.b2[0] = pi_b2; a.b3[0] = pi_b3; a.b1[1] = vf_b1; a.b2[1] = vf_b2; a.b3[1] = vf_b3;
  a.log_std = log_std; a.lo = lo; a.hi = hi; a.seed = seed; a.counter = counter; a.draw_offset = draw_offset;
  a.deterministic = deterministic;
  a.mean_out = mean_out; a.act = act; a.act_env = act_env; a.logp = logp; a.val = val; a.obs_copy = obs_copy;
  hipLaunchKernelGGL(pol_forward_kernel, dim3((N + POL_R - 1) / POL_R, 2), dim3(POL_THREADS), lds, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
