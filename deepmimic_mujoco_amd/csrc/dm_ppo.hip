// Fused PPO clipped-surrogate loss, forward and backward in one pass (gfx950).
//
// Replaces, in the learner of the rollout loop (reference: src/sb3_ppo.py:307-313 -> [EXT] SB3 PPO.train,
// DiagGaussianDistribution.log_prob / entropy, F.mse_loss), the ~70 elementwise / reduction kernels PyTorch
// launches per minibatch between the policy head and the loss scalar — the optimizer step of the reference's
// [256,128] network is launch-bound.  Semantics (SB3 defaults): per-minibatch advantage normalisation
// (adv - mean) / (std_unbiased + 1e-8); ratio = exp(logp - old_logp);
//   policy_loss  = -mean(min(adv * ratio, adv * clamp(ratio, 1 - eps, 1 + eps)))
//   value_loss   = mean((ret - value)^2)
//   entropy_loss = -mean(entropy),  entropy = sum_j (0.5 + 0.5 log(2 pi) + log_std_j)
//   loss = policy_loss + ent_coef * entropy_loss + vf_coef * value_loss
// Outputs the loss terms and d loss / d (mean, log_std, value).  One thread per sample; B x A <= 4096 x 28 floats, so
// the kernel is latency-, not bandwidth-bound: what matters is that it is two launches instead of seventy.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int PPO_MAXA = 32;     // action dimensions supported (28 here)
constexpr int PPO_BLOCK = 256;

__device__ __forceinline__ float ppo_wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum of `v`, result valid in thread 0
__device__ __forceinline__ float ppo_block_sum(float v, float *red) {
  v = ppo_wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
  return s;
}

// stats[0] = mean(adv), stats[1] = 1 / (std_unbiased(adv) + 1e-8); also clears the accumulators of the main kernel
__device__ __forceinline__ void ppo_prepare_body(const float *adv, int B, int normalize, float *stats, float *out8, float *grad_log_std,
                                                 int A) {
  __shared__ float red[16];
  float s = 0;
  for (int i = threadIdx.x; i < B; i += blockDim.x) s += adv[i];
  const float tot = ppo_block_sum(s, red);
  __shared__ float mean_s;
  if (threadIdx.x == 0) mean_s = tot / (float)B;
  __syncthreads();
  const float mean = mean_s;
  float q = 0;
  for (int i = threadIdx.x; i < B; i += blockDim.x) { const float d = adv[i] - mean; q += d * d; }
  const float qq = ppo_block_sum(q, red);
  if (threadIdx.x == 0) {
    if (normalize && B > 1) { stats[0] = mean; stats[1] = 1.0f / (sqrtf(qq / (float)(B - 1)) + 1e-8f); }
    else { stats[0] = 0.f; stats[1] = 1.f; }
  }
  if (threadIdx.x < 8) out8[threadIdx.x] = 0.f;
  if (grad_log_std && (int)threadIdx.x < A) grad_log_std[threadIdx.x] = 0.f;
}
__global__ void ppo_prepare_kernel(const float *adv, int B, int normalize, float *stats, float *out8, float *grad_log_std, int A) {
  ppo_prepare_body(adv, B, normalize, stats, out8, grad_log_std, A);
}

// One HALF-WAVE per minibatch row, lane = action index: the [B x A] arrays (mean, act, grad_mean) are read and written as
// contiguous 4 A-byte row segments (the first version gave every lane its own row: 64 cache lines per load instruction,
// 27 us at B = 4096, on the critical path where the two trunks of the library-GEMM learner join).  The row sum of
// ((a - mu) / sigma)^2 is a 32-lane butterfly; the per-action sums for d loss / d log_std stay in the lanes that own them.
__global__ void __launch_bounds__(PPO_BLOCK) ppo_loss_kernel(const float *mean, const float *log_std, const float *value, const float *act,
                                const float *old_logp, const float *adv, const float *ret, int B, int A, float clip,
                                float vf_coef, float ent_coef, const float *stats, float *grad_mean, float *grad_log_std,
                                float *grad_value, float *out8) {
  const int j = threadIdx.x & 31, hw = threadIdx.x >> 5, nhw_blk = PPO_BLOCK / 32;
  const bool ja = j < A;
  const float ls = ja ? log_std[j] : 0.f, iv = ja ? expf(-2.f * ls) : 0.f;
  float sum_ls = ls;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) sum_ls += __shfl_xor(sum_ls, o);
  const float invB = 1.0f / (float)B, lconst = -sum_ls - 0.5f * 1.8378770664093453f * (float)A;   // log(2 pi)
  const float amean = stats[0], ainv = stats[1];
  float g_ls = 0.f, pg = 0.f, vl = 0.f, kl = 0.f, cf = 0.f;
  for (int b = blockIdx.x * nhw_blk + hw; b < B; b += gridDim.x * nhw_blk) {
    const float d = ja ? act[(size_t)b * A + j] - mean[(size_t)b * A + j] : 0.f;
    const float z2 = d * d * iv;                               // ((a - mu) / sigma)^2, reused for d / d log_std
    float zs = z2;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) zs += __shfl_xor(zs, o);
    const float logp = -0.5f * zs + lconst;
    const float a_n = (adv[b] - amean) * ainv;
    const float lr = logp - old_logp[b];
    const float ratio = expf(lr);
    const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    const float p1 = a_n * ratio, p2 = a_n * rc;
    // d min(p1, p2) / d ratio: inside the clip range both branches carry a_n (ties split evenly by autograd: same sum);
    // outside, only the unclipped branch has a gradient and only when it is the smaller one
    const bool inside = (ratio >= 1.f - clip) && (ratio <= 1.f + clip);
    const float dr = (inside || p1 < p2) ? a_n : 0.f;
    const float dlogp = -invB * dr * ratio;
    if (ja) grad_mean[(size_t)b * A + j] = dlogp * d * iv;
    g_ls += ja ? dlogp * (z2 - 1.f) : 0.f;
    if (j == 0) {
      const float dv = value[b] - ret[b];
      grad_value[b] = vf_coef * 2.f * invB * dv;
      pg += -fminf(p1, p2);
      vl += dv * dv;
      kl += (ratio - 1.f) - lr;                                // SB3 approx_kl estimator
      cf += (fabsf(ratio - 1.f) > clip) ? 1.f : 0.f;
    }
  }
  // block reduction over the half-waves: per-action sums stay per lane j, the four scalars ride in lanes 0..3 of a fifth row
  __shared__ float part[PPO_BLOCK / 32][36];
  part[hw][j] = g_ls;
  if (j == 0) { part[hw][32] = pg; part[hw][33] = vl; part[hw][34] = kl; part[hw][35] = cf; }
  __syncthreads();
  const int q = threadIdx.x;
  if (q < 36) {
    float t = 0;
#pragma unroll
    for (int i = 0; i < PPO_BLOCK / 32; i++) t += part[i][q];
    if (q < A) atomicAdd(&grad_log_std[q], t);
    else if (q == 32) atomicAdd(&out8[1], t * invB);
    else if (q == 33) atomicAdd(&out8[2], t * invB);
    else if (q == 34) atomicAdd(&out8[4], t * invB);
    else if (q == 35) atomicAdd(&out8[5], t * invB);
  }
}

// loss = pg + ent_coef * (-entropy) + vf_coef * vl ; d loss / d log_std_j gets -ent_coef from the entropy term
__global__ void ppo_finish_kernel(const float *log_std, int A, float vf_coef, float ent_coef, const float *stats, float *grad_log_std,
                                  float *out8) {
  if (threadIdx.x == 0) {
    float ent = 0;
    for (int j = 0; j < A; j++) ent += 0.5f + 0.5f * 1.8378770664093453f + log_std[j];
    out8[3] = ent;
    out8[0] = out8[1] + vf_coef * out8[2] - ent_coef * ent;
    out8[6] = stats[0];
    out8[7] = stats[1];
  }
  if ((int)threadIdx.x < A) grad_log_std[threadIdx.x] -= ent_coef;
}

}  // namespace

// C-ABI (include/deepmimic_hip.h).  All pointers are device pointers; `scratch` holds >= 2 floats.
extern "C" int dm_ppo_loss(const float *mean, const float *log_std, const float *value, const float *act, const float *old_logp,
                           const float *adv, const float *ret, int B, int A, float clip_range, float vf_coef, float ent_coef,
                           int normalize_advantage, float *grad_mean, float *grad_log_std, float *grad_value, float *out8,
                           float *scratch, void *stream) {
  if (!mean || !log_std || !value || !act || !old_logp || !adv || !ret || !grad_mean || !grad_log_std || !grad_value || !out8 ||
      !scratch || B < 1 || A < 1 || A > PPO_MAXA)
    return -22;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ppo_prepare_kernel, dim3(1), dim3(1024), 0, s, adv, B, normalize_advantage, scratch, out8, grad_log_std, A);
  const int rows_per_block = 8 * (PPO_BLOCK / 32);            // eight rows per half-wave
  int loss_blocks = (B + rows_per_block - 1) / rows_per_block;
  if (loss_blocks > 256) loss_blocks = 256;
  hipLaunchKernelGGL(ppo_loss_kernel, dim3(loss_blocks), dim3(PPO_BLOCK), 0, s, mean, log_std, value, act,
                     old_logp, adv, ret, B, A, clip_range, vf_coef, ent_coef, scratch, grad_mean, grad_log_std, grad_value, out8);
  hipLaunchKernelGGL(ppo_finish_kernel, dim3(1), dim3(64), 0, s, log_std, A, vf_coef, ent_coef, scratch, grad_log_std, out8);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// ---------------------------------------------------------------------------------------------------------------
// Weight and bias gradient of a linear layer of the policy / value MLP:  dW[o][i] = sum_b dY[b][o] X[b][i],
// db[o] = sum_b dY[b][o], for a minibatch of B = 4096 rows and layers no larger than 256 x 256.  The output is tiny
// and the reduction long, so the library GEMM walks all of K on two dozen workgroups (25 us per layer, five layers per
// optimizer step).  Here every wave owns one 32 x 32 output tile and one slice of the batch: v_mfma_f32_32x32x2f32 with
// A[row = o][k = b] = dY[b][o0 + row] and B[k = b][col = i] = X[b][i0 + col] (both coalesced 128-byte row reads),
// partial tiles are added to dW with float atomics (dW and db zeroed by the caller on the same stream).
namespace {

typedef float ppo_f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void ppo_wgrad_body(const float *dY, const float *X, float *dW, float *db, int B, int O, int I, int kchunk,
                                               int bx, int by, int bz) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int i0 = bx * 32, o0 = by * 32, b0 = bz * kchunk;
  const bool oa = (o0 + r) < O, ia = (i0 + r) < I;
  const float *pa = dY + (size_t)(b0 + h) * O + (o0 + r);
  const float *px = X + (size_t)(b0 + h) * I + (i0 + r);
  ppo_f16v acc;
#pragma unroll
  for (int j = 0; j < 16; j++) acc[j] = 0.f;
  float dbacc = 0.f;
  constexpr int U = 32;                  // 64 loads per lane in flight: the kernel is latency-, not MFMA-bound
  for (int b = 0; b < kchunk; b += 2 * U) {
    float a[U], x[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      a[u] = oa ? pa[(size_t)(b + 2 * u) * O] : 0.f;
      x[u] = ia ? px[(size_t)(b + 2 * u) * I] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], x[u], acc, 0, 0, 0);
      dbacc += a[u];
    }
  }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int row = (j >> 2) * 8 + h * 4 + (j & 3);
    if ((o0 + row) < O && ia) atomicAdd(&dW[(size_t)(o0 + row) * I + i0 + r], acc[j]);
  }
  if (bx == 0) {
    const float v = dbacc + __shfl_xor(dbacc, 32);
    if (h == 0 && oa) atomicAdd(&db[o0 + r], v);
  }
}

__global__ void __launch_bounds__(64) ppo_wgrad_kernel(const float *dY, const float *X, float *dW, float *db, int B, int O, int I,
                                                       int kchunk) {
  ppo_wgrad_body(dY, X, dW, db, B, O, I, kchunk, blockIdx.x, blockIdx.y, blockIdx.z);
}

}  // namespace

// dW [O x I] and db [O] must be zero on entry (stream-ordered).  B must be a multiple of 64.
extern "C" int dm_linear_wgrad(const float *dY, const float *X, float *dW, float *db, int B, int O, int I, void *stream) {
  if (!dY || !X || !dW || !db || B < 64 || (B % 64) != 0 || O < 1 || I < 1) return -22;
  const int tiles = ((O + 31) / 32) * ((I + 31) / 32);
  int splitk = 1;
  while (splitk * 2 * tiles <= 1024 && B / (splitk * 2) >= 64 && (B % (splitk * 2 * 64)) == 0) splitk *= 2;
  const int kchunk = B / splitk;                       // multiple of 64 (one unrolled group of the kernel)
  if (kchunk % 64 != 0) return -22;
  hipLaunchKernelGGL(ppo_wgrad_kernel, dim3((I + 31) / 32, (O + 31) / 32, splitk), dim3(64), 0, (hipStream_t)stream, dY, X, dW, db, B,
                     O, I, kchunk);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// ---------------------------------------------------------------------------------------------------------------
// Column sums of a row-major [B x O] matrix, ACCUMULATED into out[O] (zeroed by the caller): the bias gradient
// db = sum_b dY[b][:] of the layers whose weight gradient stays on the library GEMM (beyond 256 units).  The framework's
// generic reduction reads the 16 MB of a [4096 x 1024] dY at 0.7 TB/s (22 us, five times per optimizer step of the
// [1024,512] net); here a wave reads 256 contiguous bytes per row and a block owns 64 columns x one slice of the rows.
namespace {
__global__ void __launch_bounds__(256) ppo_colsum_kernel(const float *Y, int B, int O, int rows_per_block, float *out) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(B, r0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < O) {
    int r = r0 + g;
    for (; r + 12 < r1; r += 16) {
      s0 += Y[(size_t)r * O + c];
      s1 += Y[(size_t)(r + 4) * O + c];
      s2 += Y[(size_t)(r + 8) * O + c];
      s3 += Y[(size_t)(r + 12) * O + c];
    }
    for (; r < r1; r += 4) s0 += Y[(size_t)r * O + c];
  }
  red[g][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && c < O) atomicAdd(&out[c], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}
}  // namespace

extern "C" int dm_colsum(const float *Y, int B, int O, float *out, void *stream) {
  if (!Y || !out || B < 1 || O < 1) return -22;
  const int cb = (O + 63) / 64;
  int slices = 1;
  while (slices * 2 * cb <= 1024 && B / (slices * 2) >= 64) slices *= 2;
  const int rpb = (B + slices - 1) / slices;
  hipLaunchKernelGGL(ppo_colsum_kernel, dim3(cb, slices), dim3(256), 0, (hipStream_t)stream, Y, B, O, rpb, out);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// ---------------------------------------------------------------------------------------------------------------
// First layer of a wide policy / value trunk ([1024,512] of BASELINE configs 3-5): Y = tanh(X W^T + b) with X [B x I], I <= 128
// (the observation: 67 / 72 / 85 / 98 wide), W [O x I], Y [B x O].  K is so short that the layer is bound by writing Y; the
// library runs the GEMM at 30 TFLOP/s (unaligned K) and the framework adds a separate 2 x 16 MB tanh pass.  Here a workgroup
// stages a 64-row block of X and a 128-row block of W in LDS (both are CONTIGUOUS in memory: flat 16-byte copies; row stride
// I + 1 odd-ised in LDS = conflict-free column reads), each of its four waves owns 32 x 64 outputs on v_mfma_f32_32x32x2f32,
// and bias + tanh are applied on the accumulators: Y is written once, coalesced, and never read back by an activation pass.
namespace {

constexpr int LT_ROWS = 64, LT_COLS = 128, LT_MAXI = 128;

// tanh without branches (the library routine is ~90 instructions with six branches: 4 M activations of one launch were 11 us
// of VALU issue).  |x| < 0.25: odd Taylor polynomial to x^9 (next term 9e-9 relative); otherwise 1 - 2 / (exp(2x) + 1) on
// v_exp_f32 / v_rcp_f32 (absolute error ~1e-7 where |tanh| >= 0.24; exp overflow gives exactly +-1).
__device__ __forceinline__ float lt_tanh(float x) {
  const float s = x * x;
  const float poly = x * (1.f + s * (-0.33333333f + s * (0.13333333f + s * (-0.053968254f + s * 0.021869488f))));
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);          // exp(2x)
  const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
  return fabsf(x) < 0.25f ? poly : big;
}

__global__ void __launch_bounds__(256) ppo_linear_tanh_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                              const float *__restrict__ bias, float *__restrict__ Y, int B, int O,
                                                              int I, int vec) {
  extern __shared__ float lt_lds[];
  const int ld = I | 1;                                // odd row stride: lanes r = 0..31 hit 32 different banks
  float *xs = lt_lds, *ws = lt_lds + LT_ROWS * ld;
  const int b0 = blockIdx.x * LT_ROWS, o0 = blockIdx.y * LT_COLS;
  const int nx = min(LT_ROWS, B - b0) * I, nw = min(LT_COLS, O - o0) * I;
  const float *xg = X + (size_t)b0 * I, *wg = W + (size_t)o0 * I;
  // both blocks are contiguous and 16-byte aligned in memory (64 | b0, 128 | o0): 16-byte loads, all requested before the
  // first LDS write (a plain copy loop waits for every load before the next)
  const bool odd = I & 1;                              // ld == I: the LDS image is the memory image
  if (vec && nx == LT_ROWS * I && nw == LT_COLS * I) {
    const float4 *x4 = (const float4 *)xg, *w4 = (const float4 *)wg;
    const int n4x = (LT_ROWS * I) >> 2, n4 = n4x + ((LT_COLS * I) >> 2);
    for (int base = 0; base < n4; base += 256 * 16) {  // one round trip up to I = 85, two beyond
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int t = base + u * 256 + (int)threadIdx.x;
        if (t < n4) v[u] = t < n4x ? x4[t] : w4[t - n4x];
      }
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int t = base + u * 256 + (int)threadIdx.x;
        if (t < n4) {
          float *dst = t < n4x ? xs : ws;
          const int e = 4 * (t < n4x ? t : t - n4x);
          if (odd) {
            *(float4 *)(dst + e) = v[u];
          } else {                                     // a 16-byte group may straddle two rows (I even, >= 4)
            const int row = e / I, c = e - row * I;
            const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int cc = c + q, rr = row + (cc >= I);
              dst[rr * ld + (cc >= I ? cc - I : cc)] = vv[q];
            }
          }
        }
      }
    }
  } else {
    auto fill = [&](float *dst, const float *src, int total, int valid) {
      for (int base = 0; base < total; base += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int t = base + u * 256 + (int)threadIdx.x;
          v[u] = t < valid ? src[t] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int t = base + u * 256 + (int)threadIdx.x;
          if (t < total) dst[odd ? t : t + t / I] = v[u];
        }
      }
    };
    fill(xs, xg, LT_ROWS * I, nx);
    fill(ws, wg, LT_COLS * I, nw);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int wr = (w & 1) * 32, wc = (w >> 1) * 64;
  ppo_f16v acc0, acc1;
#pragma unroll
  for (int j = 0; j < 16; j++) acc0[j] = acc1[j] = 0.f;
  const float *xa = xs + (wr + r) * ld + h, *wb0 = ws + (wc + r) * ld + h, *wb1 = ws + (wc + 32 + r) * ld + h;
  const int K2 = I & ~1;
  int k = 0;
  for (; k + 8 <= K2; k += 8) {                        // twelve LDS reads requested, then eight matrix instructions
    float a[4], p0[4], p1[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      a[u] = xa[k + 2 * u];
      p0[u] = wb0[k + 2 * u];
      p1[u] = wb1[k + 2 * u];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], p0[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], p1[u], acc1, 0, 0, 0);
    }
  }
  for (; k < K2; k += 2) {
    const float a = xa[k];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb0[k], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb1[k], acc1, 0, 0, 0);
  }
  if (I & 1) {                                         // odd K: the second half of the last k pair is zero
    const float a = h == 0 ? xa[K2] : 0.f;
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h == 0 ? wb0[K2] : 0.f, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h == 0 ? wb1[K2] : 0.f, acc1, 0, 0, 0);
  }
  const int c0 = o0 + wc + r, c1 = c0 + 32;
  const float bb0 = c0 < O ? bias[c0] : 0.f, bb1 = c1 < O ? bias[c1] : 0.f;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int row = b0 + wr + (j >> 2) * 8 + h * 4 + (j & 3);
    if (row < B) {
      if (c0 < O) Y[(size_t)row * O + c0] = lt_tanh(acc0[j] + bb0);
      if (c1 < O) Y[(size_t)row * O + c1] = lt_tanh(acc1[j] + bb1);
    }
  }
}

// Backward of that layer when its input needs no gradient (the observation): dW[o][i] += sum_b g[b][o] X[b][i] and
// db[o] += sum_b g[b][o] with g = dY * (1 - Y^2) formed on the fly — the tanh-backward pass (read 2 x 16 MB, write 16 MB) and
// the 16 MB intermediate disappear, and dY / Y are read once: a wave owns 32 outputs x ALL input tiles (NT x 32 columns).
template <int NT>
__global__ void __launch_bounds__(512) ppo_tanh_wgrad_kernel(const float *__restrict__ dY, const float *__restrict__ Yt,
                                                             const float *__restrict__ X, float *dW, float *db, int B, int O, int I,
                                                             int rows_per_group) {
  // eight waves share one 32-output tile and split the group's rows, so the global float atomics carry one tile per WORKGROUP
  __shared__ float part[8][1024];
  __shared__ float dbpart[8][32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int o0 = blockIdx.x * 32;
  const int g0 = blockIdx.y * rows_per_group, g1 = min(B, g0 + rows_per_group);
  const int per_wave = (((g1 - g0) + 7) / 8 + 31) & ~31;                 // multiple of the 32-row load group
  const int wb0 = g0 + w * per_wave, wb1 = min(g1, wb0 + per_wave);
  const bool oa = (o0 + r) < O;
  bool ia[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) ia[t] = (t * 32 + r) < I;
  ppo_f16v acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc[t][j] = 0.f;
  float dbacc = 0.f;
  constexpr int U = 16;
  for (int b = wb0; b < wb1; b += 2 * U) {
    float g[U], y[U], x[NT][U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int row = b + 2 * u + h;
      const bool ra = row < wb1;
      g[u] = (oa && ra) ? dY[(size_t)row * O + o0 + r] : 0.f;
      y[u] = (oa && ra) ? Yt[(size_t)row * O + o0 + r] : 0.f;
#pragma unroll
      for (int t = 0; t < NT; t++) x[t][u] = (ia[t] && ra) ? X[(size_t)row * I + t * 32 + r] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float a = g[u] * (1.f - y[u] * y[u]);
#pragma unroll
      for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[t][u], acc[t], 0, 0, 0);
      dbacc += a;
    }
  }
  // the eight partial tiles meet in LDS with plain stores, one input tile at a time, and every thread sums two elements
  // (float atomics on LDS cost 30 us here: eight waves adding into the same 3 072 words)
  const float v = dbacc + __shfl_xor(dbacc, 32);
  if (h == 0) dbpart[w][r] = v;
#pragma unroll
  for (int t = 0; t < NT; t++) {
    if (t) __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int row = (j >> 2) * 8 + h * 4 + (j & 3);
      part[w][row * 32 + r] = acc[t][j];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int e = threadIdx.x + q * 512, row = e >> 5, c = t * 32 + (e & 31);
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 8; k++) sum += part[k][e];
      if ((o0 + row) < O && c < I) atomicAdd(&dW[(size_t)(o0 + row) * I + c], sum);
    }
  }
  if (threadIdx.x < 32 && (o0 + (int)threadIdx.x) < O) {
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) sum += dbpart[k][threadIdx.x];
    atomicAdd(&db[o0 + threadIdx.x], sum);
  }
}

// Backward of tanh for the layers that keep the library GEMMs: dZ = dY * (1 - Y^2) (dZ may alias dY) and db[o] += sum_b dZ[b][o]
// in one pass — the framework's tanh-backward launch plus the separate column-sum read of dZ.
__global__ void __launch_bounds__(256) ppo_tanh_bwd_colsum_kernel(const float *dY, const float *__restrict__ Yt, float *dZ, int B, int O,
                                                                  int rows_per_block, float *db) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(B, r0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < O) {
    int r = r0 + g;
    for (; r + 12 < r1; r += 16) {
      const size_t i0 = (size_t)r * O + c, i1 = i0 + (size_t)4 * O, i2 = i0 + (size_t)8 * O, i3 = i0 + (size_t)12 * O;
      const float g0 = dY[i0], g1 = dY[i1], g2 = dY[i2], g3 = dY[i3];
      const float y0 = Yt[i0], y1 = Yt[i1], y2 = Yt[i2], y3 = Yt[i3];
      const float z0 = g0 * (1.f - y0 * y0), z1 = g1 * (1.f - y1 * y1), z2 = g2 * (1.f - y2 * y2), z3 = g3 * (1.f - y3 * y3);
      dZ[i0] = z0; dZ[i1] = z1; dZ[i2] = z2; dZ[i3] = z3;
      s0 += z0; s1 += z1; s2 += z2; s3 += z3;
    }
    for (; r < r1; r += 4) {
      const size_t i0 = (size_t)r * O + c;
      const float y0 = Yt[i0], z0 = dY[i0] * (1.f - y0 * y0);
      dZ[i0] = z0;
      s0 += z0;
    }
  }
  red[g][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && c < O) atomicAdd(&db[c], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// the same with 16-byte accesses (O a multiple of 4): a block owns 64 columns x 64 rows, a thread four columns x four rows
__global__ void __launch_bounds__(256) ppo_tanh_bwd_colsum4_kernel(const float *dY, const float *__restrict__ Yt, float *dZ, int B, int O,
                                                                   float *db) {
  __shared__ float red[16][64];
  const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cg * 4, r0 = blockIdx.y * 64 + rg;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < O) {
    float4 g[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = r0 + 16 * u;
      if (r < B) {
        g[u] = *(const float4 *)(dY + (size_t)r * O + c);
        y[u] = *(const float4 *)(Yt + (size_t)r * O + c);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = r0 + 16 * u;
      if (r < B) {
        float4 z;
        z.x = g[u].x * (1.f - y[u].x * y[u].x);
        z.y = g[u].y * (1.f - y[u].y * y[u].y);
        z.z = g[u].z * (1.f - y[u].z * y[u].z);
        z.w = g[u].w * (1.f - y[u].w * y[u].w);
        *(float4 *)(dZ + (size_t)r * O + c) = z;
        s.x += z.x; s.y += z.y; s.z += z.z; s.w += z.w;
      }
    }
  }
  red[rg][cg * 4 + 0] = s.x; red[rg][cg * 4 + 1] = s.y; red[rg][cg * 4 + 2] = s.z; red[rg][cg * 4 + 3] = s.w;
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + (int)threadIdx.x < O) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += red[k][threadIdx.x];
    atomicAdd(&db[blockIdx.x * 64 + threadIdx.x], t);
  }
}

}  // namespace

extern "C" int dm_linear_tanh(const float *X, const float *W, const float *bias, float *Y, int B, int O, int I, void *stream) {
  if (!X || !W || !bias || !Y || B < 1 || O < 1 || I < 1 || I > LT_MAXI) return -22;
  const size_t lds = (size_t)(LT_ROWS + LT_COLS) * (I | 1) * sizeof(float);         // <= 99 KB at I = 128
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -5;
  if (!attr_done[dev]) {
    if (hipFuncSetAttribute((const void *)ppo_linear_tanh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) != hipSuccess)
      return -5;
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL(ppo_linear_tanh_kernel, dim3((B + LT_ROWS - 1) / LT_ROWS, (O + LT_COLS - 1) / LT_COLS), dim3(256), lds,
                     (hipStream_t)stream, X, W, bias, Y, B, O, I, ((((uintptr_t)X | (uintptr_t)W) & 15) == 0 && I >= 4) ? 1 : 0);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// dW [O x I] and db [O] must be zero on entry (stream-ordered).  I <= 128.
extern "C" int dm_tanh_linear_wgrad(const float *dY, const float *Y, const float *X, float *dW, float *db, int B, int O, int I,
                                    void *stream) {
  if (!dY || !Y || !X || !dW || !db || B < 1 || O < 1 || I < 1 || I > 128) return -22;
  const int ot = (O + 31) / 32, nt = (I + 31) / 32;
  int groups = 1;                                      // workgroups per output tile: enough to cover the chip once
  while (groups * 2 * ot <= 256 && B / (groups * 2) >= 256) groups *= 2;
  const int rpg = (B + groups - 1) / groups;
  const dim3 grid(ot, groups), block(512);
  hipStream_t s = (hipStream_t)stream;
  switch (nt) {
    case 1: hipLaunchKernelGGL(ppo_tanh_wgrad_kernel<1>, grid, block, 0, s, dY, Y, X, dW, db, B, O, I, rpg); break;
    case 2: hipLaunchKernelGGL(ppo_tanh_wgrad_kernel<2>, grid, block, 0, s, dY, Y, X, dW, db, B, O, I, rpg); break;
    case 3: hipLaunchKernelGGL(ppo_tanh_wgrad_kernel<3>, grid, block, 0, s, dY, Y, X, dW, db, B, O, I, rpg); break;
    default: hipLaunchKernelGGL(ppo_tanh_wgrad_kernel<4>, grid, block, 0, s, dY, Y, X, dW, db, B, O, I, rpg); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// dZ = dY * (1 - Y^2) (dZ may be dY), db[o] += column sums of dZ (db zeroed by the caller, stream-ordered).
extern "C" int dm_tanh_bwd_colsum(const float *dY, const float *Y, float *dZ, float *db, int B, int O, void *stream) {
  if (!dY || !Y || !dZ || !db || B < 1 || O < 1) return -22;
  const int cb = (O + 63) / 64;
  if ((O & 3) == 0 && ((uintptr_t)dY & 15) == 0 && ((uintptr_t)Y & 15) == 0 && ((uintptr_t)dZ & 15) == 0) {
    hipLaunchKernelGGL(ppo_tanh_bwd_colsum4_kernel, dim3(cb, (B + 63) / 64), dim3(256), 0, (hipStream_t)stream, dY, Y, dZ, B, O, db);
    return hipGetLastError() == hipSuccess ? 0 : -5;
  }
  int slices = 1;
  while (slices * 2 * cb <= 2048 && B / (slices * 2) >= 32) slices *= 2;
  const int rpb = (B + slices - 1) / slices;
  hipLaunchKernelGGL(ppo_tanh_bwd_colsum_kernel, dim3(cb, slices), dim3(256), 0, (hipStream_t)stream, dY, Y, dZ, B, O, rpb, db);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// ---------------------------------------------------------------------------------------------------------------
// Minibatch gather of the rollout buffer: out_x[r] = x[idx[r]] for the five per-sample arrays of PPO.train
// (observations [n x D], actions [n x A], advantages, returns, old log-probs) in one launch instead of five
// index_select kernels.  idx is int64 (torch.randperm).
namespace {
__global__ void ppo_gather_kernel(const long long *idx, int B, const float *obs, int D, const float *act, int A, const float *adv,
                                  const float *ret, const float *logp, float *o_obs, float *o_act, float *o_adv, float *o_ret,
                                  float *o_logp) {
  const int r = blockIdx.x;              // one 128-thread block per gathered row
  if (r >= B) return;
  const long long s = idx[r];
  for (int c = threadIdx.x; c < D; c += blockDim.x) o_obs[(size_t)r * D + c] = obs[(size_t)s * D + c];
  for (int c = threadIdx.x; c < A; c += blockDim.x) o_act[(size_t)r * A + c] = act[(size_t)s * A + c];
  if (threadIdx.x == 0) { o_adv[r] = adv[s]; o_ret[r] = ret[s]; o_logp[r] = logp[s]; }
}
}  // namespace

extern "C" int dm_ppo_gather(const long long *idx, int B, const float *obs, int D, const float *act, int A, const float *adv,
                             const float *ret, const float *logp, float *o_obs, float *o_act, float *o_adv, float *o_ret,
                             float *o_logp, void *stream) {
  if (!idx || B < 1 || !obs || !act || !adv || !ret || !logp || !o_obs || !o_act || !o_adv || !o_ret || !o_logp) return -22;
  hipLaunchKernelGGL(ppo_gather_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, idx, B, obs, D, act, A, adv, ret, logp, o_obs,
                     o_act, o_adv, o_ret, o_logp);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// ---------------------------------------------------------------------------------------------------------------
// Gradient-norm clipping + Adam on one flat parameter / gradient / moment buffer (all tensors of the policy are views
// of it): two launches instead of the ~9 of clip_grad_norm_ + torch.optim.Adam.  Semantics of
// torch.nn.utils.clip_grad_norm_(max_norm) followed by torch.optim.Adam(lr, betas, eps) (no weight decay, no amsgrad).
// state[0] = scratch, state[1] = step count (float), state[2 .. 2 + DM_ADAM_PARTIALS) = per-block partial sums of
// squares — all on the device so the sequence can be replayed from a captured hipGraph.  The squared norm is reduced
// in a FIXED order (per-block partials, then one tree every block repeats): data-parallel replicas that hold the same
// all-reduced gradient then compute bit-identical updates (a float-atomic sum would let them drift apart by ulps).
namespace {
// Blocks [0, nsum): the partial sums of squares.  Blocks behind them (optional): the gather of the NEXT minibatch — it depends on
// nothing this optimizer step computes, and as blocks of this launch it costs no launch of its own (a trivial launch is ~4.7 us
// inside a graph; three other ways of hiding it lost: DESIGN 6).  Two rows per block, 128 threads each.
struct AdamGather {
  const long long *idx; int B, D, A;
  const float *obs, *act, *adv, *ret, *logp;
  float *o_obs, *o_act, *o_adv, *o_ret, *o_logp;
};
__global__ void adam_sumsq_kernel(const float *g, int n, float *state, int bump_step, int nsum, AdamGather G) {
  if ((int)blockIdx.x >= nsum) {
    const int r = 2 * ((int)blockIdx.x - nsum) + (threadIdx.x >> 7), t = threadIdx.x & 127;
    if (r >= G.B) return;
    const long long sr = G.idx[r];
    for (int c = t; c < G.D; c += 128) G.o_obs[(size_t)r * G.D + c] = G.obs[(size_t)sr * G.D + c];
    for (int c = t; c < G.A; c += 128) G.o_act[(size_t)r * G.A + c] = G.act[(size_t)sr * G.A + c];
    if (t == 0) { G.o_adv[r] = G.adv[sr]; G.o_ret[r] = G.ret[sr]; G.o_logp[r] = G.logp[sr]; }
    return;
  }
  if (bump_step && blockIdx.x == 0 && threadIdx.x == 0) state[1] += 1.f;    // Adam's step count (read by the update launch)
  // four independent chains (the loads of a thread in flight together; fixed order: a function of n and the grid only)
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  const int stride = nsum * blockDim.x;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const float a = g[i], b = g[i + stride], c = g[i + 2 * stride], d = g[i + 3 * stride];
    s0 += a * a; s1 += b * b; s2 += c * c; s3 += d * d;
  }
  for (; i < n; i += stride) s0 += g[i] * g[i];
  float s = (s0 + s1) + (s2 + s3);
  s = ppo_wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) state[2 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void adam_update_kernel(float *p, const float *g, float *m, float *v, int n, float lr, float b1, float b2, float eps,
                                   float max_norm, float grad_scale, const float *state) {
  // the per-block partials of the norm, same order in every block (and on every rank); four independent chains so that the loads
  // of a lane are in flight together — every block starts with this, and a serial loop put ~4 L2 round trips in front of the update
  float p0 = 0, p1 = 0, p2 = 0, p3 = 0;
  int ip = threadIdx.x & 63;
  for (; ip + 192 < (int)gridDim.x; ip += 256) { p0 += state[2 + ip]; p1 += state[2 + ip + 64]; p2 += state[2 + ip + 128]; p3 += state[2 + ip + 192]; }
  for (; ip < (int)gridDim.x; ip += 64) p0 += state[2 + ip];
  const float part = (p0 + p1) + (p2 + p3);
  // grad_scale: the gradient buffer holds grad_scale^-1 x the gradient (the SUM over ranks of an all-reduce: 1 / world); the
  // norm and every element are scaled here instead of by an extra elementwise launch after the collective
  const float total = grad_scale * sqrtf(ppo_wave_sum(part));
  const float coef = grad_scale * fminf(1.f, max_norm / (total + 1e-6f));          // clip_grad_norm_
  const float t = state[1];
  const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
  const float step_size = lr / bc1;
  auto upd = [&](float &pi, float gi0, float &mi_, float &vi_) {
    const float gi = gi0 * coef;
    const float mi = b1 * mi_ + (1.f - b1) * gi;
    const float vi = b2 * vi_ + (1.f - b2) * gi * gi;
    mi_ = mi;
    vi_ = vi;
    pi -= step_size * mi / (sqrtf(vi) / bc2s + eps);
  };
  // 16 bytes per lane when the four buffers allow it (they are views of 16-byte aligned flat tensors): 33 MB per step for
  // the [1024,512] net; the tail (n is rarely a multiple of 4) goes element-wise
  const bool vec = (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                      reinterpret_cast<uintptr_t>(v)) & 15) == 0);
  const int n4 = vec ? (n >> 2) : 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    float4 P = reinterpret_cast<float4 *>(p)[i], M = reinterpret_cast<float4 *>(m)[i], V = reinterpret_cast<float4 *>(v)[i];
    const float4 G = reinterpret_cast<const float4 *>(g)[i];
    upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
    reinterpret_cast<float4 *>(p)[i] = P; reinterpret_cast<float4 *>(m)[i] = M; reinterpret_cast<float4 *>(v)[i] = V;
  }
  for (int i = 4 * n4 + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) upd(p[i], g[i], m[i], v[i]);
}
}  // namespace

static int flat_adam_launch(float *p, const float *g, float *m, float *v, int n, float lr, float beta1, float beta2, float eps,
                            float max_norm, float grad_scale, float *state2, int state2_floats, void *stream, int begin,
                            const DmGatherSpec *next = nullptr) {
  if (!p || !g || !m || !v || !state2 || n < 1 || !(grad_scale > 0.f)) return -22;
  if (state2_floats < 2 + DM_ADAM_PARTIALS) return -22;   // the partial sums live behind the two scalars: a shorter buffer would be overrun
  AdamGather G;
  memset(&G, 0, sizeof G);
  int gather_blocks = 0;
  if (next) {
    if (!next->idx || next->B < 1 || next->D < 1 || next->A < 1 || !next->obs || !next->act || !next->adv || !next->ret || !next->logp ||
        !next->o_obs || !next->o_act || !next->o_adv || !next->o_ret || !next->o_logp) return -22;
    G.idx = next->idx; G.B = next->B; G.D = next->D; G.A = next->A;
    G.obs = next->obs; G.act = next->act; G.adv = next->adv; G.ret = next->ret; G.logp = next->logp;
    G.o_obs = next->o_obs; G.o_act = next->o_act; G.o_adv = next->o_adv; G.o_ret = next->o_ret; G.o_logp = next->o_logp;
    gather_blocks = (next->B + 1) / 2;
  }
  hipStream_t s = (hipStream_t)stream;
  int blocks = (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > DM_ADAM_PARTIALS) blocks = DM_ADAM_PARTIALS;
  hipLaunchKernelGGL(adam_sumsq_kernel, dim3(blocks + gather_blocks), dim3(256), 0, s, g, n, state2, begin, blocks, G);   // step count folded in: two launches
  hipLaunchKernelGGL(adam_update_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, max_norm, grad_scale, state2);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
extern "C" int dm_flat_adam_step(float *p, const float *g, float *m, float *v, int n, float lr, float beta1, float beta2, float eps,
                                 float max_norm, float grad_scale, float *state2, int state2_floats, void *stream) {
  return flat_adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, max_norm, grad_scale, state2, state2_floats, stream, 1);
}
extern "C" int dm_flat_adam_update(float *p, const float *g, float *m, float *v, int n, float lr, float beta1, float beta2, float eps,
                                   float max_norm, float grad_scale, float *state2, int state2_floats, void *stream) {
  return flat_adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, max_norm, grad_scale, state2, state2_floats, stream, 0);
}
extern "C" int dm_flat_adam_step_gather(float *p, const float *g, float *m, float *v, int n, float lr, float beta1, float beta2, float eps,
                                        float max_norm, float grad_scale, float *state2, int state2_floats, int begin,
                                        const DmGatherSpec *next, void *stream) {
  return flat_adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, max_norm, grad_scale, state2, state2_floats, stream, begin ? 1 : 0, next);
}

// ---------------------------------------------------------------------------------------------------------------
// Rollout side of the PPO loop (reference: src/sb3_ppo.py:307-313 -> [EXT] SB3 collect_rollouts): per env step the
// policy head output becomes a sampled action, its log-probability and the clipped action handed to the env, and the
// step's data go into the rollout buffer.  Two launches instead of ~20 elementwise PyTorch kernels.
namespace {

__device__ __forceinline__ unsigned ppo_hash32(unsigned long long seed, unsigned a, unsigned b, unsigned c) {
  unsigned long long x = seed ^ ((unsigned long long)a * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)b * 0xBF58476D1CE4E5B9ull) ^
                         ((unsigned long long)c * 0x94D049BB133111EBull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (unsigned)(x >> 32);
}

// act = mean + exp(log_std) * eps, eps ~ N(0,1) (Box-Muller on a counter-based hash: env, draw counter, action index);
// logp = sum_j -0.5 eps_j^2 - log_std_j - 0.5 log(2 pi); act_env = clamp(act, lo, hi).  counter[0] is read here and
// advanced by ppo_store_kernel, which runs later on the same stream.
__global__ void ppo_sample_kernel(const float *mean, const float *log_std, int N, int A, unsigned long long seed,
                                  const unsigned *counter, const float *lo, const float *hi, float *act, float *act_env,
                                  float *logp) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const unsigned ctr = counter[0];
  float lp = 0.f;
  for (int j = 0; j < A; j += 2) {
    const float u1 = ((float)(ppo_hash32(seed, (unsigned)e, ctr, (unsigned)j) >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
    const float u2 = (float)(ppo_hash32(seed, (unsigned)e, ctr, (unsigned)j + 1u) >> 8) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    const float eps[2] = {rad * cs, rad * sn};
    for (int q = 0; q < 2 && j + q < A; q++) {
      const float ls = log_std[j + q];
      const float a = mean[(size_t)e * A + j + q] + expf(ls) * eps[q];
      act[(size_t)e * A + j + q] = a;
      act_env[(size_t)e * A + j + q] = fminf(fmaxf(a, lo[j + q]), hi[j + q]);
      lp += -0.5f * eps[q] * eps[q] - ls - 0.9189385332046727f;
    }
  }
  logp[e] = lp;
}

// rollout buffer row t <- (obs the policy saw, action, value, logp, reward, done); last_obs <- the env's new obs
__global__ void ppo_store_kernel(int N, int D, int A, const float *last_obs, const float *act, const float *val, const float *logp,
                                 const float *rew, const unsigned char *done, const float *new_obs, float *b_obs, float *b_act,
                                 float *b_val, float *b_logp, float *b_rew, float *b_done, float *last_obs_out, unsigned *counter) {
  const int e = blockIdx.x;
  if (e >= N) return;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    b_obs[(size_t)e * D + c] = last_obs[(size_t)e * D + c];
    last_obs_out[(size_t)e * D + c] = new_obs[(size_t)e * D + c];
  }
  for (int c = threadIdx.x; c < A; c += blockDim.x) b_act[(size_t)e * A + c] = act[(size_t)e * A + c];
  if (threadIdx.x == 0) {
    b_val[e] = val[e]; b_logp[e] = logp[e]; b_rew[e] = rew[e]; b_done[e] = done[e] ? 1.f : 0.f;
    if (e == 0 && counter) counter[0] += 1u;
  }
}

}  // namespace

extern "C" int dm_policy_sample(const float *mean, const float *log_std, int N, int A, unsigned long long seed,
                                const unsigned *counter, const float *lo, const float *hi, float *act, float *act_env,
                                float *logp, void *stream) {
  if (!mean || !log_std || !counter || !lo || !hi || !act || !act_env || !logp || N < 1 || A < 1) return -22;
  hipLaunchKernelGGL(ppo_sample_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, log_std, N, A, seed, counter,
                     lo, hi, act, act_env, logp);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int dm_rollout_store(int N, int D, int A, const float *last_obs, const float *act, const float *val, const float *logp,
                                const float *rew, const unsigned char *done, const float *new_obs, float *b_obs, float *b_act,
                                float *b_val, float *b_logp, float *b_rew, float *b_done, float *last_obs_out, unsigned *counter,
                                void *stream) {
  if (N < 1 || !last_obs || !act || !val || !logp || !rew || !done || !new_obs || !b_obs || !b_act || !b_val || !b_logp || !b_rew ||
      !b_done || !last_obs_out)
    return -22;
  hipLaunchKernelGGL(ppo_store_kernel, dim3(N), dim3(128), 0, (hipStream_t)stream, N, D, A, last_obs, act, val, logp, rew, done, new_obs,
                     b_obs, b_act, b_val, b_logp, b_rew, b_done, last_obs_out, counter);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
