// One PPO minibatch gradient of the two-hidden-layer actor-critic MLP in THREE launches (gfx950).
//
// Reference: src/sb3_ppo.py:254-271,307-312 -> [EXT] SB3 PPO.train on MlpPolicy(net_arch=[256,128]): per minibatch
// evaluate_actions (both trunks), the clipped-surrogate / value / entropy loss, backward through both trunks.  With
// PyTorch this is ~40 graph nodes of ~5 us (library GEMMs of 4096 x 256 x 128, elementwise tanh backward, the loss
// kernels): the optimizer step of the reference's net is launch-bound at 0.22 ms.  Here:
//   1. mlp_pack_kernel   weights -> MFMA operand order (forward and transposed), advantage statistics; optional folds:
//                        clearing the gradient arena, Adam's begin                                             (1 launch)
//   2. mlp_fwdbwd_kernel a workgroup of eight waves carries 32 minibatch rows of one trunk through the forward, the
//                        loss head (same arithmetic as ppo_loss_kernel) and the input-gradient chain; activations and
//                        pre-activation gradients stay in LDS and go to HBM in one copy-out at the very end, with the
//                        workgroup's loss / log-std partial sums (no global write before the last load)        (1 launch)
//   3. mlp_wgrad_kernel  dW = dZ^T X, db = sum dZ for all six layers: a workgroup per 64 x 64 rectangle of dW and batch
//                        slice, four waves splitting the slice, LDS reduction, split-K 8 over workgroups with float
//                        atomics; one extra block sums the loss partials (loss scalar, entropy gradient)        (1 launch)
// All products are v_mfma_f32_32x32x2f32 (fp32 in, fp32 accumulate).  Gradients are ACCUMULATED into caller-owned
// buffers that must be zero on entry (the optimizer's flat gradient arena: pass it as zero_ptr, or clear it on the stream).
// Included by dm_abi.hip after dm_ppo.hip and dm_policy.hip (uses their device helpers).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/deepmimic_hip.h"

namespace {

struct MlpPackJob { const float *W; float4 *P; int O, K, so, sk, tiles, KB, first; };
struct MlpPackArgs {
  MlpPackJob j[10];
  int njobs, nblocks;               // block nblocks computes the advantage statistics, the blocks after it clear zero_ptr
  const float *adv; int B, normalize; float *stats, *out8;
  float *zero_ptr; long long zero_floats; float *adam_state2;
};

// Advantage statistics of the minibatch by ONE block of 256 threads (stats[0] = mean, stats[1] = 1 / (std_unbiased + 1e-8)), as
// ppo_prepare_body, but with every load of the block in flight at once: the values are read into registers by independent
// loads instead of 2 x B / 256 dependent round trips — this block is the critical path of the launch.  B <= 8192.
__device__ __forceinline__ void mlp_adv_stats(const float *adv, int B, int normalize, float *stats, float *out8) {
  __shared__ float red[16];
  __shared__ float mean_s;
  constexpr int ITEMS = 32;
  float v[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    const int i = threadIdx.x + k * 256;
    v[k] = (i < B) ? adv[i] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < ITEMS; k++) s += v[k];
  const float tot = ppo_block_sum(s, red);
  if (threadIdx.x == 0) mean_s = tot / (float)B;
  __syncthreads();
  const float mean = mean_s;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < ITEMS; k++) { const float d = v[k] - mean; q += ((int)threadIdx.x + k * 256 < B) ? d * d : 0.f; }
  const float qq = ppo_block_sum(q, red);
  if (threadIdx.x == 0) {
    if (normalize && B > 1) { stats[0] = mean; stats[1] = 1.0f / (sqrtf(qq / (float)(B - 1)) + 1e-8f); }
    else { stats[0] = 0.f; stats[1] = 1.f; }
  }
  if (threadIdx.x < 8) out8[threadIdx.x] = 0.f;
}

__global__ void __launch_bounds__(256) mlp_pack_kernel(MlpPackArgs a) {
  const int blk = blockIdx.x;
  if (blk == a.nblocks) {
    if (a.B <= 8192) mlp_adv_stats(a.adv, a.B, a.normalize, a.stats, a.out8);
    else ppo_prepare_body(a.adv, a.B, a.normalize, a.stats, a.out8, nullptr, 0);
    if (a.adam_state2 && threadIdx.x == 0) { a.adam_state2[0] = 0.f; a.adam_state2[1] += 1.f; }     // adam_begin_kernel
    return;
  }
  if (blk > a.nblocks) {
    const long long i = ((long long)(blk - a.nblocks - 1) * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int c = 0; c < 4; c++) if (i + c < a.zero_floats) a.zero_ptr[i + c] = 0.f;
    return;
  }
  int q = 0;
  for (int i = 1; i < a.njobs; i++) if (blk >= a.j[i].first) q = i;
  const MlpPackJob &J = a.j[q];
  pol_pack_one(J.W, J.O, J.K, J.so, J.sk, J.tiles, J.KB, J.P, (blk - J.first) * 256 + (int)threadIdx.x);
}

// tanh(x) = 1 - 2 / (1 + e^{2x}) on the hardware exp / rcp: absolute error ~1e-7 (the accurate tanhf costs ~40 instructions
// and the forward evaluates 12 k of them per workgroup)
__device__ __forceinline__ float mlp_tanh(float x) { return 1.f - __fdividef(2.f, 1.f + __expf(2.f * x)); }

struct MlpTrainArgs {
  int B, D, Dp, H1, H2, A;
  const float *obs, *act, *adv, *ret, *old_logp, *log_std;
  const float4 *pkF[2], *pkW2T[2], *pkW3T[2];
  const float *b1[2], *b2[2], *b3[2];
  float *h1g[2], *h2g[2], *dz1g[2], *dz2g[2], *d3g[2];
  float *part;                      // [2][B / 32][36] per-workgroup loss / log-std partial sums (summed by mlp_wgrad_kernel's epilogue)
  const float *stats;
  float *out8, *g_log_std;
  float clip, vf_coef;
};

// Eight waves per workgroup (two per SIMD): one wave's weight-stream latency, tanh and stores run under the other's
// MFMAs (with four waves the kernel spent a third of its time in MFMA and as much waiting on loads, nothing overlapping).
constexpr int MLP_NW = 8, MLP_THREADS = 64 * MLP_NW;

// -DMLP_PROFILE (diagnostic build only, scripts/phase_profile_learner.py): s_memtime stamps of workgroup (0, 0) per wave and phase
#ifdef MLP_PROFILE
__device__ long long mlp_prof_buf[MLP_NW * 16];
#define MLP_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) mlp_prof_buf[wave * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define MLP_STAMP(i) do { } while (0)
#endif

__global__ void __launch_bounds__(MLP_THREADS) mlp_fwdbwd_kernel(MlpTrainArgs a) {
  extern __shared__ __align__(16) float mlp_lds[];
  const int trunk = blockIdx.y, b0 = blockIdx.x * POL_R;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int sx = a.Dp + POL_PAD, s1 = a.H1 + POL_PAD, s2 = a.H2 + POL_PAD, s3 = 32 + POL_PAD;
  float *xs = mlp_lds;
  float *h1 = xs + POL_R * sx;
  float *h2 = h1 + POL_R * s1;
  float *dz2 = h2 + POL_R * s2;
  float *d3 = dz2 + POL_R * s2;
  float *red = d3 + POL_R * s3;           // MLP_NW x 32 x 32 layer-3 partial sums; before that the layer-2 K-half partials
  float *accs = red + MLP_NW * 1024;      // [0..3] pg, vl, kl, clip fraction; [4..35] d loss / d log_std partials
  const int KB1 = a.Dp >> 3, T1 = a.H1 >> 5, KB2 = a.H1 >> 3, T2 = a.H2 >> 5, KB3 = a.H2 >> 3;
  const float4 *P1 = a.pkF[trunk];
  const float4 *P2 = P1 + (size_t)T1 * KB1 * 64;
  const float4 *P3 = P2 + (size_t)T2 * KB2 * 64;
  const int r = lane & 31, h = lane >> 5;
  MLP_STAMP(0);
  // every phase's first weight batch is requested before the barrier / staging work that precedes the phase (no global
  // write precedes it any more, so nothing holds the request back)
  float4 w1pre[9];                          // D = 67 / 72: the nine k-blocks are the whole layer-1 tile
  const bool pre1 = wave < T1 && KB1 >= 9;
  if (pre1) pol_prefetch<9>(P1 + (size_t)wave * KB1 * 64, lane, 0, w1pre);
  // the biases of this wave's first tiles too (a load at the top of an epilogue is one more exposed round trip)
  const float b1pre = wave < T1 ? a.b1[trunk][wave * 32 + r] : 0.f;
  const float b2pre = (wave >> 1) < T2 ? a.b2[trunk][(wave >> 1) * 32 + r] : 0.f;
  for (int i = tid; i < POL_R * a.D; i += MLP_THREADS) {
    const int row = i / a.D, c = i - row * a.D;
    xs[row * sx + c] = a.obs[(size_t)b0 * a.D + i];
  }
  for (int i = tid; i < POL_R * (a.Dp - a.D); i += MLP_THREADS) {
    const int row = i / (a.Dp - a.D), c = a.D + i - row * (a.Dp - a.D);
    xs[row * sx + c] = 0.f;
  }
  if (tid < 36) accs[tid] = 0.f;
  MLP_STAMP(1);
  __syncthreads();
  MLP_STAMP(2);

  // ---- forward layer 1: one tile per wave (activations to LDS, and to HBM for the weight gradients)
  for (int to = wave; to < T1; to += MLP_NW) {
    pol_f16v acc;
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    if (pre1 && to == wave) pol_tile_pre<9>(xs, sx, P1 + (size_t)to * KB1 * 64, lane, 0, KB1, acc, w1pre);
    else pol_tile<3>(xs, sx, P1 + (size_t)to * KB1 * 64, lane, 0, KB1, acc);
    const float b = to == wave ? b1pre : a.b1[trunk][to * 32 + r];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int row = (j >> 2) * 8 + h * 4 + (j & 3);
      const float v = mlp_tanh(acc[j] + b);
      h1[row * s1 + to * 32 + r] = v;
    }
  }
  float4 w2pre[8];
  const bool pre2 = wave < 2 * T2 && (KB2 >> 1) >= 8;
  if (pre2) pol_prefetch<8>(P2 + (size_t)(wave >> 1) * KB2 * 64, lane, (wave & 1) * (KB2 >> 1), w2pre);
  MLP_STAMP(3);
  __syncthreads();
  MLP_STAMP(4);
  // ---- forward layer 2: a tile is shared by two waves (K halves; KB2 = H1 / 8 is even), the odd wave hands its partial
  // sums over through LDS
  for (int base = 0; base < 2 * T2; base += MLP_NW) {
    const int u = base + wave, tile = u >> 1, half = u & 1;
    const bool on = u < 2 * T2;
    pol_f16v acc;
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    if (on) {
      const int kh = KB2 >> 1;
      if (pre2 && base == 0) pol_tile_pre<8>(h1, s1, P2 + (size_t)tile * KB2 * 64, lane, half * kh, (half + 1) * kh, acc, w2pre);
      else pol_tile<8>(h1, s1, P2 + (size_t)tile * KB2 * 64, lane, half * kh, (half + 1) * kh, acc);
      if (half) {
#pragma unroll
        for (int j = 0; j < 16; j++) red[tile * 1024 + j * 64 + lane] = acc[j];
      }
    }
    __syncthreads();
    if (on && !half) {
      const float b = base == 0 ? b2pre : a.b2[trunk][tile * 32 + r];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int row = (j >> 2) * 8 + h * 4 + (j & 3);
        const float v = mlp_tanh(acc[j] + red[tile * 1024 + j * 64 + lane] + b);
        h2[row * s2 + tile * 32 + r] = v;
      }
    }
  }
  MLP_STAMP(5);
  __syncthreads();
  MLP_STAMP(6);
  // ---- forward layer 3: split-K over the waves
  {
    pol_f16v acc3;
#pragma unroll
    for (int j = 0; j < 16; j++) acc3[j] = 0.f;
    const int nsp = (KB3 % MLP_NW) == 0 ? MLP_NW : 4;      // KB3 = H2 / 8 is a multiple of 4
    const int per = KB3 / nsp;
    if (wave < nsp) pol_tile<4>(h2, s2, P3, lane, wave * per, (wave + 1) * per, acc3);
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int row = (j >> 2) * 8 + h * 4 + (j & 3);
      red[wave * 1024 + row * 32 + r] = acc3[j];
    }
  }
  MLP_STAMP(7);
  __syncthreads();
  MLP_STAMP(8);

  // ---- loss head: eight threads per row (same formulas as ppo_loss_kernel), waves 0..3
  if (tid < 256) {
    const int row = tid >> 3, q = tid & 7, b = b0 + row;
    const float invB = 1.0f / (float)a.B;
    if (trunk == 0) {
      float dd[4], iv[4], z2[4];
      float lp = 0.f;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int c = q + 8 * s;
        dd[s] = 0.f; iv[s] = 0.f; z2[s] = 0.f;
        if (c < a.A) {
          float m = a.b3[0][c];
#pragma unroll
          for (int w = 0; w < MLP_NW; w++) m += red[w * 1024 + row * 32 + c];
          const float ls = a.log_std[c];
          dd[s] = a.act[(size_t)b * a.A + c] - m;
          iv[s] = expf(-2.f * ls);
          z2[s] = dd[s] * dd[s] * iv[s];
          lp += -0.5f * z2[s] - ls - 0.9189385332046727f;
        }
      }
      lp += __shfl_xor(lp, 1);
      lp += __shfl_xor(lp, 2);
      lp += __shfl_xor(lp, 4);
      const float a_n = (a.adv[b] - a.stats[0]) * a.stats[1];
      const float lr = lp - a.old_logp[b];
      const float ratio = expf(lr);
      const float rc = fminf(fmaxf(ratio, 1.f - a.clip), 1.f + a.clip);
      const float p1 = a_n * ratio, p2 = a_n * rc;
      const bool inside = (ratio >= 1.f - a.clip) && (ratio <= 1.f + a.clip);
      const float dr = (inside || p1 < p2) ? a_n : 0.f;
      const float dlogp = -invB * dr * ratio;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int c = q + 8 * s;
        const float g = dlogp * dd[s] * iv[s];               // 0 for c >= A
        d3[row * s3 + c] = g;
        if (c < a.A) atomicAdd(&accs[4 + c], dlogp * (z2[s] - 1.f));
      }
      if (q == 0) {
        atomicAdd(&accs[0], -fminf(p1, p2));
        atomicAdd(&accs[2], (ratio - 1.f) - lr);
        atomicAdd(&accs[3], (fabsf(ratio - 1.f) > a.clip) ? 1.f : 0.f);
      }
    } else {
      float v = a.b3[1][0];
#pragma unroll
      for (int w = 0; w < MLP_NW; w++) v += red[w * 1024 + row * 32];
      const float dv = v - a.ret[b];
      const float g = a.vf_coef * 2.f * invB * dv;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int c = q + 8 * s;
        d3[row * s3 + c] = (c == 0) ? g : 0.f;
      }
      if (q == 0) atomicAdd(&accs[1], dv * dv);
    }
  }
  MLP_STAMP(9);
  __syncthreads();
  MLP_STAMP(10);

  // ---- backward: dZ2 = (d3 W3) (1 - h2^2), dZ1 = (dZ2 W2) (1 - h1^2)
  for (int to = wave; to < T2; to += MLP_NW) {
    pol_f16v acc;
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    pol_tile<4>(d3, s3, a.pkW3T[trunk] + (size_t)to * 4 * 64, lane, 0, 4, acc);
    float hv[16];                       // all sixteen LDS reads in flight before the first use
#pragma unroll
    for (int j = 0; j < 16; j++) hv[j] = h2[((j >> 2) * 8 + h * 4 + (j & 3)) * s2 + to * 32 + r];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int row = (j >> 2) * 8 + h * 4 + (j & 3);
      const float g = acc[j] * (1.f - hv[j] * hv[j]);
      dz2[row * s2 + to * 32 + r] = g;
    }
  }
  float4 w2tpre[8];
  const bool pre2t = wave < T1 && KB3 >= 8;
  if (pre2t) pol_prefetch<8>(a.pkW2T[trunk] + (size_t)wave * KB3 * 64, lane, 0, w2tpre);
  MLP_STAMP(11);
  __syncthreads();
  MLP_STAMP(12);
  for (int to = wave; to < T1; to += MLP_NW) {
    pol_f16v acc;
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    if (pre2t && to == wave) pol_tile_pre<8>(dz2, s2, a.pkW2T[trunk] + (size_t)to * KB3 * 64, lane, 0, KB3, acc, w2tpre);
    else pol_tile<8>(dz2, s2, a.pkW2T[trunk] + (size_t)to * KB3 * 64, lane, 0, KB3, acc);
    float hv[16];
#pragma unroll
    for (int j = 0; j < 16; j++) hv[j] = h1[((j >> 2) * 8 + h * 4 + (j & 3)) * s1 + to * 32 + r];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int row = (j >> 2) * 8 + h * 4 + (j & 3);
      a.dz1g[trunk][(size_t)(b0 + row) * a.H1 + to * 32 + r] = acc[j] * (1.f - hv[j] * hv[j]);
    }
  }
  // ---- everything that writes global memory besides dZ1 comes last.  A global store or atomic followed by a load makes the
  // compiler wait for the write's acknowledgement (s_waitcnt vmcnt(0): the addresses might alias), and the loss / log-std
  // sums are written as per-workgroup partials: in the middle of the kernel (as atomics) that wait was 10 k cycles
  // in front of the backward weight loads, and 1-2 k at the head of every other phase (stamps of -DMLP_PROFILE).
  {
    const int q1 = a.H1 >> 2, q2 = a.H2 >> 2;
    for (int i = tid; i < POL_R * q1; i += MLP_THREADS) {
      const int row = i / q1, c = (i - row * q1) * 4;
      *reinterpret_cast<float4 *>(a.h1g[trunk] + (size_t)(b0 + row) * a.H1 + c) = *reinterpret_cast<const float4 *>(h1 + row * s1 + c);
    }
    for (int i = tid; i < POL_R * q2; i += MLP_THREADS) {
      const int row = i / q2, c = (i - row * q2) * 4;
      *reinterpret_cast<float4 *>(a.h2g[trunk] + (size_t)(b0 + row) * a.H2 + c) = *reinterpret_cast<const float4 *>(h2 + row * s2 + c);
      *reinterpret_cast<float4 *>(a.dz2g[trunk] + (size_t)(b0 + row) * a.H2 + c) = *reinterpret_cast<const float4 *>(dz2 + row * s2 + c);
    }
    const int A3 = trunk == 0 ? a.A : 1;
    for (int i = tid; i < POL_R * A3; i += MLP_THREADS) {
      const int row = i / A3, c = i - row * A3;
      a.d3g[trunk][(size_t)(b0 + row) * A3 + c] = d3[row * s3 + c];
    }
    // per-workgroup partial sums, plain stores: 256 workgroups adding to the same two cache lines with atomics queued for
    // ~10 k cycles at the L2 (and a wave cannot retire before its atomics are acknowledged)
    if (tid < 36) a.part[((size_t)trunk * gridDim.x + blockIdx.x) * 36 + tid] = accs[tid];
  }
  MLP_STAMP(13);
}

// Weight / bias gradients of all six layers in one launch.  A workgroup of four waves owns a rectangle of 2 x 2 output
// tiles (64 x 64 of dW) of one layer and one slice of `kchunk` minibatch rows; the waves split the slice four ways,
// each accumulating the four 32 x 32 tiles from operands read straight from global memory (128-byte row segments per
// half-wave, every operand feeding two MFMAs, next batch of loads in flight under the current MFMAs), then the partial
// tiles are summed through LDS and each wave adds one tile to dW with float atomics (split-K over workgroups: 8).
constexpr int WG3_U = 8;            // k-pairs per load batch: 4 U loads per lane in flight, twice that with the prefetch
struct MlpWgradJob { const float *dY, *X; float *dW, *db; int O, I, nro, nri, kchunk, first; };
struct MlpWgradArgs {
  MlpWgradJob j[6];
  int B, nblocks;                   // block nblocks (the last one) finishes the loss scalar and the entropy gradient
  const float *log_std, *stats, *part; float *g_log_std, *out8, *loss_acc; int A, nwg; float vf_coef, ent_coef;
};

__global__ void __launch_bounds__(256, 2) mlp_wgrad_kernel(MlpWgradArgs a) {
  __shared__ __align__(16) float part[4][4][1024];       // [wave][tile][32 x 32]
  const int blk = blockIdx.x;
  if (blk == a.nblocks) {
    // loss epilogue: sum the per-workgroup partials of mlp_fwdbwd_kernel (thread q < 36 owns quantity q: 0 pg (pi), 1 vl (vf),
    // 2 kl, 3 clip fraction, 4.. d loss / d log_std), then the loss scalar and the entropy term
    float *tot = &part[0][0][0];
    const int q = threadIdx.x;
    {
      // 252 = 7 x 36 threads: thread t sums quantity t % 36 over the workgroups w = t / 36, t / 36 + 7, ... (independent loads)
      const int col = q % 36, grp = q / 36;
      float t = 0.f;
      if (q < 252) {
        const float *p = a.part + (size_t)(col == 1 ? a.nwg : 0) * 36 + col;
        float t1 = 0.f, t2 = 0.f, t3 = 0.f;             // four chains: the thread's loads are in flight together (fixed order)
        int w = grp;
        for (; w + 21 < a.nwg; w += 28) { t += p[(size_t)w * 36]; t1 += p[(size_t)(w + 7) * 36]; t2 += p[(size_t)(w + 14) * 36]; t3 += p[(size_t)(w + 21) * 36]; }
        for (; w < a.nwg; w += 7) t += p[(size_t)w * 36];
        tot[64 + q] = (t + t1) + (t2 + t3);
      }
      __syncthreads();
      if (q < 36) {
        float u = 0.f;
#pragma unroll
        for (int g = 0; g < 7; g++) u += tot[64 + g * 36 + q];
        tot[q] = u;
      }
    }
    __syncthreads();
    const float invB = 1.0f / (float)a.B;
    if (q >= 4 && q < 4 + a.A) atomicAdd(&a.g_log_std[q - 4], tot[q] - a.ent_coef);
    if (q == 0) {
      float ent = 0;
      for (int j = 0; j < a.A; j++) ent += 0.5f + 0.5f * 1.8378770664093453f + a.log_std[j];
      a.out8[1] = tot[0] * invB;
      a.out8[2] = tot[1] * invB;
      a.out8[4] = tot[2] * invB;
      a.out8[5] = tot[3] * invB;
      a.out8[3] = ent;
      a.out8[0] = a.out8[1] + a.vf_coef * a.out8[2] - a.ent_coef * ent;
      a.out8[6] = a.stats[0];
      a.out8[7] = a.stats[1];
      if (a.loss_acc) { a.loss_acc[0] += a.out8[0]; a.loss_acc[1] += 1.f; }
    }
    return;
  }
  int q = 0;
#pragma unroll
  for (int i = 1; i < 6; i++) if (blk >= a.j[i].first) q = i;
  const MlpWgradJob &J = a.j[q];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int O = J.O, I = J.I;
  const int local = blk - J.first, nrect = J.nro * J.nri;
  const int rect = local % nrect, slice = local / nrect;
  const int o0 = (rect / J.nri) * 64, i0 = (rect % J.nri) * 64;
  const int kq = J.kchunk >> 2;                               // rows per wave, a multiple of 2 WG3_U
  const int k0 = slice * J.kchunk + wave * kq;
  const bool vo1 = o0 + 32 < O, vi1 = i0 + 32 < I;            // second tile row / column exists (wave-uniform)
  const bool la0 = (o0 + r) < O, la1 = (o0 + 32 + r) < O;
  // out-of-range rows / columns of a tile are never stored, so their operand lanes only need a valid address: clamp
  const float *pa = J.dY + (size_t)(k0 + h) * O, *px = J.X + (size_t)(k0 + h) * I;
  const int ca0 = min(o0 + r, O - 1), ca1 = min(o0 + 32 + r, O - 1), cx0 = min(i0 + r, I - 1), cx1 = min(i0 + 32 + r, I - 1);
  pol_f16v acc00, acc01, acc10, acc11;
#pragma unroll
  for (int j = 0; j < 16; j++) { acc00[j] = 0.f; acc01[j] = 0.f; acc10[j] = 0.f; acc11[j] = 0.f; }
  float db0 = 0.f, db1 = 0.f;
  // ping-pong over two register sets: the 4 U loads of batch it + 1 are in flight under the MFMAs of batch it
  struct Ops { float a0[WG3_U], a1[WG3_U], x0[WG3_U], x1[WG3_U]; };
  auto load = [&](Ops &o, int k) {
#pragma unroll
    for (int u = 0; u < WG3_U; u++) {
      const size_t kk = (size_t)(k + 2 * u);
      o.a0[u] = pa[kk * O + ca0];
      o.a1[u] = pa[kk * O + ca1];
      o.x0[u] = px[kk * I + cx0];
      o.x1[u] = px[kk * I + cx1];
    }
  };
  auto mfma = [&](const Ops &o) {
#pragma unroll
    for (int u = 0; u < WG3_U; u++) {
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a0[u], o.x0[u], acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a0[u], o.x1[u], acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a1[u], o.x0[u], acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a1[u], o.x1[u], acc11, 0, 0, 0);
      db0 += o.a0[u];
      db1 += o.a1[u];
    }
  };
  Ops A, Bq;
  const int nb = kq / (2 * WG3_U);
  load(A, 0);
  int it = 0;
  for (; it + 2 <= nb; it += 2) {
    load(Bq, (it + 1) * 2 * WG3_U);
    mfma(A);
    if (it + 2 < nb) load(A, (it + 2) * 2 * WG3_U);
    mfma(Bq);
  }
  if (it < nb) mfma(A);
  // partial tiles -> LDS in accumulator order (lane-major: conflict-free), then wave w sums tile w and adds it to dW
#pragma unroll
  for (int j = 0; j < 16; j++) {
    part[wave][0][j * 64 + lane] = acc00[j];
    part[wave][1][j * 64 + lane] = acc01[j];
    part[wave][2][j * 64 + lane] = acc10[j];
    part[wave][3][j * 64 + lane] = acc11[j];
  }
  __syncthreads();
  {
    const int t = wave, to = t >> 1, ti = t & 1;
    const bool valid = (to == 0 || vo1) && (ti == 0 || vi1);
    if (valid) {
      const int col = i0 + ti * 32 + r;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const float v = part[0][t][j * 64 + lane] + part[1][t][j * 64 + lane] + part[2][t][j * 64 + lane] + part[3][t][j * 64 + lane];
        const int row = o0 + to * 32 + (j >> 2) * 8 + h * 4 + (j & 3);
        if (row < O && col < I) atomicAdd(&J.dW[(size_t)row * I + col], v);
      }
    }
  }
  if (i0 == 0) {            // bias gradient: every wave adds its K quarter (two rows of 32 outputs)
    const float v0 = db0 + __shfl_xor(db0, 32), v1 = db1 + __shfl_xor(db1, 32);
    if (h == 0 && la0) atomicAdd(&J.db[o0 + r], v0);
    if (h == 0 && la1) atomicAdd(&J.db[o0 + 32 + r], v1);
  }
}

inline bool mlp_dims_ok(int B, int D, int H1, int H2, int A) {
  return B >= 64 && (B % (8 * WG3_U)) == 0 && pol_dims_ok(D, H1, H2, A) && H1 <= 256 && H2 <= 256;
}
inline size_t mlp_lds_bytes(int D, int H1, int H2) {
  return (size_t)(POL_R * (pol_dp(D) + POL_PAD + H1 + POL_PAD + 2 * (H2 + POL_PAD) + 32 + POL_PAD) + MLP_NW * 1024 + 64) * sizeof(float);
}
// per-trunk workspace (floats): forward pack | W2^T pack | W3^T pack | h1 | h2 | dz1 | dz2 | d3
struct MlpWsLayout { size_t pkF, pkW2T, pkW3T, h1, h2, dz1, dz2, d3, per_trunk, total; };
inline MlpWsLayout mlp_layout(int B, int D, int H1, int H2, int A) {
  MlpWsLayout L;
  size_t o = 0;
  L.pkF = o;   o += 256 * ((size_t)(H1 / 32) * (pol_dp(D) / 8) + (size_t)(H2 / 32) * (H1 / 8) + (size_t)(H2 / 8));
  L.pkW2T = o; o += 256 * (size_t)(H1 / 32) * (H2 / 8);
  L.pkW3T = o; o += 256 * (size_t)(H2 / 32) * 4;
  L.h1 = o;    o += (size_t)B * H1;
  L.h2 = o;    o += (size_t)B * H2;
  L.dz1 = o;   o += (size_t)B * H1;
  L.dz2 = o;   o += (size_t)B * H2;
  L.d3 = o;    o += (size_t)B * 32;
  L.per_trunk = (o + 3) & ~(size_t)3;
  L.total = 2 * L.per_trunk + 8 + 2 * (size_t)(B / 32) * 36;     // + advantage statistics + per-workgroup loss partials
  return L;
}

}  // namespace

extern "C" long long dm_ppo_mlp_workspace_floats(int B, int D, int H1, int H2, int A) {
  if (!mlp_dims_ok(B, D, H1, H2, A)) return -22;
  if (mlp_lds_bytes(D, H1, H2) > 160 * 1024) return -22;
  return (long long)mlp_layout(B, D, H1, H2, A).total;
}

extern "C" int dm_ppo_mlp_grad(const DmPpoMlpStep *s, void *stream) {
  if (!s) return -22;
  const int B = s->B, D = s->D, H1 = s->H1, H2 = s->H2, A = s->A;
  if (!mlp_dims_ok(B, D, H1, H2, A)) return -22;
  if (!s->obs || !s->act || !s->adv || !s->ret || !s->old_logp || !s->log_std || !s->g_log_std || !s->out8 || !s->workspace) return -22;
  for (int t = 0; t < 2; t++)
    for (int l = 0; l < 3; l++)
      if (!s->W[t][l] || !s->b[t][l] || !s->gW[t][l] || !s->gb[t][l]) return -22;
  if (reinterpret_cast<uintptr_t>(s->workspace) & 15) return -22;
  const MlpWsLayout L = mlp_layout(B, D, H1, H2, A);
  if (s->workspace_floats < (long long)L.total) return -22;
  const size_t lds = mlp_lds_bytes(D, H1, H2);
  if (lds > 160 * 1024) return -22;
  // hipFuncSetAttribute applies to the CURRENT device: remember the raised limit per device ordinal
  static size_t lds_allowed[64];
  int dev_id = 0;
  if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) return -5;
  const size_t allowed = lds_allowed[dev_id] ? lds_allowed[dev_id] : (size_t)64 * 1024;
  if (lds > allowed) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_fwdbwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return -5;
    lds_allowed[dev_id] = lds;
  }
  hipStream_t st = (hipStream_t)stream;
  float *stats = s->workspace + 2 * L.per_trunk;
  const int Dp = pol_dp(D), T1 = H1 / 32, KB1 = Dp / 8, T2 = H2 / 32, KB2 = H1 / 8, KB3 = H2 / 8;

  // ---- 1. pack + advantage statistics
  MlpPackArgs pa;
  int nb = 0, nj = 0;
  auto add_pack = [&](const float *W, float *P, int O, int K, int so, int sk, int tiles, int KB) {
    MlpPackJob &J = pa.j[nj++];
    J.W = W; J.P = reinterpret_cast<float4 *>(P); J.O = O; J.K = K; J.so = so; J.sk = sk; J.tiles = tiles; J.KB = KB; J.first = nb;
    nb += (tiles * KB * 64 + 255) / 256;
  };
  for (int t = 0; t < 2; t++) {
    float *ws = s->workspace + t * L.per_trunk;
    const int Aout = t == 0 ? A : 1;
    float *P1 = ws + L.pkF, *P2 = P1 + (size_t)T1 * KB1 * 256, *P3 = P2 + (size_t)T2 * KB2 * 256;
    add_pack(s->W[t][0], P1, H1, D, D, 1, T1, KB1);
    add_pack(s->W[t][1], P2, H2, H1, H1, 1, T2, KB2);
    add_pack(s->W[t][2], P3, Aout, H2, H2, 1, 1, KB3);
    add_pack(s->W[t][1], ws + L.pkW2T, H1, H2, 1, H1, T1, KB3);      // W2^T: (i, o) -> W2[o][i]
    add_pack(s->W[t][2], ws + L.pkW3T, H2, Aout, 1, H2, T2, 4);       // W3^T: (i, c) -> W3[c][i], c padded to 32
  }
  pa.njobs = nj; pa.nblocks = nb;
  pa.adv = s->adv; pa.B = B; pa.normalize = (s->normalize_advantage && B > 1) ? 1 : 0; pa.stats = stats; pa.out8 = s->out8;
  pa.zero_ptr = s->zero_ptr; pa.zero_floats = s->zero_ptr ? s->zero_floats : 0; pa.adam_state2 = s->adam_state2;
  const int nzero = (int)((pa.zero_floats + 1023) / 1024);
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(nb + 1 + nzero), dim3(256), 0, st, pa);

  // ---- 2. forward + loss head + input gradients
  MlpTrainArgs ta;
  ta.B = B; ta.D = D; ta.Dp = Dp; ta.H1 = H1; ta.H2 = H2; ta.A = A;
  ta.obs = s->obs; ta.act = s->act; ta.adv = s->adv; ta.ret = s->ret; ta.old_logp = s->old_logp; ta.log_std = s->log_std;
  for (int t = 0; t < 2; t++) {
    float *ws = s->workspace + t * L.per_trunk;
    ta.pkF[t] = reinterpret_cast<const float4 *>(ws + L.pkF);
    ta.pkW2T[t] = reinterpret_cast<const float4 *>(ws + L.pkW2T);
    ta.pkW3T[t] = reinterpret_cast<const float4 *>(ws + L.pkW3T);
    ta.b1[t] = s->b[t][0]; ta.b2[t] = s->b[t][1]; ta.b3[t] = s->b[t][2];
    ta.h1g[t] = ws + L.h1; ta.h2g[t] = ws + L.h2; ta.dz1g[t] = ws + L.dz1; ta.dz2g[t] = ws + L.dz2; ta.d3g[t] = ws + L.d3;
  }
  ta.part = stats + 8;
  ta.stats = stats; ta.out8 = s->out8; ta.g_log_std = s->g_log_std; ta.clip = s->clip_range; ta.vf_coef = s->vf_coef;
  hipLaunchKernelGGL(mlp_fwdbwd_kernel, dim3(B / POL_R, 2), dim3(MLP_THREADS), lds, st, ta);

  // ---- 3. weight / bias gradients of the six layers + loss epilogue
  MlpWgradArgs wa;
  int nw = 0, nq = 0;
  static const int sk_env = getenv("DM_WGRAD_SPLITK") ? atoi(getenv("DM_WGRAD_SPLITK")) : 8;   // workgroup-level split-K (experiments)
  int splitk = s->reserved > 0 ? s->reserved : (sk_env > 0 ? sk_env : 8);
  while (splitk > 1 && (B % (splitk * 8 * WG3_U)) != 0) splitk >>= 1;                         // kchunk / 4 a multiple of 2 WG3_U
  if ((B % (splitk * 8 * WG3_U)) != 0) return -22;
  auto add_wg = [&](const float *dY, const float *X, float *dW, float *db, int O, int I) {
    MlpWgradJob &J = wa.j[nq++];
    J.dY = dY; J.X = X; J.dW = dW; J.db = db; J.O = O; J.I = I; J.nro = (O + 63) / 64; J.nri = (I + 63) / 64;
    J.kchunk = B / splitk; J.first = nw;
    nw += J.nro * J.nri * splitk;
  };
  for (int t = 0; t < 2; t++) {
    float *ws = s->workspace + t * L.per_trunk;
    const int Aout = t == 0 ? A : 1;
    add_wg(ws + L.dz1, s->obs, s->gW[t][0], s->gb[t][0], H1, D);
    add_wg(ws + L.dz2, ws + L.h1, s->gW[t][1], s->gb[t][1], H2, H1);
    add_wg(ws + L.d3, ws + L.h2, s->gW[t][2], s->gb[t][2], Aout, H2);
  }
  wa.B = B; wa.nblocks = nw; wa.log_std = s->log_std; wa.stats = stats; wa.g_log_std = s->g_log_std; wa.out8 = s->out8; wa.loss_acc = s->loss_acc; wa.A = A; wa.part = stats + 8; wa.nwg = B / POL_R;
  wa.vf_coef = s->vf_coef; wa.ent_coef = s->ent_coef;
  hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(nw + 1), dim3(256), 0, st, wa);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

#ifdef MLP_PROFILE
extern "C" int dm_ppo_mlp_prof(long long *host128) {
  return hipMemcpyFromSymbol(host128, HIP_SYMBOL(mlp_prof_buf), sizeof(long long) * MLP_NW * 16) == hipSuccess ? 0 : -5;
}
#endif
