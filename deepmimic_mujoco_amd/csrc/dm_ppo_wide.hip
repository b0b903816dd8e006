// One PPO minibatch of the WIDE actor-critic MLP (net_arch = [1024, 512], the net BASELINE configs 3-5 name) on the bf16 matrix pipe.
//
// Reference: src/sb3_ppo.py:254-271,307-312 -> [EXT] SB3 PPO.train on MlpPolicy: per minibatch evaluate_actions (both trunks), the
// clipped-surrogate / value / entropy loss, backward through both trunks.  The library path of this net is six ~55 us fp32 GEMMs per
// trunk on a two-stream dependency chain plus a dozen small launches (0.39 ms per optimizer step; with bf16 library GEMMs 0.37 ms: the
// chain, not the GEMMs, bounds it).  Here the whole forward / loss / input-gradient chain of BOTH trunks is ONE launch:
//   1. wide_pack_kernel    fp32 master weights -> bf16 in the two layouts the chain reads (W for X W^T, W^T for dZ W); advantage
//                          statistics; optional folds: clearing the gradient arena, Adam's begin                            (1 launch)
//   2. wide_fwdbwd_kernel  a workgroup of eight waves carries 32 minibatch rows of one trunk through layer 1, layer 2, the head, the
//                          loss (the arithmetic of ppo_loss_kernel), and back: d head, d layer 2 (x tanh'), d layer 1 (x tanh');
//                          activations live in LDS as bf16 (64 + 32 + 32 KB), every product is v_mfma_f32_32x32x16_bf16 with fp32
//                          accumulation, weights stream from L2 in FRAGMENT order (below): one operand load of a wave is 1 KB of
//                          consecutive bytes; outputs: the bf16 activations / pre-activation gradients the weight gradients need,
//                          per-workgroup loss partials, and on narrow nets the bias gradients (column sums, fp32 atomics)     (1 launch)
//   3. wide_wgrad_kernel   dW = dZ^T X of all six layers: the chain leaves its activations and pre-activation gradients TRANSPOSED
//                          (features x batch) in the same fragment order, written straight from the accumulator registers (four
//                          consecutive batch rows of a column are 8 bytes of a fragment); a workgroup of four waves owns four
//                          32 x 32 tiles of dW, the waves split the batch and meet in LDS (plain stores on wide nets, split-K
//                          with fp32 atomics only on narrow ones); the bias gradients of wide nets come from one more MFMA per
//                          k-step against a fragment of ones; block 0 sums the loss partials                                 (1 launch)
// Fragment order of a matrix M[N][K] (N % 32 == 0, K % 16 == 0) that feeds v_mfma_f32_32x32x16_bf16 as the operand with row / column
// index n and reduction index k: element (n, k) lives at ((((n >> 5) * (K >> 4) + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (n & 31)) * 8
// + (k & 7)) — tile of 32 rows, k-step of 16, then the 64 lanes' 16-byte fragments in lane order.  A wave's load of one fragment is
// base + lane * 16 bytes: eight full 128-byte lines.  (Row-major operands made every load touch 32 lines for 32 bytes each and lean
// on the 32 KB vector L1 to hold 512 half-used lines across k-steps: 92 us for the chain, 58 us for the weight gradients; now 43 + 38.)
// fp32 master weights, fp32 loss arithmetic, fp32 gradients and Adam; bf16 operands of the products only (north_star: "MFMA used
// only for the policy-MLP GEMMs").  Included by dm_abi.hip after dm_ppo_mlp.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/deepmimic_hip.h"

namespace {

typedef short wide_b8 __attribute__((ext_vector_type(8)));      // 8 bf16 = one MFMA operand fragment (4 VGPRs)
typedef float wide_f16 __attribute__((ext_vector_type(16)));    // 32 x 32 accumulator: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)

__device__ __forceinline__ unsigned short wide_f2bf(float x) {  // round to nearest even
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
// two floats -> two bf16 in one dword (low half = a): gfx950's v_cvt_pk_bf16_f32, round to nearest even like wide_f2bf
typedef __bf16 wide_bf2 __attribute__((ext_vector_type(2)));
typedef float wide_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned wide_pk2(float a, float b) {
  const wide_f2 v = {a, b};
  const wide_bf2 r = __builtin_convertvector(v, wide_bf2);
  return *reinterpret_cast<const unsigned *>(&r);
}
__device__ __forceinline__ float wide_bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ float wide_tanh(float x) { return 1.f - __fdividef(2.f, 1.f + __expf(2.f * x)); }
__device__ __forceinline__ int wide_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
// fragment order (file header): element index of (n, k) in a matrix with K / 16 = nks k-steps
__device__ __forceinline__ size_t wide_frag(int n, int k, int nks) {
  return ((((size_t)(n >> 5) * nks + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (n & 31)) << 3) + (k & 7);
}
// the fragment of tile t, k-step ks for this lane
__device__ __forceinline__ wide_b8 wide_ldfrag(const unsigned short *M, int t, int ks, int nks, int lane) {
  return *reinterpret_cast<const wide_b8 *>(M + ((((size_t)t * nks + ks) * 64 + lane) << 3));
}

#ifdef WIDE_PROFILE   // diagnostic build: s_memtime at the phase boundaries of workgroup 0 (thread 0), printed at the end
#define WPROF(k) do { if (tid == 0 && blockIdx.x == 0) wprof[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WPROF(k) do {} while (0)
#endif
constexpr int WIDE_BIAS_WGRAD_H1 = 512; // from this first-layer width up the bias gradients come from the weight-gradient launch
constexpr int WIDE_WG_WAVES = 4;        // waves of a weight-gradient workgroup: each takes a quarter of the workgroup's batch slice
constexpr int WIDE_MAX_SPLITK = 8;      // more slices than this and the fp32 atomics into one tile queue up (sk = 64 on a [256,128] net: 46 us)
constexpr int WIDE_R = 32, WIDE_NW = 8, WIDE_THREADS = 64 * WIDE_NW, WIDE_PART = 40;

struct WidePackArgs {
  const float *W[2][3];
  unsigned short *pk[2];          // per trunk: W1 [H1][Dp] | W2 [H2][H1] | W2T [H1][H2] | W3 [32][H2] | W3T [H2][32], each in fragment order
  int D, Dp, H1, H2, A[2];
  long long total;                // elements of one trunk's packed block
  int pack_blocks;                // blocks [0, 2 * pack_blocks) pack, block 2 * pack_blocks = statistics, the rest clear zero_ptr
  const float *adv; int B, normalize; float *stats, *out8;
  float *zero_ptr; long long zero_floats; float *adam_state2;
};

__global__ void __launch_bounds__(256) wide_pack_kernel(WidePackArgs a) {
  const int blk = blockIdx.x;
  if (blk == 2 * a.pack_blocks) {
    if (a.B <= 8192) mlp_adv_stats(a.adv, a.B, a.normalize, a.stats, a.out8);     // every load in flight at once (dm_ppo_mlp.hip)
    else ppo_prepare_body(a.adv, a.B, a.normalize, a.stats, a.out8, nullptr, 0);
    if (a.adam_state2 && threadIdx.x == 0) { a.adam_state2[0] = 0.f; a.adam_state2[1] += 1.f; }     // Adam's begin
    return;
  }
  if (blk > 2 * a.pack_blocks) {
    const long long i = ((long long)(blk - 2 * a.pack_blocks - 1) * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int c = 0; c < 4; c++) if (i + c < a.zero_floats) a.zero_ptr[i + c] = 0.f;
    return;
  }
  const int t = blk / a.pack_blocks;
  const long long n1 = (long long)a.H1 * a.Dp, n2 = (long long)a.H2 * a.H1, n3 = 32ll * a.H2;
  // one thread per 16-byte fragment (8 consecutive k of one row n), every block in fragment order
  for (long long c = (long long)(blk % a.pack_blocks) * 256 + threadIdx.x; c < (a.total >> 3); c += (long long)a.pack_blocks * 256) {
    long long i = c << 3;
    int which, nks;
    if (i < n1) { which = 0; nks = a.Dp >> 4; }
    else if (i < n1 + n2) { which = 1; nks = a.H1 >> 4; i -= n1; }
    else if (i < n1 + 2 * n2) { which = 2; nks = a.H2 >> 4; i -= n1 + n2; }
    else if (i < n1 + 2 * n2 + n3) { which = 3; nks = a.H2 >> 4; i -= n1 + 2 * n2; }
    else { which = 4; nks = 2; i -= n1 + 2 * n2 + n3; }
    const long long ch = i >> 3;
    const int l = (int)(ch & 63), ks = (int)((ch >> 6) % nks), tt = (int)((ch >> 6) / nks);
    const int n = tt * 32 + (l & 31), k0 = ks * 16 + 8 * (l >> 5);
    unsigned short o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = k0 + j;
      float v;
      switch (which) {
        case 0: v = k < a.D ? a.W[t][0][(size_t)n * a.D + k] : 0.f; break;                 // W1 [H1][Dp]
        case 1: v = a.W[t][1][(size_t)n * a.H1 + k]; break;                                // W2 [H2][H1]
        case 2: v = a.W[t][1][(size_t)k * a.H1 + n]; break;                                // W2^T [H1][H2]
        case 3: v = n < a.A[t] ? a.W[t][2][(size_t)n * a.H2 + k] : 0.f; break;             // W3 [32][H2]
        default: v = k < a.A[t] ? a.W[t][2][(size_t)k * a.H2 + n] : 0.f; break;            // W3^T [H2][32]
      }
      o[j] = wide_f2bf(v);
    }
    uint4 u;
    u.x = o[0] | ((unsigned)o[1] << 16); u.y = o[2] | ((unsigned)o[3] << 16); u.z = o[4] | ((unsigned)o[5] << 16); u.w = o[6] | ((unsigned)o[7] << 16);
    *reinterpret_cast<uint4 *>(a.pk[t] + (c << 3)) = u;
  }
}

struct WideArgs {
  int B, D, Dp, H1, H2, A;
  const float *obs, *act, *adv, *ret, *old_logp, *log_std, *stats;
  const unsigned short *pk[2];
  const float *b1[2], *b2[2], *b3[2];
  float *gb1[2], *gb2[2], *gb3[2];
  unsigned short *xbT, *h1T[2], *dz1T[2], *h2T[2], *dz2T[2], *dz3T[2];   // [features][B] bf16
  float *part;
  float clip, vf_coef;
  int bias_in_chain;               // 1: the chain adds its 32-row column sums of dZ to the bias gradients (small nets); 0: the weight-gradient launch forms them
};

// four consecutive batch rows row0 .. row0 + 3 (row0 % 4 == 0) of one feature column — accumulator registers 4 q .. 4 q + 3 of a
// lane — as 8 bytes of the transposed (features x batch) array in fragment order
__device__ __forceinline__ void wide_store_t4(unsigned short *T, size_t B, int col, int row0, float v0, float v1, float v2, float v3) {
  uint2 u;
  u.x = wide_pk2(v0, v1);
  u.y = wide_pk2(v2, v3);
  *reinterpret_cast<uint2 *>(T + wide_frag(col, row0, (int)(B >> 4))) = u;
}

// the same four values also into the LDS copy of the activations (rows row0 .. row0 + 3 of column col, row stride `stride` bytes)
__device__ __forceinline__ void wide_put4(char *L, int stride, int lrow0, int col, unsigned short *T, size_t B, int grow0, float v0, float v1,
                                          float v2, float v3) {
  uint2 u;
  u.x = wide_pk2(v0, v1);
  u.y = wide_pk2(v2, v3);
  char *p = L + lrow0 * stride + 2 * col;
  *reinterpret_cast<unsigned short *>(p) = (unsigned short)u.x;
  *reinterpret_cast<unsigned short *>(p + stride) = (unsigned short)(u.x >> 16);
  *reinterpret_cast<unsigned short *>(p + 2 * stride) = (unsigned short)u.y;
  *reinterpret_cast<unsigned short *>(p + 3 * stride) = (unsigned short)(u.y >> 16);
  *reinterpret_cast<uint2 *>(T + wide_frag(col, grow0, (int)(B >> 4))) = u;
}

__global__ void __launch_bounds__(WIDE_THREADS) wide_fwdbwd_kernel(WideArgs a) {
  extern __shared__ __align__(16) char wide_lds[];
  // workgroups go round the eight XCDs in launch order: even XCDs take the policy trunk, odd ones the value trunk, so an XCD's L2
  // streams ONE trunk's 2.3 MB of weights (both trunks: 4.6 MB against 4 MB of L2)
#ifdef WIDE_PROFILE
  unsigned long long wprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const int nt = a.B / WIDE_R, id = blockIdx.x;
  int trunk, tile;
  if ((nt & 3) == 0) { const int xcd = id & 7; trunk = xcd & 1; tile = (id >> 3) * 4 + (xcd >> 1); }
  else { trunk = id / nt; tile = id % nt; }
  const int b0 = tile * WIDE_R;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int Dp = a.Dp, H1 = a.H1, H2 = a.H2, At = trunk ? 1 : a.A;      // head width: actions (policy trunk) / 1 (value trunk)
  const size_t B = (size_t)a.B;
  const int SX = 2 * Dp + 16, S1 = 2 * H1 + 16, S2 = 2 * H2 + 16, S3 = 64 + 16;     // row strides (bytes): 16-byte reads of 32 rows spread over the banks
  const int RZ = WIDE_R * S2 > WIDE_NW * 4096 ? WIDE_R * S2 : WIDE_NW * 4096;      // dZ2s shares its block with the head's partial tiles
  char *Xs = wide_lds, *H1s = Xs + WIDE_R * SX, *H2s = H1s + WIDE_R * S1, *dZ2s = H2s + WIDE_R * S2, *dZ3s = dZ2s + RZ;
  float *red = reinterpret_cast<float *>(dZ2s);                 // head: eight K-slices of the 32 x 32 output (32 KB; dZ2s is not live yet)
  float *outs = reinterpret_cast<float *>(dZ3s + WIDE_R * S3);  // [32][33] head outputs (+ bias)
  float *accs = outs + 32 * 33;                                 // [16][36] loss partials of the half-waves
  const unsigned short *W1 = a.pk[trunk], *W2 = W1 + (size_t)H1 * Dp, *W2T = W2 + (size_t)H2 * H1, *W3 = W2T + (size_t)H1 * H2,
                       *W3T = W3 + 32 * (size_t)H2;

  WPROF(0);
  // ---- observations -> bf16 rows (zero-padded to Dp); trunk 0 also files them, transposed, for the weight gradient of layer 1
  for (int i = tid; i < WIDE_R * Dp; i += WIDE_THREADS) {
    const int k = i >> 5, m = i & 31;                                  // (consecutive threads: consecutive rows of one column)
    const unsigned short v = wide_f2bf(k < a.D ? a.obs[(size_t)(b0 + m) * a.D + k] : 0.f);
    *reinterpret_cast<unsigned short *>(Xs + m * SX + 2 * k) = v;
    if (trunk == 0) a.xbT[wide_frag(k, b0 + m, (int)(B >> 4))] = v;
  }
  __syncthreads();

  WPROF(1);
  // ---- layer 1: H1 = tanh(X W1^T + b1).  A = X rows from LDS (lane: row r, k = 8 h + j), B = W1 rows from L2 (lane: column r)
  // (the weight fragments of a tile — at most seven k-steps — are requested together, and those of the wave's NEXT tile before this
  // tile's MFMAs: with one load per k-step inside the loop every MFMA waited for its own L2 round trip and this layer, a ninth of
  // layer 2's work, took as long as layer 2: 25 k of the kernel's 102 k ticks)
  {
    const int nks1 = Dp >> 4, ntile = H1 >> 5;
    const wide_b8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    wide_b8 pa[7], pb[7];
    float ba = 0.f, bb = 0.f;                     // the tile's bias travels with its weights (a load after the MFMAs is an exposed round trip)
    auto ld = [&](wide_b8 (&P)[7], float &bias, const int t) {
#pragma unroll
      for (int ks = 0; ks < 7; ks++) P[ks] = wide_ldfrag(W1, t, ks < nks1 ? ks : nks1 - 1, nks1, lane);   // no branch: a repeated fragment, unused
      bias = a.b1[trunk][t * 32 + r];
    };
    auto run = [&](const wide_b8 (&P)[7], const float bias, const int t) {
      wide_f16 acc;
#pragma unroll
      for (int j = 0; j < 16; j++) acc[j] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 7; ks++) {   // straight-line: the k-steps beyond Dp multiply by a zero A fragment
        const wide_b8 a0 = *reinterpret_cast<const wide_b8 *>(Xs + r * SX + ((ks < nks1 ? ks : 0) * 16 + 8 * h) * 2);
        const wide_b8 av = ks < nks1 ? a0 : zero8;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, P[ks], acc, 0, 0, 0);
      }
      const int n = t * 32 + r;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = wide_tanh(acc[4 * q + i] + bias);
        wide_put4(H1s, S1, 8 * q + 4 * h, n, a.h1T[trunk], B, b0 + 8 * q + 4 * h, v[0], v[1], v[2], v[3]);
      }
    };
    int t = wave;
    if (t < ntile) ld(pa, ba, t);
    for (; t < ntile; t += 2 * WIDE_NW) {
      if (t + WIDE_NW < ntile) ld(pb, bb, t + WIDE_NW);
      run(pa, ba, t);
      if (t + 2 * WIDE_NW < ntile) ld(pa, ba, t + 2 * WIDE_NW);
      if (t + WIDE_NW < ntile) run(pb, bb, t + WIDE_NW);
    }
  }
  __syncthreads();

  WPROF(2);
  // ---- layer 2: H2 = tanh(H1 W2^T + b2): two 32-column tiles per wave per pass share every A fragment; the weight rows stream
  // from L2 through a two-deep ring of register blocks (eight k-steps each), so sixteen loads per lane are always in flight
  for (int t0 = 2 * wave; t0 < (H2 >> 5); t0 += 2 * WIDE_NW) {
    wide_f16 acc0, acc1;
#pragma unroll
    for (int j = 0; j < 16; j++) { acc0[j] = 0.f; acc1[j] = 0.f; }
    const char *ap = H1s + r * S1 + 16 * h;
    const int n0 = t0 * 32 + r, n1 = n0 + 32;
    const float bias0 = a.b2[trunk][n0], bias1 = a.b2[trunk][n1];      // requested with the first weights, used after the k loop
    // fragments of tile t0 / t0 + 1: k-step ks at wr + ks * 512 elements (1 KB per wave), consecutive k-steps consecutive in memory
    const unsigned short *wr0 = W2 + ((size_t)t0 * (H1 >> 4) * 64 + lane) * 8, *wr1 = wr0 + (size_t)(H1 >> 4) * 512;
    constexpr int KB = 8;
    wide_b8 p0[KB], p1[KB], q0[KB], q1[KB];
    const int nblk = (H1 >> 4) / KB;                       // H1 % 256 == 0: an even number of blocks
#pragma unroll
    for (int i = 0; i < KB; i++) { p0[i] = *reinterpret_cast<const wide_b8 *>(wr0 + i * 512); p1[i] = *reinterpret_cast<const wide_b8 *>(wr1 + i * 512); }
    for (int kb = 0; kb < nblk; kb += 2) {
#pragma unroll
      for (int i = 0; i < KB; i++) { q0[i] = *reinterpret_cast<const wide_b8 *>(wr0 + ((kb + 1) * KB + i) * 512); q1[i] = *reinterpret_cast<const wide_b8 *>(wr1 + ((kb + 1) * KB + i) * 512); }
#pragma unroll
      for (int i = 0; i < KB; i++) {
        const wide_b8 av = *reinterpret_cast<const wide_b8 *>(ap + (kb * KB + i) * 32);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, p0[i], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, p1[i], acc1, 0, 0, 0);
      }
      if (kb + 2 < nblk) {
#pragma unroll
        for (int i = 0; i < KB; i++) { p0[i] = *reinterpret_cast<const wide_b8 *>(wr0 + ((kb + 2) * KB + i) * 512); p1[i] = *reinterpret_cast<const wide_b8 *>(wr1 + ((kb + 2) * KB + i) * 512); }
      }
#pragma unroll
      for (int i = 0; i < KB; i++) {
        const wide_b8 av = *reinterpret_cast<const wide_b8 *>(ap + ((kb + 1) * KB + i) * 32);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, q0[i], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, q1[i], acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      float v[4], w[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { v[i] = wide_tanh(acc0[4 * q + i] + bias0); w[i] = wide_tanh(acc1[4 * q + i] + bias1); }
      wide_put4(H2s, S2, 8 * q + 4 * h, n0, a.h2T[trunk], B, b0 + 8 * q + 4 * h, v[0], v[1], v[2], v[3]);
      wide_put4(H2s, S2, 8 * q + 4 * h, n1, a.h2T[trunk], B, b0 + 8 * q + 4 * h, w[0], w[1], w[2], w[3]);
    }
  }
  __syncthreads();

  WPROF(3);
  // ---- head: out = H2 W3^T (32 columns, padded): K split over the eight waves, partial tiles summed through LDS
  {
    wide_f16 acc;
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    const int kper = (H2 >> 4) / WIDE_NW;                     // k-steps per wave (H2 = 512: 4)
    for (int q = 0; q < kper; q++) {
      const int ks = wave * kper + q;
      const wide_b8 av = *reinterpret_cast<const wide_b8 *>(H2s + r * S2 + (ks * 16 + 8 * h) * 2);
      const wide_b8 bv = wide_ldfrag(W3, 0, ks, H2 >> 4, lane);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) red[wave * 1024 + j * 64 + lane] = acc[j];
  }
  __syncthreads();
  for (int e = tid; e < 1024; e += WIDE_THREADS) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WIDE_NW; w++) s += red[w * 1024 + e];
    const int l = e & 63, j = e >> 6, n = l & 31;
    outs[wide_row(j, l >> 5) * 33 + n] = s + (n < At ? a.b3[trunk][n] : 0.f);
  }
  __syncthreads();

  WPROF(4);
  // ---- loss (arithmetic of ppo_loss_kernel): a half-wave per row, lane = action index; d out -> dZ3s (bf16) and, transposed, HBM
  {
    const int j = tid & 31, hw = tid >> 5;                    // 16 half-waves, two rows each
    const bool ja = j < a.A;
    const float invB = 1.0f / (float)a.B, amean = a.stats[0], ainv = a.stats[1];
    float g_ls = 0.f, pg = 0.f, vl = 0.f, kl = 0.f, cf = 0.f;
    float ls = 0.f, iv = 0.f, lconst = 0.f;
    if (trunk == 0) {
      ls = ja ? a.log_std[j] : 0.f;
      iv = ja ? expf(-2.f * ls) : 0.f;
      float sum_ls = ls;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) sum_ls += __shfl_xor(sum_ls, o);
      lconst = -sum_ls - 0.5f * 1.8378770664093453f * (float)a.A;
    }
    for (int m = hw; m < WIDE_R; m += 16) {
      const int b = b0 + m;
      float dz = 0.f;
      if (trunk == 0) {
        const float d = ja ? a.act[(size_t)b * a.A + j] - outs[m * 33 + j] : 0.f;
        const float z2 = d * d * iv;
        float zs = z2;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) zs += __shfl_xor(zs, o);
        const float logp = -0.5f * zs + lconst;
        const float a_n = (a.adv[b] - amean) * ainv;
        const float lr = logp - a.old_logp[b];
        const float ratio = expf(lr);
        const float rc = fminf(fmaxf(ratio, 1.f - a.clip), 1.f + a.clip);
        const float p1 = a_n * ratio, p2 = a_n * rc;
        const bool inside = (ratio >= 1.f - a.clip) && (ratio <= 1.f + a.clip);
        const float dr = (inside || p1 < p2) ? a_n : 0.f;
        const float dlogp = -invB * dr * ratio;
        dz = ja ? dlogp * d * iv : 0.f;
        g_ls += ja ? dlogp * (z2 - 1.f) : 0.f;
        if (j == 0) { pg += -fminf(p1, p2); kl += (ratio - 1.f) - lr; cf += (fabsf(ratio - 1.f) > a.clip) ? 1.f : 0.f; }
      } else {
        const float dv = outs[m * 33] - a.ret[b];
        dz = (j == 0) ? a.vf_coef * 2.f * invB * dv : 0.f;
        if (j == 0) vl += dv * dv;
      }
      const unsigned short dzb = wide_f2bf(dz);
      *reinterpret_cast<unsigned short *>(dZ3s + m * S3 + 2 * j) = dzb;
      a.dz3T[trunk][wide_frag(j, b, (int)(B >> 4))] = dzb;
    }
    accs[hw * 36 + j] = g_ls;
    if (j == 0) { accs[hw * 36 + 32] = pg; accs[hw * 36 + 33] = vl; accs[hw * 36 + 34] = kl; accs[hw * 36 + 35] = cf; }
  }
  __syncthreads();
  if (tid < 36) {   // this workgroup's partial sums (summed in fixed order by the weight-gradient launch's last block)
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) t += accs[i * 36 + tid];
    a.part[((size_t)trunk * nt + tile) * WIDE_PART + tid] = t;
  }
  // bias gradients = column sums of dZ.  Wide nets: the weight-gradient launch forms them with one more MFMA per k-step against a
  // fragment of ones (128 workgroups adding 32-row partial sums to the same 1 024 addresses from here cost d layer 1 ~15 % of its
  // time: 117.5 -> 112.4 us per optimizer step); narrow nets keep the sums here (their few weight-gradient tiles would carry the
  // extra MFMAs on the critical path: [256,128] 59.6 -> 65.2 us)
  if (a.bias_in_chain && tid < At) {
    float s = 0.f;
    for (int m = 0; m < WIDE_R; m++) s += wide_bf2f(*reinterpret_cast<const unsigned short *>(dZ3s + m * S3 + 2 * tid));
    atomicAdd(&a.gb3[trunk][tid], s);
  }

  WPROF(5);
  // ---- d layer 2: dZ2 = (dZ3 W3) x (1 - H2^2).  A = dZ3 rows (K = 32: two k-steps), B[k = a][col = n] = W3T row n
  for (int t0 = 2 * wave; t0 < (H2 >> 5); t0 += 2 * WIDE_NW) {
    wide_f16 acc0, acc1;
#pragma unroll
    for (int j = 0; j < 16; j++) { acc0[j] = 0.f; acc1[j] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const wide_b8 av = *reinterpret_cast<const wide_b8 *>(dZ3s + r * S3 + (ks * 16 + 8 * h) * 2);
      const wide_b8 b0v = wide_ldfrag(W3T, t0, ks, 2, lane), b1v = wide_ldfrag(W3T, t0 + 1, ks, 2, lane);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b0v, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b1v, acc1, 0, 0, 0);
    }
    const int n0 = t0 * 32 + r, n1 = n0 + 32;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      float v[4], w[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int m = 8 * q + 4 * h + i;
        const float x0 = wide_bf2f(*reinterpret_cast<const unsigned short *>(H2s + m * S2 + 2 * n0));
        const float x1 = wide_bf2f(*reinterpret_cast<const unsigned short *>(H2s + m * S2 + 2 * n1));
        v[i] = acc0[4 * q + i] * (1.f - x0 * x0); w[i] = acc1[4 * q + i] * (1.f - x1 * x1);
      }
      wide_put4(dZ2s, S2, 8 * q + 4 * h, n0, a.dz2T[trunk], B, b0 + 8 * q + 4 * h, v[0], v[1], v[2], v[3]);
      wide_put4(dZ2s, S2, 8 * q + 4 * h, n1, a.dz2T[trunk], B, b0 + 8 * q + 4 * h, w[0], w[1], w[2], w[3]);
      s0 += (v[0] + v[1]) + (v[2] + v[3]); s1 += (w[0] + w[1]) + (w[2] + w[3]);
    }
    if (a.bias_in_chain) {
      s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32);
      if (h == 0) { atomicAdd(&a.gb2[trunk][n0], s0); atomicAdd(&a.gb2[trunk][n1], s1); }
    }
  }
  __syncthreads();

  WPROF(6);
  // ---- d layer 1: dZ1 = (dZ2 W2) x (1 - H1^2).  A = dZ2 rows (K = H2), B[k = n][col = k1] = W2T row k1; four 32-column tiles per
  // wave share every A fragment; weight rows through a two-deep ring of four-k-step register blocks
  for (int t0 = 4 * wave; t0 < (H1 >> 5); t0 += 4 * WIDE_NW) {
    wide_f16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int j = 0; j < 16; j++) acc[q][j] = 0.f;
    const char *ap = dZ2s + r * S2 + 16 * h;
    const unsigned short *wrow = W2T + ((size_t)t0 * (H2 >> 4) * 64 + lane) * 8;     // tile t0 + q: + q * (H2 / 16) * 512 elements
    const size_t tstride = (size_t)(H2 >> 4) * 512;
    constexpr int KB = 4;
    wide_b8 pb[4][KB], qb[4][KB];
    const int nblk = (H2 >> 4) / KB;                       // H2 % 128 == 0: an even number of blocks
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int i = 0; i < KB; i++) pb[q][i] = *reinterpret_cast<const wide_b8 *>(wrow + q * tstride + i * 512);
    for (int kb = 0; kb < nblk; kb += 2) {
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < KB; i++) qb[q][i] = *reinterpret_cast<const wide_b8 *>(wrow + q * tstride + ((kb + 1) * KB + i) * 512);
#pragma unroll
      for (int i = 0; i < KB; i++) {
        const wide_b8 av = *reinterpret_cast<const wide_b8 *>(ap + (kb * KB + i) * 32);
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, pb[q][i], acc[q], 0, 0, 0);
      }
      if (kb + 2 < nblk) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
          for (int i = 0; i < KB; i++) pb[q][i] = *reinterpret_cast<const wide_b8 *>(wrow + q * tstride + ((kb + 2) * KB + i) * 512);
      }
#pragma unroll
      for (int i = 0; i < KB; i++) {
        const wide_b8 av = *reinterpret_cast<const wide_b8 *>(ap + ((kb + 1) * KB + i) * 32);
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, qb[q][i], acc[q], 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int n = (t0 + q) * 32 + r;
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float x = wide_bf2f(*reinterpret_cast<const unsigned short *>(H1s + (8 * g + 4 * h + i) * S1 + 2 * n));
          v[i] = acc[q][4 * g + i] * (1.f - x * x);
        }
        wide_store_t4(a.dz1T[trunk], B, n, b0 + 8 * g + 4 * h, v[0], v[1], v[2], v[3]);
        s += (v[0] + v[1]) + (v[2] + v[3]);
      }
      if (a.bias_in_chain) {
        s += __shfl_xor(s, 32);
        if (h == 0) atomicAdd(&a.gb1[trunk][n], s);
      }
    }
  }
#ifdef WIDE_PROFILE
  __syncthreads();
  WPROF(7);
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("wide_fwdbwd wg0 ticks: obs %llu layer1 %llu layer2 %llu head %llu loss %llu dlayer2 %llu dlayer1 %llu\n", wprof[1] - wprof[0], wprof[2] - wprof[1],
           wprof[3] - wprof[2], wprof[4] - wprof[3], wprof[5] - wprof[4], wprof[6] - wprof[5], wprof[7] - wprof[6]);
#endif
}

// dW = dZ^T X for the six layers.  Operands are the TRANSPOSED arrays the chain wrote, in fragment order: AT = dZ^T (O x B),
// XT = X^T (I x B): a wave's fragment load is 1 KB of consecutive bytes, the k-steps of a tile follow each other.  A workgroup of
// four waves owns four 32 x 32 tiles of dW — 64 x 64 (RO = CI = 2), or 32 x 128 for the heads (RO = 1) — and the whole batch (or
// 1 / splitk of it on small nets): every wave accumulates all four tiles over its quarter of the batch rows, the four partial
// blocks meet in LDS (64 KB: two workgroups per CU) and each wave finishes one tile — a plain store when splitk == 1: no atomics,
// fixed summation order.  Measured on the way (us per optimizer step of the [1024,512] learner, 4 096 rows):
//   one wave per 32 x 128, four waves per workgroup re-reading X                                         216
//   one wave per 128 x 128 (16 accumulators: 928 spilled VGPRs)                                            375
//   one wave per 64 x 128, split-K 4-16 over workgroups with fp32 atomics                                  185 -> 132.6 (fragment order)
//   four waves per 64 x 128 splitting the batch inside the workgroup, no atomics (128 KB LDS: 164 workgroups)  125.6
//   ... with global split-K 2 on top (292 workgroups, atomics back)                                        136.2
//   four waves per 64 x 64, 64 KB LDS, two workgroups per CU (328 workgroups), two-k-step ring             122.7  <- this
// Block 0 sums the loss partials (fixed order).
struct WideWgradJob { const unsigned short *AT, *XT; float *dW, *db; int O, I, ldw, ro, otiles, itiles, splitk, first, per; };
struct WideWgradArgs {
  WideWgradJob j[6];
  int njobs, nblocks, B;
  const float *part; int nblk, A; const float *log_std; float vf_coef, ent_coef; const float *stats; float *g_log_std, *out8, *loss_acc;
};

template <int RO, int CI>
__device__ __forceinline__ void wide_wgrad_tile(const WideWgradJob &J, const int Bn, const int ot, const int it, const int ks, const int wave, const int lane,
                                                float *wl) {
  constexpr int NT = RO * CI;                    // 32 x 32 tiles of dW per wave (a multiple of 4)
  const int r = lane & 31, h = lane >> 5;
  const int o0 = ot * 32 * RO, i0 = it * 32 * CI;
  const size_t B = (size_t)Bn;
  // batch rows of this workgroup's split, in units of 64 rows; the four waves take a quarter of the units each
  const int units = (Bn / J.splitk) >> 6, u0 = wave * units / WIDE_WG_WAVES, u1 = (wave + 1) * units / WIDE_WG_WAVES;
  const int k0 = ks * (Bn / J.splitk) + u0 * 64;
  const size_t tstride = (B >> 4) * 512;                        // elements of one 32-feature tile: (B / 16) k-steps of 512
  const unsigned short *ap = J.AT + (size_t)(o0 >> 5) * tstride + ((size_t)(k0 >> 4) * 64 + lane) * 8;
  const unsigned short *xp = J.XT + (size_t)(i0 >> 5) * tstride + ((size_t)(k0 >> 4) * 64 + lane) * 8;
  bool oa[RO], ia[CI];
#pragma unroll
  for (int p = 0; p < RO; p++) oa[p] = (o0 + 32 * p + r) < J.O;
#pragma unroll
  for (int q = 0; q < CI; q++) ia[q] = (i0 + 32 * q + r) < J.I;
  wide_f16 acc[RO][CI];
#pragma unroll
  for (int p = 0; p < RO; p++)
#pragma unroll
    for (int q = 0; q < CI; q++)
#pragma unroll
      for (int j = 0; j < 16; j++) acc[p][q][j] = 0.f;
  // bias gradient db = dZ^T 1: the workgroups of the first input tile multiply their dZ^T fragments by a fragment of ones as well
  // (every column of that 32 x 32 product is the row sum; bf16 1.0 = 0x3F80)
  const bool bias = J.db != nullptr && it == 0;
  const wide_b8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  wide_f16 accb[RO];
#pragma unroll
  for (int p = 0; p < RO; p++)
#pragma unroll
    for (int j = 0; j < 16; j++) accb[p][j] = 0.f;
  const wide_b8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  // two-deep ring of register blocks of two k-steps: the loads of the next block are in flight under the 16 MFMAs of this one
  constexpr int KB = 2;       // (a four-k-step ring: 126.0 against 122.7 us per optimizer step)
  wide_b8 av[RO][KB], xv[CI][KB], aw[RO][KB], xw[CI][KB];
  auto load_blk = [&](wide_b8 (&A_)[RO][KB], wide_b8 (&X_)[CI][KB], const int kk) {
#pragma unroll
    for (int i = 0; i < KB; i++) {
#pragma unroll
      for (int p = 0; p < RO; p++) A_[p][i] = oa[p] ? *reinterpret_cast<const wide_b8 *>(ap + p * tstride + (size_t)(kk + i) * 512) : zero;
#pragma unroll
      for (int q = 0; q < CI; q++) X_[q][i] = ia[q] ? *reinterpret_cast<const wide_b8 *>(xp + q * tstride + (size_t)(kk + i) * 512) : zero;
    }
  };
  auto mma_blk = [&](const wide_b8 (&A_)[RO][KB], const wide_b8 (&X_)[CI][KB]) {
#pragma unroll
    for (int i = 0; i < KB; i++)
#pragma unroll
      for (int p = 0; p < RO; p++)
#pragma unroll
        for (int q = 0; q < CI; q++) acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[p][i], X_[q][i], acc[p][q], 0, 0, 0);
    if (bias) {
#pragma unroll
      for (int i = 0; i < KB; i++)
#pragma unroll
        for (int p = 0; p < RO; p++) accb[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[p][i], ones, accb[p], 0, 0, 0);
    }
  };
  const int nks = (u1 - u0) * 4;                     // k-steps of 16 rows: a multiple of 4 (of KB), possibly 0
  if (nks > 0) load_blk(av, xv, 0);
  for (int kk = 0; kk < nks; kk += 2 * KB) {
    if (kk + KB < nks) load_blk(aw, xw, kk + KB);
    mma_blk(av, xv);
    if (kk + 2 * KB < nks) load_blk(av, xv, kk + 2 * KB);
    if (kk + KB < nks) mma_blk(aw, xw);
  }
  if (bias && r == 0) {
#pragma unroll
    for (int p = 0; p < RO; p++)
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int row = o0 + 32 * p + wide_row(j, h);
        if (row < J.O) atomicAdd(&J.db[row], accb[p][j]);      // 4 waves x splitk adds per address
      }
  }
  // the four partial blocks meet in LDS, four tiles at a time ([wave][tile][register][lane]: lane-contiguous, conflict-free); wave w
  // then owns tile 4 ph + w: fixed summation order, and with splitk == 1 a plain store (no atomics, bit-reproducible gradients)
#pragma unroll
  for (int ph = 0; ph < NT / 4; ph++) {
    if (ph) __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
      constexpr int dummy = 0; (void)dummy;
      const int t = 4 * ph + tt;
#pragma unroll
      for (int j = 0; j < 16; j++) wl[((wave * 4 + tt) << 10) + j * 64 + lane] = acc[t / CI][t % CI][j];
    }
    __syncthreads();
    const int t = 4 * ph + wave, p = t / CI, q = t % CI;
    if ((i0 + 32 * q + r) < J.I) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WIDE_WG_WAVES; w++) v += wl[((w * 4 + wave) << 10) + j * 64 + lane];
        const int row = o0 + 32 * p + wide_row(j, h);
        if (row < J.O) {
          float *dst = &J.dW[(size_t)row * J.ldw + i0 + 32 * q + r];
          if (J.splitk == 1) *dst = v; else atomicAdd(dst, v);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(64 * WIDE_WG_WAVES, 2) wide_wgrad_kernel(WideWgradArgs a) {
  extern __shared__ __align__(16) float wide_wl[];      // 4 waves x 4 tiles x 4 KB
  const int blk = (int)blockIdx.x - 1, tid = threadIdx.x;
  if (blk < 0) {   // block 0 (dispatched first): loss scalars and the log_std gradient from the per-workgroup partials, in a fixed order
    // thread l sums the rows l, l + 256, .. of a trunk's [nblk][40] table (ten independent 16-byte loads per row: the loads of
    // all rows are in flight together; a first version walked the rows with 36 lanes and paid ~32 dependent L2 misses: 40 us,
    // the whole launch's length for a small net), the 64 lane sums meet in LDS
    __shared__ float red[2][36];
    float (*lsum)[41] = reinterpret_cast<float (*)[41]>(wide_wl);      // [256][41]
    for (int t = 0; t < 2; t++) {
      float acc[WIDE_PART];
#pragma unroll
      for (int e = 0; e < WIDE_PART; e++) acc[e] = 0.f;
      for (int i = tid; i < a.nblk; i += 64 * WIDE_WG_WAVES) {
        const float4 *row = reinterpret_cast<const float4 *>(a.part + ((size_t)t * a.nblk + i) * WIDE_PART);
#pragma unroll
        for (int e = 0; e < WIDE_PART / 4; e++) { const float4 v = row[e]; acc[4 * e] += v.x; acc[4 * e + 1] += v.y; acc[4 * e + 2] += v.z; acc[4 * e + 3] += v.w; }
      }
#pragma unroll
      for (int e = 0; e < 36; e++) lsum[tid][e] = acc[e];
      __syncthreads();
      if (tid < 36) {
        float s0 = 0.f;
        for (int l = 0; l < 64 * WIDE_WG_WAVES; l++) s0 += lsum[l][tid];
        red[t][tid] = s0;
      }
      __syncthreads();
    }
    __syncthreads();
    if (tid < a.A) a.g_log_std[tid] += red[0][tid] - a.ent_coef;
    if (tid == 0) {
      const float invB = 1.0f / (float)a.B;
      float ent = 0.f;
      for (int j = 0; j < a.A; j++) ent += 0.5f + 0.5f * 1.8378770664093453f + a.log_std[j];
      const float pg = red[0][32] * invB, vl = red[1][33] * invB;
      a.out8[1] = pg; a.out8[2] = vl; a.out8[3] = ent; a.out8[4] = red[0][34] * invB; a.out8[5] = red[0][35] * invB;
      a.out8[0] = pg + a.vf_coef * vl - a.ent_coef * ent;
      a.out8[6] = a.stats[0]; a.out8[7] = a.stats[1];
      if (a.loss_acc) { a.loss_acc[0] += a.out8[0]; a.loss_acc[1] += 1.f; }
    }
    return;
  }
  int jq = 0;
  for (int i = 1; i < a.njobs; i++) if (blk >= a.j[i].first) jq = i;
  const WideWgradJob &J = a.j[jq];
  // workgroups go round the eight XCDs in launch order.  A job's tiles are numbered (split, output tile, input tile), input tile
  // fastest, and XCD x takes the x-th eighth of that order: the workgroups that read one slice of the batch — and, inside it, one
  // block of dZ^T rows — share an L2, which then holds their operands once (dW2 of a trunk: 2.5 MB per XCD)
  const int loc = blk - J.first;
  int rem = (loc & 7) * J.per + (loc >> 3);
  if ((loc >> 3) >= J.per || rem >= J.splitk * J.otiles * J.itiles) return;
  const int it = rem % J.itiles; rem /= J.itiles;
  const int ot = rem % J.otiles, ks = rem / J.otiles;
  if (J.ro == 2) wide_wgrad_tile<2, 2>(J, a.B, ot, it, ks, tid >> 6, tid & 63, wide_wl);
  else wide_wgrad_tile<1, 4>(J, a.B, ot, it, ks, tid >> 6, tid & 63, wide_wl);
}

inline int wide_dp(int D) { return (D + 15) & ~15; }
inline long long wide_packed_elems(int D, int H1, int H2) { return (long long)H1 * wide_dp(D) + 2ll * H2 * H1 + 64ll * H2; }
inline int wide_lds_bytes(int D, int H1, int H2) {
  const int z2 = WIDE_R * (2 * H2 + 16), rz = z2 > WIDE_NW * 4096 ? z2 : WIDE_NW * 4096;
  return WIDE_R * (2 * wide_dp(D) + 16) + WIDE_R * (2 * H1 + 16) + z2 + rz + WIDE_R * 80 + (32 * 33 + 16 * 36) * 4;
}
inline bool wide_supported(int B, int D, int H1, int H2, int A) {
  return B >= 64 && B % 64 == 0 && D >= 1 && D <= 112 && H1 % 256 == 0 && H1 >= 256 && H1 <= 1024 && H2 % 128 == 0 && H2 >= 128 && H2 <= 512 && A >= 1 &&
         A <= 32 && wide_lds_bytes(D, H1, H2) <= 160 * 1024;
}

}  // namespace

extern "C" long long dm_ppo_wide_packed_elems(int D, int H1, int H2) { return wide_packed_elems(D, H1, H2); }
extern "C" int dm_ppo_wide_dp(int D) { return wide_dp(D); }
extern "C" int dm_ppo_wide_supported(int B, int D, int H1, int H2, int A) { return wide_supported(B, D, H1, H2, A) ? 1 : 0; }

extern "C" int dm_ppo_wide_grad(const DmPpoWideStep *s, void *stream) {
  if (!s || !wide_supported(s->B, s->D, s->H1, s->H2, s->A)) return -22;
  if (!s->obs || !s->act || !s->adv || !s->ret || !s->old_logp || !s->log_std || !s->g_log_std || !s->xbT || !s->part || !s->stats8 || !s->out8) return -22;
  for (int t = 0; t < 2; t++) {
    if (!s->wpk[t] || !s->h1T[t] || !s->dz1T[t] || !s->h2T[t] || !s->dz2T[t] || !s->dz3T[t]) return -22;
    for (int l = 0; l < 3; l++) if (!s->W[t][l] || !s->b[t][l] || !s->gW[t][l] || !s->gb[t][l]) return -22;
  }
  hipStream_t st = (hipStream_t)stream;
  static int lds_set_for = -1;
  const int lds = wide_lds_bytes(s->D, s->H1, s->H2);
  int dev = 0;
  hipGetDevice(&dev);
  if (lds_set_for != dev) {   // (per device: ADVICE r1)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(wide_fwdbwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -5;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(wide_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_WG_WAVES * 4 * 4096) != hipSuccess) return -5;
    lds_set_for = dev;
  }
  const int Dp = wide_dp(s->D);
  WidePackArgs p;
  memset(&p, 0, sizeof p);
  for (int t = 0; t < 2; t++) { for (int l = 0; l < 3; l++) p.W[t][l] = s->W[t][l]; p.pk[t] = (unsigned short *)s->wpk[t]; }
  p.D = s->D; p.Dp = Dp; p.H1 = s->H1; p.H2 = s->H2; p.A[0] = s->A; p.A[1] = 1;
  p.total = wide_packed_elems(s->D, s->H1, s->H2);
  p.pack_blocks = 1152;      // ~one 16-byte fragment per thread for the [1024,512] net (256 blocks: 11.4 us, the loop serialised four strided reads per thread)
  p.adv = s->adv; p.B = s->B; p.normalize = s->normalize_advantage; p.stats = s->stats8; p.out8 = s->out8;
  p.zero_ptr = s->zero_ptr; p.zero_floats = s->zero_ptr ? s->zero_floats : 0; p.adam_state2 = s->adam_state2;
  const int zero_blocks = (int)((p.zero_floats + 1023) / 1024);
  hipLaunchKernelGGL(wide_pack_kernel, dim3(2 * p.pack_blocks + 1 + zero_blocks), dim3(256), 0, st, p);
  WideArgs a;
  memset(&a, 0, sizeof a);
  a.B = s->B; a.D = s->D; a.Dp = Dp; a.H1 = s->H1; a.H2 = s->H2; a.A = s->A;
  a.obs = s->obs; a.act = s->act; a.adv = s->adv; a.ret = s->ret; a.old_logp = s->old_logp; a.log_std = s->log_std; a.stats = s->stats8;
  for (int t = 0; t < 2; t++) {
    a.pk[t] = (const unsigned short *)s->wpk[t];
    a.b1[t] = s->b[t][0]; a.b2[t] = s->b[t][1]; a.b3[t] = s->b[t][2];
    a.gb1[t] = s->gb[t][0]; a.gb2[t] = s->gb[t][1]; a.gb3[t] = s->gb[t][2];
    a.h1T[t] = (unsigned short *)s->h1T[t]; a.dz1T[t] = (unsigned short *)s->dz1T[t]; a.h2T[t] = (unsigned short *)s->h2T[t];
    a.dz2T[t] = (unsigned short *)s->dz2T[t]; a.dz3T[t] = (unsigned short *)s->dz3T[t];
  }
  a.xbT = (unsigned short *)s->xbT; a.part = s->part; a.clip = s->clip_range; a.vf_coef = s->vf_coef;
  const bool bias_wgrad = s->H1 >= WIDE_BIAS_WGRAD_H1;
  a.bias_in_chain = bias_wgrad ? 0 : 1;
  hipLaunchKernelGGL(wide_fwdbwd_kernel, dim3(2 * (s->B / WIDE_R)), dim3(WIDE_THREADS), lds, st, a);
  // weight gradients: per trunk dW2 (the big one), dW1, dW3; split-K chosen so that every job brings ~64-128 workgroups
  WideWgradArgs g;
  memset(&g, 0, sizeof g);
  int first = 0, nj = 0;
  auto add = [&](const void *AT, const void *XT, float *dW, float *db, int O, int I, int ldw, int ro, int want) {
    WideWgradJob &J = g.j[nj++];
    J.AT = (const unsigned short *)AT; J.XT = (const unsigned short *)XT; J.dW = dW; J.db = db; J.O = O; J.I = I; J.ldw = ldw; J.ro = ro;
    J.otiles = (O + 32 * ro - 1) / (32 * ro);
    J.itiles = (I + 32 * (4 / ro) - 1) / (32 * (4 / ro));
    int sk = 1;
    while (sk < WIDE_MAX_SPLITK && J.otiles * J.itiles * sk * 2 <= want && (s->B / (sk * 2)) % 64 == 0) sk *= 2;     // want: workgroups
    J.splitk = sk; J.first = first;                       // first % 8 == 0: a job's local block id & 7 is its XCD
    J.per = (J.otiles * J.itiles * sk + 7) / 8;
    first += 8 * J.per;
  };
  for (int t = 0; t < 2; t++) {
    add(s->dz2T[t], s->h1T[t], s->gW[t][1], bias_wgrad ? s->gb[t][1] : nullptr, s->H2, s->H1, s->H1, 2, 128);
    add(s->dz1T[t], s->xbT, s->gW[t][0], bias_wgrad ? s->gb[t][0] : nullptr, s->H1, s->D, s->D, 2, 32);
    add(s->dz3T[t], s->h2T[t], s->gW[t][2], bias_wgrad ? s->gb[t][2] : nullptr, t ? 1 : s->A, s->H2, s->H2, 1, 4);
  }
  g.njobs = nj; g.nblocks = first; g.B = s->B;
  g.part = s->part; g.nblk = s->B / WIDE_R; g.A = s->A; g.log_std = s->log_std; g.vf_coef = s->vf_coef; g.ent_coef = s->ent_coef; g.stats = s->stats8;
  g.g_log_std = s->g_log_std; g.out8 = s->out8; g.loss_acc = s->loss_acc;
  hipLaunchKernelGGL(wide_wgrad_kernel, dim3(first + 1), dim3(64 * WIDE_WG_WAVES), WIDE_WG_WAVES * 4 * 4096, st, g);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
