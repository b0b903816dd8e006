// v_mfma_f32_32x32x2f32 issue rate on this box: wall-clock per MFMA for short and long kernels, 1 / 2 waves per SIMD,
// 1 / 2 independent accumulators.  hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k(float *out, int iters, float a, float b) {
  f16v acc[NACC];
  for (int q = 0; q < NACC; q++) for (int j = 0; j < 16; j++) acc[q][j] = threadIdx.x * 0.001f + q;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int q = 0; q < NACC; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
  }
  float s = 0;
  for (int q = 0; q < NACC; q++) for (int j = 0; j < 16; j++) s += acc[q][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(const char *name, int blocks, int threads, int iters, int reps) {
  float *out; hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, mfma_per_wave = (double)iters * NACC;
  const double waves_per_simd = (double)blocks * threads / 64 / 1024;
  printf("%-28s %7.1f us/kernel  %6.1f ns per MFMA per SIMD  (= %.0f cycles at 2.4 GHz)  %.1f TFLOP/s\n", name, us,
         us * 1e3 / (mfma_per_wave * (waves_per_simd < 1 ? 1 : waves_per_simd)), us * 1e3 / (mfma_per_wave * (waves_per_simd < 1 ? 1 : waves_per_simd)) * 2.4,
         (double)blocks * threads / 64 * mfma_per_wave * 4096 / (us * 1e-6) / 1e12);
  hipFree(out);
}
int main() {
  run<1>("1 wave/SIMD 1 acc 360 mfma", 256, 256, 360, 200);
  run<1>("1 wave/SIMD 1 acc 3600 mfma", 256, 256, 3600, 50);
  run<2>("1 wave/SIMD 2 acc 2x1800", 256, 256, 1800, 50);
  run<1>("2 waves/SIMD 1 acc 1800", 256, 512, 1800, 50);
  run<1>("2 waves/SIMD 1 acc 180", 256, 512, 180, 200);
  run<1>("4 waves/SIMD 1 acc 900", 1024, 256, 900, 50);
  return 0;
}
