// Microbenchmark: issue cadence of fp32 MFMAs (16x16x4 vs 32x32x2), dependent chains vs 2/4 interleaved accumulators,
// with 1 or 2 waves per SIMD (256- or 512-thread workgroups, one per CU). Prints cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC>
__global__ void k32(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <typename K>
void run(const char* name, K kern, int threads, int nacc, float* out, unsigned long long* cyc, double flops) {
  const int iters = 2000, blocks = 256;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double flop_per_mfma = flops;
  const double tf = (double)blocks * (threads / 64) * iters * 16.0 * nacc * flop_per_mfma / (ms * 1e-3) / 1e12;
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double per_wave = (double)h[0] / ((double)iters * 16 * nacc);
  const int waves_per_simd = threads / 256;
  // memtime counts at 100 MHz? report raw ticks per MFMA per wave and per SIMD
  printf("%-28s threads %4d: %.2f ticks per MFMA per wave, %.2f per MFMA per SIMD, %.3f ms, %.1f TFLOP/s wall\n", name, threads, per_wave,
         per_wave / waves_per_simd, ms, tf);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  for (int threads : {256, 512, 1024}) {
    run("16x16x4 1 acc (dependent)", k16<1>, threads, 1, out, cyc, 2048.0);
    run("16x16x4 2 acc", k16<2>, threads, 2, out, cyc, 2048.0);
    run("16x16x4 4 acc", k16<4>, threads, 4, out, cyc, 2048.0);
    run("32x32x2 1 acc (dependent)", k32<1>, threads, 1, out, cyc, 4096.0);
    run("32x32x2 2 acc", k32<2>, threads, 2, out, cyc, 4096.0);
    run("32x32x2 4 acc", k32<4>, threads, 4, out, cyc, 4096.0);
  }
  return 0;
}
