// Does the float32 MFMA (v_mfma_f32_16x16x4_f32) issue beside vector work on gfx950, or in its place?
// Waves run a loop of V independent v_fma_f32 and M dependent MFMAs per iteration; 4 waves per SIMD.
// Prints the time of vector work alone, matrix work alone, and both together in every wave.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int V, int M>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  float am = a, bm = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < M; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(a), "v"(b));
  }
  float s = acc[0] + acc[1] + acc[2] + acc[3];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V, int M>
float run(float* out) {
  const int blocks = 4096, iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<V, M><<<blocks, 256>>>(out, 50, 0.999f, 0.001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<V, M><<<blocks, 256>>>(out, iters, 0.999f, 0.001f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: blocks * 4 waves / 1024 SIMDs wave-iterations in sequence-equivalents
  const double wave_iters_per_simd = (double)blocks * 4 * iters / 1024.0;
  printf("V=%3d fma  M=%2d mfma per iteration: %8.3f ms  = %7.1f ns per wave-iteration per SIMD\n", V, M, ms,
         ms * 1e6 / wave_iters_per_simd);
  return ms;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 4096 * 4);
  run<64, 0>(out);
  run<0, 2>(out);
  run<64, 2>(out);
  run<0, 7>(out);
  run<64, 7>(out);
  run<200, 7>(out);
  run<200, 0>(out);
  return 0;
}
