// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs transcendental ops on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  v2f va = {a, a}, vb = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent scalar FMAs
      x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
      x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    } else if (MODE == 1) {  // 4 packed FMAs = 8 FMAs
      p0 = __builtin_elementwise_fma(p0, va, vb); p1 = __builtin_elementwise_fma(p1, va, vb);
      p2 = __builtin_elementwise_fma(p2, va, vb); p3 = __builtin_elementwise_fma(p3, va, vb);
    } else if (MODE == 2) {  // 8 exp2
      x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
      x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7);
    } else if (MODE == 3) {  // 8 med3
      x0 = __builtin_amdgcn_fmed3f(x0, a, b); x1 = __builtin_amdgcn_fmed3f(x1, a, b); x2 = __builtin_amdgcn_fmed3f(x2, a, b); x3 = __builtin_amdgcn_fmed3f(x3, a, b);
      x4 = __builtin_amdgcn_fmed3f(x4, a, b); x5 = __builtin_amdgcn_fmed3f(x5, a, b); x6 = __builtin_amdgcn_fmed3f(x6, a, b); x7 = __builtin_amdgcn_fmed3f(x7, a, b);
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char* name, float* out, int blocks) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 0.999f, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // waves per SIMD = blocks*4 waves / (256 CUs * 4 SIMDs)
  double wave_ops = (double)blocks * 4 * iters * 8;  // 8 lane-ops per iteration per lane (FMA-equivalents)
  double per_simd = wave_ops / 1024.0;
  printf("%-14s blocks=%5d  %8.3f ms  -> %.2f ns per wave-op-equivalent per SIMD (x2.4GHz = %.2f cyc)\n", name, blocks, ms,
         ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}

int main() {
  float* out; hipMalloc(&out, 256 * 4096 * 4);
  for (int blocks : {256, 1024, 2048}) {
    run<0>("v_fma_f32", out, blocks);
    run<1>("v_pk_fma_f32", out, blocks);
    run<2>("v_exp_f32", out, blocks);
    run<3>("v_med3_f32", out, blocks);
  }
  return 0;
}
