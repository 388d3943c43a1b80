// Issue cost of gfx950 vector instructions by occupancy: 64 instructions (8 independent chains) per loop trip, so the
// loop's own scalar instructions do not count; W waves per SIMD set by a dynamic LDS request.
// Prints ns and cycles (at the clock GRBM would show ~2.1-2.4 GHz: use ns) per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PK(n) "v_pk_fma_f32 %" #n ", %" #n ", %10, %11\n"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define I_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_MED(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MIN(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_ADD(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define I_CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_IADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_CVT(n) "v_cvt_u32_f32 %" #n ", %" #n "\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define I_MAXC(n) "v_max_f32_e64 %" #n ", %" #n ", %" #n " clamp\n"
#define I_FMACL(n) "v_fma_f32 %" #n ", %" #n ", %8, %9 clamp\n"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %10\n"
#define I_LOG(n) "v_log_f32 %" #n ", %" #n "\n"
#define I_SEL32(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define I_SEL64(n) "v_cmp_lt_f32_e64 s[20:21], %" #n ", %8\n v_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\n"
#define I_SELFAR(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n v_mul_f32 %" #n ", %" #n ", %8\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  extern __shared__ float lds[];
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  const v2f a2 = {a, a}, b2 = {b, b};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) asm volatile(REP8(I_FMA) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      if (MODE == 1) asm volatile(REP8(I_PK) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(a), "v"(b), "v"(a2), "v"(b2));
      if (MODE == 2) asm volatile(REP8(I_EXP) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      if (MODE == 3) asm volatile(REP8(I_CND) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
      if (MODE == 4) asm volatile(REP8(I_MED) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
#define ONE(M, I) if (MODE == M) asm volatile(REP8(I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc", "s20", "s21");
      ONE(6, I_MIN) ONE(7, I_MUL) ONE(8, I_ADD) ONE(9, I_CMP) ONE(10, I_CND64) ONE(11, I_MOV) ONE(12, I_IADD) ONE(13, I_CVT)
      ONE(14, I_AND) ONE(15, I_FMAC) ONE(16, I_MAXC) ONE(17, I_FMACL) ONE(19, I_LOG) ONE(20, I_SEL32) ONE(21, I_SEL64) ONE(22, I_SELFAR)
      if (MODE == 18) asm volatile(REP8(I_PKMUL) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(a), "v"(b), "v"(a2), "v"(b2));
      if (MODE == 5) {  // the kernel's mix: 2 plain, 1 packed, 1 transcendental in 8
        asm volatile(I_FMA(0) I_PK(1) I_FMA(2) I_EXP(3) I_FMA(4) I_PK(5) I_MED(6) I_PK(7)
                     : "+v"(x0), "+v"(p1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(p5), "+v"(x6), "+v"(p7) : "v"(a), "v"(b), "v"(a2), "v"(b2));
      }
    }
  }
  if (iters < 0) lds[threadIdx.x] = x0;
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
void run(const char* name, float* out, int waves_per_simd) {
  // blocks of 4 waves (one per SIMD); LDS request limits the blocks per CU to waves_per_simd
  const int lds = waves_per_simd >= 8 ? 0 : (160 * 1024 / waves_per_simd) - 1024;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int blocks = 256 * waves_per_simd * 4, iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256, lds>>>(out, 50, 0.999f, 0.001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256, lds>>>(out, iters, 0.999f, 0.001f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)blocks * 4 * iters * 64 / 1024.0;
  printf("%-14s %d waves/SIMD  %8.3f ms  %.2f ns per wave-instruction per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / per_simd);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 8192 * 4);
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", out, w);
    run<1>("v_pk_fma_f32", out, w);
    run<2>("v_exp_f32", out, w);
    run<4>("v_med3_f32", out, w);
    run<5>("mix", out, w);
  }
  const int w = 4;
  run<3>("v_cndmask vcc", out, w); run<10>("v_cndmask sgpr", out, w); run<6>("v_min_f32", out, w); run<7>("v_mul_f32", out, w);
  run<8>("v_add_f32", out, w); run<9>("v_cmp_lt_f32", out, w); run<11>("v_mov_b32", out, w); run<12>("v_add_u32", out, w);
  run<13>("v_cvt_u32_f32", out, w); run<14>("v_and_b32", out, w); run<15>("v_fmac_f32", out, w); run<16>("v_max clamp", out, w);
  run<17>("v_fma clamp", out, w); run<18>("v_pk_mul_f32", out, w); run<19>("v_log_f32", out, w);
  run<20>("cmp+cnd vcc (x2 instr)", out, w); run<21>("cmp+cnd sgpr (x2)", out, w); run<22>("cmp,mul,cnd vcc (x3)", out, w);
  return 0;
}
