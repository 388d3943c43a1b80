// Issue cost of individual gfx950 vector instructions (8 independent chains per lane, inline asm so the
// compiler cannot fold them).  Prints ns per wave-instruction per SIMD at 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHAIN8(INS)                                                                                  \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)       \
               : "v"(a), "v"(b), "s"(sa));

#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_MED3(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MAXC(n) "v_max_f32_e64 %" #n ", %" #n ", %" #n " clamp\n"
#define I_CNDM(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define I_CVT(n) "v_cvt_u32_f32 %" #n ", %" #n "\n"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define I_LOG(n) "v_log_f32 %" #n ", %" #n "\n"
#define I_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define I_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %8\n"
#define I_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_READLANE(n) "v_readlane_b32 s20, %" #n ", 3\n"
#define I_FMAS(n) "v_fma_f32 %" #n ", %" #n ", %10, %9\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b, float sa) {
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { CHAIN8(I_FMA) }
    if (MODE == 1) { CHAIN8(I_MUL) }
    if (MODE == 2) { CHAIN8(I_MED3) }
    if (MODE == 3) { CHAIN8(I_MAXC) }
    if (MODE == 4) { CHAIN8(I_CNDM) }
    if (MODE == 5) { CHAIN8(I_CMP) }
    if (MODE == 6) { CHAIN8(I_CVT) }
    if (MODE == 7) { CHAIN8(I_EXP) }
    if (MODE == 8) { CHAIN8(I_LOG) }
    if (MODE == 9) { CHAIN8(I_SQRT) }
    if (MODE == 10) { CHAIN8(I_MULHI) }
    if (MODE == 11) { CHAIN8(I_MULLO) }
    if (MODE == 12) { CHAIN8(I_MAD24) }
    if (MODE == 13) { CHAIN8(I_LSHLADD) }
    if (MODE == 14) { CHAIN8(I_ADD) }
    if (MODE == 15) { asm volatile(I_READLANE(0) I_READLANE(1) I_READLANE(2) I_READLANE(3) I_READLANE(4) I_READLANE(5) I_READLANE(6) I_READLANE(7)
                                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(sa) : "s20"); }
    if (MODE == 16) { CHAIN8(I_FMAS) }
    if (MODE == 17) { CHAIN8(I_PERM) }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
void run(const char* name, float* out) {
  const int blocks = 2048, iters = 10000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100, 0.999f, 0.001f, 0.5f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 0.999f, 0.001f, 0.5f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)blocks * 4 * iters * 8 / 1024.0;
  printf("%-16s %8.3f ms  %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / per_simd);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 4096 * 4);
  run<0>("v_fma_f32", out); run<1>("v_mul_f32", out); run<2>("v_med3_f32", out); run<3>("v_max_f32 clamp", out);
  run<4>("v_cndmask_b32", out); run<5>("v_cmp_lt_f32", out); run<6>("v_cvt_u32_f32", out); run<7>("v_exp_f32", out);
  run<8>("v_log_f32", out); run<9>("v_sqrt_f32", out); run<10>("v_mul_hi_u32", out); run<11>("v_mul_lo_u32", out);
  run<12>("v_mad_u32_u24", out); run<13>("v_lshl_add_u32", out); run<14>("v_add_u32", out); run<15>("v_readlane_b32", out);
  run<16>("v_fma_f32 sgpr", out); run<17>("v_perm_b32", out);
  return 0;
}
