// Cycles one wave spends in careful_pixel (all 64 lanes active, different pixels), per chain.
#include <cstdio>
#include <cmath>
#include <cstring>
#include "../../scanspaceimageprocessor_amd/csrc/ss_careful.h"
using namespace ss;
__global__ void k(const ChainParams P, long long* cycles, double* sink) {
  const ChainParams* Pk = (const ChainParams*)__builtin_amdgcn_kernarg_segment_ptr();
  const float t = (float)threadIdx.x / 64.0f;
  double out[3], acc = 0.0;
  const long long t0 = clock64();
  for (int i = 0; i < 16; ++i) {
    careful_pixel(Pk, 0, 0.05f + 0.9f * t, 0.9f - 0.8f * t + 0.001f * i, 0.2f + 0.5f * t * t, 0.3f + 0.4f * t, true, out);
    acc += out[0] + out[1] + out[2];
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[0] = (t1 - t0) / 16;
  sink[threadIdx.x] = acc;
}
static void run(const char* name, ChainParams P) {
  long long* c; double* s;
  (void)hipMalloc(&c, 8); (void)hipMalloc(&s, 64 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, P, c, s);
  long long h = 0;
  (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-28s %8lld clock64 ticks per call (100 MHz ticks: x ~21 for shader cycles at 2.1 GHz)\n", name, h);
}
int main() {
  ChainParams P;
  memset(&P, 0, sizeof P);
  P.final_clip = 1;
  P.ccm_kind = SS_CCM_CHEUNG; P.ccm_terms = 3;
  for (int i = 0; i < 3; ++i) P.ccm_full[i][i] = 1.0f;
  P.n_gain = 1; P.gain[0][0] = P.gain[0][1] = P.gain[0][2] = 1.1f;
  ChainParams A = P; A.usm = 1; A.usm_amount = 50.0f;
  run("ccm + gain + usm", A);
  ChainParams B = P; B.sh = 1; B.sh_shadow_amount = 0.5f; B.sh_highlight_amount = -0.5f; B.sh_falloff = 7.0f; B.sh_ln_falloff = log(7.0); B.sh_chroma = 1;
  run("ccm + gain + s/h", B);
  ChainParams C = B; C.usm = 1; C.usm_amount = 40.0f;
  run("ccm + gain + s/h + usm", C);
  run("ccm + gain only", P);
  return 0;
}
