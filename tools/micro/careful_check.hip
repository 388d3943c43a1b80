// One pixel through careful_pixel on the GPU, printed: a debugging aid for the float64 path.
#include <cstdio>
#include <cstring>
#include "../../scanspaceimageprocessor_amd/csrc/ss_careful.h"
using namespace ss;
__global__ void k(const ChainParams P, float r, float g, float b, float Lb, double* out) {
  careful_pixel((const ChainParams*)__builtin_amdgcn_kernarg_segment_ptr(), 0, r, g, b, Lb, false, out);
}
int main() {
  ChainParams P;
  memset(&P, 0, sizeof P);
  P.usm = 1; P.usm_amount = 80.0f; P.usm_threshold = 2.0f;
  double* d; hipMalloc(&d, 24);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, P, -0.10389519f, 0.11298092f, 0.34923315f, 0.44158608f, d);
  double h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("%.9f %.9f %.9f (want 0 0 0.24301538)\n", h[0], h[1], h[2]);
  return 0;
}
