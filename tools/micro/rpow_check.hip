// careful::rpow against pow() over 1e-300 ... 1e300: prints the largest relative difference.
#include <cstdio>
#include "../../scanspaceimageprocessor_amd/csrc/ss_careful.h"
using namespace ss;
__global__ void k(double* out) {
  double worst[3] = {0, 0, 0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (1 << 22); i += gridDim.x * blockDim.x) {
    const double x = exp2(-996.0 + 1992.0 * (double)i / (double)(1 << 22));
    const bool mid = x > 1e-100 && x < 1e100;  // (x^2.4 itself leaves the range outside)
    const double a = mid ? careful::rpow<12, 5>(x) / pow(x, 2.4) - 1.0 : 0.0;
    const double b = careful::rpow<5, 12>(x) / pow(x, 1.0 / 2.4) - 1.0;
    const double c = careful::rpow<1, 3>(-x) / -pow(x, 1.0 / 3.0) - 1.0;
    worst[0] = fmax(worst[0], fabs(a)), worst[1] = fmax(worst[1], fabs(b)), worst[2] = fmax(worst[2], fabs(c));
  }
  for (int j = 0; j < 3; ++j) atomicMax((unsigned long long*)&out[j], (unsigned long long)__double_as_longlong(worst[j]));
}
int main() {
  double* d;
  (void)hipMalloc(&d, 24);
  (void)hipMemset(d, 0, 24);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d);
  double h[3];
  (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("largest relative difference to pow(): x^2.4 %.3g  x^(1/2.4) %.3g  x^(1/3) %.3g\n", h[0], h[1], h[2]);
  return 0;
}
