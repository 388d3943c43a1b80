// What does the memory system give a kernel that reads one interleaved RGB stream (12 B / pixel) and writes three
// planes (4 B / pixel each), with no arithmetic at all?  That is the access shape of rgb_to_lab_kernel (and, reversed,
// of lab_to_rgb_kernel); a 1-stream-in / 1-stream-out copy of the same bytes is timed beside it.
// Build: hipcc -O3 --offload-arch=gfx950 plane_streams.hip -o plane_streams ; run: ./plane_streams [Mpx]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

struct __attribute__((aligned(4))) F3 { float x, y, z; };
__device__ __forceinline__ F3 ld(const F3* p, bool nt) {
  const float* q = (const float*)p;
  if (nt) return F3{__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2)};
  return *p;
}
__device__ __forceinline__ void st(float* p, float v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, p); else *p = v;
}

// K pixels per lane and trip, 256 apart (the operators' shape); NT: streaming loads and stores
template <int K, bool NT, bool TO_PLANES>
__global__ __launch_bounds__(256) void split(const float* __restrict__ rgb, uint64_t n, float* __restrict__ L,
                                             float* __restrict__ a, float* __restrict__ b) {
  const F3* s = (const F3*)rgb;
  constexpr uint64_t T = 256 * K;
  for (uint64_t base = (uint64_t)blockIdx.x * T; base + T <= n; base += (uint64_t)gridDim.x * T) {
    const uint64_t i0 = base + threadIdx.x;
    if (TO_PLANES) {
      F3 p[K];
#pragma unroll
      for (int k = 0; k < K; ++k) p[k] = ld(s + i0 + 256 * k, NT);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        st(L + i0 + 256 * k, p[k].x, NT);
        st(a + i0 + 256 * k, p[k].y, NT);
        st(b + i0 + 256 * k, p[k].z, NT);
      }
    } else {
      float x[K], y[K], z[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        x[k] = NT ? __builtin_nontemporal_load(L + i0 + 256 * k) : L[i0 + 256 * k];
        y[k] = NT ? __builtin_nontemporal_load(a + i0 + 256 * k) : a[i0 + 256 * k];
        z[k] = NT ? __builtin_nontemporal_load(b + i0 + 256 * k) : b[i0 + 256 * k];
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float* q = (float*)((F3*)rgb + i0 + 256 * k);
        st(q, x[k], NT); st(q + 1, y[k], NT); st(q + 2, z[k], NT);
      }
    }
  }
}

// the same bytes as one stream in, one stream out (12 B per lane and access)
template <int K, bool NT>
__global__ __launch_bounds__(256) void copy12(const float* __restrict__ src, uint64_t n, float* __restrict__ dst) {
  const F3* s = (const F3*)src;
  F3* d = (F3*)dst;
  constexpr uint64_t T = 256 * K;
  for (uint64_t base = (uint64_t)blockIdx.x * T; base + T <= n; base += (uint64_t)gridDim.x * T) {
    const uint64_t i0 = base + threadIdx.x;
    F3 p[K];
#pragma unroll
    for (int k = 0; k < K; ++k) p[k] = ld(s + i0 + 256 * k, NT);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float* q = (float*)(d + i0 + 256 * k);
      st(q, p[k].x, NT); st(q + 1, p[k].y, NT); st(q + 2, p[k].z, NT);
    }
  }
}

template <class F>
static void timeit(const char* name, uint64_t n, F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipDeviceSynchronize();
  float best = 1e9f, sum = 0;
  const int reps = 20;
  for (int i = 0; i < reps; ++i) {
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best; sum += ms;
  }
  printf("%-44s mean %.3f ms  min %.3f ms  %.0f GB/s (mean)  = %.3f of 8 TB/s\n", name, sum / reps, best, 24.0 * n / (sum / reps) / 1e6,
         24.0 * n / (sum / reps) / 1e6 / 8000.0);
}

int main(int argc, char** argv) {
  const uint64_t n = (uint64_t)(argc > 1 ? atof(argv[1]) : 192.0) * 1000000ull / 4096 * 4096;
  float *rgb, *planes, *rgb2;
  (void)hipMalloc(&rgb, n * 12); (void)hipMalloc(&planes, n * 12); (void)hipMalloc(&rgb2, n * 12);
  (void)hipMemset(rgb, 0, n * 12); (void)hipMemset(planes, 0, n * 12);
  float *L = planes, *a = planes + n, *b = planes + 2 * n;
  for (int blocks : {32768, 65536, (int)(n / 1024), (int)(n / 2048)}) {
    printf("-- %d blocks, %.0f Mpx\n", blocks, n / 1e6);
#define RUN(K, NT) \
    timeit("rgb -> 3 planes  K=" #K " nt=" #NT, n, [&] { hipLaunchKernelGGL((split<K, NT, true>), dim3(blocks), dim3(256), 0, 0, rgb, n, L, a, b); }); \
    timeit("3 planes -> rgb  K=" #K " nt=" #NT, n, [&] { hipLaunchKernelGGL((split<K, NT, false>), dim3(blocks), dim3(256), 0, 0, rgb2, n, L, a, b); }); \
    timeit("rgb -> rgb       K=" #K " nt=" #NT, n, [&] { hipLaunchKernelGGL((copy12<K, NT>), dim3(blocks), dim3(256), 0, 0, rgb, n, rgb2); });
    RUN(4, true) RUN(8, true) RUN(4, false)
  }
  return 0;
}
