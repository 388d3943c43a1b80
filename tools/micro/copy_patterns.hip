// Copy-kernel access patterns for interleaved float32 RGB: which per-lane access shape reaches the best
// HBM rate on gfx950?  Build: hipcc -O3 --offload-arch=gfx950 copy_patterns.hip -o copy_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct F3 { float x, y, z; };

// (a) one pixel (12 B) per lane per access, K pixels per lane per trip at a stride of 256 pixels
template <int K>
__global__ __launch_bounds__(256) void copy_x3(const F3* __restrict__ s, F3* __restrict__ d, uint32_t n) {
  for (uint32_t base = blockIdx.x * 256 * K; base < n; base += gridDim.x * 256 * K) {
    F3 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t i = base + k * 256 + threadIdx.x;
      if (i < n) v[k] = s[i];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t i = base + k * 256 + threadIdx.x;
      if (i < n) d[i] = F3{v[k].x + 1.0f, v[k].y + 1.0f, v[k].z + 1.0f};
    }
  }
}

// (b) 16 B per lane per access, lanes contiguous (pixel boundaries ignored), K vectors per lane per trip
template <int K>
__global__ __launch_bounds__(256) void copy_x4(const float4* __restrict__ s, float4* __restrict__ d, uint32_t n4) {
  for (uint32_t base = blockIdx.x * 256 * K; base < n4; base += gridDim.x * 256 * K) {
    float4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t i = base + k * 256 + threadIdx.x;
      if (i < n4) v[k] = s[i];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t i = base + k * 256 + threadIdx.x;
      if (i < n4) d[i] = make_float4(v[k].x + 1.0f, v[k].y + 1.0f, v[k].z + 1.0f, v[k].w + 1.0f);
    }
  }
}

// (c) 48 B (4 whole pixels) per lane as three 16-B accesses, lanes 48 B apart
__global__ __launch_bounds__(256) void copy_3x4(const float4* __restrict__ s, float4* __restrict__ d, uint32_t n12) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n12; i += gridDim.x * 256) {
    const float4 a = s[3 * i], b = s[3 * i + 1], c = s[3 * i + 2];
    d[3 * i] = make_float4(a.x + 1.0f, a.y + 1.0f, a.z + 1.0f, a.w + 1.0f);
    d[3 * i + 1] = make_float4(b.x + 1.0f, b.y + 1.0f, b.z + 1.0f, b.w + 1.0f);
    d[3 * i + 2] = make_float4(c.x + 1.0f, c.y + 1.0f, c.z + 1.0f, c.w + 1.0f);
  }
}

// (d) 16 B per lane, lanes contiguous, three vectors 256 lanes apart = 4 pixels per lane after an LDS transpose
__global__ __launch_bounds__(256) void copy_lds(const float4* __restrict__ s, float4* __restrict__ d, uint32_t n4) {
  __shared__ float4 T[3 * 256];
  for (uint32_t base = blockIdx.x * 768; base < n4; base += gridDim.x * 768) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t i = base + k * 256 + threadIdx.x;
      if (i < n4) T[k * 256 + threadIdx.x] = s[i];
    }
    __syncthreads();
    float4 a = T[3 * threadIdx.x], b = T[3 * threadIdx.x + 1], c = T[3 * threadIdx.x + 2];
    __syncthreads();
    a.x += 1.0f; b.y += 1.0f; c.z += 1.0f;
    T[3 * threadIdx.x] = a; T[3 * threadIdx.x + 1] = b; T[3 * threadIdx.x + 2] = c;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t i = base + k * 256 + threadIdx.x;
      if (i < n4) d[i] = T[k * 256 + threadIdx.x];
    }
    __syncthreads();
  }
}

int main() {
  const uint32_t n_px = 8u * 4000u * 6000u;
  const size_t bytes = (size_t)n_px * 12;
  void *s, *d;
  hipMalloc(&s, bytes);
  hipMalloc(&d, bytes);
  hipMemset(s, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6);
  };
  for (int blocks : {2048, 8192, 32768, 131072}) {
    printf("-- %d blocks\n", blocks);
    run("x3 K=1", [&] { copy_x3<1><<<blocks, 256>>>((const F3*)s, (F3*)d, n_px); });
    run("x3 K=4", [&] { copy_x3<4><<<blocks, 256>>>((const F3*)s, (F3*)d, n_px); });
    run("x4 K=1", [&] { copy_x4<1><<<blocks, 256>>>((const float4*)s, (float4*)d, n_px / 4 * 3); });
    run("x4 K=3", [&] { copy_x4<3><<<blocks, 256>>>((const float4*)s, (float4*)d, n_px / 4 * 3); });
    run("3x4 per lane", [&] { copy_3x4<<<blocks, 256>>>((const float4*)s, (float4*)d, n_px / 4); });
    run("x4 + LDS transpose", [&] { copy_lds<<<blocks, 256>>>((const float4*)s, (float4*)d, n_px / 4 * 3); });
  }
  return 0;
}
