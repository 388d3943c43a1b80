// The reference's two Lab helpers as operators of their own (editingTools.py:367-408):
//   _rgb_to_lab: colour.sRGB_to_XYZ (decodes the sRGB transfer function first) -> colour.XYZ_to_Lab (D65),
//                returned as three planes L, a, b;
//   _lab_to_rgb: colour.Lab_to_XYZ -> colour.XYZ_to_sRGB (encodes) -> clip to [0, 1].
// Inside the fused chain the same arithmetic lives in ss_device.h, written for data already clipped to [0, 1].
// These two take whatever the caller hands them, so every piecewise function selects its branch the way
// numpy.where does in colour-science: negative and out-of-range values take the linear segments.
//
// Pointwise and HBM-bound: 12 bytes in and 12 out per pixel.  A lane owns four pixels per trip, 256 apart; a wave's 64
// interleaved pixels are 768 contiguous bytes (whole cache lines, streamed), the planes are plain coalesced words.

#include "ss_device.h"
#include "ss_internal.h"

namespace ss {
namespace {

constexpr float kLabJunction = 216.0f / 24389.0f;  // (24/116)^3

__device__ __forceinline__ float lab_f_any(float t) {
  const float hi = hw_exp2(hw_log2(t) * (1.0f / 3.0f));  // NaN for t < 0, where the line is selected
  const float lo = fmaf(t, 841.0f / 108.0f, 16.0f / 116.0f);
  return t > kLabJunction ? hi : lo;
}

constexpr int kPxPerLane = 4;  // four independent pixels per lane and trip keep enough loads in flight

__device__ __forceinline__ void rgb_to_lab_px(const F3 p, float& L, float& a, float& b) {
  const float r = srgb_decode(p.x), g = srgb_decode(p.y), bl = srgb_decode(p.z);
  const float fx = lab_f_any(fmaf(kLab.fwd[0], r, fmaf(kLab.fwd[1], g, bl * kLab.fwd[2])));
  const float fy = lab_f_any(fmaf(kLab.fwd[3], r, fmaf(kLab.fwd[4], g, bl * kLab.fwd[5])));
  const float fz = lab_f_any(fmaf(kLab.fwd[6], r, fmaf(kLab.fwd[7], g, bl * kLab.fwd[8])));
  L = fmaf(fy, 116.0f, -16.0f);
  a = (fx - fy) * 500.0f;
  b = (fy - fz) * 200.0f;
}

__device__ __forceinline__ F3 lab_to_rgb_px(const float Lv, const float av, const float bv) {
  const float fy = __fdiv_rn(Lv + 16.0f, 116.0f);
  const float fx = fmaf(av, 1.0f / 500.0f, fy);
  const float fz = fmaf(bv, -1.0f / 200.0f, fy);
  const float x = lab_finv(fx), y = lab_finv(fy), z = lab_finv(fz);
  F3 o;
  o.x = clamp01(srgb_encode(fmaf(kLab.inv[0], x, fmaf(kLab.inv[1], y, z * kLab.inv[2]))));
  o.y = clamp01(srgb_encode(fmaf(kLab.inv[3], x, fmaf(kLab.inv[4], y, z * kLab.inv[5]))));
  o.z = clamp01(srgb_encode(fmaf(kLab.inv[6], x, fmaf(kLab.inv[7], y, z * kLab.inv[8]))));
  return o;
}

// A workgroup takes 1024 consecutive pixels per trip; lane t owns pixels t, t + 256, t + 512, t + 768 of them.
__global__ __launch_bounds__(256) void rgb_to_lab_kernel(const float* __restrict__ rgb, uint64_t n,
                                                         float* __restrict__ L, float* __restrict__ a,
                                                         float* __restrict__ b) {
  const F3* __restrict__ s = reinterpret_cast<const F3*>(rgb);
  constexpr uint64_t kTrip = 256 * kPxPerLane;
  for (uint64_t base = (uint64_t)blockIdx.x * kTrip; base < n; base += (uint64_t)gridDim.x * kTrip) {
    const uint64_t i0 = base + threadIdx.x;
    if (base + kTrip <= n) {
      F3 p[kPxPerLane];
#pragma unroll
      for (int k = 0; k < kPxPerLane; ++k) p[k] = load_stream(s + i0 + 256 * k);
#pragma unroll
      for (int k = 0; k < kPxPerLane; ++k) {
        float Lv, av, bv;
        rgb_to_lab_px(p[k], Lv, av, bv);
        __builtin_nontemporal_store(Lv, L + i0 + 256 * k);
        __builtin_nontemporal_store(av, a + i0 + 256 * k);
        __builtin_nontemporal_store(bv, b + i0 + 256 * k);
      }
    } else {
      for (uint64_t i = i0; i < n; i += 256) {
        float Lv, av, bv;
        rgb_to_lab_px(load_stream(s + i), Lv, av, bv);
        L[i] = Lv;
        a[i] = av;
        b[i] = bv;
      }
    }
  }
}

__global__ __launch_bounds__(256) void lab_to_rgb_kernel(const float* __restrict__ L, const float* __restrict__ a,
                                                         const float* __restrict__ b, uint64_t n,
                                                         float* __restrict__ rgb) {
  F3* __restrict__ d = reinterpret_cast<F3*>(rgb);
  constexpr uint64_t kTrip = 256 * kPxPerLane;
  for (uint64_t base = (uint64_t)blockIdx.x * kTrip; base < n; base += (uint64_t)gridDim.x * kTrip) {
    const uint64_t i0 = base + threadIdx.x;
    if (base + kTrip <= n) {
      float Lv[kPxPerLane], av[kPxPerLane], bv[kPxPerLane];
#pragma unroll
      for (int k = 0; k < kPxPerLane; ++k) {
        Lv[k] = __builtin_nontemporal_load(L + i0 + 256 * k);
        av[k] = __builtin_nontemporal_load(a + i0 + 256 * k);
        bv[k] = __builtin_nontemporal_load(b + i0 + 256 * k);
      }
#pragma unroll
      for (int k = 0; k < kPxPerLane; ++k) store_stream(d + i0 + 256 * k, lab_to_rgb_px(Lv[k], av[k], bv[k]));
    } else {
      for (uint64_t i = i0; i < n; i += 256) d[i] = lab_to_rgb_px(L[i], a[i], b[i]);
    }
  }
}

// One trip per workgroup: the workgroups in flight then cover one window of each stream that moves through memory in
// dispatch order.  (A copy of this access shape with no arithmetic at all, tools/micro/plane_streams.hip, 192 Mpx:
// 5.1 TB/s with 32,768 workgroups of six trips each, 5.6 TB/s with one trip each.)  The loop in the kernels only
// runs again beyond 2^31 workgroups' worth of pixels.
unsigned grid_for(uint64_t n) {
  uint64_t blocks = (n + 256 * kPxPerLane - 1) / (256 * kPxPerLane);
  if (blocks < 1) blocks = 1;
  if (blocks > 0x7FFFFFFFull) blocks = 0x7FFFFFFFull;
  return (unsigned)blocks;
}

}  // namespace

hipError_t launch_rgb_to_lab(const float* rgb, uint64_t n_px, float* L, float* a, float* b, hipStream_t st) {
  hipLaunchKernelGGL(rgb_to_lab_kernel, dim3(grid_for(n_px)), dim3(256), 0, st, rgb, n_px, L, a, b);
  return hipGetLastError();
}

hipError_t launch_lab_to_rgb(const float* L, const float* a, const float* b, uint64_t n_px, float* rgb,
                             hipStream_t st) {
  hipLaunchKernelGGL(lab_to_rgb_kernel, dim3(grid_for(n_px)), dim3(256), 0, st, L, a, b, n_px, rgb);
  return hipGetLastError();
}

}  // namespace ss
