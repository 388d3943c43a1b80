// Reduction-type kernels of the correction chain for gfx950:
//   - exact per-channel percentiles of the strided sample of invert_negative_auto
//     (three 10-bit radix-select passes over LDS-privatised histograms),
//   - masked-luminance statistics of calculate_average_exposure and the multipliers,
//   - the min/max heuristic of _is_lab_image.

#include "ss_internal.h"

namespace ss {

// ------------------------------------------------------------------------------------------------
// percentiles
// ------------------------------------------------------------------------------------------------
// Keys are the bit patterns of the clipped samples: for floats in [0, 1] the pattern is a
// monotone unsigned integer <= 0x3F800000 < 2^30, so three 10-bit digits select exactly.
// Per (image, channel) there are four targets: the two order statistics bracketing each of the
// two percentiles.  state[img][c][t] = {key prefix found so far, rank inside that prefix}.

constexpr int PCT_BINS = 1024;
constexpr int PCT_TARGETS = 4;
#ifndef SS_PCT_SAMPLE_BLOCKS
#define SS_PCT_SAMPLE_BLOCKS 4096u
#endif
#ifndef SS_PCT_REFINE_BLOCKS
#define SS_PCT_REFINE_BLOCKS 2048u
#endif
constexpr int PCT_HISTS_PER_IMAGE = 3 + 3 * PCT_TARGETS + 3 * PCT_TARGETS;  // level 1, 2, 3

size_t percentile_samples_bytes(int batch, int64_t n) { return (size_t)batch * 3 * (size_t)n * sizeof(uint32_t); }
size_t percentile_hist_bytes(int batch) { return (size_t)batch * PCT_HISTS_PER_IMAGE * PCT_BINS * sizeof(uint32_t); }
size_t percentile_state_bytes(int batch) { return (size_t)batch * 3 * PCT_TARGETS * 2 * sizeof(uint32_t); }

__device__ __forceinline__ uint32_t* hist_level1(uint32_t* hist, int img, int c) {
  return hist + ((size_t)img * PCT_HISTS_PER_IMAGE + c) * PCT_BINS;
}
__device__ __forceinline__ uint32_t* hist_level23(uint32_t* hist, int img, int level, int c, int t) {
  return hist + ((size_t)img * PCT_HISTS_PER_IMAGE + 3 + (level - 2) * 3 * PCT_TARGETS + c * PCT_TARGETS + t) * PCT_BINS;
}

// pass 1: gather the strided sample, clip, store keys channel-planar, histogram the top digit
__global__ __launch_bounds__(256) void pct_sample_kernel(const float* __restrict__ src, int height, int width, int step,
                                                         int ws, uint32_t n, uint32_t* __restrict__ samples,
                                                         uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[3 * PCT_BINS];
  const int img = blockIdx.y;
  for (int i = threadIdx.x; i < 3 * PCT_BINS; i += 256) h[i] = 0;
  __syncthreads();
  const F3* __restrict__ s = reinterpret_cast<const F3*>(src + (size_t)img * height * width * 3);
  uint32_t* smp = samples + (size_t)img * 3 * n;
  // four samples per trip: the gathers (one 12-byte load per sample, 9 or more pixels apart) are all issued before
  // the first key is binned
#ifndef SS_PCT_GATHER
#define SS_PCT_GATHER 4
#endif
  constexpr int kGather = SS_PCT_GATHER;
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t i0 = blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += kGather * stride) {
    F3 v[kGather];
#pragma unroll
    for (int j = 0; j < kGather; ++j) {
      const uint32_t i = i0 + j * stride;
      const uint32_t ii = i < n ? i : i0;  // (past the end: reload the first one, not used)
      const uint32_t sy = ii / ws, sx = ii - sy * ws;
      v[j] = s[(sy * step) * (uint32_t)width + sx * step];
    }
#pragma unroll
    for (int j = 0; j < kGather; ++j) {
      const uint32_t i = i0 + j * stride;
      if (i >= n) break;
      // clip to [0,1]; "+ 0.0f" turns a -0.0 that survives the clip into +0.0 so keys stay monotone
      const uint32_t k0 = __float_as_uint(clamp01(v[j].x) + 0.0f);
      const uint32_t k1 = __float_as_uint(clamp01(v[j].y) + 0.0f);
      const uint32_t k2 = __float_as_uint(clamp01(v[j].z) + 0.0f);
      smp[i] = k0;
      smp[n + i] = k1;
      smp[2 * (size_t)n + i] = k2;
      atomicAdd(&h[k0 >> 20], 1u);
      atomicAdd(&h[PCT_BINS + (k1 >> 20)], 1u);
      atomicAdd(&h[2 * PCT_BINS + (k2 >> 20)], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * PCT_BINS; i += 256) {
    const uint32_t c = h[i];
    if (c) atomicAdd(hist_level1(hist, img, 0) + i, c);
  }
}

// passes 2 and 3: histogram the next digit of the keys that match a target's prefix
template <int LEVEL>
__global__ __launch_bounds__(256) void pct_refine_kernel(const uint32_t* __restrict__ samples, uint32_t n,
                                                         const uint32_t* __restrict__ state,
                                                         uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[3 * PCT_TARGETS * PCT_BINS];
  __shared__ uint32_t prefix[3 * PCT_TARGETS];
  const int img = blockIdx.y;
  for (int i = threadIdx.x; i < 3 * PCT_TARGETS * PCT_BINS; i += 256) h[i] = 0;
  if (threadIdx.x < 3 * PCT_TARGETS) prefix[threadIdx.x] = state[((size_t)img * 3 * PCT_TARGETS + threadIdx.x) * 2];
  __syncthreads();
  constexpr int SHIFT_MATCH = LEVEL == 2 ? 20 : 10;
  constexpr int SHIFT_DIGIT = LEVEL == 2 ? 10 : 0;
  const uint32_t* smp = samples + (size_t)img * 3 * n;
  uint32_t want[3 * PCT_TARGETS];
#pragma unroll
  for (int j = 0; j < 3 * PCT_TARGETS; ++j) want[j] = prefix[j] >> SHIFT_MATCH;
  auto bin = [&](int c, uint32_t k) {
    const uint32_t digit = (k >> SHIFT_DIGIT) & (PCT_BINS - 1);
#pragma unroll
    for (int t = 0; t < PCT_TARGETS; ++t) {
      if ((k >> SHIFT_MATCH) == want[c * PCT_TARGETS + t]) atomicAdd(&h[(c * PCT_TARGETS + t) * PCT_BINS + digit], 1u);
    }
  };
  // four keys per lane and channel in one 16-byte load, the three channels' loads issued together: with the
  // 48 KB of histograms limiting a CU to three workgroups, one key per load left the memory pipe nearly empty
  struct __attribute__((aligned(4))) K4 {
    uint32_t x, y, z, w;
  };
  const uint32_t n4 = n / 4;
  for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < n4; q += gridDim.x * 256) {
    K4 k[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) k[c] = *reinterpret_cast<const K4*>(smp + (size_t)c * n + 4 * (size_t)q);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bin(c, k[c].x);
      bin(c, k[c].y);
      bin(c, k[c].z);
      bin(c, k[c].w);
    }
  }
  if (blockIdx.x == 0) {
    const uint32_t i = n4 * 4 + threadIdx.x;
    if (i < n) {
#pragma unroll
      for (int c = 0; c < 3; ++c) bin(c, smp[(size_t)c * n + i]);
    }
  }
  __syncthreads();
  uint32_t* out = hist_level23(hist, img, LEVEL, 0, 0);
  for (int i = threadIdx.x; i < 3 * PCT_TARGETS * PCT_BINS; i += 256) {
    const uint32_t c = h[i];
    if (c) atomicAdd(out + i, c);
  }
}

// One wave locates the bin that holds `rank` in a 1024-bin histogram: returns the bin and the
// rank inside it.  All 64 lanes must call it.
__device__ __forceinline__ void wave_select(const uint32_t* __restrict__ hist, uint32_t rank, uint32_t& bin,
                                            uint32_t& rank_in_bin) {
  const int lane = threadIdx.x & 63;
  uint32_t v[16];
  uint32_t local = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = hist[lane * 16 + i];
    local += v[i];
  }
  uint32_t incl = local;  // inclusive scan over lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  const uint32_t excl = incl - local;
  const bool mine = rank >= excl && rank < incl;
  uint32_t b = 0, rib = 0;
  if (mine) {
    uint32_t cum = excl;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (rank >= cum && rank < cum + v[i]) {
        b = lane * 16 + i;
        rib = rank - cum;
      }
      cum += v[i];
    }
  }
  const unsigned long long m = __ballot(mine);
  const int src_lane = m ? __ffsll((long long)m) - 1 : 0;
  bin = __shfl(b, src_lane, 64);
  rank_in_bin = __shfl(rib, src_lane, 64);
}

struct PctRanks {
  uint32_t k[PCT_TARGETS];  // order-statistic indices: low lo/hi, high lo/hi
  double g_low, g_high;
  int legacy;
  int step;
  uint32_t n;
};

// select after level 1 / 2: one workgroup per image, one wave per (channel, target) in turn
template <int LEVEL>
__global__ __launch_bounds__(256) void pct_select_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ state,
                                                         const PctRanks R) {
  const int img = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  for (int ct = wave; ct < 3 * PCT_TARGETS; ct += 4) {
    const int c = ct / PCT_TARGETS, t = ct % PCT_TARGETS;
    uint32_t* st = state + ((size_t)img * 3 * PCT_TARGETS + ct) * 2;
    const uint32_t rank = LEVEL == 1 ? R.k[t] : st[1];
    const uint32_t prefix = LEVEL == 1 ? 0u : st[0];
    const uint32_t* h = LEVEL == 1 ? hist_level1(hist, img, c) : hist_level23(hist, img, LEVEL, c, t);
    uint32_t bin, rib;
    wave_select(h, rank, bin, rib);
    if ((threadIdx.x & 63) == 0) {
      st[0] = prefix | (bin << (LEVEL == 1 ? 20 : 10));
      st[1] = rib;
    }
  }
}

// numpy's _lerp: a + (b-a)*t, switching to b - (b-a)*(1-t) for t >= 0.5
__device__ __forceinline__ double lerp_legacy(float a, float b, double t) {
  const double d = (double)(b - a);  // the subtraction happens in float32 first
  // (numpy multiplies, then adds: two roundings, so no fused multiply-add here)
  return t >= 0.5 ? sub_rn((double)b, mul_rn(d, sub_rn(1.0, t))) : add_rn((double)a, mul_rn(d, t));
}
__device__ __forceinline__ float lerp_f32(float a, float b, float t) {
  const float d = sub_rn(b, a);
  return t >= 0.5f ? sub_rn(b, mul_rn(d, sub_rn(1.0f, t))) : add_rn(a, mul_rn(d, t));
}

// select after level 3, interpolation, inversion parameters
__global__ __launch_bounds__(256) void pct_final_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ state,
                                                        const PctRanks R, float* __restrict__ params,
                                                        ss_percentile_detail* __restrict__ detail) {
  __shared__ float val[3 * PCT_TARGETS];
  const int img = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  for (int ct = wave; ct < 3 * PCT_TARGETS; ct += 4) {
    const int c = ct / PCT_TARGETS, t = ct % PCT_TARGETS;
    const uint32_t* st = state + ((size_t)img * 3 * PCT_TARGETS + ct) * 2;
    uint32_t bin, rib;
    wave_select(hist_level23(hist, img, 3, c, t), st[1], bin, rib);
    if ((threadIdx.x & 63) == 0) val[ct] = __uint_as_float(st[0] | bin);
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    const float a0 = val[c * PCT_TARGETS + 0], b0 = val[c * PCT_TARGETS + 1];
    const float a1 = val[c * PCT_TARGETS + 2], b1 = val[c * PCT_TARGETS + 3];
    double p_low, p_high;
    float lo, inv;
    if (R.legacy) {
      // numpy < 2: float64 percentiles, float64 normalisation
      p_low = lerp_legacy(a0, b0, R.g_low);
      p_high = lerp_legacy(a1, b1, R.g_high);
      const double denom = fmax(p_high - p_low, 1e-3);
      lo = (float)p_low;
      inv = (float)(1.0 / denom);
    } else {
      const float pl = lerp_f32(a0, b0, (float)R.g_low);
      const float ph = lerp_f32(a1, b1, (float)R.g_high);
      const float denom = fmaxf(sub_rn(ph, pl), 1e-3f);
      p_low = pl;
      p_high = ph;
      lo = pl;
      inv = 1.0f / denom;
    }
    params[img * SS_NEG_PARAM_STRIDE + c] = lo;
    params[img * SS_NEG_PARAM_STRIDE + 3 + c] = inv;
    if (c == 0) {
      params[img * SS_NEG_PARAM_STRIDE + 6] = 0.f;
      params[img * SS_NEG_PARAM_STRIDE + 7] = 0.f;
    }
    if (detail) {
      ss_percentile_detail* dd = detail + img;
      dd->x_low_lo[c] = a0;
      dd->x_low_hi[c] = b0;
      dd->x_high_lo[c] = a1;
      dd->x_high_hi[c] = b1;
      dd->p_low[c] = p_low;
      dd->p_high[c] = p_high;
      if (c == 0) {
        dd->step = R.step;
        dd->n_samples = (int32_t)R.n;
        dd->k_low = R.k[0];
        dd->k_high = R.k[2];
      }
    }
  }
}

hipError_t launch_negative_params(const float* src, int height, int width, int batch, int step, int64_t n,
                                  int64_t k_low, int64_t k_low_hi, double g_low, int64_t k_high, int64_t k_high_hi,
                                  double g_high, int legacy, const PercentileWork& w, float* params,
                                  ss_percentile_detail* detail, hipStream_t st) {
  hipError_t e = hipMemsetAsync(w.hist, 0, percentile_hist_bytes(batch), st);
  if (e != hipSuccess) return e;
  const int ws = (width + step - 1) / step;
  PctRanks R;
  R.k[0] = (uint32_t)k_low;
  R.k[1] = (uint32_t)k_low_hi;
  R.k[2] = (uint32_t)k_high;
  R.k[3] = (uint32_t)k_high_hi;
  R.g_low = g_low;
  R.g_high = g_high;
  R.legacy = legacy;
  R.step = step;
  R.n = (uint32_t)n;
  unsigned blocks = (unsigned)((n + 256 * 8 - 1) / (256 * 8));
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  // Every workgroup clears and flushes its own LDS histograms (3 K words when sampling, 12 K when refining): with
  // many images in the batch, fewer and longer workgroups per image keep that fixed cost small against the
  // samples a workgroup handles, and the batch still fills the GPU.
  auto per_image = [&](unsigned total_target, unsigned at_least) {
    unsigned b = total_target / (unsigned)batch;
    if (b < at_least) b = at_least;
    return b < blocks ? b : blocks;
  };
  const dim3 grid(per_image(SS_PCT_SAMPLE_BLOCKS, 16), batch), grid_refine(per_image(SS_PCT_REFINE_BLOCKS, 4), batch);
  hipLaunchKernelGGL(pct_sample_kernel, grid, dim3(256), 0, st, src, height, width, step, ws, (uint32_t)n, w.samples,
                     w.hist);
  hipLaunchKernelGGL(pct_select_kernel<1>, dim3(batch), dim3(256), 0, st, w.hist, w.state, R);
  hipLaunchKernelGGL(pct_refine_kernel<2>, grid_refine, dim3(256), 0, st, w.samples, (uint32_t)n, w.state, w.hist);
  hipLaunchKernelGGL(pct_select_kernel<2>, dim3(batch), dim3(256), 0, st, w.hist, w.state, R);
  hipLaunchKernelGGL(pct_refine_kernel<3>, grid_refine, dim3(256), 0, st, w.samples, (uint32_t)n, w.state, w.hist);
  hipLaunchKernelGGL(pct_final_kernel, dim3(batch), dim3(256), 0, st, w.hist, w.state, R, params, detail);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// exposure statistics
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// partials[img][blk][4] = {masked sum, masked count, total sum, total count}; fixed summation
// order (lane tree, then waves in order, then blocks in order) so results are reproducible.
__global__ __launch_bounds__(256) void exposure_partial_kernel(const float* __restrict__ thumbs, uint32_t n_px,
                                                               float lo, float hi, double* __restrict__ partials) {
  __shared__ double red[4][4];
  const int img = blockIdx.y;
  const F3* __restrict__ s = reinterpret_cast<const F3*>(thumbs + (size_t)img * n_px * 3);
  double ms = 0.0, ts = 0.0;
  uint32_t mc = 0, tc = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_px; i += gridDim.x * 256) {
    const F3 v = s[i];
    // numpy evaluates (0.2126*R + 0.7152*G) + 0.0722*B in float32 without fused multiply-add
    const float lum = add_rn(add_rn(mul_rn(0.2126f, v.x), mul_rn(0.7152f, v.y)), mul_rn(0.0722f, v.z));
    ts += (double)lum;
    tc += 1;
    if (lum > lo && lum < hi) {
      ms += (double)lum;
      mc += 1;
    }
  }
  double q[4] = {ms, (double)mc, ts, (double)tc};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double r = wave_sum(q[k]);
    if (lane == 0) red[wave][k] = r;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    partials[((size_t)img * gridDim.x + blockIdx.x) * 4 + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
  }
}

__global__ void exposure_final_kernel(const double* __restrict__ partials, int nblk, int batch,
                                      double* __restrict__ stats, float* __restrict__ brightness) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= batch) return;
  double q[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = 0; b < nblk; ++b)
    for (int k = 0; k < 4; ++k) q[k] += partials[((size_t)img * nblk + b) * 4 + k];
  if (stats)
    for (int k = 0; k < 4; ++k) stats[(size_t)img * 4 + k] = q[k];
  const double mean = q[1] > 0.0 ? q[0] / q[1] : (q[3] > 0.0 ? q[2] / q[3] : 0.0);
  brightness[img] = (float)mean;
}

hipError_t launch_exposure_stats(const float* thumbs, int height, int width, int batch, float lo, float hi,
                                 double* partials, double* stats, float* brightness, hipStream_t st) {
  const uint32_t n_px = (uint32_t)height * (uint32_t)width;
  hipLaunchKernelGGL(exposure_partial_kernel, dim3(kStatsBlocksPerImage, batch), dim3(256), 0, st, thumbs, n_px, lo, hi,
                     partials);
  hipLaunchKernelGGL(exposure_final_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, partials, kStatsBlocksPerImage,
                     batch, stats, brightness);
  return hipGetLastError();
}

// reference = brightness[ref_index] or the mean of all (np.mean over float32 scalars, summed here
// in float64 in list order); multiplier = reference / brightness, 1.0 where brightness <= 0
__global__ __launch_bounds__(256) void exposure_mult_kernel(const float* __restrict__ brightness, int n, int ref_index,
                                                            int first, int count, float* __restrict__ mult) {
  __shared__ double red[4];
  __shared__ float ref_s;
  if (ref_index >= 0) {
    if (threadIdx.x == 0) ref_s = brightness[ref_index];
  } else {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)brightness[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ref_s = (float)((((red[0] + red[1]) + red[2]) + red[3]) / (double)n);
  }
  __syncthreads();
  const float ref = ref_s;
  for (int i = threadIdx.x; i < count; i += 256) {
    const float b = brightness[first + i];
    mult[i] = b > 0.0f ? ref / b : 1.0f;
  }
}

hipError_t launch_exposure_multipliers(const float* brightness, int n, int ref_index, int first, int count,
                                       float* mult, hipStream_t st) {
  hipLaunchKernelGGL(exposure_mult_kernel, dim3(1), dim3(256), 0, st, brightness, n, ref_index, first, count, mult);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// _is_lab_image
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_down(v, off, 64));
  return v;
}

__global__ __launch_bounds__(256) void lab_partial_kernel(const float* __restrict__ src, int64_t n_px,
                                                          float* __restrict__ partials) {
  __shared__ float red[4][3];
  const F3* __restrict__ s = reinterpret_cast<const F3*>(src);
  float mx0 = -INFINITY, mn1 = INFINITY, mn2 = INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_px; i += (int64_t)gridDim.x * 256) {
    const F3 v = s[i];
    mx0 = fmaxf(mx0, v.x);
    mn1 = fminf(mn1, v.y);
    mn2 = fminf(mn2, v.z);
  }
  mx0 = wave_max(mx0);
  mn1 = wave_min(mn1);
  mn2 = wave_min(mn2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[wave][0] = mx0;
    red[wave][1] = mn1;
    red[wave][2] = mn2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 3 + 0] = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    partials[blockIdx.x * 3 + 1] = fminf(fminf(red[0][1], red[1][1]), fminf(red[2][1], red[3][1]));
    partials[blockIdx.x * 3 + 2] = fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2]));
  }
}

__global__ __launch_bounds__(64) void lab_final_kernel(const float* __restrict__ partials, int nblk,
                                                       int32_t* __restrict__ out, float* __restrict__ minmax) {
  float mx0 = -INFINITY, mn1 = INFINITY, mn2 = INFINITY;
  for (int i = threadIdx.x; i < nblk; i += 64) {
    mx0 = fmaxf(mx0, partials[i * 3 + 0]);
    mn1 = fminf(mn1, partials[i * 3 + 1]);
    mn2 = fminf(mn2, partials[i * 3 + 2]);
  }
  mx0 = wave_max(mx0);
  mn1 = wave_min(mn1);
  mn2 = wave_min(mn2);
  if (threadIdx.x == 0) {
    *out = (mx0 > 1.5f && (mn1 < -1.0f || mn2 < -1.0f)) ? 1 : 0;
    if (minmax) {
      minmax[0] = mx0;
      minmax[1] = mn1;
      minmax[2] = mn2;
    }
  }
}

hipError_t launch_is_lab(const float* src, int64_t n_px, float* partials, int32_t* out, float* minmax,
                         hipStream_t st) {
  int64_t want = (n_px + 256 * 4 - 1) / (256 * 4);
  const int nblk = (int)(want < 1 ? 1 : (want > kLabDetectBlocks ? kLabDetectBlocks : want));
  hipLaunchKernelGGL(lab_partial_kernel, dim3(nblk), dim3(256), 0, st, src, n_px, partials);
  hipLaunchKernelGGL(lab_final_kernel, dim3(1), dim3(64), 0, st, partials, nblk, out, minmax);
  return hipGetLastError();
}

}  // namespace ss
