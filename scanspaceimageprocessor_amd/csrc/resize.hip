// Area-average downscale of interleaved float32 RGB images for gfx950: the preview / thumbnail producers
// of the reference (cv2.resize(..., interpolation=cv2.INTER_AREA) at imageLoader.py:121, imageLoader.py:302,
// image_space.py:2983).  Every destination pixel averages the source rectangle it covers; partially
// covered source pixels are weighted by the covered fraction (OpenCV's computeResizeAreaTab rules,
// including its 1e-3 coverage threshold).  Lanes run along destination x, so a wave reads one contiguous run
// of each source row; the source is read once (HBM-bound, 12 B per source pixel).

#include "ss_internal.h"

namespace ss {

// One axis of the weight table for destination index d
struct AreaSpan {
  int first;     // first fully covered source index
  int last;      // one past the last fully covered source index (also the index of the trailing partial)
  float a_head;  // weight of source first-1 (0: not covered)
  float a_full;  // weight of every fully covered source index
  float a_tail;  // weight of source `last` (0: not covered)
};

__device__ __forceinline__ AreaSpan area_span(int d, int ssize, double scale) {
  const double f1 = d * scale, f2 = f1 + scale;
  const double cell = fmin(scale, (double)ssize - f1);
  int s1 = (int)ceil(f1), s2 = (int)floor(f2);
  s2 = s2 < ssize - 1 ? s2 : ssize - 1;
  s1 = s1 < s2 ? s1 : s2;
  AreaSpan a;
  a.first = s1;
  a.last = s2;
  a.a_head = (s1 - f1 > 1e-3) ? (float)((s1 - f1) / cell) : 0.0f;
  a.a_full = (float)(1.0 / cell);
  a.a_tail = (f2 - s2 > 1e-3) ? (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell) : 0.0f;
  return a;
}

// G threads share one destination pixel: thread g takes the covered source rows g, g + G, ... and the G
// partial sums are added in a fixed order through LDS.  (OpenCV adds the rows one after the other; the
// grouping changes the float32 rounding by an ulp or two, nothing else.)  G > 1 is for strong reductions,
// where one thread per destination pixel leaves too few loads in flight.
template <int G>
__global__ __launch_bounds__(64 * G) void resize_area_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int sh, int sw, int dh, int dw, double scale_y,
                                                             double scale_x) {
  __shared__ float part[G > 1 ? G : 1][3][64];
  const int lx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int dx = blockIdx.x * 64 + lx;
  const int dy = blockIdx.y;
  const bool live = dx < dw;
  const F3* __restrict__ s = reinterpret_cast<const F3*>(src) + (size_t)blockIdx.z * sh * sw;
  const AreaSpan ax = area_span(live ? dx : dw - 1, sw, scale_x);
  const AreaSpan ay = area_span(dy, sh, scale_y);  // uniform over the workgroup
  // horizontal average of source row sy over this pixel's span (OpenCV reduces rows first, in float32)
  auto row_avg = [&](int sy) -> F3 {
    const F3* r = s + (size_t)sy * sw;
    F3 acc{0.0f, 0.0f, 0.0f};
    if (ax.a_head != 0.0f) {
      const F3 v = r[ax.first - 1];
      acc = F3{v.x * ax.a_head, v.y * ax.a_head, v.z * ax.a_head};
    }
    int sx = ax.first;
    for (; sx + 4 <= ax.last; sx += 4) {  // four loads in flight
      const F3 v0 = r[sx], v1 = r[sx + 1], v2 = r[sx + 2], v3 = r[sx + 3];
      acc.x = fmaf(v3.x, ax.a_full, fmaf(v2.x, ax.a_full, fmaf(v1.x, ax.a_full, fmaf(v0.x, ax.a_full, acc.x))));
      acc.y = fmaf(v3.y, ax.a_full, fmaf(v2.y, ax.a_full, fmaf(v1.y, ax.a_full, fmaf(v0.y, ax.a_full, acc.y))));
      acc.z = fmaf(v3.z, ax.a_full, fmaf(v2.z, ax.a_full, fmaf(v1.z, ax.a_full, fmaf(v0.z, ax.a_full, acc.z))));
    }
    for (; sx < ax.last; ++sx) {
      const F3 v = r[sx];
      acc.x = fmaf(v.x, ax.a_full, acc.x);
      acc.y = fmaf(v.y, ax.a_full, acc.y);
      acc.z = fmaf(v.z, ax.a_full, acc.z);
    }
    if (ax.a_tail != 0.0f) {
      const F3 v = r[ax.last];
      acc.x = fmaf(v.x, ax.a_tail, acc.x);
      acc.y = fmaf(v.y, ax.a_tail, acc.y);
      acc.z = fmaf(v.z, ax.a_tail, acc.z);
    }
    return acc;
  };
  // the covered source rows as one list: [head], full rows, [tail]
  const int has_head = ay.a_head != 0.0f ? 1 : 0, has_tail = ay.a_tail != 0.0f ? 1 : 0;
  const int n_rows = has_head + (ay.last - ay.first) + has_tail;
  F3 sum{0.0f, 0.0f, 0.0f};
  for (int i = g; i < n_rows; i += G) {
    const int sy = ay.first - has_head + i;
    const float beta = (i < has_head) ? ay.a_head : (sy < ay.last ? ay.a_full : ay.a_tail);
    const F3 b = row_avg(sy);
    sum.x = fmaf(b.x, beta, sum.x);
    sum.y = fmaf(b.y, beta, sum.y);
    sum.z = fmaf(b.z, beta, sum.z);
  }
  if constexpr (G > 1) {
    part[g][0][lx] = sum.x;
    part[g][1][lx] = sum.y;
    part[g][2][lx] = sum.z;
    __syncthreads();
    if (g != 0) return;
#pragma unroll
    for (int k = 1; k < G; ++k) {
      sum.x += part[k][0][lx];
      sum.y += part[k][1][lx];
      sum.z += part[k][2][lx];
    }
  }
  if (live) reinterpret_cast<F3*>(dst)[((size_t)blockIdx.z * dh + dy) * dw + dx] = sum;
}

// Coalesced variant for moderate horizontal reductions.  A workgroup makes kTileDx adjacent destination pixels of
// one destination row.  Its threads sweep the covered source rows with lane-contiguous 4-byte loads (every
// source float is read once, by the thread that owns its column position), folding the rows with their
// vertical weights into per-thread registers; the folded strip goes to LDS and the first 3 * kTileDx threads
// reduce it horizontally.  Columns are reduced after rows here and before rows in OpenCV: the same sums in
// another order (a float32 rounding difference, nothing else).
constexpr int kTileThreads = 256;
constexpr int kTileMaxScale = 16;                                       // with 64 destination pixels per workgroup
constexpr int kTileFloats = (64 * kTileMaxScale + 2) * 3;               // strips up to 1026 source pixels: 3078 floats
constexpr int kTileVecs = (kTileFloats + 4 * kTileThreads - 1) / (4 * kTileThreads);  // 4 vectors of 4 floats per thread

struct __attribute__((aligned(4))) F4 {  // 16 bytes at 4-byte alignment: one global_load_dwordx4
  float x, y, z, w;
};

// kTileDx destination pixels per workgroup: 64 / 128 / 256 for scales up to 16 / 8 / 4, so that the strip (and with
// it every thread's accumulators) is well filled at any scale.  kRowBatch: source rows whose loads are issued together
// (a strong reduction covers many rows and gains from more loads in flight; a weak one covers few).
template <int kTileDx, int kRowBatch>
__global__ __launch_bounds__(kTileThreads) void resize_area_tiled_kernel(const float* __restrict__ src,
                                                                         float* __restrict__ dst, int sh, int sw, int dh,
                                                                         int dw, double scale_y, double scale_x) {
  __shared__ float strip[kTileFloats];
  const int tid = threadIdx.x;
  const int dx0 = blockIdx.x * kTileDx;
  const int dx1 = dx0 + kTileDx < dw ? dx0 + kTileDx : dw;
  const int dy = blockIdx.y;
  const float* __restrict__ s = src + (size_t)blockIdx.z * sh * sw * 3;
  const AreaSpan ay = area_span(dy, sh, scale_y);
  const AreaSpan a_first = area_span(dx0, sw, scale_x), a_last = area_span(dx1 - 1, sw, scale_x);
  const int sx_begin = a_first.first - (a_first.a_head != 0.0f ? 1 : 0);
  const int sx_end = a_last.last + (a_last.a_tail != 0.0f ? 1 : 0);  // one past the last source pixel touched
  const int n_floats = (sx_end - sx_begin) * 3;                         // <= kTileFloats by the launcher's choice
  float acc[kTileVecs][4];
#pragma unroll
  for (int k = 0; k < kTileVecs; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0f;
  const int has_head = ay.a_head != 0.0f ? 1 : 0, has_tail = ay.a_tail != 0.0f ? 1 : 0;
  const int n_rows = has_head + (ay.last - ay.first) + has_tail;
  // kRowBatch source rows per trip: all their loads are issued before the first is folded, so a thread keeps up to
  // kRowBatch * kTileVecs 16-byte loads in flight.  Rows past the end are not loaded and fold zeros with weight 0; the
  // folding order (row after row) is unchanged.
  for (int i0 = 0; i0 < n_rows; i0 += kRowBatch) {
    F4 v[kRowBatch][kTileVecs];
    float beta[kRowBatch];
#pragma unroll
    for (int r = 0; r < kRowBatch; ++r) {
      const bool on = i0 + r < n_rows;
      const int i = on ? i0 + r : n_rows - 1;
      const int sy = ay.first - has_head + i;
      beta[r] = !on ? 0.0f : (i < has_head) ? ay.a_head : (sy < ay.last ? ay.a_full : ay.a_tail);
      const float* __restrict__ row = s + ((size_t)sy * sw + sx_begin) * 3;
#pragma unroll
      for (int k = 0; k < kTileVecs; ++k) {
        const int f = (tid + k * kTileThreads) * 4;
        if (!on) {  // (uniform over the workgroup)
          v[r][k] = F4{0.0f, 0.0f, 0.0f, 0.0f};
        } else if (f + 4 <= n_floats) {
          // (cached loads: the strips' 16-byte accesses are not line-aligned, neighbouring ones complete a line)
          v[r][k] = *reinterpret_cast<const F4*>(row + f);
        } else {
          v[r][k] = F4{f < n_floats ? row[f] : 0.0f, f + 1 < n_floats ? row[f + 1] : 0.0f,
                       f + 2 < n_floats ? row[f + 2] : 0.0f, 0.0f};
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kRowBatch; ++r) {
#pragma unroll
      for (int k = 0; k < kTileVecs; ++k) {
        acc[k][0] = fmaf(v[r][k].x, beta[r], acc[k][0]);
        acc[k][1] = fmaf(v[r][k].y, beta[r], acc[k][1]);
        acc[k][2] = fmaf(v[r][k].z, beta[r], acc[k][2]);
        acc[k][3] = fmaf(v[r][k].w, beta[r], acc[k][3]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kTileVecs; ++k) {
    const int f = (tid + k * kTileThreads) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (f + j < n_floats) strip[f + j] = acc[k][j];
  }
  __syncthreads();
  const int n_out = (dx1 - dx0) * 3;
  for (int o = tid; o < n_out; o += kTileThreads) {
    const int dxl = o / 3, ch = o - dxl * 3;
    const AreaSpan ax = area_span(dx0 + dxl, sw, scale_x);
    const float* col = strip + ch - sx_begin * 3;
    float sum = 0.0f;
    if (ax.a_head != 0.0f) sum = col[(ax.first - 1) * 3] * ax.a_head;
    for (int sx = ax.first; sx < ax.last; ++sx) sum = fmaf(col[sx * 3], ax.a_full, sum);
    if (ax.a_tail != 0.0f) sum = fmaf(col[ax.last * 3], ax.a_tail, sum);
    dst[(((size_t)blockIdx.z * dh + dy) * dw + dx0) * 3 + o] = sum;
  }
}

hipError_t launch_resize_area(const float* src, int sh, int sw, int batch, float* dst, int dh, int dw, hipStream_t st) {
  if (dh < 1 || dw < 1 || dh > sh || dw > sw || batch < 1 || dh > 65535 || batch > 65535) return hipErrorInvalidValue;
  // cv::resize forms the inverse scale first and inverts it again
  const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);
  if (scale_x >= 2.0 && scale_x <= (double)kTileMaxScale) {
    auto tiled = [&](auto kern, int tile_dx) {
      const dim3 tgrid((dw + tile_dx - 1) / tile_dx, dh, batch);
      hipLaunchKernelGGL(kern, tgrid, dim3(kTileThreads), 0, st, src, dst, sh, sw, dh, dw, scale_y, scale_x);
    };
    if (scale_x <= 4.0) {
      tiled(resize_area_tiled_kernel<256, 1>, 256);
    } else if (scale_x <= 8.0) {
      tiled(resize_area_tiled_kernel<128, 2>, 128);
    } else {
      tiled(resize_area_tiled_kernel<64, 4>, 64);
    }
    return hipGetLastError();
  }
  const dim3 grid((dw + 63) / 64, dh, batch);
  if (scale_y >= 8.0) {
    hipLaunchKernelGGL(resize_area_kernel<8>, grid, dim3(512), 0, st, src, dst, sh, sw, dh, dw, scale_y, scale_x);
  } else if (scale_y >= 3.0) {
    hipLaunchKernelGGL(resize_area_kernel<4>, grid, dim3(256), 0, st, src, dst, sh, sw, dh, dw, scale_y, scale_x);
  } else {
    hipLaunchKernelGGL(resize_area_kernel<1>, grid, dim3(64), 0, st, src, dst, sh, sw, dh, dw, scale_y, scale_x);
  }
  return hipGetLastError();
}

}  // namespace ss
