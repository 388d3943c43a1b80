// Shadow / highlight region masking for gfx950: the worker's `_apply_masking`
// (imageProcessorWorker.py:979-1030).  A pixel is a candidate when its luminance is <= shadow_limit or
// >= highlight_limit; candidates are grouped into 8-connected components
// (cv2.connectedComponentsWithStats(mask, connectivity=8)) and every component of at least `min_area`
// pixels is set to zero in the image.  The result is a set of pixels: it does not depend on how components
// are numbered, so it is reproduced bit for bit.
//
// Connected components by union-find over row runs:
//   mask_runs_kernel    one workgroup per image row: luminance test, then every candidate pixel gets the
//                       index of the first pixel of its horizontal run (segmented max-scan along the row).
//   mask_merge_kernel   runs of adjacent rows that touch (8-connectivity) are united, once per contact:
//                       lock-free union with path halving.  Every write to a parent word in this kernel is an
//                       atomicMin: parents only ever decrease, so no writer can undo another one's link
//                       (plain stores, which reach memory late from a remote XCD's L2, did exactly that).
//   mask_flatten_kernel pointer jumping, one launch per round until a round changes nothing: each word has
//                       a single writer here, so plain stores are safe.  Afterwards every label is a root.
//   mask_area_kernel    areas per root, one atomic per group of lanes that share a root.
//   mask_apply_kernel   pixels whose component is large enough are zeroed and counted.
// Labels are pixel indices inside their own image (int32: images up to 2^31 pixels).

#include "ss_internal.h"

namespace ss {

namespace {

constexpr int kRowThreads = 256;
constexpr int kPxPerThread = 4;
constexpr int kChunk = kRowThreads * kPxPerThread;

// np: 0.2126 * c0 + 0.7152 * c1 + 0.0722 * c2 on float64 data, no contraction
__device__ __forceinline__ bool is_candidate(const F3 p, double lo, double hi) {
  const double lum = __dadd_rn(__dadd_rn(__dmul_rn(0.2126, (double)p.x), __dmul_rn(0.7152, (double)p.y)),
                               __dmul_rn(0.0722, (double)p.z));
  return lum <= lo || lum >= hi;
}

__device__ __forceinline__ int wave_scan_max(int v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d);
    if ((int)(threadIdx.x & 63) >= d) v = v > o ? v : o;
  }
  return v;
}

__global__ __launch_bounds__(kRowThreads) void mask_runs_kernel(const float* __restrict__ img, int32_t* __restrict__ label,
                                                                int h, int w, double lo, double hi) {
  __shared__ int wave_last[kRowThreads / 64];
  __shared__ int carry_s;
  const int y = blockIdx.x, b = blockIdx.y;
  const size_t row0 = ((size_t)b * h + y) * w;
  const F3* __restrict__ src = reinterpret_cast<const F3*>(img) + row0;
  int32_t* __restrict__ out = label + row0;
  const int base_label = y * w;  // labels are indices within the image
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // `brk` = x + 1 of the last non-candidate pixel at or before a position (0: none yet in this row), so the
  // run of a candidate pixel starts at the running maximum of brk
  int carry = 0;
  for (int x0 = 0; x0 < w; x0 += kChunk) {
    const int xt = x0 + tid * kPxPerThread;
    bool m[kPxPerThread];
    int local = 0;
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) {
      const int x = xt + k;
      m[k] = x < w && is_candidate(src[x], lo, hi);
      if (!m[k]) local = x + 1;
    }
    // exclusive prefix of `local` over the threads of the chunk
    const int incl = wave_scan_max(local);
    if (lane == 63) wave_last[wave] = incl;
    __syncthreads();
    int before = carry;
#pragma unroll
    for (int k = 0; k < kRowThreads / 64; ++k)
      if (k < wave) before = before > wave_last[k] ? before : wave_last[k];
    int excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 0;
    int run = before > excl ? before : excl;
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) {
      const int x = xt + k;
      if (x < w) {
        if (m[k]) {
          out[x] = base_label + run;
        } else {
          out[x] = -1;
          run = x + 1;
        }
      }
    }
    if (tid == kRowThreads - 1) carry_s = before > incl ? before : incl;
    __syncthreads();
    carry = carry_s;
  }
}

__device__ __forceinline__ int find_root(int32_t* parent, int a) {
  // path halving with atomicMin: a parent word is only ever lowered, whoever writes it
  int p = parent[a];
  while (p != a) {
    const int g = parent[p];
    if (g != p) atomicMin(&parent[a], g);
    a = p;
    p = g;
  }
  return a;
}

__device__ __forceinline__ void unite(int32_t* parent, int a, int b) {
  for (;;) {
    a = find_root(parent, a);
    b = find_root(parent, b);
    if (a == b) return;
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }
    // a > b: hang root a under b, unless somebody re-parented a in the meantime
    const int old = atomicMin(&parent[a], b);
    if (old == a) return;
    a = old;
  }
}

// One thread per pixel.  A run pair is united at its first contact only (scanning left to right):
//   upper pixel N is a candidate: first contact iff the left pixel or its upper pixel is not;
//   otherwise NW counts iff the left pixel is not a candidate (else it already touched it as its N),
//   and NE always counts (the right pixel would see it as its N only if N were a candidate too).
__global__ __launch_bounds__(256) void mask_merge_kernel(int32_t* label, int h, int w) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= w || y == 0) return;
  int32_t* L = label + (size_t)blockIdx.z * h * w;
  const int i = y * w + x;
  const int me = L[i];
  if (me < 0) return;
  const int up = i - w;
  const int n = L[up];
  const bool left = x > 0 && L[i - 1] >= 0;
  if (n >= 0) {
    const bool left_up = x > 0 && L[up - 1] >= 0;
    if (!(left && left_up)) unite(L, me, n);
  } else {
    if (x > 0 && !left) {
      const int nw = L[up - 1];
      if (nw >= 0) unite(L, me, nw);
    }
    if (x + 1 < w) {
      const int ne = L[up + 1];
      if (ne >= 0) unite(L, me, ne);
    }
  }
}

// One round of pointer jumping: label[i] <- its ancestor up to 8 levels up.  `flags[round]` is raised while
// some label may still not be a root;
// a round whose predecessor moved nothing returns at once (the launcher issues a fixed number of rounds, on
// a fixed modest grid so that the idle ones cost microseconds).
__global__ __launch_bounds__(256) void mask_flatten_kernel(int32_t* label, int n_px, int* flags, int round) {
  if (round > 0 && flags[round - 1] == 0) return;
  int32_t* L = label + (size_t)blockIdx.y * n_px;
  bool moved = false;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_px; i += gridDim.x * 256) {
    const int p = L[i];
    if (p < 0) continue;
    int a = p, g = L[a];
#pragma unroll 1
    for (int hop = 0; hop < 7 && g != a; ++hop) {  // up to 8 levels per round
      a = g;
      g = L[a];
    }
    if (g != p) {
      L[i] = g;
      if (L[g] != g) moved = true;  // not at a root yet (as far as this thread can see): another round
    }
  }
  if (moved) flags[round] = 1;
}

constexpr int kAreaRun = 16;  // consecutive pixels per thread in mask_area_kernel (runs are horizontal)

// area[root] += 1 for every candidate pixel.  A thread first folds its own 16 consecutive pixels (those that
// share the root of its first candidate pixel), lanes with the same root then fold through the wave, and one
// atomic goes out per distinct root; the few leftovers at component borders go out one by one.
__global__ __launch_bounds__(256) void mask_area_kernel(const int32_t* __restrict__ label, int32_t* __restrict__ area,
                                                        int n_px) {
  const int32_t* __restrict__ L = label + (size_t)blockIdx.y * n_px;
  int32_t* A = area + (size_t)blockIdx.y * n_px;
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * kAreaRun;
  int r[kAreaRun];
#pragma unroll
  for (int k = 0; k < kAreaRun; ++k) r[k] = i0 + k < n_px ? L[i0 + k] : -1;
  int first = -1;
#pragma unroll
  for (int k = kAreaRun - 1; k >= 0; --k)
    if (r[k] >= 0) first = r[k];
  int mine = 0;
#pragma unroll
  for (int k = 0; k < kAreaRun; ++k) {
    if (r[k] == first) {
      mine += 1;
    } else if (r[k] >= 0) {
      atomicAdd(&A[r[k]], 1);
    }
  }
  // fold `mine` over the lanes that share `first`
  bool pending = first >= 0;
  while (__any(pending)) {
    const unsigned long long todo = __ballot(pending);
    const int leader = __ffsll((long long)todo) - 1;
    const int k = __shfl(first, leader);
    const bool same = pending && first == k;
    int sum = same ? mine : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&A[k], sum);
    if (same) pending = false;
  }
}

// Fixed grid, grid-stride: a workgroup counts its hits locally and sends one atomic per image (one atomic
// per wave would put hundreds of thousands of them on a single address).
__global__ __launch_bounds__(256) void mask_apply_kernel(float* __restrict__ img, const int32_t* __restrict__ label,
                                                         const int32_t* __restrict__ area, int n_px, int min_area,
                                                         unsigned long long* __restrict__ count) {
  __shared__ unsigned int wave_hits[4];
  const size_t off = (size_t)blockIdx.y * n_px;
  unsigned int hits = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_px; i += gridDim.x * 256) {
    const int root = label[off + i];
    if (root >= 0 && area[off + root] >= min_area) {
      reinterpret_cast<F3*>(img)[off + i] = F3{0.0f, 0.0f, 0.0f};
      hits += 1;
    }
  }
  if (!count) return;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) hits += __shfl_xor(hits, d);
  if ((threadIdx.x & 63) == 0) wave_hits[threadIdx.x >> 6] = hits;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int total = wave_hits[0] + wave_hits[1] + wave_hits[2] + wave_hits[3];
    if (total) atomicAdd(&count[blockIdx.y], (unsigned long long)total);
  }
}

}  // namespace

// rounds of pointer jumping issued: a round divides every label's distance to its root by 8, so a chain of
// 2^31 nodes (more than an image with int32 labels can hold) is flat after 11 rounds; one more confirms it
constexpr int kFlattenRounds = 12;

size_t mask_workspace_bytes(int batch, int64_t n_px) {
  return (size_t)batch * n_px * 2 * sizeof(int32_t) + (kFlattenRounds + 1) * sizeof(int);
}

hipError_t launch_mask_regions(float* img, int h, int w, int batch, double shadow_limit, double highlight_limit,
                               int min_area, void* workspace, unsigned long long* count_dev, hipStream_t st) {
  const int64_t n_px = (int64_t)h * w;
  if (h < 1 || w < 1 || batch < 1 || batch > 65535 || h > 65535 || n_px > INT32_MAX) return hipErrorInvalidValue;
  int32_t* label = reinterpret_cast<int32_t*>(workspace);
  int32_t* area = label + (size_t)batch * n_px;
  int* flags = reinterpret_cast<int*>(area + (size_t)batch * n_px);
  hipError_t e = hipMemsetAsync(area, 0, (size_t)batch * n_px * sizeof(int32_t) + (kFlattenRounds + 1) * sizeof(int), st);
  if (e != hipSuccess) return e;
  if (count_dev) {
    e = hipMemsetAsync(count_dev, 0, (size_t)batch * sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(mask_runs_kernel, dim3(h, batch), dim3(kRowThreads), 0, st, img, label, h, w, shadow_limit,
                     highlight_limit);
  hipLaunchKernelGGL(mask_merge_kernel, dim3((w + 255) / 256, h, batch), dim3(256), 0, st, label, h, w);
  const unsigned blocks = (unsigned)((n_px + 255) / 256);
  const unsigned flat_blocks = blocks < 4096 ? blocks : 4096;
  for (int round = 0; round < kFlattenRounds; ++round)
    hipLaunchKernelGGL(mask_flatten_kernel, dim3(flat_blocks, batch), dim3(256), 0, st, label, (int)n_px, flags, round);
  const unsigned area_blocks = (unsigned)((n_px + 256 * kAreaRun - 1) / (256 * kAreaRun));
  hipLaunchKernelGGL(mask_area_kernel, dim3(area_blocks, batch), dim3(256), 0, st, label, area, (int)n_px);
  hipLaunchKernelGGL(mask_apply_kernel, dim3(flat_blocks, batch), dim3(256), 0, st, img, label, area, (int)n_px,
                     min_area, count_dev);
  return hipGetLastError();
}

}  // namespace ss
