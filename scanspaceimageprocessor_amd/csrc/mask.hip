// Shadow / highlight region masking for gfx950: the worker's `_apply_masking`
// (imageProcessorWorker.py:979-1030).  A pixel is a candidate when its luminance is <= shadow_limit or
// >= highlight_limit; candidates are grouped into 8-connected components
// (cv2.connectedComponentsWithStats(mask, connectivity=8)) and every component of at least `min_area`
// pixels is set to zero in the image.  The result is a set of pixels: it does not depend on how components
// are numbered, so it is reproduced bit for bit.
//
// Connected components by union-find over row runs.  The image is read once; everything after that works
// on the list of runs (maximal horizontal stretches of candidate pixels), which for the masks this is made
// for - a backdrop, a blown highlight - is a few entries per row:
//   mask_runs_kernel    one workgroup per image row: luminance test, run starts and ends, the run list of
//                       the row (run k of row y lives in slot y * cap + k, cap = ceil(w / 2) = the most runs
//                       a row can hold).
//   mask_merge_kernel   every run is united with the runs of the row above it touches (8-connectivity:
//                       columns [start - 1, end + 1]), found by bisection in that row's sorted list.
//                       Lock-free union with path halving.  Every write to a parent word in this kernel is an
//                       atomicMin: parents only ever decrease, so no writer can undo another one's link
//                       (plain stores, which reach memory late from a remote XCD's L2, did exactly that).
//   mask_area_kernel    every run's parent becomes the root of its tree (each word has a single writer here, so
//                       plain stores are safe), and areas per root: run lengths summed per lane while the root
//                       stays the same, then over the lanes of a wave that share a root, one atomic per
//                       distinct root.
//   mask_apply_kernel   per row, the runs of components that are large enough are painted into a bitmap in
//                       LDS, then the row's pixels under the bitmap are zeroed (coalesced) and counted.

#include "ss_internal.h"

namespace ss {

namespace {

constexpr int kRowThreads = 256;
constexpr int kPxPerThread = 4;  // (2 ... 12 perform alike, 24 is 20 % slower)
constexpr int kChunk = kRowThreads * kPxPerThread;

// np: 0.2126 * c0 + 0.7152 * c1 + 0.0722 * c2 and its comparison with the limits, no contraction.
// F32 = false: float64 arithmetic on the float32 pixel values (the reference's image is float64: chart mode, or
// any Lab edit).  F32 = true: what numpy < 2 does while the image is still float32 (no chart, exposure / white
// balance only): the Python scalars do not widen the array, products, sums and limits are float32.
template <bool F32>
__device__ __forceinline__ bool is_candidate(const F3 p, double lo, double hi) {
  if constexpr (F32) {
    const float lum = add_rn(add_rn(mul_rn(0.2126f, p.x), mul_rn(0.7152f, p.y)), mul_rn(0.0722f, p.z));
    return lum <= (float)lo || lum >= (float)hi;
  } else {
    const double lum = add_rn(add_rn(mul_rn(0.2126, (double)p.x), mul_rn(0.7152, (double)p.y)),
                              mul_rn(0.0722, (double)p.z));
    return lum <= lo || lum >= hi;
  }
}

// The run lists of one image.  Slots are numbered inside the image: slot = y * cap + k.
struct RunLists {
  uint16_t* span;     // [slots][2]: first and last column of the run
  int32_t* parent;    // [slots]
  int32_t* area;      // [slots]
  int32_t* rowcount;  // [h]
};

struct MaskWork {
  uint16_t* span;
  int32_t* parent;
  int32_t* area;
  int32_t* rowcount;
  int cap;
  size_t slots;  // per image
  __host__ __device__ RunLists image(int b, int h) const {
    return RunLists{span + (size_t)b * slots * 2, parent + (size_t)b * slots, area + (size_t)b * slots,
                    rowcount + (size_t)b * h};
  }
};

// A run starts at a candidate pixel whose left neighbour is none and ends left of a non-candidate pixel whose
// left neighbour is one: both are decided looking left only.  Column w (one past the row) is walked as a
// non-candidate pixel so that a run touching the right border ends too.
// A chunk is kPxPerThread stripes of 256 adjacent pixels, lane = pixel inside the stripe (loads of a wave are
// contiguous).  The left flag comes from the neighbouring lane, across waves and stripes through LDS, across
// chunks in a register; the number of a run inside its row is the count of starts to its left: lanes below
// in the same wave (ballot), plus the totals of the waves and stripes before (LDS).
template <bool F32>
__global__ __launch_bounds__(kRowThreads) void mask_runs_kernel(const float* __restrict__ img, const MaskWork work,
                                                                int h, int w, double lo, double hi) {
  constexpr int NW = kRowThreads / 64, P = kPxPerThread;
  __shared__ int starts_of[P * NW];  // starts per (stripe, wave), in pixel order
  __shared__ int edge_of[P * NW];    // candidate flag of the last pixel of each (stripe, wave)
  const int y = blockIdx.x, b = blockIdx.y;
  const F3* __restrict__ src = reinterpret_cast<const F3*>(img) + ((size_t)b * h + y) * w;
  const RunLists R = work.image(b, h);
  const int slot0 = y * work.cap;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const unsigned long long below = (1ull << lane) - 1;
  int runs_before = 0;  // runs that start left of this chunk
  int chunk_edge = 0;   // candidate flag of the last pixel of the previous chunk
  for (int x0 = 0; x0 <= w; x0 += kChunk) {
    bool m[P];
    int left[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const int x = x0 + k * kRowThreads + tid;
      m[k] = x < w && is_candidate<F32>(load_stream(src + x), lo, hi);
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {
      left[k] = __shfl_up(m[k] ? 1 : 0, 1);
      if (lane == 63) edge_of[k * NW + wave] = m[k] ? 1 : 0;
    }
    __syncthreads();
    const int next_edge = edge_of[P * NW - 1];
    unsigned long long start_mask[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
      if (lane == 0) left[k] = (k == 0 && wave == 0) ? chunk_edge : edge_of[k * NW + wave - 1];
      start_mask[k] = __ballot(m[k] && !left[k]);
      if (lane == 0) starts_of[k * NW + wave] = __popcll(start_mask[k]);
    }
    __syncthreads();
    int before = runs_before;  // starts left of stripe k, wave `wave`
#pragma unroll
    for (int k = 0; k < P; ++k) {
#pragma unroll
      for (int v = 0; v < NW; ++v) {
        const int n = starts_of[k * NW + v];
        if (v == wave) {
          // this thread's pixel of stripe k
          const int x = x0 + k * kRowThreads + tid;
          const int idx = before + __popcll(start_mask[k] & below);  // starts left of x
          if (m[k] && !left[k]) {
            const int slot = slot0 + idx;
            R.span[2 * (size_t)slot] = (uint16_t)x;
            R.parent[slot] = slot;
            R.area[slot] = 0;
          }
          if (!m[k] && left[k]) R.span[2 * (size_t)(slot0 + idx - 1) + 1] = (uint16_t)(x - 1);
        }
        before += n;
      }
    }
    runs_before = before;
    chunk_edge = next_edge;
    __syncthreads();  // the shared words are rewritten by the next chunk
  }
  if (tid == 0) R.rowcount[y] = runs_before;
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

// Workgroups walk rows; a thread takes a run, records its length as the area of its slot, and unites it
// with every run of the row above whose columns meet [start - 1, end + 1].  Runs of a row are disjoint and
// sorted, so their ends are sorted too: the first partner is found by bisection (in a copy of that row's list
// in LDS when `stage` says it fits), the others follow it.
__global__ __launch_bounds__(256) void mask_merge_kernel(const MaskWork work, int h, int stage) {
  extern __shared__ uint32_t up_lds[];
  const RunLists R = work.image(blockIdx.y, h);
  const uint32_t* __restrict__ span = reinterpret_cast<const uint32_t*>(R.span);  // start | end << 16
  for (int y = blockIdx.x; y < h; y += gridDim.x) {
    const int n_cur = R.rowcount[y], n_up = y > 0 ? R.rowcount[y - 1] : 0;
    if (n_cur == 0) continue;  // (uniform over the workgroup)
    const int cur0 = y * work.cap, up0 = cur0 - work.cap;
    const uint32_t* up = span + up0;
    if (stage && n_up > 0) {
      __syncthreads();  // the previous row's copy is no longer read
      for (int j = threadIdx.x; j < n_up; j += 256) up_lds[j] = span[up0 + j];
      __syncthreads();
      up = up_lds;
    }
    for (int k = threadIdx.x; k < n_cur; k += 256) {
      const uint32_t se = span[cur0 + k];
      const int s = (int)(se & 0xFFFF) - 1, e = (int)(se >> 16) + 1;
      R.area[cur0 + k] = e - s - 1;
      int lo = 0, hi = n_up;  // first run above whose end is >= s
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int)(up[mid] >> 16) < s) {
          lo = mid + 1;
        } else {
          hi = mid;
        }
      }
      for (int j = lo; j < n_up && (int)(up[j] & 0xFFFF) <= e; ++j) unite(R.parent, cur0 + k, up0 + j);
    }
  }
}

// parent[slot] <- the root of its tree, and area[root] += length of every run below it (every slot starts out
// with its own length).
// Roots: no union happens any more, so the forest only gets flatter while this kernel runs.  A thread follows
// its run's path to the end (indices strictly decrease along a path, so the walk terminates) and writes its
// own word, of which it is the only writer (plain store); whatever mix of old and new parents it reads on the
// way, each is an ancestor.
// Areas: a thread keeps the sum for the root it saw last and sends it when the root changes; what is left at
// the end is folded over the lanes that share a root, one atomic per distinct root: a component that covers
// half the image (a backdrop) would otherwise put every one of its runs on a single address.
__global__ __launch_bounds__(256) void mask_area_kernel(const MaskWork work, int h) {
  const RunLists R = work.image(blockIdx.y, h);
  int root = -1, sum = 0;
  constexpr int U = 4;  // runs per thread in flight (the loads of one do not wait for those of another)
  for (int y = blockIdx.x; y < h; y += gridDim.x) {
    const int n = R.rowcount[y];
    for (int k0 = threadIdx.x; k0 < n; k0 += 256 * U) {
      int p[U], len[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u * 256, slot = y * work.cap + k;
        p[u] = k < n ? R.parent[slot] : slot;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int slot = y * work.cap + k0 + u * 256;
        if (p[u] == slot) continue;
        int a = p[u], g = R.parent[a];
        while (g != a) {
          a = g;
          g = R.parent[a];
        }
        if (a != p[u]) R.parent[slot] = a;
        p[u] = a;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int slot = y * work.cap + k0 + u * 256;
        len[u] = p[u] != slot ? R.area[slot] : 0;  // a root: its own length is there already
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int slot = y * work.cap + k0 + u * 256;
        if (p[u] == slot) continue;
        if (p[u] == root) {
          sum += len[u];
        } else {
          if (root >= 0) atomicAdd(&R.area[root], sum);
          root = p[u];
          sum = len[u];
        }
      }
    }
  }
  bool pending = root >= 0;
  while (__any(pending)) {
    const unsigned long long todo = __ballot(pending);
    const int leader = __ffsll((long long)todo) - 1;
    const int k = __shfl(root, leader);
    const bool same = pending && root == k;
    int total = same ? sum : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&R.area[k], total);
    if (same) pending = false;
  }
}

constexpr int kMaxWidth = 65535;                   // columns are stored in 16 bits
constexpr int kBitmapWords = (kMaxWidth + 32) / 32;  // 8 KB of LDS

// Workgroups walk rows.  Runs whose component is large enough set their columns in a bitmap of the row (whole
// words by plain stores - no other run can own a bit of them - and the partial words at either end by
// atomicOr); then every thread zeroes the pixels of its columns whose bit is set.  A workgroup counts what it
// zeroes and sends one atomic per image at the end (one per row would put thousands on a single address).
__global__ __launch_bounds__(256) void mask_apply_kernel(float* __restrict__ img, const MaskWork work, int h, int w,
                                                         int min_area, unsigned long long* __restrict__ count) {
  __shared__ uint32_t bitmap[kBitmapWords];
  __shared__ unsigned int wave_hits[4];
  const RunLists R = work.image(blockIdx.y, h);
  const uint32_t* __restrict__ span = reinterpret_cast<const uint32_t*>(R.span);
  const int words = (w + 31) >> 5;
  unsigned int hits = 0;
  for (int y = blockIdx.x; y < h; y += gridDim.x) {
    const int n = R.rowcount[y];
    if (n == 0) continue;  // (uniform over the workgroup)
    for (int i = threadIdx.x; i < words; i += 256) bitmap[i] = 0;
    __syncthreads();
    bool any = false;
    for (int k = threadIdx.x; k < n; k += 256) {
      const int slot = y * work.cap + k;
      if (R.area[R.parent[slot]] < min_area) continue;
      const uint32_t se = span[slot];
      const int s = (int)(se & 0xFFFF), e = (int)(se >> 16);
      any = true;
      hits += (unsigned)(e - s + 1);
      const int ws = s >> 5, we = e >> 5;
      const uint32_t first = 0xFFFFFFFFu << (s & 31), last = 0xFFFFFFFFu >> (31 - (e & 31));
      if (ws == we) {
        atomicOr(&bitmap[ws], first & last);
      } else {
        atomicOr(&bitmap[ws], first);
        for (int i = ws + 1; i < we; ++i) bitmap[i] = 0xFFFFFFFFu;
        atomicOr(&bitmap[we], last);
      }
    }
    if (__syncthreads_or(any ? 1 : 0)) {
      F3* __restrict__ row = reinterpret_cast<F3*>(img) + ((size_t)blockIdx.y * h + y) * w;
      for (int x = threadIdx.x; x < w; x += 256)
        if ((bitmap[x >> 5] >> (x & 31)) & 1u) row[x] = F3{0.0f, 0.0f, 0.0f};
    }
    __syncthreads();  // the bitmap is cleared for the next row
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

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// byte offsets of the arrays inside the workspace
struct MaskLayout {
  size_t parent, area, rowcount, total;  // (span sits at offset 0)
};

MaskLayout layout_of(int batch, int h, int w) {
  const size_t slots = (size_t)h * ((w + 1) / 2);
  MaskLayout l;
  l.parent = align256((size_t)batch * slots * 2 * sizeof(uint16_t));
  l.area = l.parent + align256((size_t)batch * slots * sizeof(int32_t));
  l.rowcount = l.area + align256((size_t)batch * slots * sizeof(int32_t));
  l.total = l.rowcount + align256((size_t)batch * h * sizeof(int32_t));
  return l;
}

MaskWork carve(void* workspace, int batch, int h, int w) {
  const MaskLayout l = layout_of(batch, h, w);
  char* p = reinterpret_cast<char*>(workspace);
  MaskWork m;
  m.cap = (w + 1) / 2;
  m.slots = (size_t)h * m.cap;
  m.span = reinterpret_cast<uint16_t*>(p);
  m.parent = reinterpret_cast<int32_t*>(p + l.parent);
  m.area = reinterpret_cast<int32_t*>(p + l.area);
  m.rowcount = reinterpret_cast<int32_t*>(p + l.rowcount);
  return m;
}

}  // namespace

size_t mask_workspace_bytes(int batch, int h, int w) { return layout_of(batch, h, w).total; }

hipError_t launch_mask_regions(float* img, int h, int w, int batch, double shadow_limit, double highlight_limit,
                               int min_area, bool float32_luminance, void* workspace, unsigned long long* count_dev,
                               hipStream_t st) {
  if (h < 1 || w < 1 || w > kMaxWidth || batch < 1 || batch > 65535 || (int64_t)h * ((w + 1) / 2) > INT32_MAX)
    return hipErrorInvalidValue;
  const MaskWork work = carve(workspace, batch, h, w);
  if (count_dev) {
    hipError_t e = hipMemsetAsync(count_dev, 0, (size_t)batch * sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
  }
  // (rows go to the x dimension of the grid, which has no 65535 limit)
  if (float32_luminance)
    hipLaunchKernelGGL(mask_runs_kernel<true>, dim3(h, batch), dim3(kRowThreads), 0, st, img, work, h, w, shadow_limit,
                       highlight_limit);
  else
    hipLaunchKernelGGL(mask_runs_kernel<false>, dim3(h, batch), dim3(kRowThreads), 0, st, img, work, h, w, shadow_limit,
                       highlight_limit);
  const unsigned row_blocks = (unsigned)(h < 2048 ? h : 2048);
  const size_t up_bytes = (size_t)work.cap * sizeof(uint32_t);
  const int stage = up_bytes <= 48 * 1024;
  hipLaunchKernelGGL(mask_merge_kernel, dim3(row_blocks, batch), dim3(256), stage ? up_bytes : 0, st, work, h, stage);
  const unsigned area_blocks = (unsigned)(h < 256 ? h : 256);
  hipLaunchKernelGGL(mask_area_kernel, dim3(area_blocks, batch), dim3(256), 0, st, work, h);
  hipLaunchKernelGGL(mask_apply_kernel, dim3(row_blocks, batch), dim3(256), 0, st, img, work, h, w, min_area,
                     count_dev);
  return hipGetLastError();
}

}  // namespace ss
