// Fused correction chain WITH the unsharp mask (stage 5) for gfx950.
//
// A 512-thread workgroup owns one 128 x 24 pixel tile at a time and walks vertical runs of tiles.
//   phase 1  stages 1-4, sRGB decode and the lightness Ln = L/100 for the tile's new rows plus r halo
//            columns on either side (lanes along x, so each wave-instruction is a contiguous run of
//            12-byte pixels).  Ln of every slot and the decoded (linear) RGB of the tile's own pixels
//            go to LDS.  Halo pixels are recomputed rather than exchanged: no intermediate plane in
//            HBM, no inter-workgroup dependency.
//   phase 2  row pass of the separable Gaussian (cv2 filters rows first), 4 outputs per thread from
//            ds_read_b128 windows, into a second LDS plane.
//   phase 3  column pass for the tile's pixels (pairs), mask logic, the Lab edit applied to the linear
//            pixel as a difference (see ss_device.h), encode, final encoding, quantisation, one 12-byte
//            store per lane for 16-bit output.
// Algorithmic HBM traffic: 12 B/px in + 12 / 6 / 3 B/px out.
// LDS per workgroup at r = 6: (36 x 140 + 36 x 128 + 3 x 24 x 128) x 4 B = 75,456 B -> two workgroups
// (16 waves) per CU.

#include "ss_internal.h"

namespace ss {

constexpr int USM_TW = 128;  // tile width: 64 lanes x 2 adjacent pixels
// SS_USM_MFMA: both blur passes on the matrix pipe, 16-row tiles (see the kernel); else the vector form, 24-row tiles
#ifdef SS_USM_MFMA
#define SS_USM_TH 16
constexpr bool USM_MFMA_BUILD = true;
#else
constexpr bool USM_MFMA_BUILD = false;
#endif
#ifndef SS_USM_TH
#define SS_USM_TH 24
#endif
constexpr int USM_TH = SS_USM_TH;  // tile height
#ifndef SS_USM_NT
#define SS_USM_NT 512
#endif
constexpr int USM_NT = SS_USM_NT;  // threads per workgroup
constexpr int USM_WAVES_PER_SIMD = 2 * USM_NT / 256;  // two workgroups per CU

#ifdef SS_ABL_NO_BARRIER  // timing experiment only (results are wrong without the barriers)
#define USM_SYNC() __builtin_amdgcn_s_waitcnt(0)
#else
#define USM_SYNC() __syncthreads()
#endif

// UNIT: the values are known to lie in [0, 1] (no clamp needed before the truncating conversion)
// SCALED (16-bit output): the values are the codes themselves, 0 ... 65535 with their fraction (finish_scaled16)
template <int OUT, bool UNIT, bool SCALED = false>
__device__ __forceinline__ void usm_store_pair(void* dst_img, int W, int H, int y, int x, const Px o0, const Px o1) {
  if (y >= H || x >= W) return;
  auto q16 = [](float v) -> uint32_t { return SCALED ? (uint32_t)v : (UNIT ? quant16_unit(v) : quant16(v)); };
  auto q8 = [](float v) -> uint32_t { return UNIT ? quant8_unit(v) : quant8(v); };
  const bool two = x + 1 < W;
  const size_t px = (size_t)y * W + x;
  if constexpr (OUT == SS_OUT_F32_LINEAR || OUT == SS_OUT_F32_SRGB) {
    F3* d = reinterpret_cast<F3*>(dst_img) + px;
    store_stream(d, F3{o0.r, o0.g, o0.b});
    if (two) store_stream(d + 1, F3{o1.r, o1.g, o1.b});
  } else if constexpr (OUT == SS_OUT_U16_SRGB) {
    char* d = reinterpret_cast<char*>(dst_img) + px * 6;
    if (two && ((px * 6) & 3) == 0) {
      U3 q;
      q.x = q16(o0.r) | (q16(o0.g) << 16);
      q.y = q16(o0.b) | (q16(o1.r) << 16);
      q.z = q16(o1.g) | (q16(o1.b) << 16);
      store_stream(reinterpret_cast<U3*>(d), q);
    } else {
      uint16_t* h = reinterpret_cast<uint16_t*>(d);
      h[0] = (uint16_t)q16(o0.r);
      h[1] = (uint16_t)q16(o0.g);
      h[2] = (uint16_t)q16(o0.b);
      if (two) {
        h[3] = (uint16_t)q16(o1.r);
        h[4] = (uint16_t)q16(o1.g);
        h[5] = (uint16_t)q16(o1.b);
      }
    }
  } else {
    uint8_t* d = reinterpret_cast<uint8_t*>(dst_img) + px * 3;
    d[0] = (uint8_t)q8(o0.r);
    d[1] = (uint8_t)q8(o0.g);
    d[2] = (uint8_t)q8(o0.b);
    if (two) {
      d[3] = (uint8_t)q8(o1.r);
      d[4] = (uint8_t)q8(o1.g);
      d[5] = (uint8_t)q8(o1.b);
    }
  }
}

// The same for a pair that is known to lie inside the image with its 12-byte slot 4-byte aligned (a tile that the
// image covers whole, in an image of even width): no bounds tests, no alignment test, a 32-bit byte offset on the
// image's base address (an image has at most 2^28 pixels).
template <int OUT, bool UNIT, bool SCALED = false>
__device__ __forceinline__ void usm_store_pair_inside(void* dst_img, uint32_t byte_offset, const Px o0, const Px o1) {
  auto q16 = [](float v) -> uint32_t { return SCALED ? (uint32_t)v : (UNIT ? quant16_unit(v) : quant16(v)); };
  auto q8 = [](float v) -> uint32_t { return UNIT ? quant8_unit(v) : quant8(v); };
  char* at = reinterpret_cast<char*>(dst_img) + byte_offset;
  if constexpr (OUT == SS_OUT_F32_LINEAR || OUT == SS_OUT_F32_SRGB) {
    F3* d = reinterpret_cast<F3*>(at);
    store_stream(d, F3{o0.r, o0.g, o0.b});
    store_stream(d + 1, F3{o1.r, o1.g, o1.b});
  } else if constexpr (OUT == SS_OUT_U16_SRGB) {
    U3 q;
    q.x = q16(o0.r) | (q16(o0.g) << 16);
    q.y = q16(o0.b) | (q16(o1.r) << 16);
    q.z = q16(o1.g) | (q16(o1.b) << 16);
    store_stream(reinterpret_cast<U3*>(at), q);
  } else {
    uint8_t* d = reinterpret_cast<uint8_t*>(at);
    d[0] = (uint8_t)q8(o0.r);
    d[1] = (uint8_t)q8(o0.g);
    d[2] = (uint8_t)q8(o0.b);
    d[3] = (uint8_t)q8(o1.r);
    d[4] = (uint8_t)q8(o1.g);
    d[5] = (uint8_t)q8(o1.b);
  }
}

// R_T : compile-time blur radius (3 <-> sigma 1, 6 <-> sigma 2), 0 = runtime radius P.usm_r.
// CCMC: CCM_LIN3 for the reference's 3-term correction, CCM_NONE for chains without one.
// CHROMA: 0 / 1 = the launcher knows the chroma boost (amount > 30) to be off / on, -1 = P.usm_chroma decides.
//
// Strip walk: the tile grid of every image is cut into vertical segments of `seg_tiles` tiles, one tile
// wide.  A workgroup walks a segment top to bottom, then takes the next free segment off a device-side
// ticket counter of its XCD (workgroups that share a CU do not progress at the same speed, so a fixed deal
// would leave the fast ones idle at the end).  Consecutive tiles of a segment share 2r rows of halo: after a
// tile's phase 3 the bottom 2r rows of both LDS planes are moved to the top (through registers, across
// the tile's closing barrier), so only the first tile of a segment computes its upper halo.  Any
// assignment of segments to workgroups is correct; the order only affects speed.
struct StripWalk {
  int tiles_x, tiles_y, seg_tiles, segs_y;
  uint32_t n_segments;  // whole batch
  uint32_t chunk;       // segments per XCD: XCD k owns segments [k * chunk, (k + 1) * chunk)
  uint32_t* ticket;     // [8] segments already handed out of each XCD's chunk; the launcher clears them
};

#ifdef SS_USM_TRACE  // timing experiment: per-workgroup start / end clock and tile count
__device__ unsigned long long ss_usm_trace_buf[1024][4];
extern "C" int ss_debug_usm_trace(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ss_usm_trace_buf), sizeof(ss_usm_trace_buf));
}
#endif

template <int R_T, int CCMC, int OUT, bool PLAIN, int CHROMA = -1>
__global__ __launch_bounds__(USM_NT, USM_WAVES_PER_SIMD) void usm_kernel(const ChainParams P, const StripWalk T_arg,
                                                     const float* __restrict__ src, void* __restrict__ dst) {
#ifdef SS_USM_WALK_IN_REGISTERS
  constexpr bool PREFETCH_WALK = false;
#else
  constexpr bool PREFETCH_WALK = true;
#endif
  constexpr int TW = USM_TW, TH = USM_TH, NT = USM_NT;
  extern __shared__ __attribute__((aligned(16))) float S[];
  __shared__ float4 M[ccm_needs_table<CCMC> ? SS_MAX_CCM_TERMS : 1];
  __shared__ uint32_t next_seg;
  if constexpr (ccm_needs_table<CCMC>) {
    stage_ccm(P, M, threadIdx.x, NT);
    __syncthreads();
  }

#ifdef SS_USM_TRACE
  const unsigned long long trace_t0 = wall_clock64();
  unsigned trace_tiles = 0;
#endif
  const int r = R_T ? R_T : P.usm_r;
  const int PWv = TW + 2 * r;     // slots per LDS row that hold data
  const int PW = (PWv + 3) & ~3;  // row pitch in floats (rows stay 16-byte aligned)
  const int PH = TH + 2 * r;
  float* __restrict__ Lt = S;             // [PH][PW]  Ln of tile + halo
  float* __restrict__ Ht = Lt + PH * PW;  // [PH][TW]  row-filtered Ln
  float* __restrict__ C0 = Ht + PH * TW;  // [3][TH][TW] the tile's pixels: linear R, G, B (L, a, b for Lab data)
  constexpr int CP = TH * TW;             // plane pitch
  const int W = P.width, H = P.height;
  const int tid = threadIdx.x;
  // Both passes of the separable blur on the matrix pipe (v_mfma_f32_16x16x4_f32: float32 in, float32 accumulate, one
  // fused multiply-add per term in k order, i.e. the fmaf chain of the vector form bit for bit), which issues
  // beside the vector ALU that limits this kernel.  A 16 x 16 block of outputs is the product of a 16 x (16 + 2r)
  // band matrix of taps with a (16 + 2r) x 16 block of the plane; 16-row tiles give every wave one block per pass.
  constexpr bool MFMA = USM_MFMA_BUILD && R_T != 0 && !ccm_needs_table<CCMC> && TH == 16 && NT == 512 && TW == 128;
  constexpr int KS = (16 + 2 * (R_T ? R_T : 1) + 3) / 4;  // k steps of 4 per block
  typedef float v4f __attribute__((ext_vector_type(4)));
  float band[KS];  // lane l holds w(4 s + l / 16 - l % 16): the band matrix as the A (row pass) / B (column pass) operand
  if constexpr (MFMA) {
    const float* taps = kernarg_params().taps;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int j = 4 * s + ((tid & 63) >> 4) - (tid & 15);
      band[s] = (j >= 0 && j <= 2 * R_T) ? taps[j <= R_T ? j : 2 * R_T - j] : 0.0f;
    }
    // zero-weight terms still multiply what they read: every word of the planes must be a finite number from the
    // start (padding columns are never written afterwards)
    const int n_lds = (TH + 2 * R_T) * (((TW + 2 * R_T + 3) & ~3) + TW) + 3 * TH * TW;
    for (int i = tid; i < n_lds; i += NT) S[i] = 0.0f;
    __syncthreads();
  }
  // Rows every tile computes itself: LDS rows [2r, PH) ("new rows", TH of them: image rows y0+r ... y0+TH+r-1).
  // Thread -> slot map of the new rows, affine in the slot number k so that no per-slot index lives in a
  // register: MAIN slots cover the TW tile columns (row 4k + tid/128, column tid%128 at NT = 512), one SIDE
  // slot per thread covers the 2r halo columns.
  const int n_new = TH * PWv;
  constexpr int ROWS_PER_PASS = NT / TW;        // 4
  constexpr int MAIN = TH / ROWS_PER_PASS;      // 8 main slots per thread
  static_assert(NT % TW == 0 && TH % (NT / TW) == 0, "tile shape must divide the workgroup");
  static_assert(2 * (R_T ? R_T : 1) * TH <= NT, "one side slot per thread at most");
  // The slot coordinates are cheap to derive and expensive to keep: every phase derives its own from an opaque
  // copy of the thread index, so that the compiler cannot hoist them (and the LDS addresses built on them)
  // out of the tile loop into registers that would then spill.
  auto fresh_tid = [&]() -> int {
    int t = tid;
    asm volatile("" : "+v"(t));
    return t;
  };
  struct Lane {
    int m_row, m_col;  // main slots: new row 4k + m_row, tile column m_col
    int s_row, s_lx;   // side slot: new row s_row, LDS column s_lx
    bool s_on;
  };
  auto lane_of = [&](int t) -> Lane {
    Lane l;
    l.m_row = t / TW;
    l.m_col = t - l.m_row * TW;
    l.s_row = t / (2 * r);
    const int s_j = t - l.s_row * (2 * r);
    l.s_lx = s_j < r ? s_j : s_j + TW;
    l.s_on = t < 2 * r * TH;
    return l;
  };
  // The parameter block as the kernel argument segment holds it, behind a pointer the compiler cannot see
  // through: scalars read through it are loaded (scalar loads) in the phase that uses them instead of all
  // living in scalar registers, spilled to vector lanes, across the whole walk.  (The parameter block is the
  // kernel's first argument: offset 0 of the argument segment.)
  auto params_now = [&]() -> const ChainParams& {
    if constexpr (!PLAIN) {  // (the plain chain uses few enough parameters to keep them: measured, no gain there)
      typedef const __attribute__((address_space(4))) ChainParams* KernargParams;
      KernargParams k = (KernargParams)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(k));
      return *(const ChainParams*)k;
    }
    return P;
  };
  // float4 moves per thread for the carried rows: 2r rows of Lt (PW floats) and of Ht (TW floats)
  constexpr int CARRY_MAX = (2 * SS_MAX_USM_RADIUS * ((USM_TW + 2 * SS_MAX_USM_RADIUS + 3) / 4 + USM_TW / 4) + NT - 1) / NT;
  constexpr int CARRY = R_T ? (2 * R_T * ((TW + 2 * R_T + 3) / 4 + TW / 4) + NT - 1) / NT : CARRY_MAX;
  const int carry_lt = 2 * r * (PW / 4), carry_n = carry_lt + 2 * r * (TW / 4);
  // ---- the walk: tile positions of this workgroup, flattened so that the pixel loads of the
  // NEXT tile can be issued while the current tile is still in its phases 2 and 3 ----
  struct TilePos {
    uint32_t seg;
    int img, x0, ty, ty_end;
    bool valid, first;
  };
  // The walk's description as the kernel-argument segment holds it (the second argument, behind the parameter
  // block): it is read where a segment is drawn - once per eight tiles - and its seven scalars need not occupy
  // registers in between (the scalar file is full: every value kept there pushes another into vector lanes).
  auto walk_now = [&]() -> const StripWalk& {
    if constexpr (PREFETCH_WALK) {
      typedef const __attribute__((address_space(4))) char* KernargBytes;
      KernargBytes k = (KernargBytes)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(k));
      constexpr size_t at = (sizeof(ChainParams) + alignof(StripWalk) - 1) / alignof(StripWalk) * alignof(StripWalk);
      return *(const StripWalk*)(const char*)(k + at);
    }
    return T_arg;
  };
  auto seg_pos = [&](uint32_t seg) -> TilePos {
    const StripWalk& T = walk_now();
    TilePos t;
    t.seg = seg;
    t.valid = seg < T.n_segments;
    const int per_img = T.tiles_x * T.segs_y;
    t.img = (int)(seg / per_img);
    const int s_in = (int)(seg - (uint32_t)t.img * per_img);
    const int sy = s_in / T.tiles_x;
    t.x0 = (s_in - sy * T.tiles_x) * TW;
    t.ty = sy * T.seg_tiles;
    t.ty_end = t.ty + T.seg_tiles < T.tiles_y ? t.ty + T.seg_tiles : T.tiles_y;
    t.first = true;
    return t;
  };
  // source pixel of LDS slot (ly, lx) of the tile at (x0, y0), as a BYTE offset into the image (an image has at
  // most 2^28 pixels of 12 bytes: 32 bits hold it, and the load takes the image's base address as a scalar)
  auto slot_src = [&](int x0, int y0, bool interior, int ly, int lx) -> uint32_t {
    int py = y0 - r + ly, px = x0 - r + lx;
    if (!interior) {
      py = reflect_coord(py, H);
      px = reflect_coord(px, W);
    }
    return ((uint32_t)py * W + px) * 12u;
  };
  auto pixel_at = [](const F3* image, uint32_t byte_offset) -> F3 {
    return *reinterpret_cast<const F3*>(reinterpret_cast<const char*>(image) + byte_offset);
  };
  auto is_interior = [&](int x0, int y0) -> bool {
    // tiles whose halo stays inside the image skip the border arithmetic
    return x0 >= r && y0 >= r && x0 + TW + r <= W && y0 + TH + r <= H;
  };
  // Fast path: compile-time radius and a colour correction small enough to unroll.  The other kernels walk
  // their slots with plain loops and never carry rows between tiles (the launcher clears StripWalk::carry for them).
  constexpr bool PREFETCH = R_T != 0 && !ccm_needs_table<CCMC>;
  F3 v[MAIN + 1];
  // issue every pixel load of a tile's new rows
  auto issue_loads = [&](const TilePos& t) {
    if constexpr (PREFETCH) {
      const int y0 = t.ty * TH;
      const bool interior = is_interior(t.x0, y0);
      const F3* __restrict__ s = reinterpret_cast<const F3*>(src + (size_t)t.img * P.n_px * 3);
#ifdef SS_ABL_NO_LOAD
#pragma unroll
      for (int k = 0; k <= MAIN; ++k) v[k] = F3{(float)(tid + k) * 1e-4f, (float)k * 1e-2f, 0.5f};
#else
      const Lane l = lane_of(fresh_tid());
      if (interior) {
        // no reflection: a lane's pixels lie ROWS_PER_PASS rows apart, one multiplication serves them all
        const uint32_t first = slot_src(t.x0, y0, true, 2 * r + l.m_row, r + l.m_col);
        const uint32_t step = (uint32_t)ROWS_PER_PASS * (uint32_t)W * 12u;
#pragma unroll
        for (int k = 0; k < MAIN; ++k) v[k] = pixel_at(s, first + (uint32_t)k * step);
      } else {
#pragma unroll
        for (int k = 0; k < MAIN; ++k)
          v[k] = pixel_at(s, slot_src(t.x0, y0, false, 2 * r + ROWS_PER_PASS * k + l.m_row, r + l.m_col));
      }
      if (l.s_on) v[MAIN] = pixel_at(s, slot_src(t.x0, y0, interior, 2 * r + l.s_row, l.s_lx));
#endif
    }
  };

  // The loop is rotated by one phase: an iteration issues the pixel loads of tile i+1 (`pend`), runs phases
  // 2 and 3 of tile i (`cur`) while they are in flight, then phase 1 of tile i+1.  The loaded pixels are
  // produced and consumed inside one iteration, so no register copy (and no wait on the loads) is needed
  // at the loop edge.
  constexpr int DEFER = PREFETCH ? (R_T + ROWS_PER_PASS - 1) / ROWS_PER_PASS + 1 : 1;
  F3 late_c[DEFER];
  int cur_img = -1;
  ImgParams ip;
  // Segments are dealt per XCD: each of the 8 XCDs has its own L2, so the workgroups of one XCD (workgroup i
  // runs on XCD i % 8) work through one contiguous range of segments - neighbouring tile columns, whose halo
  // columns overlap, and the segment rows below each other, whose halo rows do - and meet each other's halo
  // pixels in their own L2 instead of fetching them from memory once per XCD.  A workgroup whose XCD's range is
  // used up helps out in the next XCD's range, so the tail stays as short as with a single counter.  If the
  // i % 8 rule ever did not hold, only the locality would be lost.
  uint32_t probe = 0;  // ranges, counting from this workgroup's own, found exhausted so far (thread 0 only)
  auto draw = [&]() -> uint32_t {
    const StripWalk& T = walk_now();
    const uint32_t home = blockIdx.x & 7u;
    while (probe < 8) {
      const uint32_t k = (home + probe) & 7u;
      const uint32_t begin = k * T.chunk;
      if (begin < T.n_segments) {
        const uint32_t left = T.n_segments - begin;
        const uint32_t t = atomicAdd(&T.ticket[k], 1u);
        if (t < (left < T.chunk ? left : T.chunk)) return begin + t;
      }
      ++probe;
    }
    return T.n_segments;  // nothing left anywhere: seg_pos() reports it as invalid
  };
  if (tid == 0) next_seg = draw();
  __syncthreads();
  const uint32_t first_seg = next_seg;
  __syncthreads();  // (next_seg is written again at the end of the first tile's phase 1)
  TilePos cur, pend = seg_pos(first_seg);
  cur.valid = false;
  for (;;) {
    if (pend.valid) issue_loads(pend);
    if (cur.valid) {
      const int x0 = cur.x0, y0 = cur.ty * TH;
      const bool first = cur.first;
      const bool more = pend.valid && !pend.first;  // the next tile continues this segment
      const ChainParams& P = params_now();
      char* d = reinterpret_cast<char*>(dst) + (size_t)cur.img * P.n_px * 3 * OutTraits<OUT>::BYTES;

      // ---- phase 2: row pass over the rows phase 1 has just written ----
      const int row_begin = first ? 0 : 2 * r;
      const int n_items = MFMA ? 0 : (PH - row_begin) * (TW / 4);
      if constexpr (MFMA) {
        // wave w: columns [16 w, 16 w + 16) of the new rows [2r, 2r + 16); a first tile also of rows [0, 16)
        const int t2 = fresh_tid();
        const int lane = t2 & 63, c0 = (t2 >> 6) * 16;
        for (int rb = first ? 0 : 2 * r; rb <= 2 * r; rb += 2 * r) {
          const float* bp = Lt + (rb + (lane & 15)) * PW + c0 + (lane >> 4);
          v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int s2 = 0; s2 < KS; ++s2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(band[s2], bp[4 * s2], acc, 0, 0, 0);
          *reinterpret_cast<float4*>(Ht + (rb + (lane & 15)) * TW + c0 + 4 * (lane >> 4)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
      }
#ifdef SS_ABL_NO_P2
      for (int item = tid; item < 0; item += NT) {
#else
      for (int item = fresh_tid(); item < n_items; item += NT) {
#endif
        const int row = row_begin + (item >> 5), q = item & 31;
        const float* base = Lt + row * PW + 4 * q;
        float4 acc;
        if constexpr (R_T != 0) {
          constexpr int NV = (4 + 2 * R_T + 3) / 4;
          float v[NV * 4];
#pragma unroll
          for (int k = 0; k < NV; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(base + 4 * k);
            v[4 * k + 0] = t.x;
            v[4 * k + 1] = t.y;
            v[4 * k + 2] = t.z;
            v[4 * k + 3] = t.w;
          }
          acc = make_float4(v[0] * P.taps[0], v[1] * P.taps[0], v[2] * P.taps[0], v[3] * P.taps[0]);
#pragma unroll
          for (int j = 1; j <= 2 * R_T; ++j) {
            const float t = tap_at<R_T>(P, j);
            acc.x = fmaf(v[j + 0], t, acc.x);
            acc.y = fmaf(v[j + 1], t, acc.y);
            acc.z = fmaf(v[j + 2], t, acc.z);
            acc.w = fmaf(v[j + 3], t, acc.w);
          }
        } else {
          float w0 = base[0], w1 = base[1], w2 = base[2], w3 = base[3];
          const float t0 = P.taps[0];
          acc = make_float4(w0 * t0, w1 * t0, w2 * t0, w3 * t0);
          for (int j = 1; j <= 2 * r; ++j) {
            w0 = w1;
            w1 = w2;
            w2 = w3;
            w3 = base[j + 3];
            const float t = P.taps[j <= r ? j : 2 * r - j];
            acc.x = fmaf(w0, t, acc.x);
            acc.y = fmaf(w1, t, acc.y);
            acc.z = fmaf(w2, t, acc.z);
            acc.w = fmaf(w3, t, acc.w);
          }
        }
        *reinterpret_cast<float4*>(Ht + row * TW + 4 * q) = acc;
      }
      USM_SYNC();

      // the image covers the whole tile and every pair's slot is 4-byte aligned: stores without per-lane tests
      // (a call with a second output takes the general path throughout: one scalar flag decides both)
      const bool inside = x0 + TW <= W && y0 + TH <= H && (W & 1) == 0 && kernarg_params().dst2 == nullptr;
      // ---- phase 3: column pass, mask logic, Lab -> sRGB, encode, store ----
      // one pixel pair at a time: everything after the blur
      // (`inside_at`: byte offset of the pair's slot in the output image, used for tiles the image covers whole)
      constexpr uint32_t PX_BYTES = 3u * OutTraits<OUT>::BYTES;
      // 16-bit output of the plain chain: the quantiser's factor 65535 is folded into the constants of the output encoding
      constexpr bool SCALED16 = PLAIN && OUT == SS_OUT_U16_SRGB;
      auto slot_at = [&](int ly, int px) -> uint32_t {
        return ((uint32_t)(y0 + ly) * (uint32_t)W + (uint32_t)(x0 + px)) * PX_BYTES;
      };
      auto pair_out = [&](int ly, int px, v2f blur, uint32_t inside_at) {
        const float2 lc = *reinterpret_cast<const float2*>(Lt + (ly + r) * PW + r + px);
        const float* cq = C0 + ly * TW + px;
        const float2 c0 = *reinterpret_cast<const float2*>(cq), c1 = *reinterpret_cast<const float2*>(cq + CP),
                     c2 = *reinterpret_cast<const float2*>(cq + 2 * CP);
        const v2f Ln = v2f{lc.x, lc.y};
#ifdef SS_ABL_NO_P3  // timing experiment: phase 3 without its arithmetic
        Px2 o = Px2{blur + v2f{c0.x, c0.y}, Ln + v2f{c1.x, c1.y}, v2f{c2.x, c2.y}};
#else
        Px2 o;
        RiskT<v2f> risk{};
        if (!PLAIN && P.lab_io) {
          const Lab2 q = usm_on_lab<PLAIN, v2f>(Ln, v2f{c1.x, c1.y}, v2f{c2.x, c2.y}, blur, P);
          o = Px2{q.L, q.a, q.b};
        } else if constexpr (PLAIN) {
          o = usm_on_rgb<PLAIN, v2f, CHROMA>(Px2{v2f{c0.x, c0.y}, v2f{c1.x, c1.y}, v2f{c2.x, c2.y}}, Ln, blur, P, &risk,
                                             RiskCfg<OUT>::ENABLED ? RiskCfg<OUT>::MIN_MX : 0.0f);
        } else {
          // phase 1 hands the size of a shadows/highlights edit over in the signs of the channel values (which
          // that stage has clipped to [0, 1]; without the stage a value may be negative in its own right)
          Px2 lin{v2f{c0.x, c0.y}, v2f{c1.x, c1.y}, v2f{c2.x, c2.y}};
          v2f carried = v2f{0.0f, 0.0f};
          if (P.sh) {
            carried = v2f{risk_carried(c0.x, c1.x, c2.x, P.sh_risk_bound), risk_carried(c0.y, c1.y, c2.y, P.sh_risk_bound)};
            lin = Px2{v2f{fabsf(c0.x), fabsf(c0.y)}, v2f{fabsf(c1.x), fabsf(c1.y)}, v2f{fabsf(c2.x), fabsf(c2.y)}};
          }
          o = usm_on_rgb<PLAIN, v2f>(lin, Ln, blur, P, &risk);
          risk.mx = risk.mx + carried;
        }
        if (!inside) {  // the second output, when the call has one
          const ChainParams& Pk = kernarg_params();
          if (Pk.dst2 != nullptr) {
            const int Wk = Pk.width, gy = y0 + ly, gx = x0 + px;
            const size_t at = ((size_t)cur.img * Pk.height + gy) * Wk + gx;
            const bool enc = Pk.dst2_encoded != 0;
            if (gy < Pk.height && gx < Wk) store_second(Pk.dst2, at, px_lo(o), false, enc);
            if (gy < Pk.height && gx + 1 < Wk) store_second(Pk.dst2, at + 1, px_hi(o), false, enc);
          }
        }
        // the sharpened pixel is in [0,1]; the LAB branch has no final clip: either way the clip stage is a no-op here
        if constexpr (SCALED16)
          o = finish_scaled16<v2f>(o);
        else
          o = finish<OUT, true, v2f>(o, P);
        // Pixels whose Lab edits float32 cannot follow (see RiskCfg) are recorded for the float64 kernel, their
        // blurred lightness waiting in the first four bytes of their own output slot.  Almost never taken.
        if constexpr (RiskCfg<OUT>::ENABLED) {
          const bool big0 = risk_big<OUT>(risk.mx.x), big1 = risk_big<OUT>(risk.mx.y);
          // (risk.any: wave-uniform, false when the edit's own pre-test has ruled a large edit out for the whole wave)
          if (risk.any && __ballot(big0 || big1) != 0ull) {  // a scalar branch: the whole wave skips it
            const ChainParams& Pr = kernarg_params();  // (nothing this branch needs stays in registers outside it)
            const int Wr = Pr.width, Hr = Pr.height;
            const int gy = y0 + ly, gx = x0 + px;
            const bool f0 = big0 && risk_cancelled<OUT>(risk.mx.x, risk.o.r.x, risk.o.g.x, risk.o.b.x) && gy < Hr && gx < Wr;
            const bool f1 = big1 && risk_cancelled<OUT>(risk.mx.y, risk.o.r.y, risk.o.g.y, risk.o.b.y) && gy < Hr && gx + 1 < Wr;
            // what the slot's first two values must be for the stored bytes to be the float's bits
            auto stash = [](float Lb, float r, float g) -> float2 {
              if constexpr (OUT == SS_OUT_U16_SRGB) {  // two 16-bit codes
                const uint32_t u = __float_as_uint(Lb);
                if constexpr (SCALED16) return float2{(float)(u & 0xFFFFu) + 0.5f, (float)(u >> 16) + 0.5f};
                return float2{((float)(u & 0xFFFFu) + 0.5f) * (1.0f / 65535.0f), ((float)(u >> 16) + 0.5f) * (1.0f / 65535.0f)};
              } else {
                return float2{Lb, g};
              }
            };
            if (f0) {
              const float2 k = stash(blur.x, o.r.x, o.g.x);
              o.r.x = k.x, o.g.x = k.y;
            }
            if (f1) {
              const float2 k = stash(blur.y, o.r.y, o.g.y);
              o.r.y = k.x, o.g.y = k.y;
            }
            // The record was cleared by the launcher.  A wave's 64 lanes hold the 64 pixel pairs of one row of the
            // tile (lane l: pixels 2 l and 2 l + 1), which is exactly the row's two words of the record: one lane
            // stores the wave's two ballots, no atomics.  (On an input where an eighth of the pixels is recorded an
            // atomic per pixel nearly doubled this kernel's time.)
            const size_t w = (((size_t)cur.img * Hr + gy) * ((Wr + TW - 1) / TW) + x0 / TW) * 2;
            if constexpr (MFMA) {  // (that build deals rows to lanes differently)
              if (f0) atomicOr(Pr.risk_bits + w, 1ull << (px >> 1));
              if (f1) atomicOr(Pr.risk_bits + w + 1, 1ull << (px >> 1));
              if (f0 || f1) *Pr.risk_any = 1u;
            } else {
              const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
              if ((m0 | m1) != 0ull && (px >> 1) == 0) {
                if (m0 != 0ull) Pr.risk_bits[w] = m0;
                if (m1 != 0ull) Pr.risk_bits[w + 1] = m1;
                *Pr.risk_any = 1u;
              }
            }
          }
        }
#endif
#ifdef SS_ABL_NO_STORE
        asm volatile("" ::"v"(o.r.x), "v"(o.r.y), "v"(o.g.x), "v"(o.g.y), "v"(o.b.x), "v"(o.b.y));
#else
        if (inside)
          usm_store_pair_inside<OUT, PLAIN, SCALED16>(d, inside_at, px_lo(o), px_hi(o));
        else
          usm_store_pair<OUT, PLAIN, SCALED16>(d, W, H, y0 + ly, x0 + px, px_lo(o), px_hi(o));
#endif
      };
      if constexpr (MFMA) {
        // wave w: the 16 x 16 block of columns [16 w, 16 w + 16); lane l ends up with the blurred values of row
        // l % 16, columns 16 w + 4 (l / 16) ... + 3: two pixel pairs
        const int t3 = fresh_tid();
        const int lane = t3 & 63, c0 = (t3 >> 6) * 16;
        const float* ap = Ht + (lane >> 4) * TW + c0 + (lane & 15);
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s2 * TW], band[s2], acc, 0, 0, 0);
        pair_out(lane & 15, c0 + 4 * (lane >> 4), v2f{acc[0], acc[1]}, slot_at(lane & 15, c0 + 4 * (lane >> 4)));
        __builtin_amdgcn_sched_barrier(0);
        pair_out(lane & 15, c0 + 4 * (lane >> 4) + 2, v2f{acc[2], acc[3]}, slot_at(lane & 15, c0 + 4 * (lane >> 4) + 2));
      } else if constexpr (R_T != 0) {
        // a thread owns one pixel-pair column and RUN consecutive rows: the RUN + 2r row-filtered values it
        // needs are read once, all reads in flight together, and every output reuses 2r of them
        constexpr int RUN = TH / (NT / (TW / 2));
        static_assert(RUN * (NT / (TW / 2)) == TH, "runs must tile the rows");
        const int t3 = fresh_tid();
        const int px = (t3 & (TW / 2 - 1)) * 2, ly0 = (t3 / (TW / 2)) * RUN;
        // rows are read in batches (all reads of a batch in flight together) and folded at once into every
        // output they belong to, so only the batch and the RUN accumulators occupy registers
        constexpr int ROWS = RUN + 2 * R_T, BATCH = 8;
        v2f blur[RUN];
#pragma unroll
        for (int j0 = 0; j0 < ROWS; j0 += BATCH) {
          v2f win[BATCH];
#pragma unroll
          for (int j = j0; j < j0 + BATCH && j < ROWS; ++j) {
            const float2 t = *reinterpret_cast<const float2*>(Ht + (ly0 + j) * TW + px);
            win[j - j0] = v2f{t.x, t.y};
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = j0; j < j0 + BATCH && j < ROWS; ++j) {
#pragma unroll
            for (int q = 0; q < RUN; ++q) {
              const int tap = j - q;  // row j is tap `tap` of output q
              if (tap == 0) blur[q] = win[j - j0] * P.taps[0];
              if (tap > 0 && tap <= 2 * R_T) blur[q] = mad<v2f>(win[j - j0], tap_at<R_T>(P, tap), blur[q]);
            }
          }
        }
        // the run's output slots lie one image row apart: one multiplication for the first, additions after it
        uint32_t at = slot_at(ly0, px);
        const uint32_t row_bytes = (uint32_t)W * PX_BYTES;
#pragma unroll
        for (int q = 0; q < RUN; ++q) {
          pair_out(ly0 + q, px, blur[q], at);
          at += row_bytes;
          __builtin_amdgcn_sched_barrier(0);  // one pair's arithmetic at a time keeps the register count down
        }
      } else {
        for (int pair = tid; pair < TH * TW / 2; pair += NT) {
          const int ly = pair >> 6, px = (pair & 63) * 2;
          const float* col = Ht + ly * TW + px;
          float2 t = *reinterpret_cast<const float2*>(col);
          v2f blur = v2f{t.x, t.y} * P.taps[0];
          for (int j = 1; j <= 2 * r; ++j) {
            t = *reinterpret_cast<const float2*>(col + j * TW);
            blur = mad<v2f>(v2f{t.x, t.y}, P.taps[j <= r ? j : 2 * r - j], blur);
          }
          pair_out(ly, px, blur, slot_at(ly, px));
        }
      }

      // ---- carry the bottom 2r rows of both planes to the top for the next tile of the column ----
      // read before the barrier (phase 3 only reads the planes), write after it (nobody reads any more)
      // (every thread reads: lanes beyond the carried range re-read its last vector and write nothing)
      static_assert(CARRY <= 2 || !PREFETCH, "two vectors per thread cover the compile-time radii");
      // float offsets into S of the vectors this thread moves; derived from an opaque copy of the thread
      // index so that they are recomputed per tile instead of occupying registers across the whole loop
      int carry_from[CARRY], carry_to[CARRY];
      {
        int t_op = tid;
        asm volatile("" : "+v"(t_op));
#pragma unroll
        for (int k = 0; k < CARRY; ++k) {
          int i = t_op + k * NT;
          if (i >= carry_n) i = carry_n - 1;
          const bool in_lt = i < carry_lt;
          carry_to[k] = in_lt ? 4 * i : PH * PW + 4 * (i - carry_lt);
          carry_from[k] = carry_to[k] + (in_lt ? TH * PW : TH * TW);
        }
      }
      float4 carry0, carry1;
      if constexpr (PREFETCH) {
        carry0 = *reinterpret_cast<const float4*>(S + carry_from[0]);
        if constexpr (CARRY > 1) carry1 = *reinterpret_cast<const float4*>(S + carry_from[1]);
      }
      USM_SYNC();  // phase 3 is done with the planes
      if (PREFETCH && more) {
        if (tid < carry_n) *reinterpret_cast<float4*>(S + carry_to[0]) = carry0;
        if constexpr (CARRY > 1) {
          if (tid + NT < carry_n) *reinterpret_cast<float4*>(S + carry_to[1]) = carry1;
        }
        if constexpr (PREFETCH) {
          const Lane l = lane_of(fresh_tid());
#pragma unroll
          for (int k = MAIN - DEFER; k < MAIN; ++k) {
            const int c_row = r + ROWS_PER_PASS * k + l.m_row - TH;  // row in the next tile's planes
            if (c_row >= 0) {
              float* q = C0 + c_row * TW + l.m_col;
              q[0] = late_c[k - (MAIN - DEFER)].x;
              q[CP] = late_c[k - (MAIN - DEFER)].y;
              q[2 * CP] = late_c[k - (MAIN - DEFER)].z;
            }
          }
        }
        // the next tile's phase 1 writes Lt rows [2r, PH) and plane rows [r, TH): disjoint from what is written
        // here, and its own barrier orders these writes before any read
      }
    }
    if (!pend.valid) break;
    cur = pend;
    {
      // the last tile of a segment draws the workgroup's next segment; the ticket travels through LDS at
      // the barrier that closes phase 1
      const ChainParams& P = params_now();
      const bool last_of_seg = cur.ty + 1 >= cur.ty_end;
      uint32_t drawn = 0;
      if (last_of_seg && tid == 0) drawn = draw();
      const int img = cur.img, x0 = cur.x0, y0 = cur.ty * TH;
      if (img != cur_img) {
        ip = load_img_params(P, img);
        cur_img = img;
      }
      const F3* __restrict__ s = reinterpret_cast<const F3*>(src + (size_t)img * P.n_px * 3);
      const bool first = cur.first;
      const bool interior = is_interior(x0, y0);

      // ---- phase 1: stages 1-4, decode, Ln to LDS for every slot, the pixel itself for the tile's own ----
      struct Slot {
        float Ln;
        F3 c;  // what phase 3 needs of the pixel: linear RGB, or (L, a, b) when the data is Lab
      };
      auto slot_eval = [&](const F3 px) -> Slot {
#ifdef SS_ABL_NO_P1  // timing experiment: phase 1 without its arithmetic
        return Slot{px.x, px};
#else
        RiskT<float> risk{};
        const Px lin = front_decoded<CCMC, PLAIN, float>(Px{px.x, px.y, px.z}, P, ip, M, PLAIN ? nullptr : &risk);
        if (!PLAIN && P.lab_io) return Slot{lin.r * 0.01f, F3{lin.r, lin.g, lin.b}};
        const float r = lin.r, g = lin.g, b = lin.b;
        const float fy = lab_f(fmaf(kLab.fwd[3], r, fmaf(kLab.fwd[4], g, b * kLab.fwd[5])));
        // the size of a shadows/highlights edit travels to phase 3 in the signs of the (non-negative) values
        if (!PLAIN && P.sh) {
          const uint32_t l = risk_level(risk.mx);
          return Slot{fmaf(fy, 1.16f, -0.16f), F3{risk_tag(r, l, 0), risk_tag(g, l, 1), risk_tag(b, l, 2)}};
        }
        return Slot{fmaf(fy, 1.16f, -0.16f), F3{r, g, b}};  // Ln = (116 fy - 16) / 100
#endif
      };
      // two slots in lock step (packed arithmetic)
      auto slot_eval2 = [&](const F3 pa, const F3 pb, Slot& qa, Slot& qb) {
#ifdef SS_ABL_NO_P1
        qa = Slot{pa.x, pa};
        qb = Slot{pb.x, pb};
#else
        RiskT<v2f> risk{};
        const Px2 p = front_decoded<CCMC, PLAIN, v2f>(Px2{v2f{pa.x, pb.x}, v2f{pa.y, pb.y}, v2f{pa.z, pb.z}}, P, ip, M,
                                                      PLAIN ? nullptr : &risk);
        if (!PLAIN && P.lab_io) {
          qa = Slot{p.r.x * 0.01f, F3{p.r.x, p.g.x, p.b.x}};
          qb = Slot{p.r.y * 0.01f, F3{p.r.y, p.g.y, p.b.y}};
          return;
        }
        const v2f r = p.r, g = p.g, b = p.b;
        const v2f fy = lab_f(mad<v2f>(kLab.fwd[3], r, mad<v2f>(kLab.fwd[4], g, b * kLab.fwd[5])));
        const v2f Ln = mad<v2f>(fy, 1.16f, -0.16f);
        if (!PLAIN && P.sh) {
          const uint32_t la = risk_level(risk.mx.x), lb = risk_level(risk.mx.y);
          qa = Slot{Ln.x, F3{risk_tag(r.x, la, 0), risk_tag(g.x, la, 1), risk_tag(b.x, la, 2)}};
          qb = Slot{Ln.y, F3{risk_tag(r.y, lb, 0), risk_tag(g.y, lb, 1), risk_tag(b.y, lb, 2)}};
          return;
        }
        qa = Slot{Ln.x, F3{r.x, g.x, b.x}};
        qb = Slot{Ln.y, F3{r.y, g.y, b.y}};
#endif
      };
      auto plane_write = [&](int cy, int cx, const F3 c) {
        float* q = C0 + cy * TW + cx;
        q[0] = c.x;
        q[CP] = c.y;
        q[2 * CP] = c.z;
      };
      auto slot_compute = [&](const F3 px, int ly, int lx) {
        const int cy = ly - r, cx = lx - r;
        const bool own = (unsigned)cy < (unsigned)TH && (unsigned)cx < (unsigned)TW;
        const Slot q = slot_eval(px);
        Lt[ly * PW + lx] = q.Ln;
        if (own) plane_write(cy, cx, q.c);
      };
      if (first) {  // upper halo rows [0, 2r): only a first tile has to compute them
        for (int idx = tid; idx < 2 * r * PWv; idx += NT) {
          const int ly = idx / PWv, lx = idx - ly * PWv;
          slot_compute(pixel_at(s, slot_src(x0, y0, interior, ly, lx)), ly, lx);
        }
      }
      // the pixels of the last r new rows belong to the NEXT tile's first r plane rows: they wait in
      // registers until this tile's phase 3 is done with the planes
      if constexpr (PREFETCH) {
        const Lane l = lane_of(fresh_tid());
        static_assert(MAIN % 2 == 0, "main slots are evaluated in pairs");
        Slot qs[MAIN];
        // pairs pay off when shadows/highlights etc. are in the chain; the plain chain is too short for the
        // packing moves to amortise (measured: 218 vs 210 VALU instructions per pixel)
        if constexpr (PLAIN) {
#pragma unroll
          for (int k = 0; k < MAIN; ++k) qs[k] = slot_eval(v[k]);
        } else {
#pragma unroll
          for (int k = 0; k < MAIN; k += 2) slot_eval2(v[k], v[k + 1], qs[k], qs[k + 1]);
        }
#pragma unroll
        for (int k = 0; k < MAIN; ++k) {
          const int ln = ROWS_PER_PASS * k + l.m_row;  // new-row number, plane row r + ln
          const Slot q = qs[k];
          Lt[(2 * r + ln) * PW + r + l.m_col] = q.Ln;
          if (k < MAIN - DEFER) {
            plane_write(r + ln, l.m_col, q.c);
          } else {
            if (r + ln < TH) plane_write(r + ln, l.m_col, q.c);
            late_c[k - (MAIN - DEFER)] = q.c;
          }
        }
        if (l.s_on) Lt[(2 * r + l.s_row) * PW + l.s_lx] = slot_eval(v[MAIN]).Ln;
      } else {
        for (int idx = tid; idx < n_new; idx += NT) {
          const int ly = idx / PWv, lx = idx - ly * PWv;
          slot_compute(pixel_at(s, slot_src(x0, y0, interior, ly + 2 * r, lx)), ly + 2 * r, lx);
        }
      }
      if (last_of_seg && tid == 0) next_seg = drawn;
      USM_SYNC();
      if (last_of_seg) {
        pend = seg_pos(next_seg);
      } else {
        pend.ty += 1;
        pend.first = false;
      }
    }
#ifdef SS_USM_TRACE
    ++trace_tiles;
#endif
  }
#ifdef SS_USM_TRACE
  if (tid == 0 && blockIdx.x < 1024) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    ss_usm_trace_buf[blockIdx.x][0] = trace_t0;
    ss_usm_trace_buf[blockIdx.x][1] = wall_clock64();
    ss_usm_trace_buf[blockIdx.x][2] = trace_tiles;
    ss_usm_trace_buf[blockIdx.x][3] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif
}

// Workgroups launched per CU (two fit in LDS at r <= 6); SS_USM_WG_PER_CU overrides it for tuning.
static int usm_wg_per_cu() {
  static const int v = [] {
    const char* e = getenv("SS_USM_WG_PER_CU");
    const int x = e ? atoi(e) : 0;
    return x >= 1 && x <= 16 ? x : 2;
  }();
  return v;
}

// Segment length (tiles) and an upper limit on the number of workgroups.  Read at every launch so that tests
// can force long segments and few workgroups on small images, and tuning runs can vary the length.
static int usm_seg_tiles() {
  const char* e = getenv("SS_USM_SEG_TILES");
  const int x = e ? atoi(e) : 0;
  return x >= 1 ? x : 8;
}
static int64_t usm_max_wg() {
  const char* e = getenv("SS_USM_MAX_WG");
  const int x = e ? atoi(e) : 0;
  return x >= 1 ? x : INT32_MAX;
}

static size_t usm_lds_bytes(int r) {
  const int PW = (USM_TW + 2 * r + 3) & ~3, PH = USM_TH + 2 * r;
  return (size_t)(PH * PW + PH * USM_TW + USM_TH * USM_TW * 3) * sizeof(float);
}

template <int R_T, int CCMC, int OUT, bool PLAIN, int CHROMA = -1>
static hipError_t launch_usm_one(const ChainParams& P, int batch, const float* src, void* dst, uint32_t* ticket,
                                  hipStream_t st) {
  if constexpr (PLAIN && R_T != 0 && CHROMA < 0) {  // the plain chains know their chroma branch at compile time
    return P.usm_chroma ? launch_usm_one<R_T, CCMC, OUT, PLAIN, 1>(P, batch, src, dst, ticket, st)
                        : launch_usm_one<R_T, CCMC, OUT, PLAIN, 0>(P, batch, src, dst, ticket, st);
  } else {
    const size_t lds = usm_lds_bytes(P.usm_r);
    auto kern = usm_kernel<R_T, CCMC, OUT, PLAIN, CHROMA>;
    if (lds > 64 * 1024) {  // sticky per device and kernel; cheap enough to repeat
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    StripWalk T;
    T.tiles_x = (P.width + USM_TW - 1) / USM_TW;
    T.tiles_y = (P.height + USM_TH - 1) / USM_TH;
    int64_t cap = (int64_t)256 * usm_wg_per_cu();  // persistent workgroups
    if (cap > usm_max_wg()) cap = usm_max_wg();
    // segment length: long enough that the first tile's extra halo rows are noise, short enough that the
    // workgroups finish close together
    int seg = usm_seg_tiles();
    if (seg > T.tiles_y) seg = T.tiles_y;
    if (R_T == 0 || ccm_needs_table<CCMC>) seg = 1;  // only the fast path hands rows from tile to tile
    T.seg_tiles = seg;
    T.segs_y = (T.tiles_y + seg - 1) / seg;
    const int64_t n_segments = (int64_t)T.tiles_x * T.segs_y * batch;
    if (n_segments > INT32_MAX) return hipErrorInvalidValue;
    T.n_segments = (uint32_t)n_segments;
    T.chunk = (T.n_segments + 7) / 8;
    T.ticket = ticket;
    const unsigned blocks = (unsigned)(n_segments < cap ? n_segments : cap);
    hipError_t me = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ticket), 0, 8, st);
    if (me != hipSuccess) return me;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(USM_NT), lds, st, P, T, src, dst);
    return hipGetLastError();
  }
}

template <int R_T, int CCMC, bool PLAIN>
static hipError_t launch_usm_out(const ChainParams& P, int batch, int out_kind, const float* src, void* dst,
                                 uint32_t* ticket, hipStream_t st) {
  switch (out_kind) {
    case SS_OUT_F32_LINEAR:
      return launch_usm_one<R_T, CCMC, SS_OUT_F32_LINEAR, PLAIN>(P, batch, src, dst, ticket, st);
    case SS_OUT_F32_SRGB:
      return launch_usm_one<R_T, CCMC, SS_OUT_F32_SRGB, PLAIN>(P, batch, src, dst, ticket, st);
    case SS_OUT_U16_SRGB:
      return launch_usm_one<R_T, CCMC, SS_OUT_U16_SRGB, PLAIN>(P, batch, src, dst, ticket, st);
    case SS_OUT_U8_SRGB:
      return launch_usm_one<R_T, CCMC, SS_OUT_U8_SRGB, PLAIN>(P, batch, src, dst, ticket, st);
    case SS_OUT_U8_LINEAR:
      return launch_usm_one<R_T, CCMC, SS_OUT_U8_LINEAR, PLAIN>(P, batch, src, dst, ticket, st);
    default:
      return hipErrorInvalidValue;
  }
}

template <int R_T>
static hipError_t launch_usm_ccm(const ChainParams& P, int batch, int out_kind, const float* src, void* dst,
                                 uint32_t* ticket, hipStream_t st) {
  const bool lin3 = P.ccm_kind != SS_CCM_NONE && P.ccm_terms == 3;
  // the worker's chart path without further adjustments: CCM, one gain stage, sharpen
  const bool plain = lin3 && !P.neg && P.n_gain == 1 && !P.sh && !P.lab_io && P.usm_thr == 0.0f;
  if (plain) return launch_usm_out<R_T, CCM_LIN3, true>(P, batch, out_kind, src, dst, ticket, st);
  if (lin3) return launch_usm_out<R_T, CCM_LIN3, false>(P, batch, out_kind, src, dst, ticket, st);
  // colour corrections with more terms run as a pointwise pass of their own in front (see ss_chain_run)
  if (P.ccm_kind != SS_CCM_NONE) return hipErrorInvalidValue;
  // the worker's no-chart path with exposure and sharpen only (and the second pass of a staged chain)
  const bool plain_none = !P.neg && P.n_gain == 1 && !P.sh && !P.lab_io && P.usm_thr == 0.0f;
  if (plain_none) return launch_usm_out<R_T, CCM_NONE, true>(P, batch, out_kind, src, dst, ticket, st);
  return launch_usm_out<R_T, CCM_NONE, false>(P, batch, out_kind, src, dst, ticket, st);
}

hipError_t launch_usm(const ChainParams& P, int batch, int out_kind, const float* src, void* dst, uint32_t* ticket,
                      hipStream_t st) {
  if (P.usm_r < 1 || P.usm_r > SS_MAX_USM_RADIUS) return hipErrorInvalidValue;
  if (P.usm_r == 3) return launch_usm_ccm<3>(P, batch, out_kind, src, dst, ticket, st);
  if (P.usm_r == 6) return launch_usm_ccm<6>(P, batch, out_kind, src, dst, ticket, st);
  return launch_usm_ccm<0>(P, batch, out_kind, src, dst, ticket, st);
}

}  // namespace ss
