// Fused correction chain WITH the unsharp mask (stage 5) for gfx950.
//
// A 512-thread workgroup owns one 128 x 32 pixel tile at a time and walks a list of tiles.
//   phase 1  every thread walks the tile plus its halo of r pixels (flat index, lanes along x, so
//            each wave-instruction is a contiguous run of 12-byte pixels), runs stages 1-4 and the
//            RGB -> Lab conversion, and leaves Ln = L/100 of every slot and (a, b) of the tile's own
//            pixels in LDS.  Halo pixels are recomputed rather than exchanged: 1.5x the per-pixel
//            front work at r = 6, no intermediate plane in HBM, no inter-workgroup dependency.
//   phase 2  row pass of the separable Gaussian (cv2 filters rows first), 4 outputs per thread from a
//            sliding register window fed by ds_read_b128, into a second LDS plane.
//   phase 3  column pass for the tile's pixels (pairs, ds_read_b64), mask logic, Lab -> sRGB, final
//            encoding, quantisation, one 12-byte store per lane for 16-bit output.
// Algorithmic HBM traffic: 12 B/px in + 12 / 6 / 3 B/px out; halo re-reads are served by L2.
// LDS per workgroup at r = 6: (44 x 140 + 44 x 128 + 32 x 128 x 2) x 4 B = 79,936 B -> two workgroups
// (16 waves) per CU.

#include "ss_internal.h"

namespace ss {

constexpr int USM_TW = 128;  // tile width: 64 lanes x 2 adjacent pixels
constexpr int USM_TH = 32;   // tile height
#ifndef SS_USM_NT
#define SS_USM_NT 512
#endif
constexpr int USM_NT = SS_USM_NT;  // threads per workgroup
constexpr int USM_WAVES_PER_SIMD = 2 * USM_NT / 256;  // two workgroups per CU

#ifndef SS_USM_PREFETCH
#define SS_USM_PREFETCH 13  // phase-1 pixels in flight per thread (13 = the whole tile at r = 6)
#endif

template <int OUT>
__device__ __forceinline__ void usm_store_pair(void* dst_img, int W, int H, int y, int x, const Px o0, const Px o1) {
  if (y >= H || x >= W) return;
  const bool two = x + 1 < W;
  const size_t px = (size_t)y * W + x;
  if constexpr (OUT == SS_OUT_F32_LINEAR || OUT == SS_OUT_F32_SRGB) {
    F3* d = reinterpret_cast<F3*>(dst_img) + px;
    d[0] = F3{o0.r, o0.g, o0.b};
    if (two) d[1] = F3{o1.r, o1.g, o1.b};
  } else if constexpr (OUT == SS_OUT_U16_SRGB) {
    char* d = reinterpret_cast<char*>(dst_img) + px * 6;
    if (two && ((px * 6) & 3) == 0) {
      U3 q;
      q.x = quant16(o0.r) | (quant16(o0.g) << 16);
      q.y = quant16(o0.b) | (quant16(o1.r) << 16);
      q.z = quant16(o1.g) | (quant16(o1.b) << 16);
      *reinterpret_cast<U3*>(d) = q;
    } else {
      uint16_t* h = reinterpret_cast<uint16_t*>(d);
      h[0] = (uint16_t)quant16(o0.r);
      h[1] = (uint16_t)quant16(o0.g);
      h[2] = (uint16_t)quant16(o0.b);
      if (two) {
        h[3] = (uint16_t)quant16(o1.r);
        h[4] = (uint16_t)quant16(o1.g);
        h[5] = (uint16_t)quant16(o1.b);
      }
    }
  } else {
    uint8_t* d = reinterpret_cast<uint8_t*>(dst_img) + px * 3;
    d[0] = (uint8_t)quant8(o0.r);
    d[1] = (uint8_t)quant8(o0.g);
    d[2] = (uint8_t)quant8(o0.b);
    if (two) {
      d[3] = (uint8_t)quant8(o1.r);
      d[4] = (uint8_t)quant8(o1.g);
      d[5] = (uint8_t)quant8(o1.b);
    }
  }
}

// R_T : compile-time blur radius (3 <-> sigma 1, 6 <-> sigma 2), 0 = runtime radius P.usm_r.
// CCMC: CCM_LIN3 for the reference's 3-term correction, CCM_RUNTIME for everything else.
// Tile walk: workgroup b serves "XCD" b % 8 (workgroups are dealt round-robin over the 8 XCDs, so
// b and b + 8 share an L2); every XCD gets one contiguous band of the tile sequence and its workgroups
// walk that band side by side, so that tiles sharing halo pixels meet in the same L2 close in time.
// Placement only affects speed: any assignment of tiles to workgroups is correct.
struct TileWalk {
  int tiles_x, tiles_y, tiles_per_img;
  int64_t n_tiles;  // whole batch
};

template <int R_T, int CCMC, int OUT, bool PLAIN>
__global__ __launch_bounds__(USM_NT, USM_WAVES_PER_SIMD) void usm_kernel(const ChainParams P, const TileWalk T,
                                                     const float* __restrict__ src, void* __restrict__ dst) {
  constexpr int TW = USM_TW, TH = USM_TH, NT = USM_NT;
  extern __shared__ __attribute__((aligned(16))) float S[];
  __shared__ float4 M[ccm_needs_table<CCMC> ? SS_MAX_CCM_TERMS : 1];
  if constexpr (ccm_needs_table<CCMC>) {
    stage_ccm(P, M, threadIdx.x, NT);
    __syncthreads();
  }

  const int r = R_T ? R_T : P.usm_r;
  const int PWv = TW + 2 * r;     // slots per LDS row that hold data
  const int PW = (PWv + 3) & ~3;  // row pitch in floats (rows stay 16-byte aligned)
  const int PH = TH + 2 * r;
  float* __restrict__ Lt = S;             // [PH][PW]  Ln of tile + halo
  float* __restrict__ Ht = Lt + PH * PW;  // [PH][TW]  row-filtered Ln
  float* __restrict__ AB = Ht + PH * TW;  // [TH][TW][2] a, b of the tile's pixels
  const int W = P.width, H = P.height;
  const int tid = threadIdx.x;
  const int n_slots = PH * PWv;
  // phase-1 slots per thread; known at compile time for the common radii so that every pixel load
  // of a tile can be issued before the first one is consumed
  constexpr int ITERS = R_T ? ((TH + 2 * R_T) * (TW + 2 * R_T) + NT - 1) / NT : 1;

  // this workgroup's share of the tile sequence
  const int xcd = blockIdx.x & 7, lane_in_xcd = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
  const int64_t band = (T.n_tiles + 7) >> 3;
  const int64_t band_begin = xcd * band;
  const int64_t band_end = band_begin + band < T.n_tiles ? band_begin + band : T.n_tiles;

  int cur_img = -1;
  ImgParams ip;
  for (int64_t tile = band_begin + lane_in_xcd; tile < band_end; tile += wg_per_xcd) {
    const int img = (int)(tile / T.tiles_per_img);
    const int t_in = (int)(tile - (int64_t)img * T.tiles_per_img);
    const int ty = t_in / T.tiles_x, tx = t_in - ty * T.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    if (img != cur_img) {
      ip = load_img_params(P, img);
      cur_img = img;
    }
    const F3* __restrict__ s = reinterpret_cast<const F3*>(src + (size_t)img * P.n_px * 3);
    // tiles whose halo stays inside the image skip the border arithmetic
    const bool interior = x0 >= r && y0 >= r && x0 + TW + r <= W && y0 + TH + r <= H;

    // ---- phase 1: stages 1-4 + RGB -> Lab for the tile and its halo ----
    auto slot_src = [&](int idx, int& ly, int& lx) -> uint32_t {
      ly = idx / PWv;
      lx = idx - ly * PWv;
      int sy = y0 - r + ly, sx = x0 - r + lx;
      if (!interior) {
        sy = reflect_coord(sy, H);
        sx = reflect_coord(sx, W);
      }
      return (uint32_t)sy * W + sx;
    };
    auto slot_compute = [&](const F3 v, int ly, int lx) {
      const Px p = front<CCMC, true, PLAIN>(Px{v.x, v.y, v.z}, P, ip, M);
      Lab q;
      if (!PLAIN && P.lab_io) {
        q = Lab{p.r, p.g, p.b};
      } else {
        q = rgb_to_lab(p, P);
      }
      Lt[ly * PW + lx] = q.L * 0.01f;
      const int cy = ly - r, cx = lx - r;
      if ((unsigned)cy < (unsigned)TH && (unsigned)cx < (unsigned)TW) {
        *reinterpret_cast<float2*>(AB + (cy * TW + cx) * 2) = make_float2(q.a, q.b);
      }
    };
    if constexpr (R_T != 0 && !ccm_needs_table<CCMC>) {
      // pixels are fetched in groups of PF: every load of a group is issued before its first use
      // (not for the generic colour-correction code, which is too large to unroll)
      constexpr int PF = SS_USM_PREFETCH < ITERS ? SS_USM_PREFETCH : ITERS;
#pragma unroll
      for (int k0 = 0; k0 < ITERS; k0 += PF) {
        F3 v[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
          const int idx = tid + (k0 + k) * NT;
          int ly, lx;
          if (k0 + k < ITERS && idx < n_slots) v[k] = s[slot_src(idx, ly, lx)];
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
          const int idx = tid + (k0 + k) * NT;
          if (k0 + k < ITERS && idx < n_slots) {
            const int ly = idx / PWv;
            slot_compute(v[k], ly, idx - ly * PWv);
          }
        }
      }
    } else {
      for (int idx = tid; idx < n_slots; idx += NT) {
        int ly, lx;
        const F3 v = s[slot_src(idx, ly, lx)];
        slot_compute(v, ly, lx);
      }
    }
    __syncthreads();

    // ---- phase 2: row pass ----
    const int n_items = PH * (TW / 4);
    for (int item = tid; item < n_items; item += NT) {
      const int row = item >> 5, q = item & 31;
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
    __syncthreads();

    // ---- phase 3: column pass, mask logic, Lab -> sRGB, encode, store ----
    char* d = reinterpret_cast<char*>(dst) + (size_t)img * P.n_px * 3 * OutTraits<OUT>::BYTES;
    for (int pair = tid; pair < TH * TW / 2; pair += NT) {
      const int ly = pair >> 6, px = (pair & 63) * 2;
      const float* col = Ht + ly * TW + px;
      float2 t = *reinterpret_cast<const float2*>(col);
      float b0 = t.x * P.taps[0], b1 = t.y * P.taps[0];
      if constexpr (R_T != 0) {
#pragma unroll
        for (int j = 1; j <= 2 * R_T; ++j) {
          t = *reinterpret_cast<const float2*>(col + j * TW);
          b0 = fmaf(t.x, tap_at<R_T>(P, j), b0);
          b1 = fmaf(t.y, tap_at<R_T>(P, j), b1);
        }
      } else {
        for (int j = 1; j <= 2 * r; ++j) {
          t = *reinterpret_cast<const float2*>(col + j * TW);
          const float tp = P.taps[j <= r ? j : 2 * r - j];
          b0 = fmaf(t.x, tp, b0);
          b1 = fmaf(t.y, tp, b1);
        }
      }
      const float* lc = Lt + (ly + r) * PW + r + px;
      const float4 ab = *reinterpret_cast<const float4*>(AB + (ly * TW + px) * 2);
      const Lab q0 = usm_on_lab<PLAIN>(lc[0], ab.x, ab.y, b0, P);
      const Lab q1 = usm_on_lab<PLAIN>(lc[1], ab.z, ab.w, b1, P);
      Px o0, o1;
      if (!PLAIN && P.lab_io) {
        o0 = Px{q0.L, q0.a, q0.b};
        o1 = Px{q1.L, q1.a, q1.b};
      } else {
        o0 = lab_to_rgb(q0, P);
        o1 = lab_to_rgb(q1, P);
      }
      usm_store_pair<OUT>(d, W, H, y0 + ly, x0 + px, finish<OUT, true>(o0, P), finish<OUT, true>(o1, P));
    }
    __syncthreads();  // the next tile's phase 1 overwrites the planes phase 3 has just read
  }
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

static size_t usm_lds_bytes(int r) {
  const int PW = (USM_TW + 2 * r + 3) & ~3, PH = USM_TH + 2 * r;
  return (size_t)(PH * PW + PH * USM_TW + USM_TH * USM_TW * 2) * sizeof(float);
}

template <int R_T, int CCMC, int OUT, bool PLAIN>
static hipError_t launch_usm_one(const ChainParams& P, int batch, const float* src, void* dst, hipStream_t st) {
  const size_t lds = usm_lds_bytes(P.usm_r);
  auto kern = usm_kernel<R_T, CCMC, OUT, PLAIN>;
  if (lds > 64 * 1024) {  // sticky per device and kernel; cheap enough to repeat
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  TileWalk T;
  T.tiles_x = (P.width + USM_TW - 1) / USM_TW;
  T.tiles_y = (P.height + USM_TH - 1) / USM_TH;
  T.tiles_per_img = T.tiles_x * T.tiles_y;
  T.n_tiles = (int64_t)T.tiles_per_img * batch;
  // persistent workgroups: a multiple of 8 (one share per XCD), at most usm_wg_per_cu per CU
  int64_t want = (T.n_tiles + 7) / 8 * 8;
  const int64_t cap = (int64_t)256 * usm_wg_per_cu();
  const unsigned blocks = (unsigned)(want < cap ? want : cap);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(USM_NT), lds, st, P, T, src, dst);
  return hipGetLastError();
}

template <int R_T, int CCMC, bool PLAIN>
static hipError_t launch_usm_out(const ChainParams& P, int batch, int out_kind, const float* src, void* dst,
                                 hipStream_t st) {
  switch (out_kind) {
    case SS_OUT_F32_LINEAR:
      return launch_usm_one<R_T, CCMC, SS_OUT_F32_LINEAR, PLAIN>(P, batch, src, dst, st);
    case SS_OUT_F32_SRGB:
      return launch_usm_one<R_T, CCMC, SS_OUT_F32_SRGB, PLAIN>(P, batch, src, dst, st);
    case SS_OUT_U16_SRGB:
      return launch_usm_one<R_T, CCMC, SS_OUT_U16_SRGB, PLAIN>(P, batch, src, dst, st);
    case SS_OUT_U8_SRGB:
      return launch_usm_one<R_T, CCMC, SS_OUT_U8_SRGB, PLAIN>(P, batch, src, dst, st);
    case SS_OUT_U8_LINEAR:
      return launch_usm_one<R_T, CCMC, SS_OUT_U8_LINEAR, PLAIN>(P, batch, src, dst, st);
    default:
      return hipErrorInvalidValue;
  }
}

template <int R_T>
static hipError_t launch_usm_ccm(const ChainParams& P, int batch, int out_kind, const float* src, void* dst,
                                 hipStream_t st) {
  const bool lin3 = P.ccm_kind != SS_CCM_NONE && P.ccm_terms == 3;
  // the worker's chart path without further adjustments: CCM, one gain stage, sharpen
  const bool plain = lin3 && !P.neg && P.n_gain == 1 && !P.sh && !P.lab_io && P.usm_thr == 0.0f;
  if (plain) return launch_usm_out<R_T, CCM_LIN3, true>(P, batch, out_kind, src, dst, st);
  if (lin3) return launch_usm_out<R_T, CCM_LIN3, false>(P, batch, out_kind, src, dst, st);
  return launch_usm_out<R_T, CCM_RUNTIME, false>(P, batch, out_kind, src, dst, st);
}

hipError_t launch_usm(const ChainParams& P, int batch, int out_kind, const float* src, void* dst, hipStream_t st) {
  if (P.usm_r < 1 || P.usm_r > SS_MAX_USM_RADIUS) return hipErrorInvalidValue;
  if (P.usm_r == 3) return launch_usm_ccm<3>(P, batch, out_kind, src, dst, st);
  if (P.usm_r == 6) return launch_usm_ccm<6>(P, batch, out_kind, src, dst, st);
  return launch_usm_ccm<0>(P, batch, out_kind, src, dst, st);
}

}  // namespace ss
