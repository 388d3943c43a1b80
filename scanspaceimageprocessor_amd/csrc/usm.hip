// Fused correction chain WITH the unsharp mask (stage 5) for gfx950.
//
//   usm_kernel : a workgroup owns a 128 x TH pixel tile, recomputes stages 1-4 + Lab lightness for
//                the tile plus a halo of `r` pixels into LDS, runs the separable Gaussian there (row
//                pass into registers, written back in place, then the column pass), and finishes the
//                pixel (mask logic, Lab -> sRGB, encoding, quantisation) without ever writing an
//                intermediate plane to HBM.  Algorithmic traffic: 12 B/px in + output.

#include "ss_internal.h"

namespace ss {

// ------------------------------------------------------------------------------------------------
// fused unsharp mask
// ------------------------------------------------------------------------------------------------

constexpr int USM_TW = 128;  // tile width: 64 lanes x 2 adjacent pixels

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

// R_T: compile-time blur radius (3 <-> sigma 1, 6 <-> sigma 2), 0 = runtime radius P.usm_r.
// TH : tile height; the workgroup has TH*8 threads = TH/8 waves, each wave owns 8 tile rows.
template <int R_T, int OUT, int TH>
__global__ __launch_bounds__(TH * 8) void usm_kernel(const ChainParams P, const float* __restrict__ src,
                                                     void* __restrict__ dst) {
  constexpr int TW = USM_TW;
  constexpr int NT = TH * 8;
  constexpr int NW = NT / 64;
  constexpr int ROWS = 8;  // tile rows per wave
  constexpr int RMAX = R_T ? R_T : SS_MAX_USM_RADIUS;
  constexpr int P2_ITERS = ((TH + 2 * RMAX) * (TW / 4) + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float S[];

  const int r = R_T ? R_T : P.usm_r;
  const int PWv = TW + 2 * r;      // valid LDS row width
  const int PW = (PWv + 3) & ~3;   // row pitch (16-byte aligned rows)
  const int PH = TH + 2 * r;
  const int W = P.width, H = P.height;
  const int img = blockIdx.z;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ImgParams ip = load_img_params(P, img);
  const F3* __restrict__ s = reinterpret_cast<const F3*>(src + (size_t)img * P.n_px * 3);

  // ---- phase 1a: tile pixels (pairs), keep Ln/a/b in registers, Ln also to LDS ----
  float stL[ROWS][2], stA[ROWS][2], stB[ROWS][2];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int ly = wave + NW * i;
    const int sy = reflect_coord(y0 + ly, H);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int sx = reflect_coord(x0 + 2 * lane + j, W);
      const F3 v = s[(uint32_t)sy * W + sx];
      const Px p = front<CCM_RUNTIME, true>(Px{v.x, v.y, v.z}, P, ip);
      Lab q;
      if (P.lab_io) {
        q = Lab{p.r, p.g, p.b};
      } else {
        q = rgb_to_lab(p, P);
      }
      stL[i][j] = q.L * 0.01f;
      stA[i][j] = q.a;
      stB[i][j] = q.b;
    }
    float* row = S + (ly + r) * PW + r + 2 * lane;
    row[0] = stL[i][0];
    row[1] = stL[i][1];
  }

  // ---- phase 1b: halo ring, lightness only ----
  {
    const int n_tb = 2 * r * PWv;  // top + bottom bands, full width
    const int n_ring = n_tb + TH * 2 * r;
    for (int idx = tid; idx < n_ring; idx += NT) {
      int ly, lx;
      if (idx < n_tb) {
        const int row = idx / PWv;
        lx = idx - row * PWv;
        ly = row < r ? row : row + TH;
      } else {
        const int j = idx - n_tb;
        const int row = j / (2 * r);
        const int c = j - row * (2 * r);
        ly = r + row;
        lx = c < r ? c : c + TW;
      }
      const int sy = reflect_coord(y0 - r + ly, H);
      const int sx = reflect_coord(x0 - r + lx, W);
      const F3 v = s[(uint32_t)sy * W + sx];
      const Px p = front<CCM_RUNTIME, true>(Px{v.x, v.y, v.z}, P, ip);
      const float Ln = P.lab_io ? p.r * 0.01f : fmaf(1.16f, rgb_to_fy(p, P), -0.16f);
      S[ly * PW + lx] = Ln;
    }
  }
  __syncthreads();

  // ---- phase 2: row pass (cv2 runs the row filter first), results held in registers ----
  float4 hreg[P2_ITERS];
  const int n_items = PH * (TW / 4);
#pragma unroll
  for (int it = 0; it < P2_ITERS; ++it) {
    const int item = tid + it * NT;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (item < n_items) {
      const int row = item >> 5, q = item & 31;
      const float* base = S + row * PW + 4 * q;
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
        acc.x = v[0] * P.taps[0];
        acc.y = v[1] * P.taps[0];
        acc.z = v[2] * P.taps[0];
        acc.w = v[3] * P.taps[0];
#pragma unroll
        for (int j = 1; j <= 2 * R_T; ++j) {
          acc.x = fmaf(v[j + 0], P.taps[j], acc.x);
          acc.y = fmaf(v[j + 1], P.taps[j], acc.y);
          acc.z = fmaf(v[j + 2], P.taps[j], acc.z);
          acc.w = fmaf(v[j + 3], P.taps[j], acc.w);
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
          const float t = P.taps[j];
          acc.x = fmaf(w0, t, acc.x);
          acc.y = fmaf(w1, t, acc.y);
          acc.z = fmaf(w2, t, acc.z);
          acc.w = fmaf(w3, t, acc.w);
        }
      }
    }
    hreg[it] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < P2_ITERS; ++it) {
    const int item = tid + it * NT;
    if (item < n_items) {
      const int row = item >> 5, q = item & 31;
      *reinterpret_cast<float4*>(S + row * PW + 4 * q) = hreg[it];
    }
  }
  __syncthreads();

  // ---- phase 3: column pass for the owned pixels, mask logic, Lab -> sRGB, encode, store ----
  char* d = reinterpret_cast<char*>(dst) + (size_t)img * P.n_px * 3 * OutTraits<OUT>::BYTES;
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int ly = wave + NW * i;
    const float* col = S + ly * PW + 2 * lane;
    float2 t = *reinterpret_cast<const float2*>(col);
    float b0 = t.x * P.taps[0], b1 = t.y * P.taps[0];
    if constexpr (R_T != 0) {
#pragma unroll
      for (int j = 1; j <= 2 * R_T; ++j) {
        t = *reinterpret_cast<const float2*>(col + j * PW);
        b0 = fmaf(t.x, P.taps[j], b0);
        b1 = fmaf(t.y, P.taps[j], b1);
      }
    } else {
      for (int j = 1; j <= 2 * r; ++j) {
        t = *reinterpret_cast<const float2*>(col + j * PW);
        b0 = fmaf(t.x, P.taps[j], b0);
        b1 = fmaf(t.y, P.taps[j], b1);
      }
    }
    const Lab q0 = usm_on_lab(stL[i][0], stA[i][0], stB[i][0], b0, P);
    const Lab q1 = usm_on_lab(stL[i][1], stA[i][1], stB[i][1], b1, P);
    Px o0, o1;
    if (P.lab_io) {
      o0 = Px{q0.L, q0.a, q0.b};
      o1 = Px{q1.L, q1.a, q1.b};
    } else {
      o0 = lab_to_rgb(q0, P);
      o1 = lab_to_rgb(q1, P);
    }
    usm_store_pair<OUT>(d, W, H, y0 + ly, x0 + 2 * lane, finish<OUT>(o0, P), finish<OUT>(o1, P));
  }
}

static int g_usm_tile_h = 32;  // set once from SS_USM_TH before the first launch (tuning knob)

template <int R_T, int TH>
static hipError_t launch_usm_out(const ChainParams& P, int batch, int out_kind, const float* src, void* dst,
                                 hipStream_t st) {
  const int r = P.usm_r;
  const int PW = (USM_TW + 2 * r + 3) & ~3;
  const size_t lds = (size_t)(TH + 2 * r) * PW * sizeof(float);
  const dim3 grid((P.width + USM_TW - 1) / USM_TW, (P.height + TH - 1) / TH, batch);
  const dim3 block(TH * 8);
  switch (out_kind) {
    case SS_OUT_F32_LINEAR:
      hipLaunchKernelGGL((usm_kernel<R_T, SS_OUT_F32_LINEAR, TH>), grid, block, lds, st, P, src, dst);
      break;
    case SS_OUT_F32_SRGB:
      hipLaunchKernelGGL((usm_kernel<R_T, SS_OUT_F32_SRGB, TH>), grid, block, lds, st, P, src, dst);
      break;
    case SS_OUT_U16_SRGB:
      hipLaunchKernelGGL((usm_kernel<R_T, SS_OUT_U16_SRGB, TH>), grid, block, lds, st, P, src, dst);
      break;
    case SS_OUT_U8_SRGB:
      hipLaunchKernelGGL((usm_kernel<R_T, SS_OUT_U8_SRGB, TH>), grid, block, lds, st, P, src, dst);
      break;
    case SS_OUT_U8_LINEAR:
      hipLaunchKernelGGL((usm_kernel<R_T, SS_OUT_U8_LINEAR, TH>), grid, block, lds, st, P, src, dst);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int TH>
static hipError_t launch_usm_r(const ChainParams& P, int batch, int out_kind, const float* src, void* dst,
                               hipStream_t st) {
  if (P.usm_r == 3) return launch_usm_out<3, TH>(P, batch, out_kind, src, dst, st);
  if (P.usm_r == 6) return launch_usm_out<6, TH>(P, batch, out_kind, src, dst, st);
  return launch_usm_out<0, TH>(P, batch, out_kind, src, dst, st);
}

hipError_t launch_usm(const ChainParams& P, int batch, int out_kind, const float* src, void* dst, hipStream_t st) {
  static const int th = [] {
    const char* e = getenv("SS_USM_TH");
    return (e && atoi(e) == 64) ? 64 : g_usm_tile_h;
  }();
  if (P.usm_r < 1 || P.usm_r > SS_MAX_USM_RADIUS) return hipErrorInvalidValue;
  return th == 64 ? launch_usm_r<64>(P, batch, out_kind, src, dst, st)
                  : launch_usm_r<32>(P, batch, out_kind, src, dst, st);
}

}  // namespace ss
