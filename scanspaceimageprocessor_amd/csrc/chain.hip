// Pointwise fused correction-chain kernel for gfx950.
//
//   chain_kernel : stages 1-4, 6, 7 as one pointwise pass.  A lane loads 4 pixels, 256 pixels apart
//                  (per-pixel global_load_dwordx3: 12 B, 4-byte aligned, lanes contiguous, no width
//                  restriction), and every lane stores exactly 12 bytes (one global_store_dwordx3,
//                  lanes contiguous): 1 pixel of float32, or G = 2 pixels of uint16 / 4 pixels of uint8
//                  whose quantised values reach their storing lane through LDS.  Loads and stores
//                  stream (nontemporal).  Algorithmic traffic: 12 B/px in + 12 / 6 / 3 B/px out; HBM-bound.

#include "ss_internal.h"
#include "ss_careful.h"

namespace ss {

// ------------------------------------------------------------------------------------------------
// pointwise chain
// ------------------------------------------------------------------------------------------------

template <int OUT>
__device__ __forceinline__ void store_one(void* dst_img, uint32_t px, const Px p) {
  if constexpr (OUT == SS_OUT_F32_LINEAR || OUT == SS_OUT_F32_SRGB) {
    reinterpret_cast<F3*>(dst_img)[px] = F3{p.r, p.g, p.b};
  } else if constexpr (OUT == SS_OUT_U16_SRGB) {
    uint16_t* d = reinterpret_cast<uint16_t*>(dst_img) + (size_t)px * 3;
    d[0] = (uint16_t)quant16(p.r);
    d[1] = (uint16_t)quant16(p.g);
    d[2] = (uint16_t)quant16(p.b);
  } else {
    uint8_t* d = reinterpret_cast<uint8_t*>(dst_img) + (size_t)px * 3;
    d[0] = (uint8_t)quant8(p.r);
    d[1] = (uint8_t)quant8(p.g);
    d[2] = (uint8_t)quant8(p.b);
  }
}

template <int CCMC, bool SH, int OUT>
__global__ __launch_bounds__(256) void chain_kernel(const ChainParams P_arg, const float* __restrict__ src,
                                                    void* __restrict__ dst) {
  constexpr int G = OutTraits<OUT>::G;
  constexpr int K = 4 / G;  // groups per lane per trip: 4 pixels in flight per lane
  __shared__ float4 M[ccm_needs_table<CCMC> ? SS_MAX_CCM_TERMS : 1];
  if constexpr (ccm_needs_table<CCMC>) {
    stage_ccm(P_arg, M, threadIdx.x, 256);
    __syncthreads();
  }
  const int img = blockIdx.y;
  const ImgParams ip = load_img_params(P_arg, img);
  const uint32_t n = (uint32_t)P_arg.n_px;
  // Chains with the shadows/highlights stage use more scalars than the register file holds.  They read the
  // parameter block through a pointer to the kernel argument segment the compiler cannot see through, once
  // per trip: scalar loads next to the uses instead of scalars parked in (and fetched back from) vector lanes.
  // (The parameter block is the kernel's first argument: offset 0 of the argument segment.)
  auto params_now = [&]() -> const ChainParams& {
    if constexpr (SH && !ccm_needs_table<CCMC>) {  // (no gain measured for the table-driven corrections)
      typedef const __attribute__((address_space(4))) ChainParams* KernargParams;
      KernargParams k = (KernargParams)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(k));
      return *(const ChainParams*)k;
    }
    return P_arg;
  };
  const F3* __restrict__ s = reinterpret_cast<const F3*>(src + (size_t)img * n * 3);
  char* d = reinterpret_cast<char*>(dst) + (size_t)img * n * 3 * OutTraits<OUT>::BYTES;
  const uint32_t n_groups = n / G;
  constexpr uint32_t CHUNK = 256 * K;

  // 16- and 8-bit outputs: one lane stores 12 bytes = G pixels, but loads them like the float path - lane t takes
  // pixels t, t + 256, t + 512, t + 768 of a 1024-pixel chunk, so that every load covers whole cache lines across
  // the wave and can stream - and the quantised values change lanes through LDS, laid out as the chunk's output
  // bytes.
  constexpr int BYTES = OutTraits<OUT>::BYTES;
  constexpr int STAGE_WORDS = G == 1 ? 1 : 1024 * 3 * BYTES / 4;
  __shared__ uint32_t stage[STAGE_WORDS];

  // (base * G < n rather than base < n_groups: when the groups fill their last chunk exactly, the n % G pixels behind
  // them still get a trip, which records them and stores their second output; their first output is stored below)
  for (uint32_t base = blockIdx.x * CHUNK; base * G < n; base += gridDim.x * CHUNK) {
    const ChainParams& P = params_now();
    F3 in[4] = {};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // G == 1: group = pixel; G > 1: the chunk starts at pixel base * G
      const uint32_t p = base * G + k * 256 + threadIdx.x;
      if (p < (G == 1 ? n_groups : n)) in[k] = load_stream(s + p);
    }
    // the 4 pixels of this lane go through the stages as two packed pairs (lanes past the end of the image
    // compute on zeros and store nothing)
    Px res[4];
    bool flagged[4] = {false, false, false, false};
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
      RiskT<v2f> risk{};
      const Px2 f2 = front<CCMC, SH, false, v2f>(
          pair_px(Px{in[q].x, in[q].y, in[q].z}, Px{in[q + 1].x, in[q + 1].y, in[q + 1].z}), P, ip, M, SH ? &risk : nullptr);
      {  // the second output, when the call has one (a scalar branch on a kernel argument)
        const ChainParams& Pk = kernarg_params();
        if (Pk.dst2 != nullptr) {
          const uint32_t p0 = base * G + q * 256 + threadIdx.x, p1 = p0 + 256;
          const bool clip = Pk.final_clip != 0, enc = Pk.dst2_encoded != 0;
          if (p0 < n) store_second(Pk.dst2, (size_t)img * n + p0, px_lo(f2), clip, enc);
          if (p1 < n) store_second(Pk.dst2, (size_t)img * n + p1, px_hi(f2), clip, enc);
        }
      }
      // (behind the shadows/highlights stage the pixel is already clipped - the stage ends in clip(encode(.)) - or
      // it is Lab data, for which the descriptor cannot ask for a final clip: either way that clip has nothing to do)
      const Px2 o2 = finish<OUT, SH, v2f>(f2, P);
      res[q] = px_lo(o2);
      res[q + 1] = px_hi(o2);
      if constexpr (SH && RiskCfg<OUT>::ENABLED) {
        // pixels whose shadows/highlights edit float32 cannot follow (see RiskCfg): recorded for the float64 kernel
        const bool big0 = risk_big<OUT>(risk.mx.x), big1 = risk_big<OUT>(risk.mx.y);
        if (__ballot(big0 || big1) != 0ull) {  // a scalar branch, almost never taken
          const uint32_t p0 = base * G + q * 256 + threadIdx.x, p1 = p0 + 256;
          flagged[q] = big0 && risk_cancelled<OUT>(risk.mx.x, risk.o.r.x, risk.o.g.x, risk.o.b.x) && p0 < n;
          flagged[q + 1] = big1 && risk_cancelled<OUT>(risk.mx.y, risk.o.r.y, risk.o.g.y, risk.o.b.y) && p1 < n;
        }
      }
      // long term lists: one pair's arithmetic at a time keeps the register count down (282 -> 102 vector
      // registers for the 22-term lists, i.e. one wave per SIMD -> four)
      if constexpr (ccm_needs_table<CCMC>) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SH && RiskCfg<OUT>::ENABLED) {
      // Lane t holds pixels t, t + 256, ... of the chunk: a wave's k-th pixels are 64 consecutive ones, one word
      // of the record (which the launcher has cleared).  Almost never taken.
      if (__ballot(flagged[0] || flagged[1] || flagged[2] || flagged[3]) != 0ull) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned long long m = __ballot(flagged[k]);
          const uint32_t first = base * G + k * 256 + (threadIdx.x & ~63u);  // a multiple of 64
          if ((threadIdx.x & 63u) == 0 && m != 0ull) {
            const ChainParams& Pr = kernarg_params();
            Pr.risk_bits[(size_t)img * ((n + 63u) / 64u) + first / 64u] = m;
            *Pr.risk_any = 1u;
          }
        }
      }
    }
    if constexpr (G == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t g = base + k * 256 + threadIdx.x;
        if (g < n_groups) store_stream(reinterpret_cast<F3*>(d) + g, F3{res[k].r, res[k].g, res[k].b});
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t i = k * 256 + threadIdx.x;  // pixel inside the chunk
        if constexpr (BYTES == 2) {
          uint16_t* h = reinterpret_cast<uint16_t*>(stage) + 3 * i;
          h[0] = (uint16_t)quant16(res[k].r);
          h[1] = (uint16_t)quant16(res[k].g);
          h[2] = (uint16_t)quant16(res[k].b);
        } else {
          uint8_t* b = reinterpret_cast<uint8_t*>(stage) + 3 * i;
          b[0] = (uint8_t)quant8(res[k].r);
          b[1] = (uint8_t)quant8(res[k].g);
          b[2] = (uint8_t)quant8(res[k].b);
        }
      }
      __syncthreads();
      constexpr uint32_t UNITS = 1024 * 3 * BYTES / 12;  // 12-byte units of the chunk = groups
#pragma unroll
      for (uint32_t u = threadIdx.x; u < UNITS; u += 256) {
        const uint32_t g = base + u;
        if (g < n_groups) store_stream(reinterpret_cast<U3*>(d) + g, U3{stage[3 * u], stage[3 * u + 1], stage[3 * u + 2]});
      }
      __syncthreads();  // the next trip writes the staging words again
    }
  }
  // the n % G pixels no full group covers
  if (blockIdx.x == 0) {
    const uint32_t px = n_groups * G + threadIdx.x;
    if (px < n) {
      const F3 v = s[px];
      const ChainParams& P = params_now();
      Px o = finish<OUT, SH, float>(front<CCMC, SH, false, float>(Px{v.x, v.y, v.z}, P, ip, M), P);
      // (the main loop has already recorded these pixels and stored their second output: it evaluates every pixel,
      // it just does not store the first output of these)
      store_one<OUT>(d, px, o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 16-bit RAW output -> float32 (imageProcessorWorker.py:313: np.array(rgb, dtype=np.float32) / 65535.0)
// ------------------------------------------------------------------------------------------------

// One lane converts 8 values: a 16-byte load, two 16-byte stores.  The division is IEEE float32 division,
// as numpy's is (a multiplication by the reciprocal differs in the last bit for some inputs).
__global__ __launch_bounds__(256) void unpack_u16_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst,
                                                         uint64_t n) {
  const uint64_t n8 = n / 8;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * 256) {
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(src) + 4 * i;
    const uint32_t w[4] = {__builtin_nontemporal_load(sp), __builtin_nontemporal_load(sp + 1),
                           __builtin_nontemporal_load(sp + 2), __builtin_nontemporal_load(sp + 3)};
    float f[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = __fdiv_rn((float)(w[k] & 0xFFFFu), 65535.0f);
      f[2 * k + 1] = __fdiv_rn((float)(w[k] >> 16), 65535.0f);
    }
    reinterpret_cast<float4*>(dst)[2 * i] = make_float4(f[0], f[1], f[2], f[3]);
    reinterpret_cast<float4*>(dst)[2 * i + 1] = make_float4(f[4], f[5], f[6], f[7]);
  }
  if (blockIdx.x == 0) {
    const uint64_t i = n8 * 8 + threadIdx.x;
    if (i < n) dst[i] = __fdiv_rn((float)src[i], 65535.0f);
  }
}

hipError_t launch_unpack_u16(const uint16_t* src, float* dst, uint64_t n, hipStream_t st) {
  uint64_t blocks = (n / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > (1u << 22)) blocks = 1u << 22;  // (one trip per workgroup streams best)
  hipLaunchKernelGGL(unpack_u16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n);
  return hipGetLastError();
}

template <int CCMC, bool SH>
static hipError_t launch_chain_out(const ChainParams& P, int batch, int out_kind, const float* src, void* dst,
                                   hipStream_t st) {
  auto grid_for = [&](int G) {
    const uint32_t K = 4 / G;
    const uint64_t groups = (uint64_t)P.n_px / G;
    uint64_t blocks = (groups + 256 * K - 1) / (256 * K);
    if (blocks < 1) blocks = 1;
    // one or two trips per workgroup stream best; kernels that stage a coefficient table in LDS per
    // workgroup amortise that over more trips
    const uint64_t cap = ccm_needs_table<CCMC> ? 4096 : 32768;  // (16384 ... 131072 perform alike, 4096 is 5 % slower)
    if (blocks > cap) blocks = cap;
    return dim3((unsigned)blocks, (unsigned)batch, 1);
  };
  switch (out_kind) {
    case SS_OUT_F32_LINEAR:
      hipLaunchKernelGGL((chain_kernel<CCMC, SH, SS_OUT_F32_LINEAR>), grid_for(1), dim3(256), 0, st, P, src, dst);
      break;
    case SS_OUT_F32_SRGB:
      hipLaunchKernelGGL((chain_kernel<CCMC, SH, SS_OUT_F32_SRGB>), grid_for(1), dim3(256), 0, st, P, src, dst);
      break;
    case SS_OUT_U16_SRGB:
      hipLaunchKernelGGL((chain_kernel<CCMC, SH, SS_OUT_U16_SRGB>), grid_for(2), dim3(256), 0, st, P, src, dst);
      break;
    case SS_OUT_U8_SRGB:
      hipLaunchKernelGGL((chain_kernel<CCMC, SH, SS_OUT_U8_SRGB>), grid_for(4), dim3(256), 0, st, P, src, dst);
      break;
    case SS_OUT_U8_LINEAR:
      hipLaunchKernelGGL((chain_kernel<CCMC, SH, SS_OUT_U8_LINEAR>), grid_for(4), dim3(256), 0, st, P, src, dst);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int CCMC>
static hipError_t launch_chain_sh(const ChainParams& P, int batch, int out_kind, const float* src, void* dst,
                                  hipStream_t st) {
  return P.sh ? launch_chain_out<CCMC, true>(P, batch, out_kind, src, dst, st)
              : launch_chain_out<CCMC, false>(P, batch, out_kind, src, dst, st);
}

hipError_t launch_chain(const ChainParams& P, int batch, int out_kind, const float* src, void* dst, hipStream_t st) {
  if (P.ccm_kind == SS_CCM_NONE) return launch_chain_sh<CCM_NONE>(P, batch, out_kind, src, dst, st);
  if (P.ccm_terms == 3) return launch_chain_sh<CCM_LIN3>(P, batch, out_kind, src, dst, st);
  if (P.ccm_terms <= 8) return launch_chain_sh<CCM_SMALL>(P, batch, out_kind, src, dst, st);
  if (P.ccm_kind == SS_CCM_CHEUNG) return launch_chain_sh<CCM_CHEUNG>(P, batch, out_kind, src, dst, st);
  return launch_chain_sh<CCM_ROOT>(P, batch, out_kind, src, dst, st);
}

// ------------------------------------------------------------------------------------------------
// dense float64 colour correction (stages 1-2 of every pixel), for coefficient tables float32 cannot carry
// ------------------------------------------------------------------------------------------------

// stages 6-7 of one pixel and its store, as the pointwise kernel does them
template <int OUT>
__device__ __forceinline__ void finish_and_store(const Px v, const ChainParams& P, void* dst_img, uint32_t px) {
  store_one<OUT>(dst_img, px, finish<OUT, false, float>(v, P));
}

// The coefficients' float64 values, [channel][term]: the kernel's second argument.  The products take them as
// scalar operands, loaded from the kernel-argument segment next to their uses (66 values are 132 scalar registers:
// more than the file holds, so they are fetched again for every pixel through a pointer the compiler cannot see
// through; as LDS reads they kept the LDS pipe as busy as the float64 unit).  The launcher fills it from
// ccm_full + ccm_lo.
struct F64Coefs {
  double c[3][SS_MAX_CCM_TERMS];
};
// (the kernel finds it behind its first argument in the kernel-argument segment)
static_assert(alignof(ChainParams) <= 8 && alignof(F64Coefs) == 8, "second kernel argument is not at sizeof(ChainParams) rounded up to 8");

// A fit with nearly as many terms as swatches (22 terms, root-polynomials of degree 3 and 4 on 24 swatches) has
// coefficients of size 10 ... 10^5 whose products with the terms cancel to a pixel of size 1: float32 keeps such a
// sum to ~4e-7 of the sum of |coefficient x term|, several times the 1e-5 the result is held to.  The reference
// evaluates it in float64 (colour_correction converts its input first); so does this pass: negative inversion in
// float32 as the reference does it, expansion and sum in float64 with the coefficients' float64 values,
// the result rounded to float32 (unclipped linear RGB) for the stages behind it.
// One trip of 1024 pixels per workgroup, lane t taking pixels t, t + 256, ... like the pointwise chain.
// `out_kind` < 0: the result goes to `dst` as it is (a float32 staging image for the kernels of the stages behind).
// `out_kind` >= 0: nothing but gains, the final clip and the output encoding are behind the correction, and the pass
// finishes the chain itself - the same float32 operations in the same order as the pointwise kernel applies to a
// staged pixel, so both routes give the same bits - and stores in the output's format (and the 8-bit second
// output, when the call has one), a pixel per lane.
// (KIND, T: the expansion, compile-time so that the terms live in registers)
template <int KIND, int T>
__global__ __launch_bounds__(256) void ccm_f64_kernel(const ChainParams P, const F64Coefs C_arg,
                                                      const float* __restrict__ src, void* __restrict__ dst, int out_kind) {
  const int img = blockIdx.y;
  const ImgParams ip = load_img_params(P, img);
  const uint32_t n = (uint32_t)P.n_px;
  const F3* __restrict__ s = reinterpret_cast<const F3*>(src + (size_t)img * n * 3);
  const size_t out_px_bytes = out_kind < 0 || out_kind == SS_OUT_F32_LINEAR || out_kind == SS_OUT_F32_SRGB
                                  ? 12 : (out_kind == SS_OUT_U16_SRGB ? 6 : 3);
  char* d = reinterpret_cast<char*>(dst) + (size_t)img * n * out_px_bytes;
  for (uint32_t base = blockIdx.x * 1024u; base < n; base += gridDim.x * 1024u) {
    F3 in[4] = {};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t p = base + k * 256 + threadIdx.x;
      if (p < n) in[k] = load_stream(s + p);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      __builtin_amdgcn_sched_barrier(0);  // one pixel's terms at a time: 2 T registers
      const uint32_t p = base + k * 256 + threadIdx.x;
      float x[3] = {in[k].x, in[k].y, in[k].z};
      if (P.neg) {
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = 1.0f - clamp01((clamp01(x[c]) - ip.neg_lo[c]) * ip.neg_inv[c]);
      }
      double e[SS_MAX_CCM_TERMS];
      careful::expand_shared_roots(KIND, T, (double)x[0], (double)x[1], (double)x[2], e);
      typedef const __attribute__((address_space(4))) F64Coefs* KernargCoefs;
      KernargCoefs kc = (KernargCoefs)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                       ((sizeof(ChainParams) + 7) & ~(size_t)7));
      double o0 = 0.0, o1 = 0.0, o2 = 0.0;
      asm volatile("" : "+s"(kc));
#pragma unroll
      for (int t = 0; t < T; ++t) o0 = fma(kc->c[0][t], e[t], o0);
#pragma unroll
      for (int t = 0; t < T; ++t) o1 = fma(kc->c[1][t], e[t], o1);
#pragma unroll
      for (int t = 0; t < T; ++t) o2 = fma(kc->c[2][t], e[t], o2);
      if (p >= n) continue;
      Px v{(float)o0, (float)o1, (float)o2};
      if (out_kind < 0) {
        store_stream(reinterpret_cast<F3*>(d) + p, F3{v.r, v.g, v.b});
        continue;
      }
      if (P.n_gain > 0) {
        v = Px{clamp01(v.r * ip.gain0[0]), clamp01(v.g * ip.gain0[1]), clamp01(v.b * ip.gain0[2])};
        if (P.n_gain > 1) {
          v = Px{clamp01(v.r * P.gain[1][0]), clamp01(v.g * P.gain[1][1]), clamp01(v.b * P.gain[1][2])};
          if (P.n_gain > 2) v = Px{clamp01(v.r * P.gain[2][0]), clamp01(v.g * P.gain[2][1]), clamp01(v.b * P.gain[2][2])};
        }
      }
      if (P.dst2 != nullptr) store_second(P.dst2, (size_t)img * n + p, v, P.final_clip != 0, P.dst2_encoded != 0);
      switch (out_kind) {
        case SS_OUT_F32_LINEAR: finish_and_store<SS_OUT_F32_LINEAR>(v, P, d, p); break;
        case SS_OUT_F32_SRGB: finish_and_store<SS_OUT_F32_SRGB>(v, P, d, p); break;
        case SS_OUT_U16_SRGB: finish_and_store<SS_OUT_U16_SRGB>(v, P, d, p); break;
        case SS_OUT_U8_SRGB: finish_and_store<SS_OUT_U8_SRGB>(v, P, d, p); break;
        default: finish_and_store<SS_OUT_U8_LINEAR>(v, P, d, p); break;
      }
    }
  }
}

hipError_t launch_ccm_f64(const ChainParams& P, int batch, const float* src, void* dst, int out_kind, hipStream_t st) {
  uint64_t blocks = ((uint64_t)P.n_px + 1023) / 1024;
  if (blocks < 1) blocks = 1;
  if (blocks > 32768) blocks = 32768;
  const dim3 grid((unsigned)blocks, (unsigned)batch, 1);
  F64Coefs C{};
  for (int t = 0; t < P.ccm_terms; ++t)
    for (int ch = 0; ch < 3; ++ch) C.c[ch][t] = (double)P.ccm_full[t][ch] + (double)P.ccm_lo[t][ch];
#define SS_F64_CASE(KIND, T)                                                                          \
  if (P.ccm_kind == KIND && P.ccm_terms == T) {                                                       \
    hipLaunchKernelGGL((ccm_f64_kernel<KIND, T>), grid, dim3(256), 0, st, P, C, src, dst, out_kind);  \
    return hipGetLastError();                                                                         \
  }
  SS_F64_CASE(SS_CCM_CHEUNG, 5) SS_F64_CASE(SS_CCM_CHEUNG, 7) SS_F64_CASE(SS_CCM_CHEUNG, 8) SS_F64_CASE(SS_CCM_CHEUNG, 10)
  SS_F64_CASE(SS_CCM_CHEUNG, 11) SS_F64_CASE(SS_CCM_CHEUNG, 14) SS_F64_CASE(SS_CCM_CHEUNG, 16) SS_F64_CASE(SS_CCM_CHEUNG, 17)
  SS_F64_CASE(SS_CCM_CHEUNG, 19) SS_F64_CASE(SS_CCM_CHEUNG, 20) SS_F64_CASE(SS_CCM_CHEUNG, 22)
  SS_F64_CASE(SS_CCM_ROOTPOLY, 6) SS_F64_CASE(SS_CCM_ROOTPOLY, 13) SS_F64_CASE(SS_CCM_ROOTPOLY, 22)
#undef SS_F64_CASE
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// float64 re-evaluation of the queued pixels (ss_careful.h)
// ------------------------------------------------------------------------------------------------

// One workgroup takes 256 words of the record at a time, counts their bits, and deals the recorded pixels out
// evenly over its threads (prefix sums in LDS), so a word with many recorded pixels does not hold a wave up.
// Pixels are evaluated 256 at a time, one per thread; what a chunk leaves over (fewer than 256) is carried into
// the next chunk's list instead of running as a mostly idle trip (with 2 % of the pixels recorded a chunk holds
// about 330: a full trip and one at 28 % without the carry).
__global__ __launch_bounds__(256) void careful_kernel(const ChainParams P, int layout, int batch,
                                                      const float* __restrict__ src, void* __restrict__ dst, int out_kind) {
  if (*P.risk_any == 0u) return;
  const ChainParams* Pk = (const ChainParams*)__builtin_amdgcn_kernarg_segment_ptr();  // P is the first argument
  const bool enc = out_kind != SS_OUT_F32_LINEAR && out_kind != SS_OUT_U8_LINEAR;
  const int W = P.width, H = P.height;
  const uint32_t tiles_x = (uint32_t)(W + kRiskTileWidth - 1) / kRiskTileWidth;
  const size_t wpi = risk_words_per_image(layout, H, W);
  const size_t n_words = wpi * (size_t)batch;
  __shared__ unsigned long long words[256];
  __shared__ uint32_t start[257];
  __shared__ uint32_t wave_total[4];
  __shared__ unsigned long long carry[256];  // pixels not evaluated yet: image << 32 | pixel inside the image
  uint32_t carried = 0;                      // (the same in every thread)

  // the i-th recorded pixel of the chunk whose scan is in start[] / words[]
  auto locate = [&](size_t chunk, uint32_t i) -> unsigned long long {
    uint32_t lo = 0, hi = 256;  // the word that holds it: start[lo] <= i < start[lo + 1]
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) / 2;
      if (start[mid] <= i) lo = mid; else hi = mid;
    }
    unsigned long long ww = words[lo];
    for (uint32_t k = i - start[lo]; k > 0; --k) ww &= ww - 1ull;  // drop the k lowest set bits
    const uint32_t bit = (uint32_t)__ffsll((long long)ww) - 1u;
    const size_t gw = chunk + lo;
    const uint32_t img = (uint32_t)(gw / wpi);
    const size_t wi = gw - (size_t)img * wpi;
    uint32_t px;
    if (layout == RISK_LINEAR) {
      px = (uint32_t)wi * 64u + bit;
    } else {
      const uint32_t y = (uint32_t)(wi / (tiles_x * 2u)), r2 = (uint32_t)(wi - (size_t)y * tiles_x * 2u);
      px = y * (uint32_t)W + (r2 >> 1) * (uint32_t)kRiskTileWidth + 2u * bit + (r2 & 1u);
    }
    return ((unsigned long long)img << 32) | px;
  };
  // A recorded pixel's inputs (its source pixel and, behind an unsharp mask, the blurred lightness the tile kernel left
  // in its output slot) are fetched one pixel ahead of the arithmetic: the scattered loads of the next pixel are in
  // flight during the ~thousand float64 instructions of this one, where fetching and evaluating in turn left the
  // kernel waiting for memory two thirds of the time.
  struct Fetched {
    size_t at;
    uint32_t img;
    float sr, sg, sb, Lb;
  };
  auto fetch = [&](unsigned long long entry) -> Fetched {
    Fetched f;
    f.img = (uint32_t)(entry >> 32);
    const uint32_t px = (uint32_t)entry;
    f.at = (size_t)f.img * (size_t)P.n_px + px;
    const float* sp = src + f.at * 3;
    f.sr = sp[0], f.sg = sp[1], f.sb = sp[2];
    f.Lb = 0.0f;
    if (P.usm) {  // left by the tile kernel in the pixel's output slot
      if (out_kind == SS_OUT_U16_SRGB) {
        const uint16_t* d = reinterpret_cast<const uint16_t*>(dst) + f.at * 3;
        f.Lb = __uint_as_float((uint32_t)d[0] | ((uint32_t)d[1] << 16));
      } else {
        f.Lb = reinterpret_cast<const float*>(dst)[f.at * 3];
      }
    }
    return f;
  };
  auto evaluate = [&](const Fetched& f) {
    double v3[3];
    careful_pixel(Pk, (int)f.img, f.sr, f.sg, f.sb, f.Lb, enc, v3);
    if (out_kind == SS_OUT_U16_SRGB) {
      uint16_t* d = reinterpret_cast<uint16_t*>(dst) + f.at * 3;
      for (int c = 0; c < 3; ++c) d[c] = (uint16_t)fmin(fmax(v3[c] * 65535.0, 0.0), 65535.0);
    } else {
      float* d = reinterpret_cast<float*>(dst) + f.at * 3;
      d[0] = (float)v3[0], d[1] = (float)v3[1], d[2] = (float)v3[2];
    }
  };

  for (size_t chunk = (size_t)blockIdx.x * 256; chunk < n_words; chunk += (size_t)gridDim.x * 256) {
    const size_t g = chunk + threadIdx.x;
    const unsigned long long w = g < n_words ? P.risk_bits[g] : 0ull;
    if (__syncthreads_or(w != 0ull) == 0) continue;
    words[threadIdx.x] = w;
    // inclusive scan of the bit counts: inside each wave by lane shifts, the four waves' totals through LDS
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = (uint32_t)__popcll(w);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t below = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t)d) incl += below;
    }
    if (lane == 63u) wave_total[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t k = 0; k < wave; ++k) before += wave_total[k];
    start[threadIdx.x + 1] = before + incl;
    if (threadIdx.x == 0) start[0] = 0;
    __syncthreads();
    const uint32_t total = start[256];
    if (threadIdx.x == 0) atomicAdd(P.risk_count, total);
    // the list: the carried pixels, then this chunk's; whole groups of 256 are evaluated now
    const uint32_t listed = carried + total, whole = listed & ~255u;
    {
      auto entry_of = [&](uint32_t v) -> unsigned long long { return v < carried ? carry[v] : locate(chunk, v - carried); };
      uint32_t v = threadIdx.x;
      bool have = v < whole;
      Fetched cur{};
      if (have) cur = fetch(entry_of(v));
      while (have) {
        const uint32_t vn = v + 256;
        const bool have_next = vn < whole;
        Fetched next{};
        if (have_next) next = fetch(entry_of(vn));
        evaluate(cur);
        cur = next;
        have = have_next;
        v = vn;
      }
    }
    const uint32_t rest = whole + threadIdx.x;  // what is left becomes the carry
    unsigned long long keep = 0ull;
    if (rest < listed) keep = rest < carried ? carry[rest] : locate(chunk, rest - carried);
    __syncthreads();  // carry[], words[] and start[] have been read by everyone
    if (rest < listed) carry[threadIdx.x] = keep;
    carried = listed - whole;
  }
  __syncthreads();
  if (threadIdx.x < carried) evaluate(fetch(carry[threadIdx.x]));
}

hipError_t launch_careful(const ChainParams& P, int batch, int out_kind, const float* src, void* dst, hipStream_t st) {
  const int layout = P.usm ? RISK_TILES : RISK_LINEAR;
  const size_t n_words = risk_words_per_image(layout, P.height, P.width) * (size_t)batch;
  size_t blocks = (n_words + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(careful_kernel, dim3((unsigned)blocks), dim3(256), 0, st, P, layout, batch, src, dst, out_kind);
  return hipGetLastError();
}

bool chain_needs_careful(const ChainParams& P, int out_kind) {
  // (chains on Lab data never record a pixel, but their kernels write the - empty - record all the same)
  if (!(P.sh || P.usm)) return false;
  return out_kind == SS_OUT_F32_LINEAR || out_kind == SS_OUT_F32_SRGB || out_kind == SS_OUT_U16_SRGB;
}

}  // namespace ss
