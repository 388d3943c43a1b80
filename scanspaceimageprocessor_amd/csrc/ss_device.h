// Device-side pixel pipeline shared by the pointwise chain kernel and the fused
// unsharp-mask kernel.  gfx950 only: transcendental work goes straight to the
// hardware v_log_f32 / v_exp_f32 / v_sqrt_f32 instructions (1 ulp each), which
// keeps the chain HBM-bound; all stage parameters arrive as one by-value kernel
// argument so that they live in SGPRs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/scanspace.h"

namespace ss {

// Everything a kernel needs, derived on the host from ss_chain_desc.
struct ChainParams {
  int32_t height, width;
  int64_t n_px;  // pixels per image

  int32_t neg;
  float neg_lo[3], neg_inv[3];
  const float* neg_dev;  // [batch][8]

  int32_t ccm_kind, ccm_terms;
  float ccm3[9];  // the 3-term matrix (the reference's only call form), read straight from SGPRs
  float ccm_small[3 * 8];  // row-major 3 x T for 3 < T <= 8 (Cheung 5/7/8, root-polynomial degree 2), also SGPRs

  int32_t n_gain;
  float gain[SS_MAX_GAIN_STAGES][3];
  const float* gain0_dev;  // [batch]

  int32_t sh;
  float sh_ss, sh_hs;  // amounts * 0.05
  float sh_log2f;      // log2(falloff_power)
  int32_t sh_chroma;   // |amount| > 50 on either side

  int32_t usm;
  int32_t usm_r;            // taps = 2r+1
  float usm_amount_scaled;  // amount / 50
  float usm_thr;            // threshold / 100
  float usm_inv_thr;        // 1 / usm_thr (0 when usm_thr == 0)
  int32_t usm_chroma;       // amount > 30
  float usm_chroma_k;       // amount / 100 * 0.1
  float taps[SS_MAX_USM_RADIUS + 1];  // taps[0..r]: the Gaussian is symmetric, tap j == tap 2r-j

  int32_t lab_io;
  int32_t final_clip;

  // Full 3 x T table for T > 3, as [t] = {row R, row G, row B, 0}.  Kept LAST and only ever read with a
  // runtime index (staged into LDS at kernel start) so that it never occupies scalar registers.
  float ccm_full[SS_MAX_CCM_TERMS][4];

  // Read by the float64 evaluation of single pixels only (ss_careful.h), through a pointer to the kernel-argument
  // segment: the stage arguments as the caller gave them, the low-order parts of the colour-correction
  // coefficients (ccm_full + ccm_lo = the float64 coefficient), and the largest B (see lab_edit_at_risk) the
  // shadows/highlights stage of this chain can produce.
  float sh_shadow_amount, sh_highlight_amount, sh_falloff, usm_amount, usm_threshold;
  float sh_risk_bound;
  float ccm_lo[SS_MAX_CCM_TERMS][4];
  // Record of the pixels to evaluate again (see RiskMap below): set by the launcher, read in the rare branch only
  unsigned long long* risk_bits;
  uint32_t* risk_any;
  uint32_t* risk_count;
  // Second output of the same pass: 3 bytes per pixel, [batch][H][W][3], made from the pixel as it stands before
  // the output encoding (null = none).  Set by the launcher, read where the kernels store (through kernarg_params).
  unsigned char* dst2;
  int dst2_encoded;  // 1 = SS_OUT_U8_SRGB, 0 = SS_OUT_U8_LINEAR
  double sh_ln_falloff;  // ln(sh_falloff) in float64, for the float64 re-evaluation (falloff ** x = exp(x ln falloff))
};

// Single operations with one rounding each, where the reference's numpy arithmetic must be followed to the bit.
// (HIP's __fmul_rn / __dadd_rn and friends are plain operators on this toolchain, so the compiler fuses them into
// multiply-adds like any other expression; the pragma is what keeps them apart, also after inlining.)
template <class T>
__device__ __forceinline__ T mul_rn(T x, T y) {
#pragma clang fp contract(off)
  return x * y;
}
template <class T>
__device__ __forceinline__ T add_rn(T x, T y) {
#pragma clang fp contract(off)
  return x + y;
}
template <class T>
__device__ __forceinline__ T sub_rn(T x, T y) {
#pragma clang fp contract(off)
  return x - y;
}

// The kernel's parameter block as its kernel-argument segment holds it (the block is every kernel's FIRST argument),
// behind a pointer the compiler cannot see through: scalars read through it are loaded where they are used instead
// of living in scalar registers (spilled to vector lanes, each reload a vector instruction) across a whole loop.
__device__ __forceinline__ const ChainParams& kernarg_params() {
  typedef const __attribute__((address_space(4))) ChainParams* KernargParams;
  KernargParams k = (KernargParams)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));
  return *(const ChainParams*)k;
}

// sRGB <-> XYZ of colour-science's sRGB colourspace: IEC 61966-2-1:1999 gives both matrices to four decimals
// and colour carries both as given, so they are not an exact inverse pair (XYZ->RGB . RGB->XYZ = I + E with
// |E| up to 4e-5: a pixel that goes through _rgb_to_lab and _lab_to_rgb unedited comes back as (I + E) . rgb).
// Folded with the D65 white (xy = 0.3127, 0.3290) of XYZ_to_Lab / Lab_to_XYZ: `fwd` rows are divided by the
// white, `inv` columns multiplied by it; `drift` = E.  Compile-time constants: as kernel arguments they (and
// the register pairs packed arithmetic wants for them) crowded the scalar register file.
struct LabMatrices {
  float fwd[9];
  float inv[9];
  float drift[9];
};

constexpr LabMatrices make_lab_matrices() {
  const double m[9] = {0.4124, 0.3576, 0.1805, 0.2126, 0.7152, 0.0722, 0.0193, 0.1192, 0.9505};
  const double inv[9] = {3.2406, -1.5372, -0.4986, -0.9689, 1.8758, 0.0415, 0.0557, -0.2040, 1.0570};
  const double white[3] = {0.3127 / 0.3290, 1.0, (1.0 - 0.3127 - 0.3290) / 0.3290};
  LabMatrices o{};
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) {
      o.fwd[r * 3 + k] = (float)(m[r * 3 + k] / white[r]);
      o.inv[r * 3 + k] = (float)(inv[r * 3 + k] * white[k]);
      double rt = 0.0;
      for (int j = 0; j < 3; ++j) rt += inv[r * 3 + j] * m[j * 3 + k];
      const double e = rt - (r == k ? 1.0 : 0.0);
      o.drift[r * 3 + k] = (e < 1e-12 && e > -1e-12) ? 0.0f : (float)e;  // (one entry is 0 in decimal arithmetic)
    }
  return o;
}

constexpr LabMatrices kLab = make_lab_matrices();

// symmetric tap lookup with a compile-time index
template <int R>
__device__ __forceinline__ float tap_at(const ChainParams& P, int j) {
  return P.taps[j <= R ? j : 2 * R - j];
}

// ---- value types ----------------------------------------------------------------------------
// Every stage is written once for V = float (one pixel per lane) and V = v2f (two pixels per lane in
// lock step).  On gfx950 a plain VALU instruction issues in ~4 cycles per wave64 while the packed
// v_pk_mul/add/fma_f32 forms do two lanes' worth of work in ~5, so the pair form cuts the cost of all
// multiply/add/fma work by about a third; comparisons, selects, clamps and transcendentals stay per element.

typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

template <class V>
struct PxT {
  V r, g, b;
};
template <class V>
struct LabT {
  V L, a, b;
};
using Px = PxT<float>;
using Lab = LabT<float>;
using Px2 = PxT<v2f>;
using Lab2 = LabT<v2f>;

__device__ __forceinline__ Px2 pair_px(const Px a, const Px b) { return Px2{v2f{a.r, b.r}, v2f{a.g, b.g}, v2f{a.b, b.b}}; }
__device__ __forceinline__ Px px_lo(const Px2 p) { return Px{p.r.x, p.g.x, p.b.x}; }
__device__ __forceinline__ Px px_hi(const Px2 p) { return Px{p.r.y, p.g.y, p.b.y}; }

// a * b + c with any mix of V and float operands (floats are splatted)
template <class V, class A, class B, class C>
__device__ __forceinline__ V mad(A a, B b, C c) {
  return __builtin_elementwise_fma((V)(a), (V)(b), (V)(c));
}

// ---- per-element helpers ----------------------------------------------------------------------

__device__ __forceinline__ float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ float med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ float vabs(float x) { return fabsf(x); }
__device__ __forceinline__ float vcopysign(float m, float s) { return copysignf(m, s); }

__device__ __forceinline__ v2f hw_log2(v2f x) { return v2f{hw_log2(x.x), hw_log2(x.y)}; }
__device__ __forceinline__ v2f hw_exp2(v2f x) { return v2f{hw_exp2(x.x), hw_exp2(x.y)}; }
__device__ __forceinline__ v2f hw_sqrt(v2f x) { return v2f{hw_sqrt(x.x), hw_sqrt(x.y)}; }
__device__ __forceinline__ v2f clamp01(v2f x) { return v2f{clamp01(x.x), clamp01(x.y)}; }
__device__ __forceinline__ v2f clampf(v2f x, float lo, float hi) { return v2f{clampf(x.x, lo, hi), clampf(x.y, lo, hi)}; }
__device__ __forceinline__ v2f med3(v2f x, v2f lo, v2f hi) { return v2f{med3(x.x, lo.x, hi.x), med3(x.y, lo.y, hi.y)}; }
__device__ __forceinline__ v2f vabs(v2f x) { return v2f{fabsf(x.x), fabsf(x.y)}; }
__device__ __forceinline__ v2f vcopysign(v2f m, v2f s) { return v2f{copysignf(m.x, s.x), copysignf(m.y, s.y)}; }

// clip(a - b, 0, 1), clip(a * b, 0, 1), clip(a * b + c, 0, 1) as ONE instruction: the clamp output modifier is free,
// where a separate v_max / v_min / v_med3 costs 1.8 times an add on this GPU.  The compiler folds the clip of a
// single float into its producer by itself; for pairs (v_pk_*_f32) it has to be spelled out.  `b`, `c` of the pair
// forms may be wave-uniform (scalar register pairs).
__device__ __forceinline__ float sub_clip01(float a, float b) { return clamp01(a - b); }
__device__ __forceinline__ float mul_clip01(float a, float b) { return clamp01(a * b); }
__device__ __forceinline__ float fma_clip01(float a, float b, float c) { return clamp01(fmaf(a, b, c)); }
__device__ __forceinline__ v2f sub_clip01(v2f a, v2f b) {
  v2f r;
  asm("s_nop 0\n\tv_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ v2f sub_from_const_clip01(float a, v2f b) {  // clip(a - b, 0, 1), a wave-uniform
  v2f r;
  const v2f a2 = {a, a};
  asm("s_nop 0\n\tv_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(r) : "s"(a2), "v"(b));
  return r;
}
__device__ __forceinline__ float sub_from_const_clip01(float a, float b) { return clamp01(a - b); }
__device__ __forceinline__ v2f mul_clip01(v2f a, float b) {
  v2f r;
  const v2f b2 = {b, b};
  asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(b2));
  return r;
}
__device__ __forceinline__ v2f fma_clip01(v2f a, float b, float c) {
  v2f r;
  const v2f b2 = {b, b}, c2 = {c, c};
  // (the compiler cannot see into the asm: when `a` comes straight from v_exp_f32 / v_log_f32 / v_sqrt_f32 the
  // wait state those need before their result is read has to be here)
  asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "s"(b2), "v"(c2));
  return r;
}
// clip(a * b + c, 0, 1), a and b per lane, c wave-uniform
__device__ __forceinline__ float fma2_clip01(float a, float b, float c) { return clamp01(fmaf(a, b, c)); }
__device__ __forceinline__ v2f fma2_clip01(v2f a, v2f b, float c) {
  v2f r;
  const v2f c2 = {c, c};
  asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "s"(c2));
  return r;
}
// colour.algebra.spow with p = 1/2, 1/3, 1/4: sign(x) * |x| ** p, exact zero at zero
template <class V>
__device__ __forceinline__ V sroot2(V x) { return vcopysign(hw_sqrt(vabs(x)), x); }
template <class V>
__device__ __forceinline__ V sroot3(V x) { return vcopysign(hw_exp2(hw_log2(vabs(x)) * (1.0f / 3.0f)), x); }
template <class V>
__device__ __forceinline__ V sroot4(V x) { return vcopysign(hw_sqrt(hw_sqrt(vabs(x))), x); }

// x <= k per lane as a mask in a scalar register pair of its own, and the select that reads it: the compare sits
// ahead of the logarithm and exponential of its branch instead of right in front of the select (which, reading vcc
// behind the compare that wrote it, waits for it), and several can be in flight.  Used by the plain tile kernels
// (SG = true below: 0.3-0.4 % of their time, same bits); the kernels with more stages have no scalar registers to
// spare and spill vector registers with it.
__device__ __forceinline__ unsigned long long mask_le(float x, float k) {
  unsigned long long m;
  asm("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m) : "s"(k), "v"(x));
  return m;
}
// AFTER_TRANS: an operand may come straight from v_exp_f32 / v_log_f32, whose result needs a wait state before it is
// read; the compiler cannot see into the asm, so the wait state is spelled out.
template <bool AFTER_TRANS = false>
__device__ __forceinline__ float select_by(unsigned long long m, float if_set, float if_clear) {
  float r;
  if constexpr (AFTER_TRANS) {
    asm("s_nop 0\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
  } else {
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
  }
  return r;
}
struct Mask2 { unsigned long long x, y; };
__device__ __forceinline__ Mask2 mask_le(v2f v, float k) { return Mask2{mask_le(v.x, k), mask_le(v.y, k)}; }
template <bool AFTER_TRANS = false>
__device__ __forceinline__ v2f select_by(Mask2 m, v2f if_set, v2f if_clear) {
  return v2f{select_by<AFTER_TRANS>(m.x, if_set.x, if_clear.x), select_by<AFTER_TRANS>(m.y, if_set.y, if_clear.y)};
}

// sRGB inverse EOTF (colour.cctf_encoding): linear -> encoded.  x ** p = exp2(p * log2 x) on the power
// segment only (x > 0 there); everything else, negatives included, takes the linear segment.
template <class V>
__device__ __forceinline__ V srgb_encode(V L) {
  const V hi = mad<V>(hw_exp2(hw_log2(L) * (1.0f / 2.4f)), 1.055f, -0.055f);
  const V lo = L * 12.92f;
  return L <= (V)(0.0031308f) ? lo : hi;
}

// clip(encode(L), 0, 1) with the clip applied to the two branches, where it is the clamp output modifier of the
// multiply / multiply-add that makes them (free), instead of an instruction of its own behind the select
template <class V, bool SG = false>
__device__ __forceinline__ V srgb_encode_clip(V L) {
  if constexpr (SG) {
    const auto m = mask_le(L, 0.0031308f);
    const V hi = fma_clip01(hw_exp2(hw_log2(L) * (1.0f / 2.4f)), 1.055f, -0.055f);
    const V lo = mul_clip01(L, 12.92f);
    return select_by(m, lo, hi);
  } else {
    const V hi = fma_clip01(hw_exp2(hw_log2(L) * (1.0f / 2.4f)), 1.055f, -0.055f);
    const V lo = mul_clip01(L, 12.92f);
    return L <= (V)(0.0031308f) ? lo : hi;
  }
}

// sRGB EOTF (colour.cctf_decoding): encoded -> linear
template <class V, bool SG = false>
__device__ __forceinline__ V srgb_decode(V v) {
  if constexpr (SG) {
    const auto m = mask_le(v, 0.040449936f);
    const V t = mad<V>(v, 1.0f / 1.055f, 0.055f / 1.055f);
    const V hi = hw_exp2(hw_log2(t) * 2.4f);
    const V lo = v * (1.0f / 12.92f);
    return select_by<true>(m, lo, hi);
  }
  const V t = mad<V>(v, 1.0f / 1.055f, 0.055f / 1.055f);
  const V hi = hw_exp2(hw_log2(t) * 2.4f);
  const V lo = v * (1.0f / 12.92f);
  return v <= (V)(0.040449936f) ? lo : hi;
}

// CIE 1976 intermediate lightness function and its inverse
template <class V>
__device__ __forceinline__ V lab_f(V t) {
  const V hi = hw_exp2(hw_log2(t) * (1.0f / 3.0f));
  const V lo = mad<V>(t, 841.0f / 108.0f, 16.0f / 116.0f);
  // The line is the tangent of the cube root at the junction (value 6/29 there) and the cube root is concave:
  // hi <= lo everywhere, both <= 6/29 below the junction and >= 6/29 above it, so the selected branch is the
  // median of {hi, lo, 6/29}: one instruction instead of a compare and a select.  (t >= 0 here: X, Y, Z of
  // non-negative linear RGB.)
  return med3(hi, lo, (V)(24.0f / 116.0f));
}
template <class V>
__device__ __forceinline__ V lab_finv(V f) {
  const V hi = f * f * f;
  const V lo = (f - 16.0f / 116.0f) * (108.0f / 841.0f);
  return f > (V)(24.0f / 116.0f) ? hi : lo;
}

// ---- colour space conversions (the reference treats its data as sRGB-ENCODED here) ----------

template <class V>
__device__ __forceinline__ LabT<V> rgb_to_lab(const PxT<V> p, const ChainParams& P) {
  const V r = srgb_decode(p.r), g = srgb_decode(p.g), b = srgb_decode(p.b);
  const V fx = lab_f(mad<V>(kLab.fwd[0], r, mad<V>(kLab.fwd[1], g, b * kLab.fwd[2])));
  const V fy = lab_f(mad<V>(kLab.fwd[3], r, mad<V>(kLab.fwd[4], g, b * kLab.fwd[5])));
  const V fz = lab_f(mad<V>(kLab.fwd[6], r, mad<V>(kLab.fwd[7], g, b * kLab.fwd[8])));
  LabT<V> o;
  o.L = mad<V>(fy, 116.0f, -16.0f);
  o.a = (fx - fy) * 500.0f;
  o.b = (fy - fz) * 200.0f;
  return o;
}

// _lab_to_rgb: Lab -> XYZ -> sRGB (encoded) -> clip
template <class V>
__device__ __forceinline__ PxT<V> lab_to_rgb(const LabT<V> q, const ChainParams& P) {
  const V fy = mad<V>(q.L, 1.0f / 116.0f, 16.0f / 116.0f);
  const V fx = mad<V>(q.a, 1.0f / 500.0f, fy);
  const V fz = mad<V>(q.b, -1.0f / 200.0f, fy);
  const V x = lab_finv(fx), y = lab_finv(fy), z = lab_finv(fz);
  PxT<V> o;
  o.r = clamp01(srgb_encode(mad<V>(kLab.inv[0], x, mad<V>(kLab.inv[1], y, z * kLab.inv[2]))));
  o.g = clamp01(srgb_encode(mad<V>(kLab.inv[3], x, mad<V>(kLab.inv[4], y, z * kLab.inv[5]))));
  o.b = clamp01(srgb_encode(mad<V>(kLab.inv[6], x, mad<V>(kLab.inv[7], y, z * kLab.inv[8]))));
  return o;
}

// ---- Lab edits on RGB data without the round-trip error -----------------------------------------
// The reference edits L (and scales a, b) between _rgb_to_lab and _lab_to_rgb in float64, where an
// unedited pixel comes back as (I + E) rgb to 1e-16.  In float32 the literal round trip loses ~1e-7 absolute per linear
// channel (X, Y, Z ~ 0.5 cancel to a dark channel through the inverse matrix), which two successive sRGB
// encodes amplify 167-fold near black - several 16-bit steps.  So the edit is applied as a difference:
//   linear' = (I + E) linear + M_inv * (g(f + df) - g(f)),   g = lab_finv,  f = (fx, fy, fz) of the pixel,
// whose error is proportional to the size of the edit, not to the size of XYZ.  (E: see LabMatrices; the
// reference's own unedited round trip is (I + E) linear in exact arithmetic.)

template <class V>
struct LabFT {
  V fx, fy, fz;
};

// (fx, fy, fz) of linear RGB
template <class V>
__device__ __forceinline__ LabFT<V> linear_to_labf(const PxT<V> lin, const ChainParams& P) {
  LabFT<V> f;
  f.fx = lab_f(mad<V>(kLab.fwd[0], lin.r, mad<V>(kLab.fwd[1], lin.g, lin.b * kLab.fwd[2])));
  f.fy = lab_f(mad<V>(kLab.fwd[3], lin.r, mad<V>(kLab.fwd[4], lin.g, lin.b * kLab.fwd[5])));
  f.fz = lab_f(mad<V>(kLab.fwd[6], lin.r, mad<V>(kLab.fwd[7], lin.g, lin.b * kLab.fwd[8])));
  return f;
}

__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ v2f vmax(v2f a, v2f b) { return v2f{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ v2f vmin(v2f a, v2f b) { return v2f{fminf(a.x, b.x), fminf(a.y, b.y)}; }

__device__ __forceinline__ float vmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ v2f vmin3(v2f a, v2f b, v2f c) { return v2f{vmin3(a.x, b.x, c.x), vmin3(a.y, b.y, c.y)}; }
__device__ __forceinline__ float vmax3abs(float a, float b, float c) { return fmaxf(fmaxf(fabsf(a), fabsf(b)), fabsf(c)); }
__device__ __forceinline__ v2f vmax3abs(v2f a, v2f b, v2f c) {
  return v2f{vmax3abs(a.x, b.x, c.x), vmax3abs(a.y, b.y, c.y)};
}

// ---- which Lab edits float32 cannot follow ---------------------------------------------------------
// The edited pixel is linear + M_inv (g(f + d) - g(f)): terms of a size up to B = 3.3 mx, mx = max |g(f + d) - g(f)|
// over X, Y, Z (3.3 ~ the largest row sum of |M_inv|), that can cancel to a channel value o near zero.  float32
// keeps the sum to ~3e-7 B absolute; the sRGB encodes that follow multiply that by their slopes: the edit's own
// encode (12.92 below 0.0031, ~0.44 o^-0.58 above) and, for encoded outputs, the output encode on top (together
// min(167, ~0.14 / o)).  Held against a third of the output's tolerance (1e-5 on float data, one step on 16 bits)
// this says a pixel is at risk when mx > MIN_MX and one of its edited channels lies in (-4e-6, mx / RATIO).
// Such pixels are recorded and evaluated again in float64 by a second kernel (ss_careful.h).  8-bit outputs never
// need it.
template <int OUT>
struct RiskCfg {
  static constexpr bool ENCODED = OUT == SS_OUT_F32_SRGB || OUT == SS_OUT_U16_SRGB;
  static constexpr bool ENABLED = OUT == SS_OUT_F32_LINEAR || ENCODED;
  static constexpr float MIN_MX = ENCODED ? 0.1f / 3.3f : 1.29f / 3.3f;
  static constexpr float RATIO = ENCODED ? 100.0f / 3.3f : 400.0f / 3.3f;
};
constexpr float kRiskBelowZero = -4.0e-6f;

// What an edit reports: its `mx` (the caller adds what an earlier edit of the same pixel contributed) and the
// edited linear pixel `o` before it is encoded.  The caller decides at the end of its pixel, in a branch
// that only waves holding a pixel with a large edit take (almost none), which pixels are at risk.
template <class V>
struct RiskT {
  V mx;
  PxT<V> o;
  bool any = true;  // wave-uniform: false = no pixel of this wave can have a large edit (mx was not formed)
};
__device__ __forceinline__ bool lane_above(float x, float lim) { return x > lim; }
__device__ __forceinline__ bool lane_above(v2f x, float lim) { return x.x > lim || x.y > lim; }
__device__ __forceinline__ bool risk_channel(float o, float lim) { return o > kRiskBelowZero && o < lim; }
template <int OUT>
__device__ __forceinline__ bool risk_big(float mx) {
  return RiskCfg<OUT>::ENABLED && mx > RiskCfg<OUT>::MIN_MX;
}
// (call where risk_big holds)
template <int OUT>
__device__ __forceinline__ bool risk_cancelled(float mx, float r, float g, float b) {
  const float lim = mx * (1.0f / RiskCfg<OUT>::RATIO);
  return risk_channel(r, lim) || risk_channel(g, lim) || risk_channel(b, lim);
}

// A shadows/highlights edit in front of the unsharp mask hands its mx to the sharpen stage, which sees the pixel
// only after the edit, as a 3-bit level in the signs of the three channel values (the edit returns them clipped to
// [0, 1], so the signs are free): level 0 = below 2^-6 (nothing to carry), level L = [2^(L-7), 2^(L-6)), carried as
// its upper edge, level 7 = 1 or more, carried as the largest value the stage's settings allow.
__device__ __forceinline__ uint32_t risk_level(float mx) {
  const int e = (int)((__float_as_uint(mx) >> 23) & 255u) - 127;  // floor(log2 mx)
  const int l = e + 7;
  return (uint32_t)(l < 0 ? 0 : (l > 7 ? 7 : l));
}
__device__ __forceinline__ float risk_tag(float v, uint32_t level, int bit) {
  return __uint_as_float(__float_as_uint(v) | (((level >> bit) & 1u) << 31));
}
__device__ __forceinline__ float risk_carried(float c0, float c1, float c2, float bound_max) {
  const uint32_t l = (__float_as_uint(c0) >> 31) | ((__float_as_uint(c1) >> 31) << 1) | ((__float_as_uint(c2) >> 31) << 2);
  if (l == 0) return 0.0f;
  return l == 7 ? bound_max : __uint_as_float((l + 121u) << 23);  // 2^(l - 6)
}

// Record of the pixels to evaluate again: one bit per pixel, cleared by the launcher; the float32 kernels write a
// word only when it has a bit set, as a whole 64-bit word (a wave's ballot: no atomics), and it holds any number
// of pixels.  Two layouts, after how the kernels walk an image:
//   RISK_LINEAR  (pointwise kernel)  word w of an image = pixels [64 w, 64 w + 64), bit b = pixel 64 w + b;
//   RISK_TILES   (tile kernel)       words (y * tiles_x + t) * 2 + e, e = 0 / 1, of an image = row y of tile column t:
//                                    bit b of word e = pixel x = 128 t + 2 b + e.
// `any` is raised (to 1) by every wave that records a pixel; the launcher clears it, the float64 kernel returns at
// once when it is still 0.  For chains with an unsharp mask the blurred lightness of a recorded pixel waits in the
// pixel's own output slot (its first 4 bytes), which the float64 kernel overwrites with the result.
enum RiskLayout { RISK_LINEAR = 0, RISK_TILES = 1 };
struct RiskMap {
  unsigned long long* bits;
  uint32_t* any;
  uint32_t* count;  // total of recorded pixels, added up by the float64 kernel (statistics)
};
constexpr int kRiskTileWidth = 128;
__host__ __device__ inline size_t risk_words_per_image(int layout, int height, int width) {
  return layout == RISK_LINEAR ? ((size_t)height * width + 63) / 64
                               : (size_t)height * ((width + kRiskTileWidth - 1) / kRiskTileWidth) * 2;
}

// g(f + d) - g(f) on the cubic segment of g: d (3 f (f + d) + d^2), free of cancellation
template <class V>
__device__ __forceinline__ V finv_delta_cubic(V f, V d) {
  return d * mad<V>(d, d, (f * (f + d)) * 3.0f);
}
// ... and anywhere.  g is the cube with its tangent line at T = 6/29 substituted below T, i.e.
// g(f) = max(f, T)^3 + 3 T^2 min(f - T, 0).  With m = min(f - T, 0) and c = f - m = max(f, T):
//   g(f + d) - g(f) = dc (3 c c2 + dc^2) + 3 T^2 dm,   dm = m2 - m,  dc = d - dm.
// On the cubic segment dm = 0 exactly and dc = d exactly; below T the quantities are small.  No select.
template <class V>
__device__ __forceinline__ V finv_delta_any(V f, V d) {
  constexpr float T = 24.0f / 116.0f;
  const V f2 = f + d;
  // -m = max(T - f, 0) = clip(T - f, 0, 1) (f > T - 1 always): a subtraction with the clamp output modifier, where
  // a min costs 1.8 times an add on this GPU
  const V nm = sub_from_const_clip01(T, f), nm2 = sub_from_const_clip01(T, f2);
  const V dm = nm - nm2, dc = d - dm;
  const V c = f + nm, c2 = f2 + nm2;
  // dc (3 c c2 + dc^2) + 3 T^2 dm with dm = d - dc:  dc (3 c c2 - 3 T^2 + dc^2) + 3 T^2 d
  // (below T: c = c2 = T and the bracket is dc^2 with dc = 0; on the cubic segment dc = d exactly)
  return mad<V>(d, 3.0f * T * T, dc * mad<V>(dc, dc, mad<V>(c * c2, 3.0f, -3.0f * T * T)));
}

// a + drift[I] x: nothing at all for the entry that is zero
template <int I, class V>
__device__ __forceinline__ V drift_mad(V x, V a) {
  if constexpr (kLab.drift[I] == 0.0f) {
    return a;
  } else {
    return mad<V>(kLab.drift[I], x, a);
  }
}

// linear + M_inv * (g(f + df) - g(f)): the edited pixel before _lab_to_rgb encodes and clips it.
// `same_d_floor` > 0: the caller guarantees dfx = dfy = dfz and 0 <= linear <= 1, so that f <= 1.0001 and
// |g(f + d) - g(f)| <= 3 (1.0001 + |d|)^2 |d|: when no pixel of the wave has |d| above the floor, no edit of the
// wave is large (see RiskCfg) and mx - two maxima per pixel - is not formed.
template <class V>
__device__ __forceinline__ PxT<V> shifted_linear(const PxT<V> lin, const LabFT<V> f, const V dfx, const V dfy,
                                                 const V dfz, RiskT<V>* risk = nullptr, float same_d_floor = 0.0f) {
  // (a wave-uniform shortcut for "every pixel on the cubic segment" costs as much to decide as it saves)
  const V dx = finv_delta_any(f.fx, dfx), dy = finv_delta_any(f.fy, dfy), dz = finv_delta_any(f.fz, dfz);
  // the unedited part of the round trip: (I + E) . linear, E = the mismatch of the two IEC matrices
  const V ur = drift_mad<0, V>(lin.r, drift_mad<1, V>(lin.g, drift_mad<2, V>(lin.b, lin.r)));
  const V ug = drift_mad<3, V>(lin.r, drift_mad<4, V>(lin.g, drift_mad<5, V>(lin.b, lin.g)));
  const V ub = drift_mad<6, V>(lin.r, drift_mad<7, V>(lin.g, drift_mad<8, V>(lin.b, lin.b)));
  PxT<V> o;
  o.r = mad<V>(kLab.inv[0], dx, mad<V>(kLab.inv[1], dy, mad<V>(kLab.inv[2], dz, ur)));
  o.g = mad<V>(kLab.inv[3], dx, mad<V>(kLab.inv[4], dy, mad<V>(kLab.inv[5], dz, ug)));
  o.b = mad<V>(kLab.inv[6], dx, mad<V>(kLab.inv[7], dy, mad<V>(kLab.inv[8], dz, ub)));
  if (risk) {
    if (same_d_floor > 0.0f) {
      risk->any = __ballot(lane_above(vabs(dfy), same_d_floor)) != 0ull;
      risk->mx = (V)(0.0f);
      if (risk->any) risk->mx = vmax3abs(dx, dy, dz);
    } else {
      risk->mx = vmax3abs(dx, dy, dz);
    }
    risk->o = o;
  }
  return o;
}

// clip(encode(.)) of it: what _lab_to_rgb returns for the edited pixel
template <class V, bool SG = false>
__device__ __forceinline__ PxT<V> shifted_rgb(const PxT<V> lin, const LabFT<V> f, const V dfx, const V dfy, const V dfz,
                                              const ChainParams& P, RiskT<V>* risk = nullptr, float same_d_floor = 0.0f) {
  const PxT<V> o = shifted_linear(lin, f, dfx, dfy, dfz, risk, same_d_floor);
  return PxT<V>{srgb_encode_clip<V, SG>(o.r), srgb_encode_clip<V, SG>(o.g), srgb_encode_clip<V, SG>(o.b)};
}

// ---- colour correction ---------------------------------------------------------------------

// dot product of T expansion terms with the staged table M[t] = {cR, cG, cB, 0} (LDS, broadcast reads)
template <int T, class V>
__device__ __forceinline__ PxT<V> ccm_dot(const V (&e)[T], const float4* __restrict__ M) {
  float4 c = M[0];
  PxT<V> o;
  o.r = e[0] * c.x;
  o.g = e[0] * c.y;
  o.b = e[0] * c.z;
#pragma unroll
  for (int t = 1; t < T; ++t) {
    c = M[t];
    o.r = mad<V>(e[t], c.x, o.r);
    o.g = mad<V>(e[t], c.y, o.g);
    o.b = mad<V>(e[t], c.z, o.b);
  }
  return o;
}

template <class V>
__device__ __forceinline__ PxT<V> ccm_linear3(const PxT<V> p, const ChainParams& P) {
  PxT<V> o;
  o.r = mad<V>(P.ccm3[2], p.b, mad<V>(P.ccm3[1], p.g, p.r * P.ccm3[0]));
  o.g = mad<V>(P.ccm3[5], p.b, mad<V>(P.ccm3[4], p.g, p.r * P.ccm3[3]));
  o.b = mad<V>(P.ccm3[8], p.b, mad<V>(P.ccm3[7], p.g, p.r * P.ccm3[6]));
  return o;
}

// dot product with a small table held in scalar registers (row-major 3 x T, compile-time indices)
template <int T, class V>
__device__ __forceinline__ PxT<V> ccm_dot_small(const V (&e)[T], const ChainParams& P) {
  PxT<V> o;
  o.r = e[0] * P.ccm_small[0];
  o.g = e[0] * P.ccm_small[T];
  o.b = e[0] * P.ccm_small[2 * T];
#pragma unroll
  for (int t = 1; t < T; ++t) {
    o.r = mad<V>(e[t], P.ccm_small[t], o.r);
    o.g = mad<V>(e[t], P.ccm_small[T + t], o.g);
    o.b = mad<V>(e[t], P.ccm_small[2 * T + t], o.b);
  }
  return o;
}

// 3 < T <= 8: Cheung 2004 with 5, 7 or 8 terms, Finlayson 2015 root-polynomial of degree 2 (6 terms)
template <class V>
__device__ __forceinline__ PxT<V> ccm_small_terms(const PxT<V> p, const ChainParams& P) {
  const V R = p.r, G = p.g, B = p.b;
  const V one = (V)(1.0f);
  switch (P.ccm_terms) {
    case 5: {
      const V e[5] = {R, G, B, R * G * B, one};
      return ccm_dot_small<5, V>(e, P);
    }
    case 6: {
      const V e[6] = {R, G, B, sroot2(R * G), sroot2(G * B), sroot2(R * B)};
      return ccm_dot_small<6, V>(e, P);
    }
    case 7: {
      const V e[7] = {R, G, B, R * G, R * B, G * B, one};
      return ccm_dot_small<7, V>(e, P);
    }
    default: {  // 8
      const V e[8] = {R, G, B, R * G, R * B, G * B, R * G * B, one};
      return ccm_dot_small<8, V>(e, P);
    }
  }
}

// Cooperative copy of the full table into LDS; callers synchronise the workgroup afterwards.
__device__ __forceinline__ void stage_ccm(const ChainParams& P, float4* __restrict__ M, int tid, int nthreads) {
  for (int t = tid; t < P.ccm_terms; t += nthreads)
    M[t] = make_float4(P.ccm_full[t][0], P.ccm_full[t][1], P.ccm_full[t][2], 0.0f);
}

// colour.characterisation.matrix_augmented_Cheung2004 term lists
template <class V>
__device__ __forceinline__ PxT<V> ccm_cheung(const PxT<V> p, const int terms, const float4* __restrict__ M) {
  const V R = p.r, G = p.g, B = p.b;
  const V one = (V)(1.0f);
  switch (terms) {
    case 3: {
      const V e[3] = {R, G, B};
      return ccm_dot<3, V>(e, M);
    }
    case 5: {
      const V e[5] = {R, G, B, R * G * B, one};
      return ccm_dot<5, V>(e, M);
    }
    case 7: {
      const V e[7] = {R, G, B, R * G, R * B, G * B, one};
      return ccm_dot<7, V>(e, M);
    }
    case 8: {
      const V e[8] = {R, G, B, R * G, R * B, G * B, R * G * B, one};
      return ccm_dot<8, V>(e, M);
    }
    case 10: {
      const V e[10] = {R, G, B, R * G, R * B, G * B, R * R, G * G, B * B, one};
      return ccm_dot<10, V>(e, M);
    }
    case 11: {
      const V e[11] = {R, G, B, R * G, R * B, G * B, R * R, G * G, B * B, R * G * B, one};
      return ccm_dot<11, V>(e, M);
    }
    case 14: {
      const V e[14] = {R, G, B, R * G, R * B, G * B, R * R, G * G, B * B, R * G * B,
                       R * R * R, G * G * G, B * B * B, one};
      return ccm_dot<14, V>(e, M);
    }
    case 16: {
      const V e[16] = {R, G, B, R * G, R * B, G * B, R * R, G * G, B * B, R * G * B,
                       R * R * G, G * G * B, B * B * R, R * R * R, G * G * G, B * B * B};
      return ccm_dot<16, V>(e, M);
    }
    case 17: {
      const V e[17] = {R, G, B, R * G, R * B, G * B, R * R, G * G, B * B, R * G * B,
                       R * R * G, G * G * B, B * B * R, R * R * R, G * G * G, B * B * B, one};
      return ccm_dot<17, V>(e, M);
    }
    case 19: {
      const V e[19] = {R, G, B, R * G, R * B, G * B, R * R, G * G, B * B, R * G * B,
                       R * R * G, G * G * B, B * B * R, R * R * B, G * G * R, B * B * G,
                       R * R * R, G * G * G, B * B * B};
      return ccm_dot<19, V>(e, M);
    }
    case 20: {
      const V e[20] = {R, G, B, R * G, R * B, G * B, R * R, G * G, B * B, R * G * B,
                       R * R * G, G * G * B, B * B * R, R * R * B, G * G * R, B * B * G,
                       R * R * R, G * G * G, B * B * B, one};
      return ccm_dot<20, V>(e, M);
    }
    default: {  // 22
      const V e[22] = {R, G, B, R * G, R * B, G * B, R * R, G * G, B * B, R * G * B,
                       R * R * G, G * G * B, B * B * R, R * R * B, G * G * R, B * B * G,
                       R * R * R, G * G * G, B * B * B, R * R * G * B, R * G * G * B, R * G * B * B};
      return ccm_dot<22, V>(e, M);
    }
  }
}

// colour.characterisation.polynomial_expansion_Finlayson2015(root_polynomial_expansion=True)
template <class V>
__device__ __forceinline__ PxT<V> ccm_rootpoly(const PxT<V> p, const int terms, const float4* __restrict__ M) {
  const V R = p.r, G = p.g, B = p.b;
  const V rg = R * G, gb = G * B, rb = R * B;
  switch (terms) {
    case 3: {
      const V e[3] = {R, G, B};
      return ccm_dot<3, V>(e, M);
    }
    case 6: {
      const V e[6] = {R, G, B, sroot2(rg), sroot2(gb), sroot2(rb)};
      return ccm_dot<6, V>(e, M);
    }
    // Degrees 3 and 4: spow is multiplicative (sign(x y) |x y|^p = spow(x, p) spow(y, p)), so every root of a
    // product is a product of the three channels' own signed roots: 3 cube roots instead of 7, 3 fourth roots
    // instead of 3 square and 9 fourth roots (each root a transcendental pair at 3.4 times a multiplication).
    // The rounding of a term changes by an ulp or two either way, far inside what float32 leaves of these sums.
    case 13: {
      const V cr = sroot3(R), cg = sroot3(G), cb = sroot3(B);
      const V cr2 = cr * cr, cg2 = cg * cg, cb2 = cb * cb;
      const V e[13] = {R, G, B, sroot2(rg), sroot2(gb), sroot2(rb),
                       cr * cg2, cg * cb2, cr * cb2, cg * cr2, cb * cg2, cb * cr2, (cr * cg) * cb};
      return ccm_dot<13, V>(e, M);
    }
    default: {  // 22
      const V cr = sroot3(R), cg = sroot3(G), cb = sroot3(B);
      const V cr2 = cr * cr, cg2 = cg * cg, cb2 = cb * cb;
      const V qr = sroot4(R), qg = sroot4(G), qb = sroot4(B);       // signed fourth roots
      const V hr = qr * vabs(qr), hg = qg * vabs(qg), hb = qb * vabs(qb);  // signed square roots
      const V ar = vabs(hr), ag = vabs(hg), ab = vabs(hb);          // |x|^(1/2) = q^2
      const V e[22] = {R, G, B, hr * hg, hg * hb, hr * hb,
                       cr * cg2, cg * cb2, cr * cb2, cg * cr2, cb * cg2, cb * cr2, (cr * cg) * cb,
                       (ar * qr) * qg, (ar * qr) * qb, (ag * qg) * qr,   // (R^3 G)^(1/4), (R^3 B), (G^3 R)
                       (ag * qg) * qb, (ab * qb) * qr, (ab * qb) * qg,   // (G^3 B), (B^3 R), (B^3 G)
                       ar * (qg * qb), ag * (qr * qb), ab * (qr * qg)};  // (R^2 G B), (G^2 R B), (B^2 R G)
      return ccm_dot<22, V>(e, M);
    }
  }
}

enum CcmClass { CCM_NONE = 0, CCM_LIN3 = 1, CCM_CHEUNG = 2, CCM_ROOT = 3, CCM_RUNTIME = 4, CCM_SMALL = 5 };

template <int CCMC, class V>
__device__ __forceinline__ PxT<V> apply_ccm(const PxT<V> p, const ChainParams& P, const float4* __restrict__ M) {
  if constexpr (CCMC == CCM_NONE) {
    return p;
  } else if constexpr (CCMC == CCM_LIN3) {
    return ccm_linear3(p, P);
  } else if constexpr (CCMC == CCM_SMALL) {
    return ccm_small_terms(p, P);
  } else if constexpr (CCMC == CCM_CHEUNG) {
    return ccm_cheung(p, P.ccm_terms, M);
  } else if constexpr (CCMC == CCM_ROOT) {
    return ccm_rootpoly(p, P.ccm_terms, M);
  } else {
    if (P.ccm_kind == SS_CCM_NONE) return p;
    if (P.ccm_kind == SS_CCM_CHEUNG) return ccm_cheung(p, P.ccm_terms, M);
    return ccm_rootpoly(p, P.ccm_terms, M);
  }
}

// kernels of these classes stage the full table into LDS
template <int CCMC>
constexpr bool ccm_needs_table = (CCMC == CCM_CHEUNG || CCMC == CCM_ROOT || CCMC == CCM_RUNTIME);

// ---- per-image parameters -------------------------------------------------------------------

struct ImgParams {
  float neg_lo[3], neg_inv[3];
  float gain0[3];
  int ccm3_gain[9];  // (float bits) the 3-term matrix with the first gain stage folded into its rows
};

__device__ __forceinline__ ImgParams load_img_params(const ChainParams& P, int img) {
  ImgParams q;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    q.neg_lo[c] = P.neg_dev ? P.neg_dev[img * SS_NEG_PARAM_STRIDE + c] : P.neg_lo[c];
    q.neg_inv[c] = P.neg_dev ? P.neg_dev[img * SS_NEG_PARAM_STRIDE + 3 + c] : P.neg_inv[c];
    q.gain0[c] = P.gain0_dev ? P.gain0_dev[img] * P.gain[0][c] : P.gain[0][c];
  }
  // wave-uniform by construction; said so explicitly, so that the nine products live in scalar registers
#pragma unroll
  for (int i = 0; i < 9; ++i) q.ccm3_gain[i] = __builtin_amdgcn_readfirstlane(__float_as_int(P.ccm3[i] * q.gain0[i / 3]));
  return q;
}

// ---- stage 4: shadows / highlights on Lab (editingTools.py:508-536) --------------------------

// total adjustment in L units (editingTools.py:518-522)
template <class V>
__device__ __forceinline__ V sh_total(const V L, const ChainParams& P) {
  const V Ln = L * 0.01f;
  V total = (V)(0.0f);
  if (P.sh_hs != 0.0f) total = (hw_exp2(Ln * P.sh_log2f) - 1.0f) * P.sh_hs;
  if (P.sh_ss != 0.0f) total = mad<V>(hw_exp2((1.0f - Ln) * P.sh_log2f) - 1.0f, P.sh_ss, total);
  return total * 100.0f;
}

// Lab in, Lab out: the reference's LAB-input branch
template <class V>
__device__ __forceinline__ LabT<V> sh_on_lab(LabT<V> q, const ChainParams& P) {
  const V total = sh_total(q.L, P);
  q.L = clampf(q.L + total, 0.0f, 100.0f);
  if (P.sh_chroma) {
    const V cs = 1.0f - clampf(vabs(total) * (0.05f / 20.0f), 0.0f, 0.2f);
    q.a = q.a * cs;
    q.b = q.b * cs;
  }
  return q;
}


// RGB (sRGB-encoded, as the reference assumes) in and out: _rgb_to_lab, the edit, _lab_to_rgb.
// DECODED = true returns the sRGB decoding of that result instead: decode(clip(encode(x), 0, 1)) is
// clip(x, 0, 1) (encode is increasing and fixes 0 and 1), so a consumer that would decode the pixel again
// - the next Lab edit - gets it without the encode/decode pair and without that pair's rounding.
template <bool DECODED = false, class V>
__device__ __forceinline__ PxT<V> sh_on_rgb(const PxT<V> p, const ChainParams& P, RiskT<V>* risk = nullptr) {
  const PxT<V> lin{srgb_decode(p.r), srgb_decode(p.g), srgb_decode(p.b)};
  const LabFT<V> f = linear_to_labf(lin, P);
  const V L = mad<V>(f.fy, 116.0f, -16.0f);
  const V total = sh_total(L, P);
  // clip(L + total, 0, 100) - L, exactly
  const V dfy = med3(total, -L, 100.0f - L) * (1.0f / 116.0f);
  V dfx = dfy, dfz = dfy;
  if (P.sh_chroma) {
    const V shrink = clampf(vabs(total) * (0.05f / 20.0f), 0.0f, 0.2f);  // a' - a = -a * shrink, a/500 = fx - fy
    dfx = dfy - (f.fx - f.fy) * shrink;
    dfz = dfy + (f.fy - f.fz) * shrink;
  }
  if constexpr (DECODED) {
    const PxT<V> o = shifted_linear(lin, f, dfx, dfy, dfz, risk);
    return PxT<V>{clamp01(o.r), clamp01(o.g), clamp01(o.b)};
  } else {
    return shifted_rgb(lin, f, dfx, dfy, dfz, P, risk);
  }
}

// ---- stages 1-4 --------------------------------------------------------------------------------

// PLAIN = the caller guarantees: no negative inversion, exactly one gain stage, no shadows/highlights,
// RGB data.  Everything else about a chain is then dead code and its parameters never reach registers.
template <int CCMC, bool SH, bool PLAIN = false, class V = float>
__device__ __forceinline__ PxT<V> front(PxT<V> p, const ChainParams& P, const ImgParams& ip,
                                        const float4* __restrict__ M, RiskT<V>* risk = nullptr) {
  if constexpr (PLAIN) {
    if constexpr (CCMC == CCM_LIN3) {
      // clip(gain * (M p)) as clip((gain M) p): three instructions per pixel fewer, same value to an ulp
      const float* m = reinterpret_cast<const float*>(ip.ccm3_gain);
      PxT<V> o;
      o.r = clamp01(mad<V>(m[2], p.b, mad<V>(m[1], p.g, p.r * m[0])));
      o.g = clamp01(mad<V>(m[5], p.b, mad<V>(m[4], p.g, p.r * m[3])));
      o.b = clamp01(mad<V>(m[8], p.b, mad<V>(m[7], p.g, p.r * m[6])));
      return o;
    }
    p = apply_ccm<CCMC>(p, P, M);
    p.r = clamp01(p.r * ip.gain0[0]);
    p.g = clamp01(p.g * ip.gain0[1]);
    p.b = clamp01(p.b * ip.gain0[2]);
    return p;
  }
  if (P.neg) {
    p.r = 1.0f - clamp01((clamp01(p.r) - ip.neg_lo[0]) * ip.neg_inv[0]);
    p.g = 1.0f - clamp01((clamp01(p.g) - ip.neg_lo[1]) * ip.neg_inv[1]);
    p.b = 1.0f - clamp01((clamp01(p.b) - ip.neg_lo[2]) * ip.neg_inv[2]);
  }
  p = apply_ccm<CCMC>(p, P, M);
  if (P.n_gain > 0) {
    p.r = clamp01(p.r * ip.gain0[0]);
    p.g = clamp01(p.g * ip.gain0[1]);
    p.b = clamp01(p.b * ip.gain0[2]);
    if (P.n_gain > 1) {
      p.r = clamp01(p.r * P.gain[1][0]);
      p.g = clamp01(p.g * P.gain[1][1]);
      p.b = clamp01(p.b * P.gain[1][2]);
      if (P.n_gain > 2) {
        p.r = clamp01(p.r * P.gain[2][0]);
        p.g = clamp01(p.g * P.gain[2][1]);
        p.b = clamp01(p.b * P.gain[2][2]);
      }
    }
  }
  if constexpr (SH) {
    if (P.sh) {
      if (P.lab_io) {
        const LabT<V> q = sh_on_lab(LabT<V>{p.r, p.g, p.b}, P);
        p = PxT<V>{q.L, q.a, q.b};
      } else {
        p = sh_on_rgb<false, V>(p, P, risk);
      }
    }
  }
  return p;
}

// Stages 1-4 for a consumer that works on the decoded pixel (the sharpen stage): linear RGB for RGB data,
// (L, a, b) unchanged for Lab data.
template <int CCMC, bool PLAIN = false, class V = float>
__device__ __forceinline__ PxT<V> front_decoded(PxT<V> p, const ChainParams& P, const ImgParams& ip,
                                                const float4* __restrict__ M, RiskT<V>* risk = nullptr) {
  p = front<CCMC, false, PLAIN, V>(p, P, ip, M);
  if constexpr (!PLAIN) {
    if (P.lab_io) {
      if (P.sh) {
        const LabT<V> q = sh_on_lab(LabT<V>{p.r, p.g, p.b}, P);
        p = PxT<V>{q.L, q.a, q.b};
      }
      return p;
    }
    if (P.sh) return sh_on_rgb<true, V>(p, P, risk);
  }
  return PxT<V>{srgb_decode<V, PLAIN>(p.r), srgb_decode<V, PLAIN>(p.g), srgb_decode<V, PLAIN>(p.b)};
}

// ---- stage 5 pointwise part: unsharp mask on (Ln, blurred Ln) (editingTools.py:767-809) -------

template <bool PLAIN = false, class V = float>
__device__ __forceinline__ LabT<V> usm_on_lab(const V Ln, const V a, const V b, const V Lb, const ChainParams& P) {
  V um = Ln - Lb;
  if (!PLAIN && P.usm_thr > 0.0f) {
    const V es = vabs(um);
    um = um * (es > (V)(P.usm_thr) ? (V)(1.0f) : es * P.usm_inv_thr);
  }
  const V hm = Ln > (V)(0.9f) ? mad<V>(Ln - 0.9f, -10.0f, 1.0f) : (V)(1.0f);
  const V sm = Ln < (V)(0.1f) ? Ln * 10.0f : (V)(1.0f);
  const V ad = clampf(hm * sm, 0.1f, 1.0f);
  LabT<V> o;
  o.L = clamp01(mad<V>(um * P.usm_amount_scaled, ad, Ln)) * 100.0f;
  if (P.usm_chroma) {
    const V boost = mad<V>(vabs(um), P.usm_chroma_k, 1.0f);
    o.a = clampf(a * boost, -127.0f, 127.0f);
    o.b = clampf(b * boost, -127.0f, 127.0f);
  } else {
    o.a = a;
    o.b = b;
  }
  return o;
}

// The same edit on RGB data: `lin` = decoded (linear) pixel, Ln = L/100 of it, Lb = blurred Ln.
// Returns what _lab_to_rgb gives for the sharpened pixel (sRGB-encoded, clipped).
// CHROMA: 1 / 0 = the chroma boost (amount > 30) is known to be on / off, -1 = read P.usm_chroma.
// `risk_floor` (with PLAIN and CHROMA == 0 only): RiskCfg<OUT>::MIN_MX of the caller's output, 0 = none.
template <bool PLAIN = false, class V = float, int CHROMA = -1>
__device__ __forceinline__ PxT<V> usm_on_rgb(const PxT<V> lin, const V Ln, const V Lb, const ChainParams& P,
                                             RiskT<V>* risk = nullptr, float risk_floor = 0.0f) {
  LabFT<V> f;
  f.fy = mad<V>(Ln, 100.0f / 116.0f, 16.0f / 116.0f);
  f.fx = lab_f(mad<V>(kLab.fwd[0], lin.r, mad<V>(kLab.fwd[1], lin.g, lin.b * kLab.fwd[2])));
  f.fz = lab_f(mad<V>(kLab.fwd[6], lin.r, mad<V>(kLab.fwd[7], lin.g, lin.b * kLab.fwd[8])));
  V um = Ln - Lb;
  if (!PLAIN && P.usm_thr > 0.0f) {
    const V es = vabs(um);
    um = um * (es > (V)(P.usm_thr) ? (V)(1.0f) : es * P.usm_inv_thr);
  }
  // where(Ln > 0.9, 1 - (Ln - 0.9) 10, 1) and where(Ln < 0.1, Ln 10, 1): both are min(., 1)
  // (Ln is in [0, 1], so both are >= 0 and min(., 1) is a clip to [0, 1]: the clamp modifier of the instruction)
  const V hm = fma_clip01(Ln - 0.9f, -10.0f, 1.0f);
  const V sm = mul_clip01(Ln, 10.0f);
  // clip(hm sm, 0.1, 1) as 0.1 + clip(hm sm - 0.1, 0, 1): a max costs 1.8 times an add on this GPU
  const V ad = fma2_clip01(hm, sm, -0.1f) + 0.1f;
  // clip(Ln + um * amount * ad, 0, 1) - Ln, exactly; L = 100 Ln = 116 fy - 16
  const V dfy = med3((um * P.usm_amount_scaled) * ad, -Ln, 1.0f - Ln) * (100.0f / 116.0f);
  V dfx = dfy, dfz = dfy;
  const bool chroma = CHROMA < 0 ? P.usm_chroma != 0 : CHROMA != 0;
  if (chroma) {
    // clip(a (1 + e), -127, 127) - a = med3(a e, -127 - a, 127 - a), in units of fx - fy = a / 500 and
    // fy - fz = b / 200.  Formed as a difference: for a small edit 1 + e rounds to 1 or 1 + 2^-23 and a (1 + e) to a
    // multiple of 8e-6, which moved a dark channel of a saturated pixel by 1e-7.
    const V e = vabs(um) * P.usm_chroma_k;
    const V pa = f.fx - f.fy, pb = f.fy - f.fz;
    dfx = dfy + med3(pa * e, (V)(-127.0f / 500.0f) - pa, (V)(127.0f / 500.0f) - pa);
    dfz = dfy - med3(pb * e, (V)(-127.0f / 200.0f) - pb, (V)(127.0f / 200.0f) - pb);
  }
  // one d for X, Y and Z and a pixel in [0, 1]: |d| <= floor / 4 keeps every |g(f + d) - g(f)| below the floor
  // (3 (1.0001 + d)^2 d <= 0.91 floor for floor <= 0.4)
  const float same_d_floor = (PLAIN && CHROMA == 0) ? risk_floor * 0.25f : 0.0f;
  return shifted_rgb<V, PLAIN>(lin, f, dfx, dfy, dfz, P, risk, same_d_floor);
}

// ---- stages 6-7: final clip, encoding, quantisation -------------------------------------------

template <int OUT, bool IN_RANGE = false, class V = float>
__device__ __forceinline__ PxT<V> finish(PxT<V> p, const ChainParams& P) {
  if (!IN_RANGE && P.final_clip) {
    p.r = clamp01(p.r);
    p.g = clamp01(p.g);
    p.b = clamp01(p.b);
  }
  if constexpr (OUT != SS_OUT_F32_LINEAR && OUT != SS_OUT_U8_LINEAR) {
    p.r = srgb_encode(p.r);
    p.g = srgb_encode(p.g);
    p.b = srgb_encode(p.b);
  }
  return p;
}

// Output encoding of a pixel in [0, 1] for the 16-bit quantiser, scaled: 65535 encode(x), the factor folded into the
// curve's constants (one rounding where the product behind the curve has two).  The caller truncates.
template <class V>
__device__ __forceinline__ V srgb_encode_scaled16(V L) {
  // (only the plain tile kernels come here: masks in scalar register pairs, see mask_le)
  const auto m = mask_le(L, 0.0031308f);
  const V hi = mad<V>(hw_exp2(hw_log2(L) * (1.0f / 2.4f)), (float)(1.055 * 65535.0), (float)(-0.055 * 65535.0));
  const V lo = L * (float)(12.92 * 65535.0);
  return select_by(m, lo, hi);
}
template <class V>
__device__ __forceinline__ PxT<V> finish_scaled16(const PxT<V> p) {
  return PxT<V>{srgb_encode_scaled16(p.r), srgb_encode_scaled16(p.g), srgb_encode_scaled16(p.b)};
}

// np.uint16(np.clip(x*65535, 0, 65535)) / np.uint8(np.clip(x*255, 0, 255)): truncation toward zero
__device__ __forceinline__ uint32_t quant16(float x) { return (uint32_t)clampf(x * 65535.0f, 0.0f, 65535.0f); }
__device__ __forceinline__ uint32_t quant8(float x) { return (uint32_t)clampf(x * 255.0f, 0.0f, 255.0f); }
// the same for x known to lie in [0, 1]: the product cannot leave the range, the conversion truncates
__device__ __forceinline__ uint32_t quant16_unit(float x) { return (uint32_t)(x * 65535.0f); }
__device__ __forceinline__ uint32_t quant8_unit(float x) { return (uint32_t)(x * 255.0f); }

// The second, 8-bit output of one pixel (ChainParams::dst2) from its value after stage 6: exactly what the chain
// with that output kind would store.  `px`: pixel index in the batch.
__device__ __forceinline__ void store_second(unsigned char* dst2, size_t px, Px v, bool clip, bool encoded) {
  if (clip) v = Px{clamp01(v.r), clamp01(v.g), clamp01(v.b)};
  if (encoded) v = Px{srgb_encode(v.r), srgb_encode(v.g), srgb_encode(v.b)};
  unsigned char* d = dst2 + px * 3;
  d[0] = (unsigned char)quant8(v.r);
  d[1] = (unsigned char)quant8(v.g);
  d[2] = (unsigned char)quant8(v.b);
}

// lane granularity per output kind: G adjacent pixels per lane make a 12-byte store
template <int OUT>
struct OutTraits;
template <>
struct OutTraits<SS_OUT_F32_LINEAR> {
  static constexpr int G = 1, BYTES = 4;
};
template <>
struct OutTraits<SS_OUT_F32_SRGB> {
  static constexpr int G = 1, BYTES = 4;
};
template <>
struct OutTraits<SS_OUT_U16_SRGB> {
  static constexpr int G = 2, BYTES = 2;
};
template <>
struct OutTraits<SS_OUT_U8_SRGB> {
  static constexpr int G = 4, BYTES = 1;
};
template <>
struct OutTraits<SS_OUT_U8_LINEAR> {
  static constexpr int G = 4, BYTES = 1;
};

// 12-byte, 4-byte-aligned vector types: one global_load/store_dwordx3 per lane, lanes contiguous
struct __attribute__((aligned(4))) F3 {
  float x, y, z;
};
struct __attribute__((aligned(4))) U3 {
  uint32_t x, y, z;
};

// Streaming ("nontemporal") forms for data that is touched once: the caches are told not to keep it, which is
// worth ~10 % on the pointwise chain (5.6 -> 6.1 TB/s for 12 B in + 12 B out per pixel).  The three dwords are
// merged back into one dwordx3 access by the compiler.
__device__ __forceinline__ F3 load_stream(const F3* p) {
  const float* q = reinterpret_cast<const float*>(p);
  return F3{__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2)};
}
__device__ __forceinline__ void store_stream(F3* p, const F3 v) {
  float* q = reinterpret_cast<float*>(p);
  __builtin_nontemporal_store(v.x, q);
  __builtin_nontemporal_store(v.y, q + 1);
  __builtin_nontemporal_store(v.z, q + 2);
}
__device__ __forceinline__ void store_stream(U3* p, const U3 v) {
  uint32_t* q = reinterpret_cast<uint32_t*>(p);
  __builtin_nontemporal_store(v.x, q);
  __builtin_nontemporal_store(v.y, q + 1);
  __builtin_nontemporal_store(v.z, q + 2);
}

// OpenCV BORDER_REFLECT (fedcba|abcdefgh|hgfedcb) coordinate map (cv::borderInterpolate)
__device__ __forceinline__ int reflect_coord(int p, int len) {
  if (len == 1) return 0;
  while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p - 1 : 2 * len - p - 1;
  return p;
}

}  // namespace ss
