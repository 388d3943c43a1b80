// Float64 evaluation of the correction chain for single pixels.
//
// The reference computes the chain in float64 (colour-science converts its input with as_float_array).  The
// float32 kernels follow it to a few 1e-7, which every output encoding tolerates except in one situation: a Lab
// edit (shadows/highlights, unsharp mask) that pushes one channel of a saturated pixel to almost zero.  The edited
// pixel is linear + M_inv (g(f + d) - g(f)) with terms of size ~1 that cancel to ~1e-4; float32 keeps that sum to
// ~3e-7 absolute, and the two sRGB encodes that follow have slope 12.92 each near zero: 167 x 3e-7 = 3 steps of 16
// bits.  Such pixels are rare and easy to recognise (RiskCfg in ss_device.h): the float32 kernels queue them and a
// second kernel evaluates them again here, from the source pixel, stage by stage as the reference does, in
// float64 (its powers as refined float32 estimates, see rpow).  Only the blurred lightness is taken over from the
// float32 pipeline (the reference blurs in float32 too).  About a thousand instructions per pixel, paid per queued
// pixel only.
#pragma once

#include "ss_device.h"

namespace ss {

namespace careful {

__device__ __forceinline__ double clip01(double x) { return fmin(fmax(x, 0.0), 1.0); }
__device__ __forceinline__ double sroot2(double a) { return a < 0.0 ? -sqrt(-a) : sqrt(a); }
__device__ __forceinline__ double sroot4(double a) { return a < 0.0 ? -sqrt(sqrt(-a)) : sqrt(sqrt(a)); }
__device__ __forceinline__ double spow(double a, double p) {
  if (a == 0.0) return 0.0;
  const double m = pow(fabs(a), p);
  return a < 0.0 ? -m : m;
}
// sign(a) |a|^(N/M) for the rational exponents of this chain (12/5, 5/12, 1/3): the float32 hardware estimate
// exp2(N/M log2 |a|), good to ~1e-7, and one Newton step on y^M = |a|^N in float64, which squares that error
// ((M - 1)/2 e^2: below 1e-12 relative against pow(), measured): a dozen float64 multiplications where pow() takes
// several hundred instructions.  The step's quotient (y^M - x^N) / (M y^M) is itself of size e, so the float32
// reciprocal of its denominator (good to 1e-7) serves: it puts e * 1e-7 on the result, as little as the step's own
// remainder, and a float64 division is some thirty instructions.  The argument is first scaled by 2^(M k) into
// [2^-M, 2^M], so that neither the float32 estimates nor the powers inside the step can leave their ranges
// (y^M stays inside 2^+-60), and the result by 2^(-N k).
// Divisions by constants are written as multiplications by the constant's reciprocal throughout this file for the
// same reason: a rounding of 1e-16 either way, where results are needed to 1e-9.
// v^K by squaring (12 -> 4 multiplications, 5 -> 3)
template <int K>
__device__ __forceinline__ double ipow(double v) {
  if constexpr (K == 1) {
    return v;
  } else if constexpr (K % 2 == 0) {
    const double h = ipow<K / 2>(v);
    return h * h;
  } else {
    return v * ipow<K - 1>(v);
  }
}

template <int N, int M>
__device__ __forceinline__ double rpow(double a) {
  if (a == 0.0) return 0.0;
  const double x0 = fabs(a);
  const int k = -(ilogb(x0) / M);
  const double x = ldexp(x0, M * k);
  double y = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)x) * ((float)N / (float)M));
  const double ym = ipow<M>(y), xn = ipow<N>(x);
  y -= y * (ym - xn) * (double)__builtin_amdgcn_rcpf((float)M * (float)ym);
  y = ldexp(y, -N * k);
  return a < 0.0 ? -y : y;
}
__device__ __forceinline__ double encode(double L) { return L <= 0.0031308 ? L * 12.92 : 1.055 * rpow<5, 12>(L) - 0.055; }
__device__ __forceinline__ double decode(double V) {
  return V <= 12.92 * 0.0031308 ? V * (1.0 / 12.92) : rpow<12, 5>((V + 0.055) * (1.0 / 1.055));
}
__device__ __forceinline__ double lab_f(double t) {
  const double T = 24.0 / 116.0;
  return t > T * T * T ? rpow<1, 3>(t) : (841.0 / 108.0) * t + 16.0 / 116.0;
}
__device__ __forceinline__ double lab_finv(double f) {
  return f > 24.0 / 116.0 ? f * f * f : (f - 16.0 / 116.0) * (108.0 / 841.0);
}

struct Lab3 {
  double L, a, b;
};

// colour.sRGB_to_XYZ (decode first) and colour.XYZ_to_Lab under D65
__device__ __forceinline__ Lab3 rgb_to_lab(const double rgb[3]) {
  const double m[9] = {0.4124, 0.3576, 0.1805, 0.2126, 0.7152, 0.0722, 0.0193, 0.1192, 0.9505};
  const double inv_white[3] = {0.3290 / 0.3127, 1.0, 0.3290 / (1.0 - 0.3127 - 0.3290)};
  const double r = decode(rgb[0]), g = decode(rgb[1]), b = decode(rgb[2]);
  double f[3];
  for (int k = 0; k < 3; ++k) f[k] = lab_f((m[3 * k] * r + m[3 * k + 1] * g + m[3 * k + 2] * b) * inv_white[k]);
  return Lab3{116.0 * f[1] - 16.0, 500.0 * (f[0] - f[1]), 200.0 * (f[1] - f[2])};
}

// The same from pixels that are decoded already (`lin`: what decode() gives for the stage's input)
__device__ __forceinline__ Lab3 linear_to_lab(const double lin[3]) {
  const double m[9] = {0.4124, 0.3576, 0.1805, 0.2126, 0.7152, 0.0722, 0.0193, 0.1192, 0.9505};
  const double inv_white[3] = {0.3290 / 0.3127, 1.0, 0.3290 / (1.0 - 0.3127 - 0.3290)};
  double f[3];
  for (int k = 0; k < 3; ++k)
    f[k] = lab_f((m[3 * k] * lin[0] + m[3 * k + 1] * lin[1] + m[3 * k + 2] * lin[2]) * inv_white[k]);
  return Lab3{116.0 * f[1] - 16.0, 500.0 * (f[0] - f[1]), 200.0 * (f[1] - f[2])};
}

// Lab -> XYZ -> linear RGB clipped to [0, 1]: decode() of what lab_to_rgb returns, without the encode / decode pair
// (decode(clip(encode(x), 0, 1)) = clip(x, 0, 1): encode is increasing and fixes 0 and 1)
__device__ __forceinline__ void lab_to_linear(const Lab3 q, double lin[3]) {
  const double inv[9] = {3.2406, -1.5372, -0.4986, -0.9689, 1.8758, 0.0415, 0.0557, -0.2040, 1.0570};
  const double white[3] = {0.3127 / 0.3290, 1.0, (1.0 - 0.3127 - 0.3290) / 0.3290};
  const double fy = (q.L + 16.0) * (1.0 / 116.0), fx = q.a * (1.0 / 500.0) + fy, fz = fy - q.b * (1.0 / 200.0);
  const double x = lab_finv(fx) * white[0], y = lab_finv(fy) * white[1], z = lab_finv(fz) * white[2];
  for (int k = 0; k < 3; ++k) lin[k] = clip01(inv[3 * k] * x + inv[3 * k + 1] * y + inv[3 * k + 2] * z);
}

// colour.Lab_to_XYZ, colour.XYZ_to_sRGB (the IEC matrix as published, sRGB encode), clip
__device__ __forceinline__ void lab_to_rgb(const Lab3 q, double rgb[3]) {
  const double inv[9] = {3.2406, -1.5372, -0.4986, -0.9689, 1.8758, 0.0415, 0.0557, -0.2040, 1.0570};
  const double white[3] = {0.3127 / 0.3290, 1.0, (1.0 - 0.3127 - 0.3290) / 0.3290};
  const double fy = (q.L + 16.0) * (1.0 / 116.0), fx = q.a * (1.0 / 500.0) + fy, fz = fy - q.b * (1.0 / 200.0);
  const double x = lab_finv(fx) * white[0], y = lab_finv(fy) * white[1], z = lab_finv(fz) * white[2];
  for (int k = 0; k < 3; ++k) rgb[k] = clip01(encode(inv[3 * k] * x + inv[3 * k + 1] * y + inv[3 * k + 2] * z));
}

// term lists of colour.characterisation: matrix_augmented_Cheung2004 / polynomial_expansion_Finlayson2015 (roots)
__device__ __forceinline__ int expand(int kind, int terms, double R, double G, double B, double e[SS_MAX_CCM_TERMS]) {
  int n = 0;
  e[n++] = R;
  e[n++] = G;
  e[n++] = B;
  if (terms == 3) return n;
  if (kind == SS_CCM_CHEUNG) {
    if (terms == 5) {
      e[n++] = R * G * B;
      e[n++] = 1.0;
      return n;
    }
    e[n++] = R * G;
    e[n++] = R * B;
    e[n++] = G * B;
    if (terms == 7) {
      e[n++] = 1.0;
      return n;
    }
    if (terms == 8) {
      e[n++] = R * G * B;
      e[n++] = 1.0;
      return n;
    }
    e[n++] = R * R;
    e[n++] = G * G;
    e[n++] = B * B;
    if (terms == 10) {
      e[n++] = 1.0;
      return n;
    }
    e[n++] = R * G * B;
    if (terms == 11) {
      e[n++] = 1.0;
      return n;
    }
    if (terms == 14) {
      e[n++] = R * R * R;
      e[n++] = G * G * G;
      e[n++] = B * B * B;
      e[n++] = 1.0;
      return n;
    }
    e[n++] = R * R * G;
    e[n++] = G * G * B;
    e[n++] = B * B * R;
    if (terms >= 19) {
      e[n++] = R * R * B;
      e[n++] = G * G * R;
      e[n++] = B * B * G;
    }
    e[n++] = R * R * R;
    e[n++] = G * G * G;
    e[n++] = B * B * B;
    if (terms == 17 || terms == 20) e[n++] = 1.0;
    if (terms == 22) {
      e[n++] = R * R * G * B;
      e[n++] = R * G * G * B;
      e[n++] = R * G * B * B;
    }
    return n;
  }
  e[n++] = sroot2(R * G);
  e[n++] = sroot2(G * B);
  e[n++] = sroot2(R * B);
  if (terms == 6) return n;
  e[n++] = rpow<1, 3>(R * G * G);
  e[n++] = rpow<1, 3>(G * B * B);
  e[n++] = rpow<1, 3>(R * B * B);
  e[n++] = rpow<1, 3>(G * R * R);
  e[n++] = rpow<1, 3>(B * G * G);
  e[n++] = rpow<1, 3>(B * R * R);
  e[n++] = rpow<1, 3>(R * G * B);
  if (terms == 13) return n;
  e[n++] = sroot4(R * R * R * G);
  e[n++] = sroot4(R * R * R * B);
  e[n++] = sroot4(G * G * G * R);
  e[n++] = sroot4(G * G * G * B);
  e[n++] = sroot4(B * B * B * R);
  e[n++] = sroot4(B * B * B * G);
  e[n++] = sroot4(R * R * G * B);
  e[n++] = sroot4(G * G * R * B);
  e[n++] = sroot4(B * B * R * G);
  return n;
}

// sign(a) |a|^(1/M) for the dense float64 colour-correction pass: the float32 hardware estimate (sqrt, or
// exp2(log2 |a| / M)), good to 1e-7 relative, and STEPS Newton steps on y^M = |a| in float64.  A step squares the
// error ((M - 1)/2 e^2) and its quotient (y^M - |a|) / (M y^(M-1)) is of size e, so the float32 reciprocal of the
// denominator serves (it adds e * 1e-7): one step leaves ~1e-14, two are as good as float64 gets.  A third of the
// instructions of sqrt() / pow(), and no float64 transcendental (quarter rate).  |a| below float32's range
// (1e-38, where the hardware estimates flush to 0) gives 0: an absolute error of at most 1e-19 on a term.
template <int M, int STEPS>
__device__ __forceinline__ double sroot(double a) {
  const double x = fabs(a);
  const float xf = (float)x;
  const double y0 = (double)(M == 2 ? __builtin_amdgcn_sqrtf(xf)
                                    : __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(xf) * (1.0f / (float)M)));
  double y = y0;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const double ym1 = ipow<M - 1>(y);
    // (estimate 0: the reciprocal is inf and the step NaN; selected away below)
    y -= (ym1 * y - x) * (double)__builtin_amdgcn_rcpf((float)M * (float)ym1);
  }
  y = y0 == 0.0 ? 0.0 : y;
  return a < 0.0 ? -y : y;
}

// The same term lists for the dense float64 colour-correction pass: the root-polynomial terms as products of the
// three channels' own signed roots (sign(x y) |x y|^p = spow(x, p) spow(y, p): three square, cube and fourth roots
// instead of three, seven and nine), which in float64 agree with the roots of the products to a few 1e-16.
// Degree 4 takes two Newton steps per root (its fits have coefficients of 10^5: 1e-14 on a term would show as 1e-9
// in the pixel, harmless, but the terms of a fit that bad deserve every bit), the others one.
__device__ __forceinline__ int expand_shared_roots(int kind, int terms, double R, double G, double B, double e[SS_MAX_CCM_TERMS]) {
  if (kind != SS_CCM_ROOTPOLY || terms <= 3) return expand(kind, terms, R, G, B, e);
  int n = 0;
  e[n++] = R;
  e[n++] = G;
  e[n++] = B;
  if (terms == 6) {
    const double hr = sroot<2, 1>(R), hg = sroot<2, 1>(G), hb = sroot<2, 1>(B);
    e[n++] = hr * hg;
    e[n++] = hg * hb;
    e[n++] = hr * hb;
    return n;
  }
  if (terms == 13) {
    const double hr = sroot<2, 1>(R), hg = sroot<2, 1>(G), hb = sroot<2, 1>(B);
    const double cr = sroot<3, 1>(R), cg = sroot<3, 1>(G), cb = sroot<3, 1>(B);
    const double cr2 = cr * cr, cg2 = cg * cg, cb2 = cb * cb;
    e[n++] = hr * hg;
    e[n++] = hg * hb;
    e[n++] = hr * hb;
    e[n++] = cr * cg2;
    e[n++] = cg * cb2;
    e[n++] = cr * cb2;
    e[n++] = cg * cr2;
    e[n++] = cb * cg2;
    e[n++] = cb * cr2;
    e[n++] = (cr * cg) * cb;
    return n;
  }
  const double cr = sroot<3, 2>(R), cg = sroot<3, 2>(G), cb = sroot<3, 2>(B);
  const double cr2 = cr * cr, cg2 = cg * cg, cb2 = cb * cb;
  const double qr = sroot<4, 2>(R), qg = sroot<4, 2>(G), qb = sroot<4, 2>(B);  // signed fourth roots
  const double ar = qr * qr, ag = qg * qg, ab = qb * qb;                       // |x|^(1/2)
  const double hr = qr * fabs(qr), hg = qg * fabs(qg), hb = qb * fabs(qb);     // signed square roots
  e[n++] = hr * hg;
  e[n++] = hg * hb;
  e[n++] = hr * hb;
  e[n++] = cr * cg2;
  e[n++] = cg * cb2;
  e[n++] = cr * cb2;
  e[n++] = cg * cr2;
  e[n++] = cb * cg2;
  e[n++] = cb * cr2;
  e[n++] = (cr * cg) * cb;
  e[n++] = (ar * qr) * qg;  // (R^3 G)^(1/4)
  e[n++] = (ar * qr) * qb;
  e[n++] = (ag * qg) * qr;
  e[n++] = (ag * qg) * qb;
  e[n++] = (ab * qb) * qr;
  e[n++] = (ab * qb) * qg;
  e[n++] = ar * (qg * qb);  // (R^2 G B)^(1/4)
  e[n++] = ag * (qr * qb);
  e[n++] = ab * (qr * qg);
  return n;
}

}  // namespace careful

// The whole chain for the pixel (sr, sg, sb) of image `img`, stages 1-7, in float64.  `Lb`: the blurred
// lightness at the pixel (used when the chain has an unsharp mask).  `encode_out`: stage 7 applies the sRGB
// encoding (every output kind except the two linear ones).  out[3]: the values the quantiser or the float store
// receives.  P points at the kernel's parameter block in memory (its kernel-argument segment).
__device__ inline void careful_pixel(const ChainParams* P, int img, float sr, float sg,
                                                               float sb, float Lb, bool encode_out, double out[3]) {
  using namespace careful;
  double p[3];
  {
    // stage 1: the reference inverts in float32 (invert_negative_auto works on a float32 array)
    float x[3] = {sr, sg, sb};
    if (P->neg) {
      for (int c = 0; c < 3; ++c) {
        const float lo = P->neg_dev ? P->neg_dev[img * SS_NEG_PARAM_STRIDE + c] : P->neg_lo[c];
        const float inv = P->neg_dev ? P->neg_dev[img * SS_NEG_PARAM_STRIDE + 3 + c] : P->neg_inv[c];
        x[c] = 1.0f - clamp01((clamp01(x[c]) - lo) * inv);
      }
    }
    for (int c = 0; c < 3; ++c) p[c] = (double)x[c];
  }
  if (P->ccm_kind != SS_CCM_NONE) {  // stage 2
    double e[SS_MAX_CCM_TERMS];
    const int n = expand(P->ccm_kind, P->ccm_terms, p[0], p[1], p[2], e);
    double o[3] = {0.0, 0.0, 0.0};
    for (int t = 0; t < n; ++t)
      for (int c = 0; c < 3; ++c) o[c] += ((double)P->ccm_full[t][c] + (double)P->ccm_lo[t][c]) * e[t];
    for (int c = 0; c < 3; ++c) p[c] = o[c];
  }
  for (int s = 0; s < P->n_gain; ++s) {  // stage 3
    const double g0 = (s == 0 && P->gain0_dev) ? (double)P->gain0_dev[img] : 1.0;
    for (int c = 0; c < 3; ++c) p[c] = clip01(p[c] * (g0 * (double)P->gain[s][c]));
  }
  if (P->sh) {  // stage 4 (editingTools.py:508-541)
    Lab3 q = rgb_to_lab(p);
    const double Ln = q.L * 0.01;
    const double ss = (double)P->sh_shadow_amount * 0.05, hs = (double)P->sh_highlight_amount * 0.05;
    // falloff ** x as exp(x ln falloff), the logarithm taken once on the host (pow() is three times the work)
    const double lf = P->sh_ln_falloff;
    const double h_adj = hs != 0.0 ? hs * (exp(lf * Ln) - 1.0) : 0.0;
    const double s_adj = ss != 0.0 ? ss * (exp(lf * (1.0 - Ln)) - 1.0) : 0.0;
    const double total = (h_adj + s_adj) * 100.0;
    q.L = fmin(fmax(q.L + total, 0.0), 100.0);
    if (P->sh_chroma) {
      const double cs = 1.0 - fmin(fmax(fabs(total) * (0.05 / 20.0), 0.0), 0.2);
      q.a *= cs;
      q.b *= cs;
    }
    // (in front of the sharpen stage, which decodes its input again, the pixel stays decoded: six powers fewer)
    if (P->usm)
      lab_to_linear(q, p);
    else
      lab_to_rgb(q, p);
  }
  if (P->usm) {  // stage 5 (editingTools.py:745-817)
    Lab3 q = P->sh ? linear_to_lab(p) : rgb_to_lab(p);
    const double Ln = q.L * 0.01;
    const double amount = (double)P->usm_amount;
    const double thr = (double)P->usm_threshold * 0.01;
    double um = Ln - (double)Lb;
    if (thr > 0.0) {
      const double es = fabs(um);
      um *= es > thr ? 1.0 : es / thr;
    }
    const double hm = Ln > 0.9 ? 1.0 - (Ln - 0.9) * 10.0 : 1.0;
    const double sm = Ln < 0.1 ? Ln * 10.0 : 1.0;
    const double ad = fmin(fmax(hm * sm, 0.1), 1.0);
    q.L = clip01(Ln + um * (amount * 0.02) * ad) * 100.0;
    if (amount > 30.0) {
      const double boost = 1.0 + fabs(um) * (amount * 0.01) * 0.1;
      q.a = fmin(fmax(q.a * boost, -127.0), 127.0);
      q.b = fmin(fmax(q.b * boost, -127.0), 127.0);
    }
    lab_to_rgb(q, p);
  }
  for (int c = 0; c < 3; ++c) {  // stages 6-7
    double v = P->final_clip ? clip01(p[c]) : p[c];
    out[c] = encode_out ? encode(v) : v;
  }
}

}  // namespace ss
