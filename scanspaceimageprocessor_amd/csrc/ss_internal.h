// Host-side declarations shared by the translation units of libscanspace_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ss_device.h"

namespace ss {

// ---- chain.hip ---------------------------------------------------------------------------------
// Pointwise fused chain (stages 1-4, 6, 7) over `batch` images.
// (chains with a Lab edit record the pixels to evaluate again in float64 at P.risk_bits; see RiskCfg, RiskMap)
hipError_t launch_chain(const ChainParams& P, int batch, int out_kind, const float* src, void* dst, hipStream_t st);
// does a chain with these parameters and this output kind ever queue pixels?
bool chain_needs_careful(const ChainParams& P, int out_kind);
// float64 re-evaluation of what the launch before it recorded (same P, batch, src, dst), overwriting those output
// pixels; the record's layout follows from P.usm (tile kernel) or not (pointwise kernel)
hipError_t launch_careful(const ChainParams& P, int batch, int out_kind, const float* src, void* dst, hipStream_t st);
// stages 1-2 of every pixel with the colour correction in float64 (see ccm_f64_kernel): out_kind < 0 -> float32 linear
// RGB for the stages behind; out_kind >= 0 (chains with nothing but gains, the final clip and the encoding behind the
// correction) -> the chain's output
hipError_t launch_ccm_f64(const ChainParams& P, int batch, const float* src, void* dst, int out_kind, hipStream_t st);
// Fused chain with the unsharp mask (stage 5): tile + halo recompute, separable blur in LDS.
// Colour correction: none or the 3-term matrix only (callers run larger expansions as a pointwise pass first).
// `ticket`: eight uint32 of device memory owned by the caller (the per-XCD segment counters); launches that share
// them must be stream-ordered
hipError_t launch_usm(const ChainParams& P, int batch, int out_kind, const float* src, void* dst, uint32_t* ticket,
                      hipStream_t st);

// ---- stats.hip ---------------------------------------------------------------------------------
struct PercentileWork {
  // all device pointers, sized for `batch` images of `n` samples
  uint32_t* samples;  // [batch][3][n] clipped float bit patterns
  uint32_t* hist;     // [batch][3 + 12 + 12][1024]
  uint32_t* state;    // [batch][3][4][2] = {key prefix, residual rank}
};
size_t percentile_samples_bytes(int batch, int64_t n);
size_t percentile_hist_bytes(int batch);
size_t percentile_state_bytes(int batch);
hipError_t launch_negative_params(const float* src, int height, int width, int batch, int step, int64_t n,
                                  int64_t k_low, int64_t k_low_hi, double g_low, int64_t k_high, int64_t k_high_hi,
                                  double g_high, int legacy, const PercentileWork& w, float* params,
                                  ss_percentile_detail* detail, hipStream_t st);

constexpr int kStatsBlocksPerImage = 8;
hipError_t launch_exposure_stats(const float* thumbs, int height, int width, int batch, float lo, float hi,
                                 double* partials /*[batch][kStatsBlocksPerImage][4]*/, double* stats, float* brightness,
                                 hipStream_t st);
hipError_t launch_exposure_multipliers(const float* brightness, int n, int ref_index, int first, int count,
                                       float* mult, hipStream_t st);

constexpr int kLabDetectBlocks = 1024;
// ---- mask.hip ----------------------------------------------------------------------------------
// _apply_masking: candidates by luminance, 8-connected components, components >= min_area zeroed in place.
// workspace: mask_workspace_bytes(batch, h, w) bytes of device memory (about 6 bytes per pixel: the run lists
// are sized for the worst case); count_dev: optional [batch] counters.  w <= 65535.
size_t mask_workspace_bytes(int batch, int h, int w);
hipError_t launch_mask_regions(float* img, int h, int w, int batch, double shadow_limit, double highlight_limit,
                               int min_area, bool float32_luminance, void* workspace, unsigned long long* count_dev,
                               hipStream_t st);

// uint16 -> float32 / 65535 (chain.hip); src 16-byte aligned, dst 16-byte aligned
hipError_t launch_unpack_u16(const uint16_t* src, float* dst, uint64_t n, hipStream_t st);

// area-average downscale (resize.hip); dh <= sh, dw <= sw
hipError_t launch_resize_area(const float* src, int sh, int sw, int batch, float* dst, int dh, int dw, hipStream_t st);

// _rgb_to_lab / _lab_to_rgb as operators of their own (lab.hip): interleaved RGB <-> planes L, a, b
hipError_t launch_rgb_to_lab(const float* rgb, uint64_t n_px, float* L, float* a, float* b, hipStream_t st);
hipError_t launch_lab_to_rgb(const float* L, const float* a, const float* b, uint64_t n_px, float* rgb, hipStream_t st);

hipError_t launch_is_lab(const float* src, int64_t n_px, float* partials /*[kLabDetectBlocks][3]*/, int32_t* out,
                         float* minmax, hipStream_t st);

}  // namespace ss
