// C ABI of libscanspace_hip.so (declared in include/scanspace.h): context management, descriptor
// validation, host-side parameter derivation, workspace and pinned staging.

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <thread>

#include <rccl/rccl.h>  // types and signatures only: the library is resolved at run time

#include "ss_internal.h"

using namespace ss;

namespace {

constexpr size_t kStageBytes = 32u << 20;  // each of the two pinned staging buffers
constexpr int64_t kMaxPixels = (int64_t)1 << 28;  // 32-bit element offsets inside one image

thread_local std::string g_create_error;

}  // namespace

struct ss_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  // lazily grown device workspace
  void* ws_samples = nullptr;
  size_t ws_samples_bytes = 0;
  void* ws_small = nullptr;  // histograms, select state, partials
  size_t ws_small_bytes = 0;
  uint32_t* usm_ticket = nullptr;  // per-XCD segment counters of the unsharp-mask kernel's work queue
  uint32_t* static_ticket = nullptr;  // the same counters inside this GPU's copy of `scratch` (ss_chain_run_static)
  void* ws_stage = nullptr;        // float32 images between the two passes of a staged chain
  size_t ws_stage_bytes = 0;
  void* ws_mask = nullptr;         // labels and areas of the region masking
  size_t ws_mask_bytes = 0;
  void* ws_risk = nullptr;         // queue of the pixels the float64 kernel evaluates again: counters, then entries
  size_t ws_risk_bytes = 0;
  const uint32_t* last_risk_count = nullptr;  // re-evaluated pixels of the most recent chain launch with a record
  // host path
  void* pinned[2] = {nullptr, nullptr};
  hipEvent_t pinned_free[2] = {nullptr, nullptr};
  void* host_src = nullptr;
  void* host_dst = nullptr;
  void* host_neg = nullptr;
  size_t host_src_bytes = 0, host_dst_bytes = 0;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // batch pipeline of the host path: upload and download streams beside `stream`, a small ring of events
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipEvent_t ev_in[4] = {nullptr, nullptr, nullptr, nullptr}, ev_k[4] = {nullptr, nullptr, nullptr, nullptr};
};

// Scratch of the allocation-free entry point (ss_chain_run_static), in the code object's own device memory instead
// of hipMalloc'd buffers: the eight per-XCD segment counters of the unsharp-mask kernel's work queue, and the
// record (one bit per pixel) of the pixels for the float64 re-evaluation (16 MiB + 512 B in all).  Every GPU the library runs on holds its
// own copy at its own address until the process ends.
// In host code this name is only a handle for hipGetSymbolAddress: kernels, memsets and copies get the
// per-device address resolved below, never the name.
constexpr size_t kScratchFlagsAt = 256;   // byte offsets: tickets 0, {any, count} 256, the record's words 512
constexpr size_t kScratchBitsAt = 512;
constexpr size_t kStaticRecordBytes = (size_t)16 << 20;  // record of the static scratch: one bit per pixel
constexpr size_t kScratchBytes = kScratchBitsAt + kStaticRecordBytes;
static __device__ __align__(256) uint32_t scratch[kScratchBytes / sizeof(uint32_t)];

namespace {

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
uint32_t* g_scratch_ptr[kMaxDevices] = {};

// Address of `scratch` on `device`, which must be the current device.  Resolved once per device.
hipError_t resolve_scratch(int device, uint32_t** out) {
  *out = nullptr;
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = static_cast<uint32_t*>(p);
  }
  *out = g_scratch_ptr[device];
  return hipSuccess;
}

int fail(ss_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c)
    c->err = buf;
  else
    g_create_error = buf;
  return code;
}

#define SS_HIP(c, call)                                                                             \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return fail((c), e_ == hipErrorOutOfMemory ? SS_ERR_NOMEM : SS_ERR_HIP, "%s failed: %s", #call, \
                  hipGetErrorString(e_));                                                           \
  } while (0)

int grow(ss_ctx* c, void** p, size_t* have, size_t need) {
  if (*have >= need) return SS_OK;
  if (*p) {
    SS_HIP(c, hipStreamSynchronize(c->stream));
    SS_HIP(c, hipFree(*p));
    *p = nullptr;
    *have = 0;
  }
  SS_HIP(c, hipMalloc(p, need));
  *have = need;
  return SS_OK;
}

bool cheung_terms_ok(int t) {
  switch (t) {
    case 3: case 5: case 7: case 8: case 10: case 11: case 14: case 16: case 17: case 19: case 20: case 22:
      return true;
    default:
      return false;
  }
}
bool rootpoly_terms_ok(int t) { return t == 3 || t == 6 || t == 13 || t == 22; }

int usm_kernel_size(double sigma) {
  int k = (int)(6 * sigma) + 1;  // editingTools.py:756-758
  if (k % 2 == 0) k += 1;
  return k;
}

// Validate a descriptor and derive the kernel parameters from it.
int build_params(ss_ctx* c, const ss_chain_desc* d, ChainParams* P) {
  if (!d) return fail(c, SS_ERR_INVALID, "descriptor is NULL");
  if (d->height <= 0 || d->width <= 0 || d->batch <= 0)
    return fail(c, SS_ERR_INVALID, "height, width and batch must be positive (got %d x %d x %d)", d->height, d->width,
                d->batch);
  if ((int64_t)d->height * d->width > kMaxPixels)
    return fail(c, SS_ERR_UNSUPPORTED, "image of %lld pixels exceeds the %lld-pixel limit per image",
                (long long)d->height * d->width, (long long)kMaxPixels);
  if (d->batch > 65535) return fail(c, SS_ERR_UNSUPPORTED, "batch %d exceeds 65535 images per call", d->batch);
  memset(P, 0, sizeof *P);
  P->height = d->height;
  P->width = d->width;
  P->n_px = (int64_t)d->height * d->width;

  P->neg = d->neg_enabled ? 1 : 0;
  if (P->neg) {
    P->neg_dev = d->neg_params_dev;
    for (int k = 0; k < 3; ++k) {
      P->neg_lo[k] = d->neg_p_low[k];
      P->neg_inv[k] = d->neg_inv_denom[k];
    }
  }

  P->ccm_kind = d->ccm_kind;
  P->ccm_terms = d->ccm_terms;
  if (d->ccm_kind == SS_CCM_CHEUNG) {
    if (!cheung_terms_ok(d->ccm_terms))
      return fail(c, SS_ERR_INVALID, "\"Cheung 2004\" defines no augmented matrix with %d terms", d->ccm_terms);
  } else if (d->ccm_kind == SS_CCM_ROOTPOLY) {
    if (!rootpoly_terms_ok(d->ccm_terms))
      return fail(c, SS_ERR_INVALID, "\"Finlayson 2015\" root-polynomial expansion has 3, 6, 13 or 22 terms, not %d",
                  d->ccm_terms);
  } else if (d->ccm_kind != SS_CCM_NONE) {
    return fail(c, SS_ERR_INVALID, "unknown ccm_kind %d", d->ccm_kind);
  }
  if (d->ccm_kind != SS_CCM_NONE) {
    const int T = d->ccm_terms;
    for (int t = 0; t < T; ++t)
      for (int ch = 0; ch < 3; ++ch) P->ccm_full[t][ch] = d->ccm[ch * T + t];
    for (int t = 0; t < T; ++t)
      for (int ch = 0; ch < 3; ++ch) P->ccm_lo[t][ch] = d->ccm_lo[ch * T + t];
    if (T == 3) memcpy(P->ccm3, d->ccm, sizeof(float) * 9);
    if (T > 3 && T <= 8) memcpy(P->ccm_small, d->ccm, sizeof(float) * 3 * T);
  }

  if (d->n_gain_stages < 0 || d->n_gain_stages > SS_MAX_GAIN_STAGES)
    return fail(c, SS_ERR_INVALID, "n_gain_stages must be 0..%d", SS_MAX_GAIN_STAGES);
  if (d->gain0_dev && d->n_gain_stages < 1)
    return fail(c, SS_ERR_INVALID, "gain0_dev needs at least one gain stage");
  P->n_gain = d->n_gain_stages;
  memcpy(P->gain, d->gain, sizeof P->gain);
  P->gain0_dev = d->gain0_dev;

  P->sh = (d->sh_enabled && (d->sh_shadow_amount != 0.0f || d->sh_highlight_amount != 0.0f)) ? 1 : 0;
  if (P->sh) {
    if (!(d->sh_falloff_power > 0.0f)) return fail(c, SS_ERR_INVALID, "sh_falloff_power must be positive");
    P->sh_ss = (float)((double)d->sh_shadow_amount * 0.05);
    P->sh_hs = (float)((double)d->sh_highlight_amount * 0.05);
    P->sh_log2f = (float)log2((double)d->sh_falloff_power);
    P->sh_chroma = (fabsf(d->sh_shadow_amount) > 50.0f || fabsf(d->sh_highlight_amount) > 50.0f) ? 1 : 0;
    P->sh_shadow_amount = d->sh_shadow_amount;
    P->sh_highlight_amount = d->sh_highlight_amount;
    P->sh_falloff = d->sh_falloff_power;
    P->sh_ln_falloff = log((double)d->sh_falloff_power);
    // the largest |g(f + d) - g(f)| this stage can produce: |L' - L| <= 100 (|ss| + |hs|) |falloff - 1| (and <= 100),
    // d = that / 116, f <= ~1; the chroma shrink moves fx, fz by up to 0.2 |fx - fy| more
    double dl = 100.0 * (fabs((double)P->sh_ss) + fabs((double)P->sh_hs)) * fabs((double)d->sh_falloff_power - 1.0);
    if (!(dl < 100.0)) dl = 100.0;
    const double dm = dl / 116.0 + (P->sh_chroma ? 0.1 : 0.0);
    P->sh_risk_bound = (float)(dm * (3.0 + 3.0 * dm + dm * dm));
  }

  P->usm = (d->usm_enabled && d->usm_amount > 0.0f) ? 1 : 0;
  if (P->usm) {
    if (!(d->usm_radius > 0.0f)) return fail(c, SS_ERR_INVALID, "usm_radius (sigma) must be positive");
    if (!(d->usm_radius < 1.0e6f))  // (also keeps the float -> int conversion below defined)
      return fail(c, SS_ERR_UNSUPPORTED, "usm_radius %g is beyond any supported kernel size", (double)d->usm_radius);
    const int k = usm_kernel_size(d->usm_radius);
    const int r = k / 2;
    if (r < 1 || r > SS_MAX_USM_RADIUS)
      return fail(c, SS_ERR_UNSUPPORTED, "usm_radius %.3f needs a %d-tap kernel; supported sizes are 3..%d taps",
                  d->usm_radius, k, 2 * SS_MAX_USM_RADIUS + 1);
    P->usm_r = r;
    // cv2.getGaussianKernel: taps in double, normalised, rounded to float32
    double g[2 * SS_MAX_USM_RADIUS + 1], sum = 0.0;
    const double s = (double)d->usm_radius;
    for (int i = 0; i < k; ++i) {
      const double x = i - (k - 1) * 0.5;
      g[i] = exp(-(x * x) / (2.0 * s * s));
      sum += g[i];
    }
    for (int i = 0; i <= r; ++i) P->taps[i] = (float)(g[i] / sum);  // symmetric: tap i == tap k-1-i
    P->usm_amount_scaled = (float)((double)d->usm_amount / 50.0);
    P->usm_thr = (float)((double)d->usm_threshold / 100.0);
    P->usm_inv_thr = P->usm_thr > 0.0f ? (float)(100.0 / (double)d->usm_threshold) : 0.0f;
    P->usm_chroma = d->usm_amount > 30.0f ? 1 : 0;
    P->usm_chroma_k = (float)((double)d->usm_amount / 100.0 * 0.1);
    P->usm_amount = d->usm_amount;
    P->usm_threshold = d->usm_threshold;
  }

  P->lab_io = d->lab_io ? 1 : 0;
  if (P->lab_io) {
    if (P->neg || P->ccm_kind != SS_CCM_NONE || P->n_gain || d->final_clip || d->out_kind != SS_OUT_F32_LINEAR)
      return fail(c, SS_ERR_INVALID,
                  "lab_io requires no inversion, no colour correction, no gain stages, no final clip and F32_LINEAR output");
  }
  P->final_clip = d->final_clip ? 1 : 0;
  if (d->out_kind < SS_OUT_F32_LINEAR || d->out_kind > SS_OUT_U8_LINEAR)
    return fail(c, SS_ERR_INVALID, "unknown out_kind %d", d->out_kind);

  return SS_OK;
}

size_t out_bytes_per_px(int out_kind) {
  return out_kind == SS_OUT_U16_SRGB ? 6 : (out_kind == SS_OUT_U8_SRGB || out_kind == SS_OUT_U8_LINEAR) ? 3 : 12;
}

int ensure_pinned(ss_ctx* c) {
  for (int i = 0; i < 2; ++i) {
    if (!c->pinned[i]) SS_HIP(c, hipHostMalloc(&c->pinned[i], kStageBytes, hipHostMallocDefault));
    if (!c->pinned_free[i]) SS_HIP(c, hipEventCreateWithFlags(&c->pinned_free[i], hipEventDisableTiming));
  }
  return SS_OK;
}

// Host side of the staging copies.  One thread moves ~10-15 GB/s, PCIe takes ~50: a few helper threads per
// chunk bring the pageable <-> pinned copy up to the link rate (measured on the MI355X host: 288 MB in 10.1 ms
// with one thread, 5.6 ms with four).  SS_HOST_COPY_THREADS overrides the count (1 = no helpers).
int host_copy_threads() {
  static const int n = [] {
    const char* e = getenv("SS_HOST_COPY_THREADS");
    int x = e ? atoi(e) : 4;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && x > (int)hw) x = (int)hw;
    return x < 1 ? 1 : (x > 16 ? 16 : x);
  }();
  return n;
}

void parallel_memcpy(void* dst, const void* src, size_t n) {
  const int threads = n >= ((size_t)4 << 20) ? host_copy_threads() : 1;
  if (threads == 1) {
    memcpy(dst, src, n);
    return;
  }
  const size_t part = ((n + threads - 1) / threads + 4095) & ~(size_t)4095;
  std::thread pool[16];
  int started = 0;
  for (int t = 1; t < threads; ++t) {
    const size_t o = (size_t)t * part;
    if (o >= n) break;
    const size_t len = n - o < part ? n - o : part;
    pool[started++] = std::thread([=] { memcpy((char*)dst + o, (const char*)src + o, len); });
  }
  memcpy(dst, src, part < n ? part : n);
  for (int t = 0; t < started; ++t) pool[t].join();
}

// Page-locked host memory (hipHostMalloc, hipHostRegister, torch's pinned tensors) needs no staging: the copy
// engine reads and writes it directly.
bool is_pinned_host(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // an ordinary pageable pointer: not an error of ours
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

int copy_h2d(ss_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes >= ((size_t)1 << 20) && is_pinned_host(src)) {
    SS_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return SS_OK;
  }
  int rc = ensure_pinned(c);
  if (rc) return rc;
  size_t off = 0;
  int b = 0;
  while (off < bytes) {
    const size_t n = bytes - off < kStageBytes ? bytes - off : kStageBytes;
    SS_HIP(c, hipEventSynchronize(c->pinned_free[b]));
    parallel_memcpy(c->pinned[b], (const char*)src + off, n);
    SS_HIP(c, hipMemcpyAsync((char*)dst + off, c->pinned[b], n, hipMemcpyHostToDevice, c->stream));
    SS_HIP(c, hipEventRecord(c->pinned_free[b], c->stream));
    off += n;
    b ^= 1;
  }
  return SS_OK;
}

int copy_d2h(ss_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes >= ((size_t)1 << 20) && is_pinned_host(dst)) {
    SS_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    SS_HIP(c, hipStreamSynchronize(c->stream));
    return SS_OK;
  }
  int rc = ensure_pinned(c);
  if (rc) return rc;
  // two-deep pipeline: the copy into one staging buffer overlaps the memcpy out of the other
  size_t off_issue = 0, off_done = 0;
  size_t len[2] = {0, 0};
  int bi = 0, bd = 0;
  while (off_done < bytes) {
    while (off_issue < bytes && off_issue - off_done < 2 * kStageBytes && len[bi] == 0) {
      const size_t n = bytes - off_issue < kStageBytes ? bytes - off_issue : kStageBytes;
      SS_HIP(c, hipMemcpyAsync(c->pinned[bi], (const char*)src + off_issue, n, hipMemcpyDeviceToHost, c->stream));
      SS_HIP(c, hipEventRecord(c->pinned_free[bi], c->stream));
      len[bi] = n;
      off_issue += n;
      bi ^= 1;
    }
    SS_HIP(c, hipEventSynchronize(c->pinned_free[bd]));
    parallel_memcpy((char*)dst + off_done, c->pinned[bd], len[bd]);
    off_done += len[bd];
    len[bd] = 0;
    bd ^= 1;
  }
  return SS_OK;
}

int check_ctx(ss_ctx* c) {
  if (!c) return fail(nullptr, SS_ERR_INVALID, "context is NULL");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "hipSetDevice(%d) failed: %s", c->device, hipGetErrorString(e));
  // The launchers read hipGetLastError() behind their kernel launches; that value is per thread and sticky, so an
  // error some earlier, unrelated call of the host left behind (a refused allocation, an invalidated capture) would
  // be reported as this call's launch failure.  It is taken off here.
  (void)hipGetLastError();
  return SS_OK;
}


// Where the kernels of one call keep their scratch: the unsharp-mask work-queue counters and the record of the
// pixels for the float64 kernel.
struct ChainScratch {
  uint32_t* ticket;
  RiskMap map;
};

size_t risk_record_bytes(const ChainParams& P, int batch) {
  return risk_words_per_image(P.usm ? RISK_TILES : RISK_LINEAR, P.height, P.width) * (size_t)batch * sizeof(unsigned long long);
}

int context_scratch(ss_ctx* c, const ChainParams& P, int batch, ChainScratch* out) {
  int rc = grow(c, &c->ws_risk, &c->ws_risk_bytes, kScratchBitsAt + risk_record_bytes(P, batch));
  if (rc) return rc;
  out->ticket = c->usm_ticket;
  out->map.any = (uint32_t*)((char*)c->ws_risk + kScratchFlagsAt);  // (same layout as the static scratch)
  out->map.count = out->map.any + 1;
  out->map.bits = (unsigned long long*)((char*)c->ws_risk + kScratchBitsAt);
  return SS_OK;
}

// One chain launch (pointwise or tile kernel) and, for chains with a Lab edit, the float64 kernel behind it.
int launch_chain_and_careful(ss_ctx* c, const ChainParams& P_in, int batch, int out_kind, const float* src, void* dst,
                             const ChainScratch& S) {
  ChainParams P = P_in;
  const bool careful = chain_needs_careful(P, out_kind);
  c->last_risk_count = careful ? S.map.count : nullptr;
  if (careful) {
    P.risk_bits = S.map.bits;
    P.risk_any = S.map.any;
    P.risk_count = S.map.count;
    // {any, count} and the record behind them
    SS_HIP(c, hipMemsetAsync(S.map.any, 0, (kScratchBitsAt - kScratchFlagsAt) + risk_record_bytes(P, batch), c->stream));
  }
  hipError_t e = P.usm ? launch_usm(P, batch, out_kind, src, dst, S.ticket, c->stream)
                       : launch_chain(P, batch, out_kind, src, dst, c->stream);
  // (SS_CAREFUL=0, a debugging aid: leave the recorded pixels as the float32 kernels left them - for chains with
  // an unsharp mask that is the stashed blur value, not a result)
  static const bool skip = [] {
    const char* v = getenv("SS_CAREFUL");
    return v && v[0] == '0';
  }();
  if (e == hipSuccess && careful && !skip) e = launch_careful(P, batch, out_kind, src, dst, c->stream);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
  return SS_OK;
}

// Unsharp mask behind a colour correction of more than 3 terms: the fused tile kernel only carries the
// 3-term matrix (larger expansions in it cost more registers than a tile kernel has), so the chain runs as
// two launches per group of images: pointwise {negative inversion, colour correction} into a float32
// staging buffer, then the tile kernel {gains, shadows/highlights, unsharp mask, encode} from there.
// The staging buffer holds as many images as fit in kChainStageBytes (at least one); SS_CHAIN_STAGE_BYTES
// overrides the size (tests use it to force several groups on small batches).
constexpr size_t kChainStageBytes = (size_t)1 << 30;

size_t chain_stage_bytes() {
  const char* e = getenv("SS_CHAIN_STAGE_BYTES");
  const long long x = e ? atoll(e) : 0;
  return x > 0 ? (size_t)x : kChainStageBytes;
}

// Colour corrections whose coefficient table float32 cannot carry take the dense float64 pass (ccm_f64_kernel): with
// the inputs in [0, 1] no term exceeds 1, so the sum of |coefficient x term| of an output channel is at most
// C = sum over the terms of the largest |coefficient| of the term, and float32 keeps the channel to ~4e-7 C.  Held
// against a third of the 1e-5 the result is compared at: C > 8.  (Fits of up to ~20 polynomial terms on 24 swatches stay
// well below that; 22 terms and the root-polynomials of degree 3 and 4 have coefficients of 10 ... 10^5.)
bool f64_pass_disabled() {  // SS_CCM_F64=0: a debugging aid (float32 throughout)
  static const bool never = [] {
    const char* v = getenv("SS_CCM_F64");
    return v && v[0] == '0';
  }();
  return never;
}

bool ccm_needs_f64(const ChainParams& P) {
  if (P.ccm_kind == SS_CCM_NONE || P.ccm_terms <= 3) return false;
  if (f64_pass_disabled()) return false;
  double bound = 0.0;
  for (int t = 0; t < P.ccm_terms; ++t) {
    double m = 0.0;
    for (int ch = 0; ch < 3; ++ch) m = fmax(m, fabs((double)P.ccm_full[t][ch] + (double)P.ccm_lo[t][ch]));
    bound += m;
  }
  return 4.0e-7 * bound > 1.0e-5 / 3.0;
}

// Nothing but gains, the final clip and the output encoding behind stages 1-2: the float64 pass finishes the chain
// itself (no staging image, one launch).
bool f64_pass_finishes_chain(const ChainParams& P) { return !P.usm && !P.sh && !P.lab_io; }

// Two launches per group of images through a float32 staging image: {negative inversion, colour correction}, then
// everything behind them from the staged pixels.  Used for an unsharp mask behind a colour correction of more than 3
// terms, and for every chain whose colour correction takes the float64 pass.
int run_staged(ss_ctx* c, const ss_chain_desc* d, const ChainParams& P, const float* src_dev, void* dst_dev,
               const ChainScratch& S) {
  // In front of an unsharp mask every correction of more than 3 terms takes the float64 pass: the stage blurs the
  // lightness of what it is handed, and amount / 50 times a few 1e-7 of float32 rounding in 14 or 20 products per
  // pixel, through two sRGB encodes, is more than the result may differ by on the dark channel of a saturated pixel
  // (one unit in the last place of the lightness moves such a value by 1e-5 at amount 80).  The staging image then
  // holds the correction's float64 value rounded once.
  const bool dense = ccm_needs_f64(P) || (P.usm && P.ccm_kind != SS_CCM_NONE && P.ccm_terms > 3 && !f64_pass_disabled());
  const size_t image_bytes = (size_t)P.n_px * 3 * sizeof(float);
  if (dense && f64_pass_finishes_chain(P)) {
    const hipError_t e = launch_ccm_f64(P, d->batch, src_dev, dst_dev, d->out_kind, c->stream);
    if (e != hipSuccess) return fail(c, SS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return SS_OK;
  }
  const size_t by_size = chain_stage_bytes() / image_bytes;
  int group = by_size > (size_t)d->batch ? d->batch : (int)by_size;
  if (group < 1) group = 1;
  if (group > d->batch) group = d->batch;
  int rc = grow(c, &c->ws_stage, &c->ws_stage_bytes, (size_t)group * image_bytes);
  if (rc) return rc;
  const size_t out_image_bytes = (size_t)P.n_px * out_bytes_per_px(d->out_kind);
  for (int first = 0; first < d->batch; first += group) {
    const int n = d->batch - first < group ? d->batch - first : group;
    ChainParams A = P;  // stages 1-2
    A.usm = 0;
    A.n_gain = 0;
    A.sh = 0;
    A.lab_io = 0;
    A.final_clip = 0;
    if (A.neg_dev) A.neg_dev += (size_t)first * SS_NEG_PARAM_STRIDE;
    ChainParams B = P;  // stages 3-7
    B.neg = 0;
    B.neg_dev = nullptr;
    B.ccm_kind = SS_CCM_NONE;
    B.ccm_terms = 0;
    if (B.gain0_dev) B.gain0_dev += first;
    A.dst2 = nullptr;  // the second output is made where the first is
    if (B.dst2) B.dst2 += (size_t)first * P.n_px * 3;
    if (dense) {
      const hipError_t e = launch_ccm_f64(A, n, src_dev + (size_t)first * P.n_px * 3, c->ws_stage, -1, c->stream);
      if (e != hipSuccess) return fail(c, SS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    } else {
      rc = launch_chain_and_careful(c, A, n, SS_OUT_F32_LINEAR, src_dev + (size_t)first * P.n_px * 3, c->ws_stage, S);
      if (rc) return rc;
    }
    rc = launch_chain_and_careful(c, B, n, d->out_kind, (const float*)c->ws_stage,
                                  (char*)dst_dev + (size_t)first * out_image_bytes, S);
    if (rc) return rc;
  }
  return SS_OK;
}

bool chain_is_staged(const ChainParams& P) {
  return (P.usm && P.ccm_kind != SS_CCM_NONE && P.ccm_terms != 3) || ccm_needs_f64(P);
}

// src_dev == dst_dev: every pixel is read once and written once by the same lane, which is safe for the pointwise
// kernel with a float32 output.  Not for the tile kernel (halos), not for narrower outputs (the store of one lane
// covers pixels other lanes still have to read), and not for chains whose recorded pixels are evaluated again from
// the SOURCE by the float64 kernel: in place that source is already the float32 result.
int check_in_place(ss_ctx* c, const ChainParams& P, int out_kind, const float* src_dev, const void* dst_dev) {
  if ((const void*)src_dev != dst_dev) return SS_OK;
  if (P.usm || out_bytes_per_px(out_kind) != 12)
    return fail(c, SS_ERR_INVALID, "in-place operation needs a float32 output and no unsharp mask");
  if (chain_needs_careful(P, out_kind))
    return fail(c, SS_ERR_INVALID,
                "in-place operation is not available for chains with a shadows/highlights stage: the pixels float32 "
                "cannot follow are evaluated again from the source image, which an in-place run has overwritten");
  return SS_OK;
}

}  // namespace

extern "C" {


const char* ss_version(void) { return "scanspace-hip 0.1 (gfx950)"; }

int ss_ctx_create(int device, ss_ctx** out_ctx) {
  if (!out_ctx) return fail(nullptr, SS_ERR_INVALID, "out_ctx is NULL");
  *out_ctx = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, SS_ERR_HIP, "no HIP device available (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= count) return fail(nullptr, SS_ERR_INVALID, "device %d out of range [0,%d)", device, count);
  e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, SS_ERR_HIP, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return fail(nullptr, SS_ERR_HIP, "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, SS_ERR_UNSUPPORTED, "device %d is %s; this library only carries gfx950 code", device,
                prop.gcnArchName);
  ss_ctx* c = new ss_ctx();
  c->device = device;
  e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(nullptr, SS_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  e = hipMalloc(reinterpret_cast<void**>(&c->usm_ticket), 256);
  if (e != hipSuccess) {
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return fail(nullptr, SS_ERR_NOMEM, "hipMalloc of the work-queue counter failed: %s", hipGetErrorString(e));
  }
  e = resolve_scratch(device, &c->static_ticket);  // (device is current)
  if (e != hipSuccess) {
    ss_ctx_destroy(c);
    return fail(nullptr, SS_ERR_HIP, "hipGetSymbolAddress of the static scratch failed on device %d: %s", device,
                hipGetErrorString(e));
  }
  if (hipEventCreate(&c->t0) != hipSuccess || hipEventCreate(&c->t1) != hipSuccess) {
    ss_ctx_destroy(c);
    return fail(nullptr, SS_ERR_HIP, "hipEventCreate failed");
  }
  *out_ctx = c;
  return SS_OK;
}

void ss_ctx_destroy(ss_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->ws_samples) (void)hipFree(c->ws_samples);
  if (c->ws_small) (void)hipFree(c->ws_small);
  if (c->usm_ticket) (void)hipFree(c->usm_ticket);
  if (c->ws_stage) (void)hipFree(c->ws_stage);
  if (c->ws_mask) (void)hipFree(c->ws_mask);
  if (c->ws_risk) (void)hipFree(c->ws_risk);
  if (c->host_src) (void)hipFree(c->host_src);
  if (c->host_dst) (void)hipFree(c->host_dst);
  if (c->host_neg) (void)hipFree(c->host_neg);
  for (int i = 0; i < 2; ++i) {
    if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
    if (c->pinned_free[i]) (void)hipEventDestroy(c->pinned_free[i]);
  }
  for (int i = 0; i < 4; ++i) {
    if (c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]);
    if (c->ev_k[i]) (void)hipEventDestroy(c->ev_k[i]);
  }
  if (c->s_in) (void)hipStreamDestroy(c->s_in);
  if (c->s_out) (void)hipStreamDestroy(c->s_out);
  if (c->t0) (void)hipEventDestroy(c->t0);
  if (c->t1) (void)hipEventDestroy(c->t1);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int ss_ctx_set_stream(ss_ctx* c, void* hip_stream) {
  int rc = check_ctx(c);
  if (rc) return rc;
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return SS_OK;
}

void* ss_ctx_get_stream(ss_ctx* c) { return c ? (void*)c->stream : nullptr; }

int ss_ctx_sync(ss_ctx* c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  SS_HIP(c, hipStreamSynchronize(c->stream));
  return SS_OK;
}

const char* ss_last_error(const ss_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

static int chain_run_two(ss_ctx* c, const ss_chain_desc* d, const float* src_dev, void* dst_dev, int out_kind2,
                         void* dst2_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!src_dev || !dst_dev) return fail(c, SS_ERR_INVALID, "src_dev and dst_dev must not be NULL");
  ChainParams P;
  rc = build_params(c, d, &P);
  if (rc) return rc;
  if (dst2_dev && P.lab_io)
    return fail(c, SS_ERR_INVALID, "a chain on Lab data (lab_io) has no 8-bit second output: its values are not RGB");
  P.dst2 = (unsigned char*)dst2_dev;
  P.dst2_encoded = out_kind2 == SS_OUT_U8_SRGB ? 1 : 0;
  c->last_risk_count = nullptr;  // (the workspace it points into may be reallocated below)
  rc = check_in_place(c, P, d->out_kind, src_dev, dst_dev);
  if (rc) return rc;
  ChainScratch S{c->usm_ticket, RiskMap{nullptr, nullptr, nullptr}};
  if (chain_needs_careful(P, d->out_kind)) {
    rc = context_scratch(c, P, d->batch, &S);
    if (rc) return rc;
  }
  if (chain_is_staged(P)) return run_staged(c, d, P, src_dev, dst_dev, S);
  return launch_chain_and_careful(c, P, d->batch, d->out_kind, src_dev, dst_dev, S);
}

int ss_chain_run(ss_ctx* c, const ss_chain_desc* d, const float* src_dev, void* dst_dev) {
  return chain_run_two(c, d, src_dev, dst_dev, SS_OUT_U8_LINEAR, nullptr);
}

int ss_chain_run2(ss_ctx* c, const ss_chain_desc* d, const float* src_dev, void* dst_dev, int out_kind2, void* dst2_dev) {
  if (!c) return fail(nullptr, SS_ERR_INVALID, "context is NULL");
  if (!dst2_dev) return fail(c, SS_ERR_INVALID, "dst2_dev must not be NULL");
  if (out_kind2 != SS_OUT_U8_LINEAR && out_kind2 != SS_OUT_U8_SRGB)
    return fail(c, SS_ERR_INVALID, "the second output is 8-bit: out_kind2 must be SS_OUT_U8_LINEAR or SS_OUT_U8_SRGB, not %d",
                out_kind2);
  if (dst2_dev == dst_dev || dst2_dev == (const void*)src_dev)
    return fail(c, SS_ERR_INVALID, "dst2_dev must not be src_dev or dst_dev");
  return chain_run_two(c, d, src_dev, dst_dev, out_kind2, dst2_dev);
}

int ss_chain_run_static(ss_ctx* c, const ss_chain_desc* d, const float* src_dev, void* dst_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!src_dev || !dst_dev) return fail(c, SS_ERR_INVALID, "src_dev and dst_dev must not be NULL");
  ChainParams P;
  rc = build_params(c, d, &P);
  if (rc) return rc;
  c->last_risk_count = nullptr;
  rc = check_in_place(c, P, d->out_kind, src_dev, dst_dev);
  if (rc) return rc;
  if (P.usm && P.ccm_kind != SS_CCM_NONE && P.ccm_terms != 3)
    return fail(c, SS_ERR_UNSUPPORTED,
                "an unsharp mask behind a %d-term colour correction needs a float32 staging image per image of the "
                "batch, which the static scratch of ss_chain_run_static does not hold: use ss_chain_run",
                P.ccm_terms);
  if (ccm_needs_f64(P)) {
    if (!f64_pass_finishes_chain(P))
      return fail(c, SS_ERR_UNSUPPORTED,
                  "this %d-term colour correction is evaluated in float64, and with shadows / highlights behind it "
                  "that pass writes a float32 staging image per image of the batch, which the static scratch of "
                  "ss_chain_run_static does not hold: use ss_chain_run",
                  P.ccm_terms);
    const hipError_t e = launch_ccm_f64(P, d->batch, src_dev, dst_dev, d->out_kind, c->stream);
    if (e != hipSuccess) return fail(c, SS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return SS_OK;
  }
  if (P.usm) {
    // the work queue's counters are 32-bit tickets over the segments of the whole batch
    const int64_t tiles = (int64_t)((d->width + 127) / 128) * ((d->height + 23) / 24) * d->batch;
    if (tiles > INT32_MAX)
      return fail(c, SS_ERR_UNSUPPORTED, "%lld tiles exceed the 2^31 - 1 the static work-queue counters can hand out",
                  (long long)tiles);
  }
  if (!c->static_ticket) return fail(c, SS_ERR_HIP, "the static scratch was not resolved on device %d", c->device);
  ChainScratch S;
  S.ticket = c->static_ticket;
  S.map.any = (uint32_t*)((char*)c->static_ticket + kScratchFlagsAt);
  S.map.count = S.map.any + 1;
  S.map.bits = (unsigned long long*)((char*)c->static_ticket + kScratchBitsAt);
  if (chain_needs_careful(P, d->out_kind) && risk_record_bytes(P, d->batch) > kStaticRecordBytes)
    return fail(c, SS_ERR_UNSUPPORTED,
                "a chain with a Lab edit (shadows/highlights, unsharp mask) and a float32 or 16-bit output keeps a record "
                "of one bit per pixel for its float64 re-evaluation: %zu bytes for this call, the static scratch of "
                "ss_chain_run_static holds %zu (about 134 million pixels): use ss_chain_run or split the batch",
                risk_record_bytes(P, d->batch), kStaticRecordBytes);
  return launch_chain_and_careful(c, P, d->batch, d->out_kind, src_dev, dst_dev, S);
}

int ss_chain_requeued(ss_ctx* c, uint64_t* n_pixels) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!n_pixels) return fail(c, SS_ERR_INVALID, "n_pixels is NULL");
  *n_pixels = 0;
  if (!c->last_risk_count) return SS_OK;
  uint32_t n = 0;
  SS_HIP(c, hipMemcpyAsync(&n, c->last_risk_count, sizeof n, hipMemcpyDeviceToHost, c->stream));
  SS_HIP(c, hipStreamSynchronize(c->stream));
  *n_pixels = n;
  return SS_OK;
}

int ss_percentile_rank(int64_t n, double fraction, int legacy_numpy, int64_t* lo, int64_t* hi, double* gamma) {
  if (n <= 0 || !lo || !hi || !gamma || !(fraction >= 0.0 && fraction <= 1.0)) return SS_ERR_INVALID;
  // editingTools.py:329 passes `fraction * 100.0`; np.percentile divides by 100 again
  // (in float32 for float32 data under numpy >= 2), then index = (n - 1) * q
  double v;
  if (legacy_numpy) {
    const double q = (fraction * 100.0) / 100.0;
    v = (double)(n - 1) * q;
  } else {
    // np.true_divide(python_float, np.float32(100)) -> float32; (n-1) * q -> float32
    const float q = (float)(fraction * 100.0) / 100.0f;
    v = (double)((float)(n - 1) * q);
  }
  if (v >= (double)(n - 1)) {
    *lo = *hi = n - 1;
    *gamma = 0.0;
  } else if (v < 0.0) {
    *lo = *hi = 0;
    *gamma = 0.0;
  } else {
    const double f = floor(v);
    *lo = (int64_t)f;
    *hi = *lo + 1;
    *gamma = v - f;
  }
  return SS_OK;
}

int32_t ss_negative_sample_step(int32_t height, int32_t width) {
  const int s = (int)sqrt(((double)height * (double)width) / 250000.0);  // editingTools.py:325
  return s < 1 ? 1 : s;
}

int ss_negative_params(ss_ctx* c, const float* src_dev, int32_t height, int32_t width, int32_t batch,
                       double low_percentile, double high_percentile, int legacy_numpy, float* params_dev,
                       ss_percentile_detail* detail_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!src_dev || !params_dev) return fail(c, SS_ERR_INVALID, "src_dev and params_dev must not be NULL");
  if (height <= 0 || width <= 0 || batch <= 0 || batch > 65535)
    return fail(c, SS_ERR_INVALID, "bad geometry %d x %d x %d", height, width, batch);
  if ((int64_t)height * width > kMaxPixels) return fail(c, SS_ERR_UNSUPPORTED, "image too large");
  const int step = ss_negative_sample_step(height, width);
  const int64_t hs = (height + step - 1) / step, ws = (width + step - 1) / step;
  const int64_t n = hs * ws;
  int64_t kl, klh, kh, khh;
  double gl, gh;
  if (ss_percentile_rank(n, low_percentile, legacy_numpy, &kl, &klh, &gl) ||
      ss_percentile_rank(n, high_percentile, legacy_numpy, &kh, &khh, &gh))
    return fail(c, SS_ERR_INVALID, "percentile fractions must lie in [0,1] (got %g, %g)", low_percentile,
                high_percentile);
  rc = grow(c, &c->ws_samples, &c->ws_samples_bytes, percentile_samples_bytes(batch, n));
  if (rc) return rc;
  const size_t hist_b = percentile_hist_bytes(batch), state_b = percentile_state_bytes(batch);
  rc = grow(c, &c->ws_small, &c->ws_small_bytes, hist_b + state_b);
  if (rc) return rc;
  PercentileWork w;
  w.samples = (uint32_t*)c->ws_samples;
  w.hist = (uint32_t*)c->ws_small;
  w.state = (uint32_t*)((char*)c->ws_small + hist_b);
  hipError_t e = launch_negative_params(src_dev, height, width, batch, step, n, kl, klh, gl, kh, khh, gh,
                                        legacy_numpy ? 1 : 0, w, params_dev, detail_dev, c->stream);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "percentile kernels failed to launch: %s", hipGetErrorString(e));
  return SS_OK;
}

int ss_exposure_stats(ss_ctx* c, const float* thumbs_dev, int32_t height, int32_t width, int32_t batch,
                      float shadow_threshold, float highlight_threshold, double* stats_dev, float* brightness_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!thumbs_dev || !brightness_dev) return fail(c, SS_ERR_INVALID, "thumbs_dev and brightness_dev must not be NULL");
  if (height <= 0 || width <= 0 || batch <= 0 || batch > 65535 || (int64_t)height * width > kMaxPixels)
    return fail(c, SS_ERR_INVALID, "bad geometry %d x %d x %d", height, width, batch);
  rc = grow(c, &c->ws_small, &c->ws_small_bytes, (size_t)batch * kStatsBlocksPerImage * 4 * sizeof(double));
  if (rc) return rc;
  hipError_t e = launch_exposure_stats(thumbs_dev, height, width, batch, shadow_threshold, highlight_threshold,
                                       (double*)c->ws_small, stats_dev, brightness_dev, c->stream);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "exposure kernels failed to launch: %s", hipGetErrorString(e));
  return SS_OK;
}

int ss_exposure_multipliers(ss_ctx* c, const float* brightness_dev, int32_t n, int32_t ref_index, int32_t first,
                            int32_t count, float* mult_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!brightness_dev || !mult_dev) return fail(c, SS_ERR_INVALID, "brightness_dev and mult_dev must not be NULL");
  if (n <= 0 || ref_index >= n || first < 0 || count < 0 || first + count > n)
    return fail(c, SS_ERR_INVALID, "bad range: n=%d ref_index=%d first=%d count=%d", n, ref_index, first, count);
  hipError_t e = launch_exposure_multipliers(brightness_dev, n, ref_index, first, count, mult_dev, c->stream);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "multiplier kernel failed to launch: %s", hipGetErrorString(e));
  return SS_OK;
}

int ss_is_lab_image(ss_ctx* c, const float* src_dev, int64_t n_pixels, int32_t* out_dev, float* minmax_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!src_dev || !out_dev || n_pixels <= 0) return fail(c, SS_ERR_INVALID, "bad arguments");
  rc = grow(c, &c->ws_small, &c->ws_small_bytes, (size_t)kLabDetectBlocks * 3 * sizeof(float));
  if (rc) return rc;
  hipError_t e = launch_is_lab(src_dev, n_pixels, (float*)c->ws_small, out_dev, minmax_dev, c->stream);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "lab-detect kernels failed to launch: %s", hipGetErrorString(e));
  return SS_OK;
}

int ss_rgb_to_lab(ss_ctx* c, const float* rgb_dev, uint64_t n_pixels, float* L_dev, float* a_dev, float* b_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!rgb_dev || !L_dev || !a_dev || !b_dev) return fail(c, SS_ERR_INVALID, "NULL pointer");
  if (n_pixels == 0) return SS_OK;
  hipError_t e = launch_rgb_to_lab(rgb_dev, n_pixels, L_dev, a_dev, b_dev, c->stream);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "rgb->lab kernel failed to launch: %s", hipGetErrorString(e));
  return SS_OK;
}

int ss_lab_to_rgb(ss_ctx* c, const float* L_dev, const float* a_dev, const float* b_dev, uint64_t n_pixels,
                  float* rgb_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!rgb_dev || !L_dev || !a_dev || !b_dev) return fail(c, SS_ERR_INVALID, "NULL pointer");
  if (n_pixels == 0) return SS_OK;
  hipError_t e = launch_lab_to_rgb(L_dev, a_dev, b_dev, n_pixels, rgb_dev, c->stream);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "lab->rgb kernel failed to launch: %s", hipGetErrorString(e));
  return SS_OK;
}

int ss_unpack_u16(ss_ctx* c, const uint16_t* src_dev, float* dst_dev, uint64_t n_values) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!src_dev || !dst_dev) return fail(c, SS_ERR_INVALID, "src_dev and dst_dev must not be NULL");
  if (((uintptr_t)src_dev | (uintptr_t)dst_dev) & 15)
    return fail(c, SS_ERR_INVALID, "src_dev and dst_dev must be 16-byte aligned");
  if (n_values == 0) return SS_OK;
  hipError_t e = launch_unpack_u16(src_dev, dst_dev, n_values, c->stream);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "unpack kernel failed to launch: %s", hipGetErrorString(e));
  return SS_OK;
}

int ss_mask_regions(ss_ctx* c, float* img_dev, int32_t height, int32_t width, int32_t batch, double shadow_limit,
                    double highlight_limit, int32_t min_area, uint64_t* masked_count_dev) {
  return ss_mask_regions_f32(c, img_dev, height, width, batch, shadow_limit, highlight_limit, min_area, 0,
                             masked_count_dev);
}

int ss_mask_regions_f32(ss_ctx* c, float* img_dev, int32_t height, int32_t width, int32_t batch, double shadow_limit,
                        double highlight_limit, int32_t min_area, int32_t float32_luminance, uint64_t* masked_count_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!img_dev) return fail(c, SS_ERR_INVALID, "img_dev must not be NULL");
  if (height < 1 || width < 1 || batch < 1 || batch > 65535 || width > 65535 ||
      (int64_t)height * width > INT32_MAX)
    return fail(c, SS_ERR_INVALID, "bad image shape %d x %d x %d", batch, height, width);
  if (!(shadow_limit == shadow_limit) || !(highlight_limit == highlight_limit))
    return fail(c, SS_ERR_INVALID, "limits must not be NaN");
  rc = grow(c, &c->ws_mask, &c->ws_mask_bytes, mask_workspace_bytes(batch, height, width));
  if (rc) return rc;
  hipError_t e = launch_mask_regions(img_dev, height, width, batch, shadow_limit, highlight_limit, min_area,
                                     float32_luminance != 0, c->ws_mask,
                                     reinterpret_cast<unsigned long long*>(masked_count_dev), c->stream);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "masking kernels failed to launch: %s", hipGetErrorString(e));
  return SS_OK;
}

int ss_resize_area(ss_ctx* c, const float* src_dev, int32_t height, int32_t width, int32_t batch, float* dst_dev,
                   int32_t dst_height, int32_t dst_width) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!src_dev || !dst_dev) return fail(c, SS_ERR_INVALID, "src_dev and dst_dev must not be NULL");
  if (height < 1 || width < 1 || batch < 1 || batch > 65535)
    return fail(c, SS_ERR_INVALID, "bad source shape %d x %d x %d", batch, height, width);
  if (dst_height < 1 || dst_width < 1 || dst_height > height || dst_width > width || dst_height > 65535)
    return fail(c, SS_ERR_INVALID, "area resize only shrinks: %d x %d -> %d x %d", height, width, dst_height, dst_width);
  hipError_t e = launch_resize_area(src_dev, height, width, batch, dst_dev, dst_height, dst_width, c->stream);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "resize kernel failed to launch: %s", hipGetErrorString(e));
  return SS_OK;
}

int ss_resize_area_host(ss_ctx* c, const float* src_host, int32_t height, int32_t width, float* dst_host,
                        int32_t dst_height, int32_t dst_width) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!src_host || !dst_host) return fail(c, SS_ERR_INVALID, "src_host and dst_host must not be NULL");
  if (height < 1 || width < 1 || dst_height < 1 || dst_width < 1)
    return fail(c, SS_ERR_INVALID, "bad shape");
  const size_t in_bytes = (size_t)height * width * 3 * sizeof(float);
  const size_t out_bytes = (size_t)dst_height * dst_width * 3 * sizeof(float);
  rc = grow(c, &c->host_src, &c->host_src_bytes, in_bytes);
  if (rc) return rc;
  rc = grow(c, &c->host_dst, &c->host_dst_bytes, out_bytes);
  if (rc) return rc;
  rc = copy_h2d(c, c->host_src, src_host, in_bytes);
  if (rc) return rc;
  rc = ss_resize_area(c, (const float*)c->host_src, height, width, 1, (float*)c->host_dst, dst_height, dst_width);
  if (rc) return rc;
  return copy_d2h(c, dst_host, c->host_dst, out_bytes);
}

int ss_malloc(ss_ctx* c, uint64_t bytes, void** out_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!out_dev || bytes == 0) return fail(c, SS_ERR_INVALID, "bad arguments");
  SS_HIP(c, hipMalloc(out_dev, bytes));
  return SS_OK;
}

int ss_free(ss_ctx* c, void* dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (dev) {
    SS_HIP(c, hipStreamSynchronize(c->stream));
    SS_HIP(c, hipFree(dev));
  }
  return SS_OK;
}

int ss_host_alloc(ss_ctx* c, uint64_t bytes, void** out_host) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!out_host || bytes == 0) return fail(c, SS_ERR_INVALID, "bad arguments");
  SS_HIP(c, hipHostMalloc(out_host, bytes, hipHostMallocDefault));
  return SS_OK;
}

int ss_host_free(ss_ctx* c, void* host) {
  if (!host) return SS_OK;
  if (c) {  // (a buffer may outlive its context: ctx == NULL frees without waiting for a stream)
    int rc = check_ctx(c);
    if (rc) return rc;
    SS_HIP(c, hipStreamSynchronize(c->stream));
  }
  hipError_t e = hipHostFree(host);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "hipHostFree failed: %s", hipGetErrorString(e));
  return SS_OK;
}

int ss_upload(ss_ctx* c, void* dst_dev, const void* src_host, uint64_t bytes) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!dst_dev || !src_host) return fail(c, SS_ERR_INVALID, "NULL pointer");
  rc = copy_h2d(c, dst_dev, src_host, bytes);
  if (rc) return rc;
  SS_HIP(c, hipStreamSynchronize(c->stream));
  return SS_OK;
}

int ss_download(ss_ctx* c, void* dst_host, const void* src_dev, uint64_t bytes) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!dst_host || !src_dev) return fail(c, SS_ERR_INVALID, "NULL pointer");
  return copy_d2h(c, dst_host, src_dev, bytes);
}

int ss_upload_async(ss_ctx* c, void* dst_dev, const void* src_host, uint64_t bytes) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!dst_dev || !src_host) return fail(c, SS_ERR_INVALID, "NULL pointer");
  if (bytes == 0) return SS_OK;
  if (!is_pinned_host(src_host))
    return fail(c, SS_ERR_INVALID, "ss_upload_async needs page-locked host memory (ss_host_alloc); use ss_upload");
  SS_HIP(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
  return SS_OK;
}

int ss_download_async(ss_ctx* c, void* dst_host, const void* src_dev, uint64_t bytes) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!dst_host || !src_dev) return fail(c, SS_ERR_INVALID, "NULL pointer");
  if (bytes == 0) return SS_OK;
  if (!is_pinned_host(dst_host))
    return fail(c, SS_ERR_INVALID, "ss_download_async needs page-locked host memory (ss_host_alloc); use ss_download");
  SS_HIP(c, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
  return SS_OK;
}

namespace {

constexpr int kPipeEvents = 4;

int ensure_pipeline(ss_ctx* c) {
  if (!c->s_in) SS_HIP(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
  if (!c->s_out) SS_HIP(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
  for (int i = 0; i < kPipeEvents; ++i) {
    if (!c->ev_in[i]) SS_HIP(c, hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
    if (!c->ev_k[i]) SS_HIP(c, hipEventCreateWithFlags(&c->ev_k[i], hipEventDisableTiming));
  }
  return SS_OK;
}

bool host_pipeline_enabled() {
  static const bool on = [] {
    const char* e = getenv("SS_HOST_PIPELINE");
    return !(e && e[0] == '0');
  }();
  return on;
}

// A batch in page-locked host memory, image by image: the upload of image i + 1, the kernels of image i and the
// download of image i - 1 run on three streams at once (PCIe carries both directions together), ordered by
// events.  The device buffers hold the whole batch, so no image waits for another's buffer.  Images are
// independent in every stage (the negative percentiles are per image, editingTools.py:324-330).
int run_host_pipelined(ss_ctx* c, const ss_chain_desc* d, const float* src_host, void* dst_host,
                       const ss_host_opts* opts) {
  int rc = ensure_pipeline(c);
  if (rc) return rc;
  const size_t in_img = (size_t)d->height * d->width * 12;
  const size_t out_img = (size_t)d->height * d->width * out_bytes_per_px(d->out_kind);
  const bool neg_auto = opts && opts->neg_auto && d->neg_enabled && !d->neg_params_dev;
  auto step = [&](int i) -> int {
    const int slot = i % kPipeEvents;
    const float* src_i = (const float*)((const char*)c->host_src + (size_t)i * in_img);
    void* dst_i = (char*)c->host_dst + (size_t)i * out_img;
    SS_HIP(c, hipMemcpyAsync((void*)src_i, (const char*)src_host + (size_t)i * in_img, in_img, hipMemcpyHostToDevice,
                             c->s_in));
    SS_HIP(c, hipEventRecord(c->ev_in[slot], c->s_in));
    SS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_in[slot], 0));
    ss_chain_desc dd = *d;
    dd.batch = 1;
    if (dd.gain0_dev) dd.gain0_dev += i;
    if (neg_auto) {
      float* p = (float*)c->host_neg + (size_t)i * SS_NEG_PARAM_STRIDE;
      int r = ss_negative_params(c, src_i, d->height, d->width, 1, opts->neg_low_percentile, opts->neg_high_percentile,
                                 opts->legacy_numpy, p, nullptr);
      if (r) return r;
      dd.neg_params_dev = p;
    } else if (dd.neg_params_dev) {
      dd.neg_params_dev += (size_t)i * SS_NEG_PARAM_STRIDE;
    }
    int r = ss_chain_run(c, &dd, src_i, dst_i);
    if (r) return r;
    SS_HIP(c, hipEventRecord(c->ev_k[slot], c->stream));
    SS_HIP(c, hipStreamWaitEvent(c->s_out, c->ev_k[slot], 0));
    SS_HIP(c, hipMemcpyAsync((char*)dst_host + (size_t)i * out_img, dst_i, out_img, hipMemcpyDeviceToHost, c->s_out));
    return SS_OK;
  };
  for (int i = 0; i < d->batch && rc == SS_OK; ++i) rc = step(i);
  // success or not, nothing may still be reading or writing the caller's buffers on return
  const hipError_t e0 = hipStreamSynchronize(c->s_in), e1 = hipStreamSynchronize(c->stream),
                   e2 = hipStreamSynchronize(c->s_out);
  if (rc) return rc;
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess)
    return fail(c, SS_ERR_HIP, "host pipeline failed: %s",
                hipGetErrorString(e0 != hipSuccess ? e0 : e1 != hipSuccess ? e1 : e2));
  return SS_OK;
}

}  // namespace

int ss_chain_run_host(ss_ctx* c, const ss_chain_desc* d, const float* src_host, void* dst_host, ss_host_opts* opts) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!d || !src_host || !dst_host) return fail(c, SS_ERR_INVALID, "NULL argument");
  ChainParams P;
  rc = build_params(c, d, &P);  // validate before moving any data
  if (rc) return rc;
  const size_t n_px = (size_t)d->height * d->width * d->batch;
  const size_t in_bytes = n_px * 12, out_bytes = n_px * out_bytes_per_px(d->out_kind);
  rc = grow(c, &c->host_src, &c->host_src_bytes, in_bytes);
  if (rc) return rc;
  rc = grow(c, &c->host_dst, &c->host_dst_bytes, out_bytes);
  if (rc) return rc;
  if (!c->host_neg) SS_HIP(c, hipMalloc(&c->host_neg, 65536 * SS_NEG_PARAM_STRIDE * sizeof(float)));
  if (d->batch > 1 && !(opts && opts->lab_auto) && in_bytes / d->batch >= ((size_t)1 << 20) && host_pipeline_enabled() &&
      is_pinned_host(src_host) && is_pinned_host(dst_host)) {
    if (opts) opts->lab_detected = 0;
    return run_host_pipelined(c, d, src_host, dst_host, opts);
  }
  rc = copy_h2d(c, c->host_src, src_host, in_bytes);
  if (rc) return rc;
  ss_chain_desc dd = *d;
  if (opts && opts->neg_auto && d->neg_enabled && !d->neg_params_dev) {
    rc = ss_negative_params(c, (const float*)c->host_src, d->height, d->width, d->batch, opts->neg_low_percentile,
                            opts->neg_high_percentile, opts->legacy_numpy, (float*)c->host_neg, nullptr);
    if (rc) return rc;
    dd.neg_params_dev = (const float*)c->host_neg;
  }
  if (opts) opts->lab_detected = 0;
  if (opts && opts->lab_auto && !d->lab_io) {
    // the flag word lives behind the (unused here) tail of the negative-parameter table
    int32_t* flag_dev = (int32_t*)((char*)c->host_neg + 65535 * SS_NEG_PARAM_STRIDE * sizeof(float));
    rc = ss_is_lab_image(c, (const float*)c->host_src, (int64_t)n_px, flag_dev, nullptr);
    if (rc) return rc;
    int32_t flag = 0;
    SS_HIP(c, hipMemcpyAsync(&flag, flag_dev, sizeof flag, hipMemcpyDeviceToHost, c->stream));
    SS_HIP(c, hipStreamSynchronize(c->stream));
    if (flag) {
      opts->lab_detected = 1;
      dd.lab_io = 1;
    }
  }
  rc = ss_chain_run(c, &dd, (const float*)c->host_src, c->host_dst);
  if (rc) return rc;
  return copy_d2h(c, dst_host, c->host_dst, out_bytes);
}

}  // extern "C"

// ---- multi-GPU ---------------------------------------------------------------------------------------------------

namespace {

// RCCL's entry points, looked up once: first an RCCL that the process has loaded already (a host that made its own
// communicator must meet the same library here), then the system's.
struct Rccl {
  void* handle = nullptr;
  std::string why;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclBroadcast) broadcast = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

const Rccl& rccl() {
  static const Rccl lib = [] {
    Rccl r;
    const char* names[4] = {getenv("SS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !r.handle; ++pass)
      for (const char* n : names)
        if (n && n[0] && !r.handle) r.handle = dlopen(n, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD) : (RTLD_NOW | RTLD_LOCAL));
    if (!r.handle) {
      const char* e = dlerror();
      r.why = std::string("RCCL could not be loaded (librccl.so.1; SS_RCCL_LIB names another): ") + (e ? e : "not found");
      return r;
    }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(r.handle, name);
      if (!p && r.why.empty()) r.why = std::string("RCCL lacks ") + name;
      return p;
    };
    r.get_unique_id = (decltype(r.get_unique_id))sym("ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))sym("ncclCommInitRank");
    r.comm_destroy = (decltype(r.comm_destroy))sym("ncclCommDestroy");
    r.all_gather = (decltype(r.all_gather))sym("ncclAllGather");
    r.broadcast = (decltype(r.broadcast))sym("ncclBroadcast");
    r.group_start = (decltype(r.group_start))sym("ncclGroupStart");
    r.group_end = (decltype(r.group_end))sym("ncclGroupEnd");
    r.error_string = (decltype(r.error_string))sym("ncclGetErrorString");
    return r;
  }();
  return lib;
}

#define SS_RCCL(c, R, call)                                                                                   \
  do {                                                                                                        \
    ncclResult_t r_ = (call);                                                                                 \
    if (r_ != ncclSuccess) return fail((c), SS_ERR_HIP, "%s failed: %s", #call, (R).error_string(r_));        \
  } while (0)

void shard_of(int64_t n, int rank, int world, int64_t* first, int64_t* count) {
  const int64_t base = n / world, extra = n % world;
  *first = rank * base + (rank < extra ? rank : extra);
  *count = base + (rank < extra ? 1 : 0);
}

}  // namespace

static_assert(SS_COMM_ID_BYTES == sizeof(ncclUniqueId), "the identifier is RCCL's ncclUniqueId");

extern "C" {

int ss_shard_range(int64_t n_items, int32_t rank, int32_t world, int64_t* first, int64_t* count) {
  if (!first || !count || world < 1 || rank < 0 || rank >= world || n_items < 0) return SS_ERR_INVALID;
  shard_of(n_items, rank, world, first, count);
  return SS_OK;
}

int ss_comm_unique_id(char id[SS_COMM_ID_BYTES]) {
  if (!id) return fail(nullptr, SS_ERR_INVALID, "id is NULL");
  const Rccl& R = rccl();
  if (!R.why.empty()) return fail(nullptr, SS_ERR_UNSUPPORTED, "%s", R.why.c_str());
  ncclUniqueId u;
  SS_RCCL(nullptr, R, R.get_unique_id(&u));
  memcpy(id, &u, sizeof u);
  return SS_OK;
}

int ss_comm_create(ss_ctx* c, const char id[SS_COMM_ID_BYTES], int32_t rank, int32_t world, void** comm) {
  int rc = check_ctx(c);  // (makes the context's GPU current: the communicator binds to it)
  if (rc) return rc;
  if (!id || !comm) return fail(c, SS_ERR_INVALID, "id and comm must not be NULL");
  *comm = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(c, SS_ERR_INVALID, "rank %d of %d", rank, world);
  const Rccl& R = rccl();
  if (!R.why.empty()) return fail(c, SS_ERR_UNSUPPORTED, "%s", R.why.c_str());
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclComm_t k = nullptr;
  SS_RCCL(c, R, R.comm_init_rank(&k, world, u, rank));
  *comm = (void*)k;
  return SS_OK;
}

int ss_comm_destroy(ss_ctx* c, void* comm) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!comm) return SS_OK;
  const Rccl& R = rccl();
  if (!R.why.empty()) return fail(c, SS_ERR_UNSUPPORTED, "%s", R.why.c_str());
  SS_HIP(c, hipStreamSynchronize(c->stream));
  SS_RCCL(c, R, R.comm_destroy((ncclComm_t)comm));
  return SS_OK;
}

int ss_allgather_stats(ss_ctx* c, void* comm, int32_t rank, int32_t world, int64_t n_images, const float* local_dev,
                       float* all_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!comm || !all_dev) return fail(c, SS_ERR_INVALID, "comm and all_dev must not be NULL");
  if (world < 1 || rank < 0 || rank >= world || n_images < 0)
    return fail(c, SS_ERR_INVALID, "rank %d of %d, %lld images", rank, world, (long long)n_images);
  if (n_images == 0) return SS_OK;
  {
    // a rank whose shard is empty (fewer images than ranks) has nothing to send and may pass no buffer; it still
    // has to take part in the collective, or the other ranks wait for it
    int64_t my_first, my_count;
    shard_of(n_images, rank, world, &my_first, &my_count);
    if (!local_dev && my_count > 0)
      return fail(c, SS_ERR_INVALID, "local_dev is NULL but rank %d of %d holds %lld of the %lld images", rank, world,
                  (long long)my_count, (long long)n_images);
    if (!local_dev) local_dev = all_dev;  // never read: this rank is the root of no broadcast
  }
  const Rccl& R = rccl();
  if (!R.why.empty()) return fail(c, SS_ERR_UNSUPPORTED, "%s", R.why.c_str());
  ncclComm_t k = (ncclComm_t)comm;
  // (SS_ALLGATHER_BROADCASTS=1: the uneven form for even shards too, for tests)
  const char* v = getenv("SS_ALLGATHER_BROADCASTS");
  const bool always_v = v && v[0] == '1';
  if (n_images % world == 0 && !always_v) {
    SS_RCCL(c, R, R.all_gather(local_dev, all_dev, (size_t)(n_images / world), ncclFloat, k, c->stream));
    return SS_OK;
  }
  // uneven shards: rank r's values go out from r, one group so that the launches fuse
  SS_RCCL(c, R, R.group_start());
  for (int r = 0; r < world; ++r) {
    int64_t first, count;
    shard_of(n_images, r, world, &first, &count);
    if (count == 0) continue;
    ncclResult_t e = R.broadcast(local_dev, all_dev + first, (size_t)count, ncclFloat, r, k, c->stream);
    if (e != ncclSuccess) {
      (void)R.group_end();
      return fail(c, SS_ERR_HIP, "ncclBroadcast from rank %d failed: %s", r, R.error_string(e));
    }
  }
  SS_RCCL(c, R, R.group_end());
  return SS_OK;
}

int ss_timer_start(ss_ctx* c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  SS_HIP(c, hipEventRecord(c->t0, c->stream));
  return SS_OK;
}

int ss_timer_stop(ss_ctx* c, float* elapsed_ms) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!elapsed_ms) return fail(c, SS_ERR_INVALID, "elapsed_ms is NULL");
  SS_HIP(c, hipEventRecord(c->t1, c->stream));
  SS_HIP(c, hipEventSynchronize(c->t1));
  SS_HIP(c, hipEventElapsedTime(elapsed_ms, c->t0, c->t1));
  return SS_OK;
}

}  // extern "C"
