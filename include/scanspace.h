/*
 * scanspace.h -- C ABI of libscanspace_hip.so: the per-image correction chain of
 * "Image Space" (hglmz/scanSpaceImageProcessor) as hand-written gfx950 HIP kernels.
 *
 * The reference has no FFI layer: its boundary is the Python call surface of
 * ImageProcessor/editingTools.py and ImageProcessor/imageProcessorWorker.py.
 * Each entry point below names the reference function(s) it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - every function returns 0 (SS_OK) or a negative ss_status; the text of the
 *     last failure on a context is ss_last_error(ctx).
 *   - images are interleaved RGB, C-contiguous (H, W, 3) float32; a batch is
 *     `batch` such images of identical size, back to back.
 *   - "*_dev" pointers are device (HBM) addresses on the context's GPU, "*_host"
 *     pointers are ordinary host memory.  The caller owns every buffer it passes;
 *     the library keeps no pointer after a call returns (device-pointer calls are
 *     asynchronous on the context's stream: keep the buffers alive until
 *     ss_ctx_sync or a stream synchronisation of your own).
 *   - a context is not thread-safe; use one context per worker thread (the
 *     reference runs one QRunnable / Python thread per worker:
 *     image_space.py:3483, networkProcessor.py:2603).  Different contexts share
 *     no mutable state.
 */
#ifndef SCANSPACE_H
#define SCANSPACE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ss_ctx ss_ctx;

typedef enum ss_status {
  SS_OK = 0,
  SS_ERR_INVALID = -1,     /* bad argument / descriptor */
  SS_ERR_HIP = -2,         /* a HIP runtime call failed */
  SS_ERR_NOMEM = -3,       /* device or pinned allocation failed */
  SS_ERR_UNSUPPORTED = -4  /* valid request outside what the kernels implement */
} ss_status;

/* Colour-correction expansion applied per pixel (colour.colour_correction's
 * `method`, called from imageProcessorWorker.py:337). */
typedef enum ss_ccm_kind {
  SS_CCM_NONE = 0,
  SS_CCM_CHEUNG = 1,   /* 'Cheung 2004', ccm_terms in {3,5,7,8,10,11,14,16,17,19,20,22}; the reference uses 3 */
  SS_CCM_ROOTPOLY = 2  /* 'Finlayson 2015' root-polynomial, ccm_terms in {3,6,13,22} (degree 1..4) */
} ss_ccm_kind;

/* Output encodings of ImageCorrectionWorker.encode_image (imageProcessorWorker.py:579-602). */
typedef enum ss_out_kind {
  SS_OUT_F32_LINEAR = 0, /* '.exr' branch: float array unchanged */
  SS_OUT_F32_SRGB = 1,   /* colour.cctf_encoding(x), float */
  SS_OUT_U16_SRGB = 2,   /* np.uint16(np.clip(enc*65535, 0, 65535)) : truncation */
  SS_OUT_U8_SRGB = 3,    /* np.uint8(np.clip(enc*255, 0, 255))      : truncation */
  SS_OUT_U8_LINEAR = 4   /* np.uint8(np.clip(x*255, 0, 255)) without encoding: the preview array of
                            imageProcessorWorker.py:653,922 */
} ss_out_kind;

#define SS_MAX_CCM_TERMS 22
#define SS_MAX_GAIN_STAGES 3
#define SS_MAX_USM_RADIUS 15 /* kernel size int(6*sigma)+1 (made odd) <= 31, i.e. sigma <= 5 */
#define SS_NEG_PARAM_STRIDE 8 /* floats per image in a negative-parameter table */

/*
 * One fused pass over a batch.  Stages run in this order; each is skipped at its
 * neutral setting, exactly like the reference skips the corresponding call.
 *
 *  1. negative inversion      editingTools.py:321,336-337
 *        x = 1 - clip((clip(x,0,1) - p_low[c]) * inv_denom[c], 0, 1)
 *  2. colour correction       colour.colour_correction, imageProcessorWorker.py:337
 *        rgb = CCM(3 x terms) . expand(rgb)
 *  3. gain stages, each `clip(rgb * gain[s][c], 0, 1)`:
 *        exposure-normalisation multiplier   imageProcessorWorker.py:875-876
 *        white-balance multipliers           editingTools.py:636-640
 *        exposure 2**ev                      editingTools.py:460-466
 *  4. shadows/highlights in CIE Lab          editingTools.py:469-544
 *  5. unsharp mask on L                      editingTools.py:714-820
 *  6. final clip to [0,1]                    imageProcessorWorker.py:876,894,900
 *  7. output encoding                        imageProcessorWorker.py:579-602
 */
typedef struct ss_chain_desc {
  int32_t height, width, batch;

  /* 1. negative inversion */
  int32_t neg_enabled;
  float neg_p_low[3];          /* used when neg_params_dev == NULL */
  float neg_inv_denom[3];      /* 1 / max(p_high - p_low, 1e-3) */
  const float* neg_params_dev; /* device table [batch][SS_NEG_PARAM_STRIDE] = {p_low[3], inv_denom[3], 0, 0},
                                  as written by ss_negative_params; overrides the two arrays above */

  /* 2. colour correction */
  int32_t ccm_kind;  /* ss_ccm_kind */
  int32_t ccm_terms; /* T */
  float ccm[3 * SS_MAX_CCM_TERMS]; /* row-major 3 x T */

  /* 3. gain + clip stages */
  int32_t n_gain_stages;              /* 0..SS_MAX_GAIN_STAGES */
  float gain[SS_MAX_GAIN_STAGES][3];  /* per stage, per channel */
  const float* gain0_dev;             /* optional device array [batch]: per-image factor multiplied into stage 0
                                         (dataset exposure normalisation); needs n_gain_stages >= 1 */

  /* 4. shadows / highlights (arguments of adjust_shadows_highlights_combined) */
  int32_t sh_enabled;
  float sh_shadow_amount;
  float sh_highlight_amount;
  float sh_falloff_power; /* reference default 7.0 */

  /* 5. unsharp mask (arguments of adjust_sharpen) */
  int32_t usm_enabled;
  float usm_amount;
  float usm_radius;    /* Gaussian sigma; kernel size int(6*sigma)+1, made odd */
  float usm_threshold;

  /* Data is CIE Lab (L 0..100) in and out: stages 4/5 skip their RGB<->Lab conversions
   * (the `is_lab_input` branches, editingTools.py:504-506, 542-544, 741-743, 818-820).
   * Requires ccm_kind == NONE, no negative inversion, no gain stages, F32_LINEAR output, final_clip == 0. */
  int32_t lab_io;

  /* 6./7. */
  int32_t final_clip; /* clip to [0,1] before encoding */
  int32_t out_kind;   /* ss_out_kind */

  /* Low-order parts of the colour-correction coefficients, same layout as `ccm`: ccm[i] + ccm_lo[i] is the
   * float64 coefficient colour.matrix_colour_correction produced.  Read only for the few pixels the kernels
   * re-evaluate in float64 (a Lab edit that cancels a channel to almost zero); all zero when the caller has
   * float32 coefficients only. */
  float ccm_lo[3 * SS_MAX_CCM_TERMS];
} ss_chain_desc;

/* ---- context ----------------------------------------------------------------------------- */

/* Create a context on HIP device `device` (own non-blocking stream, lazily grown workspace). */
int ss_ctx_create(int device, ss_ctx** out_ctx);
void ss_ctx_destroy(ss_ctx* ctx);
/* Launch on a caller-owned hipStream_t instead of the context's own (NULL restores the own stream). */
int ss_ctx_set_stream(ss_ctx* ctx, void* hip_stream);
/* The hipStream_t the context currently launches on (its own or the caller's), so that a host can order other
 * work - a collective, a copy - on the same stream; NULL for a NULL context. */
void* ss_ctx_get_stream(ss_ctx* ctx);
/* Block until everything queued on the context's stream has finished. */
int ss_ctx_sync(ss_ctx* ctx);
/* Text of the last error recorded on this context ("" if none); ctx == NULL -> last ss_ctx_create failure. */
const char* ss_last_error(const ss_ctx* ctx);
/* Library version string. */
const char* ss_version(void);

/* ---- correction chain -------------------------------------------------------------------- */

/* Fused chain on device-resident data (asynchronous).  src_dev: [batch][H][W][3] float32.
 * dst_dev: same shape, element type given by out_kind (float32 / uint16 / uint8).
 * src_dev == dst_dev is allowed only for float32 outputs of chains without a Lab edit (sh_enabled == 0 and
 * usm_enabled == 0): narrower outputs and the unsharp mask read pixels other lanes have overwritten by then, and the
 * float64 re-evaluation of a shadows/highlights chain (see ss_chain_requeued) reads the source again.  Anything else
 * in place is refused with SS_ERR_INVALID.
 * A colour correction whose coefficients float32 cannot carry - the sum over its terms of the largest
 * |coefficient| above 8, which on 24 swatches means 22 polynomial terms or a root-polynomial of degree 3 or 4 - is
 * evaluated in float64 for every pixel (colour.colour_correction works in float64, imageProcessorWorker.py:337):
 * a pass of its own that finishes the chain when only gains, the final clip and the output encoding follow, and
 * otherwise (shadows / highlights or an unsharp mask behind it) writes a float32 staging image owned by the context,
 * from which the rest of the chain runs.
 * Replaces, for one batch: ImageCorrectionWorker.run steps 5-7 (imageProcessorWorker.py:855-923),
 * apply_all_adjustments (editingTools.py:823-909) and its callees. */
int ss_chain_run(ss_ctx* ctx, const ss_chain_desc* desc, const float* src_dev, void* dst_dev);

/* One pass, two outputs: ss_chain_run, and from the same registers a second, 8-bit image of the corrected pixels:
 * dst2_dev [batch][H][W][3] uint8 in out_kind2 = SS_OUT_U8_LINEAR (the worker's preview,
 * np.clip(corrected, 0, 1) * 255 -> uint8, imageProcessorWorker.py:922: derived from the array that is also encoded
 * and written, imageProcessorWorker.py:907-921) or SS_OUT_U8_SRGB.  Both outputs are bit for bit what two
 * ss_chain_run calls with the two output kinds give; the second costs 3 B/px of stores instead of a second pass.
 * dst2_dev must not overlap src_dev or dst_dev. */
int ss_chain_run2(ss_ctx* ctx, const ss_chain_desc* desc, const float* src_dev, void* dst_dev, int out_kind2,
                  void* dst2_dev);

/* Options of the host-buffer entry point. */
typedef struct ss_host_opts {
  /* neg_auto != 0 with desc->neg_enabled and desc->neg_params_dev == NULL: compute the percentiles on
   * the device first (invert_negative_auto, editingTools.py:296-338) with these fractions; otherwise
   * the descriptor's neg_p_low / neg_inv_denom are used. */
  int32_t neg_auto;
  int32_t legacy_numpy; /* see ss_percentile_rank */
  double neg_low_percentile, neg_high_percentile;
  /* lab_auto != 0: evaluate _is_lab_image (editingTools.py:411-432) on the uploaded data and, when it
   * holds, run stages 4/5 on the LAB-in/LAB-out branch (as if desc->lab_io were set). */
  int32_t lab_auto;
  int32_t lab_detected; /* out: 1 when lab_auto found Lab data */
} ss_host_opts;

/* Same on host buffers (synchronous): pinned staging, async upload, kernels, async download.
 * opts may be NULL (no automatic percentiles, no Lab detection).  Page-locked buffers (ss_host_alloc) are read and
 * written by the copy engines directly, and a batch of them is pipelined image by image: upload of image i + 1,
 * kernels of image i and download of image i - 1 at once, same results (SS_HOST_PIPELINE=0 disables it). */
int ss_chain_run_host(ss_ctx* ctx, const ss_chain_desc* desc, const float* src_host, void* dst_host,
                      ss_host_opts* opts);

/* Lab edits (stages 4 and 5) that cancel one channel of a pixel to almost zero are beyond float32: the two sRGB
 * encodes behind them amplify its rounding to several 16-bit steps, where the reference (float64 throughout,
 * editingTools.py:367-408) has none.  The kernels recognise such pixels and record them (one bit per pixel, owned
 * by the context), and a second kernel evaluates them again in float64 from the source pixel (float32 and 16-bit
 * outputs; 8-bit outputs never need it).  This call reports how many pixels of the most recent ss_chain_run /
 * ss_chain_run_static on the context were evaluated again (for a staged chain: of its last group of images).
 * Synchronises the stream. */
int ss_chain_requeued(ss_ctx* ctx, uint64_t* n_pixels);

/* ss_chain_run for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  No counterpart in the reference.  Same arguments, kernels and results as ss_chain_run,
 * but the scratch of its kernels - the unsharp-mask kernel's work-queue counters (eight per-XCD tickets) and the
 * one-bit-per-pixel record for the float64 re-evaluation (see ss_chain_requeued) - lives in a static device array
 * of the library (16 MiB + 512 B) instead of memory the context allocated, and the call itself performs no
 * hipMalloc, hipFree or other allocating call.  Every GPU holds one copy of that array for the life of the process
 * and every call on a GPU shares it (what must start at zero is cleared in stream order before each launch): calls
 * that overlap on one GPU - two contexts, two streams - race on it; order them on one stream.
 * Limits: a chain with a Lab edit and a float32 or 16-bit output may cover about 134 million pixels per call (the
 * record's 16 MiB; beyond: SS_ERR_UNSUPPORTED); the combination "unsharp mask behind a colour correction of more
 * than 3 terms" needs a staging image per image of the batch and is refused (SS_ERR_UNSUPPORTED: use
 * ss_chain_run), and so does a colour correction that is evaluated in float64 (see ss_chain_run) with shadows /
 * highlights behind it; the batch may hold at most 2^31 - 1 tiles of 128 x 24 pixels.  Asynchronous. */
int ss_chain_run_static(ss_ctx* ctx, const ss_chain_desc* desc, const float* src_dev, void* dst_dev);

/* ---- negative film mode: per-channel percentiles ------------------------------------------ */

/* Per-image detail of the percentile search (all exact order statistics of the strided sample). */
typedef struct ss_percentile_detail {
  float x_low_lo[3], x_low_hi[3];   /* order statistics bracketing the low percentile, per channel */
  float x_high_lo[3], x_high_hi[3]; /* ... the high percentile */
  double p_low[3], p_high[3];       /* interpolated percentiles (float64 lerp; float32 lerp if !legacy_numpy) */
  int32_t step, n_samples;
  int64_t k_low, k_high;            /* lower order-statistic index of each percentile */
} ss_percentile_detail;

/* Index arithmetic of np.percentile(method='linear') as called at editingTools.py:329-330 with
 * `percentile * 100.0`: n samples, fraction in [0,1].  legacy_numpy != 0 -> numpy<2 (float64 index
 * arithmetic, the reference's pin); 0 -> numpy>=2 (float32 arithmetic for float32 data).
 * Pure host function. */
int ss_percentile_rank(int64_t n, double fraction, int legacy_numpy, int64_t* lo, int64_t* hi, double* gamma);

/* Sampling step of invert_negative_auto (editingTools.py:325). Pure host function. */
int32_t ss_negative_sample_step(int32_t height, int32_t width);

/* invert_negative_auto's statistics for a batch (editingTools.py:321-333): clip to [0,1], sample every
 * `step`-th pixel in both directions, exact order statistics per channel via three 10-bit radix-select
 * passes, linear interpolation, denom = max(p_high - p_low, 1e-3).
 * params_dev: device [batch][SS_NEG_PARAM_STRIDE] floats (layout of ss_chain_desc::neg_params_dev).
 * detail_dev: optional device [batch] ss_percentile_detail (may be NULL).  Asynchronous. */
int ss_negative_params(ss_ctx* ctx, const float* src_dev, int32_t height, int32_t width, int32_t batch,
                       double low_percentile, double high_percentile, int legacy_numpy,
                       float* params_dev, ss_percentile_detail* detail_dev);

/* ---- dataset exposure normalisation ------------------------------------------------------- */

/* Per-thumbnail statistics of calculate_average_exposure (image_space.py:4360-4365):
 * lum = 0.2126 R + 0.7152 G + 0.0722 B (float32), mask = (lum > shadow) & (lum < highlight),
 * brightness = mean(lum[mask]) or mean(lum) when the mask is empty.
 * thumbs_dev: [batch][h][w][3] float32.  stats_dev: optional device [batch][4] doubles
 * {masked sum, masked count, total sum, total count}.  brightness_dev: device [batch] float32.
 * Asynchronous. */
int ss_exposure_stats(ss_ctx* ctx, const float* thumbs_dev, int32_t height, int32_t width, int32_t batch,
                      float shadow_threshold, float highlight_threshold,
                      double* stats_dev, float* brightness_dev);

/* Multipliers of calculate_average_exposure (image_space.py:4376-4394) for n images whose
 * brightness values are already gathered on this GPU: reference = brightness[ref_index], or the
 * mean of all n when ref_index < 0; multiplier = reference / brightness (1.0 where brightness <= 0).
 * Writes multipliers for images [first, first+count) to mult_dev[0..count).  Asynchronous. */
int ss_exposure_multipliers(ss_ctx* ctx, const float* brightness_dev, int32_t n, int32_t ref_index,
                            int32_t first, int32_t count, float* mult_dev);

/* ---- multi-GPU: the one exchange step of the dataset exposure normalisation ------------------------------------ */

/* One process per GPU, each with the contiguous range of images ss_shard_range gives it (the reference chunks its
 * image list over worker threads the same way, image_space.py:3423-3425).  Every rank needs the brightness of ALL
 * images to form the reference value (image_space.py:4368-4394): ss_exposure_stats on the local thumbnails ->
 * ss_allgather_stats -> ss_exposure_multipliers for the local range.  The collective is RCCL's, resolved at run
 * time (the RCCL already loaded in the process, else librccl.so.1; SS_RCCL_LIB names another): hosts that never call
 * these functions need no RCCL. */

/* first / count of rank's contiguous, balanced share of n_items (the first n_items % world ranks hold one more). */
int ss_shard_range(int64_t n_items, int32_t rank, int32_t world, int64_t* first, int64_t* count);

#define SS_COMM_ID_BYTES 128
/* A fresh identifier for a communicator (ncclGetUniqueId): call on one rank, hand the bytes to every rank by any
 * means the host has (a file, a socket, MPI), then every rank calls ss_comm_create with it. */
int ss_comm_unique_id(char id[SS_COMM_ID_BYTES]);
/* Collective over all `world` ranks: the communicator (an ncclComm_t) of this rank on the context's GPU. */
int ss_comm_create(ss_ctx* ctx, const char id[SS_COMM_ID_BYTES], int32_t rank, int32_t world, void** comm);
int ss_comm_destroy(ss_ctx* ctx, void* comm);

/* Brightness of all n_images in image order on every rank: local_dev holds this rank's ss_shard_range(n_images, rank,
 * world) values (may be NULL on a rank whose range is empty, i.e. with fewer images than ranks: that rank still has
 * to make the call), all_dev receives n_images values (all_dev may not overlap local_dev).  `comm`: an ncclComm_t of
 * `world` ranks in which this process is `rank` - from ss_comm_create, or the host's own communicator when it was
 * made by the same RCCL library.  Even shards go out as one all-gather, uneven ones as one group of broadcasts;
 * 4 bytes per image, latency-bound.  Queued on the context's stream behind ss_exposure_stats (asynchronous). */
int ss_allgather_stats(ss_ctx* ctx, void* comm, int32_t rank, int32_t world, int64_t n_images, const float* local_dev,
                       float* all_dev);

/* ---- 16-bit RAW decode output -> float32 ------------------------------------------------------ */

/* load_image's `np.array(rgb, dtype=np.float32) / 65535.0` (imageProcessorWorker.py:313; imageLoader.py:92) on
 * the device, so that the 16-bit output of the RAW decoder can be uploaded as it is (half the bytes of the float
 * image, and no float conversion pass on the host): dst[i] = float(src[i]) / 65535.0f, IEEE division, bit for
 * bit numpy's result.  n_values = height * width * 3 of a batch; both pointers 16-byte aligned.  Asynchronous. */
int ss_unpack_u16(ss_ctx* ctx, const uint16_t* src_dev, float* dst_dev, uint64_t n_values);

/* ---- Lab conversions as operators of their own ---------------------------------------------- */

/* _rgb_to_lab (editingTools.py:367-386): colour.sRGB_to_XYZ (sRGB decode first) then colour.XYZ_to_Lab (D65).
 * rgb_dev: n_pixels interleaved float32 triples; L_dev, a_dev, b_dev: n_pixels floats each (the three arrays the
 * reference returns; L in 0..100).  Any input values: the piecewise functions pick their branch as numpy.where
 * does in colour-science.  Asynchronous. */
int ss_rgb_to_lab(ss_ctx* ctx, const float* rgb_dev, uint64_t n_pixels, float* L_dev, float* a_dev, float* b_dev);

/* _lab_to_rgb (editingTools.py:389-408): colour.Lab_to_XYZ, colour.XYZ_to_sRGB (IEC 61966-2-1 matrix, sRGB
 * encode), clip to [0, 1].  Planes in, n_pixels interleaved float32 triples out.  Asynchronous. */
int ss_lab_to_rgb(ss_ctx* ctx, const float* L_dev, const float* a_dev, const float* b_dev, uint64_t n_pixels,
                  float* rgb_dev);

/* ---- LAB-input heuristic ------------------------------------------------------------------ */

/* _is_lab_image (editingTools.py:411-432): max(ch0) > 1.5 and (min(ch1) < -1 or min(ch2) < -1).
 * out_dev: device int32 (1 = looks like Lab).  minmax_dev: optional device float[3] = {max ch0, min ch1, min ch2}.
 * Asynchronous. */
int ss_is_lab_image(ss_ctx* ctx, const float* src_dev, int64_t n_pixels, int32_t* out_dev, float* minmax_dev);

/* ---- preview / thumbnail producers --------------------------------------------------------- */

/* cv2.resize(src, (dst_width, dst_height), interpolation=cv2.INTER_AREA) for a batch of interleaved float32 RGB
 * images, downscaling only (dst_height <= height, dst_width <= width): the 1 MP RAW preview
 * (imageLoader.py:94-121), the <= 512 px thumbnails (imageLoader.py:298-302) and the 4 MP edit preview
 * (image_space.py:2959-2986).  Each destination pixel is the area-weighted mean of the source rectangle it
 * covers (OpenCV's computeResizeAreaTab weights, float32 accumulation, rows reduced first).
 * src_dev: [batch][height][width][3], dst_dev: [batch][dst_height][dst_width][3].  Asynchronous. */
int ss_resize_area(ss_ctx* ctx, const float* src_dev, int32_t height, int32_t width, int32_t batch, float* dst_dev,
                   int32_t dst_height, int32_t dst_width);

/* The same for one host image: upload, resize, download (synchronous, through the pinned staging ring). */
int ss_resize_area_host(ss_ctx* ctx, const float* src_host, int32_t height, int32_t width, float* dst_host,
                        int32_t dst_height, int32_t dst_width);

/* ---- shadow / highlight region masking ------------------------------------------------------ */

/* _apply_masking (imageProcessorWorker.py:979-1030), the step between the chain's last clip and the encode when
 * `export_masked` is set: luminance 0.2126 R + 0.7152 G + 0.0722 B (float64, as on the reference's float64 image);
 * candidate pixels are those with luminance <= shadow_limit or >= highlight_limit; 8-connected components of
 * candidates (cv2.connectedComponentsWithStats(mask, connectivity=8)) with an area of at least min_area pixels
 * (the reference hard-wires 200) are set to 0 in place.  img_dev: [batch][height][width][3] float32.
 * masked_count_dev: optional device uint64[batch], the number of pixels zeroed per image (the reference logs it).
 * The luminance 0.2126 r + 0.7152 g + 0.0722 b and its comparison with the limits are float64 arithmetic on the
 * float32 pixel values: the set of zeroed pixels is the reference's when its image holds these very float32 values
 * and numpy forms the luminance in float64 (chart mode: the corrected image is float64).  Where the reference's
 * image is still float32 (no chart, exposure / white balance only) numpy < 2 forms the luminance and the comparison
 * in float32, and where it is float64 with values float32 cannot hold, a pixel within one rounding of a limit can
 * fall on the other side there - which can keep or drop a whole region whose area is about min_area
 * (ss_mask_regions_f32 covers the float32 case).  Asynchronous.
 * Limits: width <= 65535, height * width < 2^31, batch <= 65535. */
int ss_mask_regions(ss_ctx* ctx, float* img_dev, int32_t height, int32_t width, int32_t batch, double shadow_limit,
                    double highlight_limit, int32_t min_area, uint64_t* masked_count_dev);
/* The same with the luminance arithmetic chosen: float32_luminance != 0 forms 0.2126 r + 0.7152 g + 0.0722 b, and
 * compares it with the limits rounded to float32, in float32 - what the reference's numpy < 2 does while its image
 * is still float32 (imageProcessorWorker.py:989-993 behind a no-chart path with exposure / white balance only);
 * 0 is ss_mask_regions. */
int ss_mask_regions_f32(ss_ctx* ctx, float* img_dev, int32_t height, int32_t width, int32_t batch, double shadow_limit,
                        double highlight_limit, int32_t min_area, int32_t float32_luminance, uint64_t* masked_count_dev);

/* ---- plain device memory helpers for hosts without another HIP binding -------------------- */

int ss_malloc(ss_ctx* ctx, uint64_t bytes, void** out_dev);
int ss_free(ss_ctx* ctx, void* dev);
/* Page-locked host memory.  The reference hands numpy arrays in pageable memory across its call boundary
 * (imageProcessorWorker.py:313, 337), which every host entry point here accepts and stages through the
 * context's pinned ring; buffers from ss_host_alloc (or any other page-locked memory: hipHostMalloc,
 * hipHostRegister, a pinned torch tensor) are recognised and copied by the DMA engines directly, without the
 * staging copy. */
int ss_host_alloc(ss_ctx* ctx, uint64_t bytes, void** out_host);
int ss_host_free(ss_ctx* ctx, void* host); /* ctx may be NULL when the context is already destroyed */
/* Synchronous copies (through the context's pinned staging ring unless the host side is page-locked). */
int ss_upload(ss_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);
int ss_download(ss_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);
/* Asynchronous copies between PAGE-LOCKED host memory and the device, queued on the context's stream in order
 * with its kernels (the device-resident use of the reference's load -> correct -> encode sequence,
 * imageProcessorWorker.py:839-923, without a host stall per step); ss_ctx_sync waits for them.  The host buffer
 * must stay valid and untouched until then.  Pageable memory is refused (SS_ERR_INVALID): it cannot be copied
 * without the calling thread staging it. */
int ss_upload_async(ss_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);
int ss_download_async(ss_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);

/* ---- timing (HIP events on the context's stream) ------------------------------------------ */

/* Record a start / stop event pair around whatever is queued in between and read the elapsed
 * milliseconds after synchronising on the stop event. */
int ss_timer_start(ss_ctx* ctx);
int ss_timer_stop(ss_ctx* ctx, float* elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* SCANSPACE_H */
