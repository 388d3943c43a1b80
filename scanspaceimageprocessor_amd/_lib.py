"""ctypes binding of libscanspace_hip.so (declared in include/scanspace.h).

The library is the only compute path of this package: if it is missing, or no
gfx950 device is available when a context is requested, the call raises --
there is no CPU fallback.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCANSPACE_LIB selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("SCANSPACE_LIB") or os.path.join(_HERE, "libscanspace_hip.so")

SS_OK = 0
SS_ERR_INVALID, SS_ERR_HIP, SS_ERR_NOMEM, SS_ERR_UNSUPPORTED = -1, -2, -3, -4

SS_CCM_NONE, SS_CCM_CHEUNG, SS_CCM_ROOTPOLY = 0, 1, 2
SS_OUT_F32_LINEAR, SS_OUT_F32_SRGB, SS_OUT_U16_SRGB, SS_OUT_U8_SRGB, SS_OUT_U8_LINEAR = 0, 1, 2, 3, 4
SS_MAX_CCM_TERMS = 22
SS_MAX_GAIN_STAGES = 3
SS_MAX_USM_RADIUS = 15
SS_NEG_PARAM_STRIDE = 8

OUT_DTYPES = {
    SS_OUT_F32_LINEAR: np.float32,
    SS_OUT_F32_SRGB: np.float32,
    SS_OUT_U16_SRGB: np.uint16,
    SS_OUT_U8_SRGB: np.uint8,
    SS_OUT_U8_LINEAR: np.uint8,
}


class ChainDesc(C.Structure):
    """Mirror of ``ss_chain_desc``."""

    _fields_ = [
        ("height", C.c_int32), ("width", C.c_int32), ("batch", C.c_int32),
        ("neg_enabled", C.c_int32),
        ("neg_p_low", C.c_float * 3),
        ("neg_inv_denom", C.c_float * 3),
        ("neg_params_dev", C.c_void_p),
        ("ccm_kind", C.c_int32), ("ccm_terms", C.c_int32),
        ("ccm", C.c_float * (3 * SS_MAX_CCM_TERMS)),
        ("n_gain_stages", C.c_int32),
        ("gain", (C.c_float * 3) * SS_MAX_GAIN_STAGES),
        ("gain0_dev", C.c_void_p),
        ("sh_enabled", C.c_int32),
        ("sh_shadow_amount", C.c_float), ("sh_highlight_amount", C.c_float), ("sh_falloff_power", C.c_float),
        ("usm_enabled", C.c_int32),
        ("usm_amount", C.c_float), ("usm_radius", C.c_float), ("usm_threshold", C.c_float),
        ("lab_io", C.c_int32),
        ("final_clip", C.c_int32),
        ("out_kind", C.c_int32),
        ("ccm_lo", C.c_float * (3 * SS_MAX_CCM_TERMS)),
    ]


class HostOpts(C.Structure):
    """Mirror of ``ss_host_opts``."""

    _fields_ = [
        ("neg_auto", C.c_int32), ("legacy_numpy", C.c_int32),
        ("neg_low_percentile", C.c_double), ("neg_high_percentile", C.c_double),
        ("lab_auto", C.c_int32), ("lab_detected", C.c_int32),
    ]


class PercentileDetail(C.Structure):
    """Mirror of ``ss_percentile_detail``."""

    _fields_ = [
        ("x_low_lo", C.c_float * 3), ("x_low_hi", C.c_float * 3),
        ("x_high_lo", C.c_float * 3), ("x_high_hi", C.c_float * 3),
        ("p_low", C.c_double * 3), ("p_high", C.c_double * 3),
        ("step", C.c_int32), ("n_samples", C.c_int32),
        ("k_low", C.c_int64), ("k_high", C.c_int64),
    ]


# name -> (restype, argtypes): every symbol include/scanspace.h declares
SIGNATURES = {
    "ss_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ss_ctx_destroy": (None, [C.c_void_p]),
    "ss_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ss_ctx_get_stream": (C.c_void_p, [C.c_void_p]),
    "ss_ctx_sync": (C.c_int, [C.c_void_p]),
    "ss_last_error": (C.c_char_p, [C.c_void_p]),
    "ss_version": (C.c_char_p, []),
    "ss_chain_run": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.c_void_p, C.c_void_p]),
    "ss_shard_range": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ss_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ss_comm_create": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "ss_comm_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ss_allgather_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "ss_chain_run2": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ss_chain_run_static": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.c_void_p, C.c_void_p]),
    "ss_chain_requeued": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "ss_chain_run_host": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.c_void_p, C.c_void_p, C.POINTER(HostOpts)]),
    "ss_percentile_rank": (C.c_int, [C.c_int64, C.c_double, C.c_int, C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "ss_negative_sample_step": (C.c_int32, [C.c_int32, C.c_int32]),
    "ss_negative_params": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "ss_exposure_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "ss_exposure_multipliers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_void_p]),
    "ss_unpack_u16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ss_is_lab_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ss_rgb_to_lab": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ss_lab_to_rgb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "ss_mask_regions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                  C.c_int32, C.c_void_p]),
    "ss_mask_regions_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                      C.c_int32, C.c_int32, C.c_void_p]),
    "ss_resize_area": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                 C.c_int32]),
    "ss_resize_area_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]),
    "ss_malloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "ss_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ss_host_alloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "ss_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ss_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ss_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ss_upload_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ss_download_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ss_timer_start": (C.c_int, [C.c_void_p]),
    "ss_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
}

_lib = None
_lib_lock = threading.Lock()


class ScanSpaceError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libscanspace_hip error {code}: {message}")
        self.code = code


_hip_runtime_choice = None  # which HIP runtime load_library() arranged for: see hip_runtime_choice()


def _share_torch_hip_runtime(share):
    """One HIP runtime per process.  PyTorch's ROCm wheels bundle their own libamdhip64 (soname libamdhip64.so.7)
    and ask for it by the unversioned file name: loaded AFTER the system runtime this library links to, that is a
    second runtime in the process, whose initialisation finds no GPU ("No HIP GPUs are available").  Loaded
    first, it satisfies this library's dependency by soname and both share it.

    Ordering rule for hosts: a process that uses torch beside this library either imports torch BEFORE the library
    is loaded (nothing to do here then: torch's runtime is in the process already), or asks for the preload
    (``load_library(share_torch_hip=True)`` or SCANSPACE_SHARE_TORCH_HIP=1).  Nothing is preloaded otherwise: a
    process that never imports torch keeps the system runtime the library was built and linked against, and no
    second copy of the HIP symbols is made global."""
    global _hip_runtime_choice
    import importlib.util
    import logging
    import sys

    log = logging.getLogger("scanspaceimageprocessor_amd")
    if "torch" in sys.modules:
        _hip_runtime_choice = "torch (imported before the library)"
        log.debug("HIP runtime: torch's, already in the process")
        return
    if share is None:
        share = os.environ.get("SCANSPACE_SHARE_TORCH_HIP") == "1"
    if not share:
        _hip_runtime_choice = "system"
        log.debug("HIP runtime: the system's (import torch first, or set SCANSPACE_SHARE_TORCH_HIP=1, to share torch's)")
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    path = None
    if spec is not None and spec.submodule_search_locations:
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if path and os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            _hip_runtime_choice = f"torch (preloaded {path})"
            log.debug("HIP runtime: preloaded %s", path)
            return
        except OSError as e:
            log.warning("could not preload %s (%s): the system HIP runtime is used; importing torch later in this "
                        "process may fail", path, e)
    _hip_runtime_choice = "system"


def hip_runtime_choice():
    """"system", or "torch (...)": the HIP runtime ``load_library`` arranged for (None before the first load)."""
    return _hip_runtime_choice


def load_library(share_torch_hip=None):
    """Load libscanspace_hip.so (built in-tree by ``__graft_entry__.build()``) and type its symbols.
    ``share_torch_hip``: preload torch's bundled HIP runtime first (see ``_share_torch_hip_runtime``; None = only when
    SCANSPACE_SHARE_TORCH_HIP=1).  Only the first call in a process loads anything."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
            _share_torch_hip_runtime(share_torch_hip)
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


class Context:
    """One ``ss_ctx``: a GPU, a stream, workspace and pinned staging.  Not thread-safe; use
    ``default_context()`` for a per-thread instance."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.ss_ctx_create(int(device), C.byref(h))
        if rc != SS_OK:
            raise ScanSpaceError(rc, self.lib.ss_last_error(None).decode())
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            for ptr, _ in getattr(self, "_workspaces", {}).values():
                self.lib.ss_free(self.handle, C.c_void_p(ptr))
            self._workspaces = {}
            self.__dict__.pop("_host_workspaces", None)  # (freed by their finalizers)
            self.lib.ss_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != SS_OK:
            raise ScanSpaceError(rc, self.lib.ss_last_error(self.handle).decode())

    # -- streams / sync / timing
    def set_stream(self, hip_stream):
        self.check(self.lib.ss_ctx_set_stream(self.handle, C.c_void_p(hip_stream or None)))

    @property
    def stream(self):
        """The raw hipStream_t (integer address) this context launches on."""
        return int(self.lib.ss_ctx_get_stream(self.handle) or 0)

    def sync(self):
        self.check(self.lib.ss_ctx_sync(self.handle))

    def timer_start(self):
        self.check(self.lib.ss_timer_start(self.handle))

    def timer_stop(self):
        ms = C.c_float()
        self.check(self.lib.ss_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    # -- memory
    def malloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.lib.ss_malloc(self.handle, int(nbytes), C.byref(p)))
        return p.value

    def free(self, ptr):
        self.check(self.lib.ss_free(self.handle, C.c_void_p(ptr)))

    def workspace(self, tag, nbytes):
        """A device buffer of at least ``nbytes`` that stays with the context under ``tag`` (grown when a larger
        one is asked for, released by ``close``): callers that process image after image on one thread reuse it
        instead of paying an allocation and a device-wide synchronising free per image."""
        ws = self.__dict__.setdefault("_workspaces", {})
        have = ws.get(tag)
        if have is not None and have[1] >= nbytes:
            return have[0]
        if have is not None:
            self.sync()
            self.free(have[0])
            del ws[tag]
        ptr = self.malloc(nbytes)
        ws[tag] = (ptr, int(nbytes))
        return ptr

    def pinned_empty(self, shape, dtype=np.float32):
        """An uninitialised numpy array in page-locked host memory: uploads from it and downloads into it skip the
        staging copy.  The memory is released when the array (and every view of it) is gone."""
        import weakref

        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes <= 0:
            return np.empty(shape, dtype=dtype)
        ptr = C.c_void_p()
        self.check(self.lib.ss_host_alloc(self.handle, nbytes, C.byref(ptr)))
        buf = (C.c_ubyte * nbytes).from_address(ptr.value)
        lib, address, owner = self.lib, ptr.value, weakref.ref(self)

        def release():
            ctx = owner()
            lib.ss_host_free(ctx.handle if ctx is not None and ctx.handle else None, C.c_void_p(address))

        weakref.finalize(buf, release)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def host_workspace(self, tag, shape, dtype=np.float32):
        """A page-locked array kept under ``tag`` for as long as the context lives and handed out again on the
        next request (grown when too small): the caller must be done with it before asking again."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        pool = self.__dict__.setdefault("_host_workspaces", {})
        have = pool.get(tag)
        if have is None or have.nbytes < nbytes:
            have = pool[tag] = self.pinned_empty((max(nbytes, 1),), np.uint8)
        return have[:nbytes].view(dtype).reshape(shape)

    def upload(self, dst_dev, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.ss_upload(self.handle, C.c_void_p(dst_dev), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def download(self, arr, src_dev):
        assert arr.flags.c_contiguous and arr.flags.writeable
        self.check(self.lib.ss_download(self.handle, arr.ctypes.data_as(C.c_void_p), C.c_void_p(src_dev), arr.nbytes))

    def upload_async(self, dst_dev, arr):
        """Queue the upload of a page-locked array (``pinned_empty``) on the context's stream; keep ``arr`` alive and
        unchanged until ``sync``."""
        assert arr.flags.c_contiguous
        self.check(self.lib.ss_upload_async(self.handle, C.c_void_p(dst_dev), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def download_async(self, arr, src_dev):
        """Queue the download into a page-locked array on the context's stream; read ``arr`` after ``sync``."""
        assert arr.flags.c_contiguous and arr.flags.writeable
        self.check(self.lib.ss_download_async(self.handle, arr.ctypes.data_as(C.c_void_p), C.c_void_p(src_dev), arr.nbytes))

    # -- entry points on device pointers
    def chain_run(self, desc, src_dev, dst_dev):
        self.check(self.lib.ss_chain_run(self.handle, C.byref(desc), C.c_void_p(src_dev), C.c_void_p(dst_dev)))

    def chain_run2(self, desc, src_dev, dst_dev, out_kind2, dst2_dev):
        """``chain_run`` with a second, 8-bit output (``SS_OUT_U8_LINEAR`` or ``SS_OUT_U8_SRGB``) from the same pass."""
        self.check(self.lib.ss_chain_run2(self.handle, C.byref(desc), C.c_void_p(src_dev), C.c_void_p(dst_dev),
                                          int(out_kind2), C.c_void_p(dst2_dev)))

    # -- multi-GPU: communicator and the one collective
    def comm_create(self, comm_id, rank, world):
        """This rank's communicator (an ``ncclComm_t`` as an integer) from the bytes of ``comm_unique_id()``.
        Collective over all ranks."""
        comm = C.c_void_p()
        self.check(self.lib.ss_comm_create(self.handle, bytes(comm_id), int(rank), int(world), C.byref(comm)))
        return comm.value

    def comm_destroy(self, comm):
        self.check(self.lib.ss_comm_destroy(self.handle, C.c_void_p(comm)))

    def allgather_stats(self, comm, rank, world, n_images, local_dev, all_dev):
        """Brightness of all ``n_images`` in image order into ``all_dev`` on every rank (asynchronous)."""
        self.check(self.lib.ss_allgather_stats(self.handle, C.c_void_p(comm), int(rank), int(world), int(n_images),
                                               C.c_void_p(local_dev), C.c_void_p(all_dev)))

    def chain_run_static(self, desc, src_dev, dst_dev):
        """``chain_run`` without any allocation inside the call (work-queue counters in the library's static device
        scratch): for callers under graph capture.  Calls on one GPU must not overlap."""
        self.check(self.lib.ss_chain_run_static(self.handle, C.byref(desc), C.c_void_p(src_dev), C.c_void_p(dst_dev)))

    def chain_requeued(self):
        """Pixels the most recent chain run on this context evaluated again in float64.  Synchronises."""
        n = C.c_uint64()
        self.check(self.lib.ss_chain_requeued(self.handle, C.byref(n)))
        return n.value

    def negative_params(self, src_dev, height, width, batch, low, high, legacy_numpy, params_dev, detail_dev=None):
        self.check(self.lib.ss_negative_params(self.handle, C.c_void_p(src_dev), height, width, batch,
                                               float(low), float(high), int(bool(legacy_numpy)),
                                               C.c_void_p(params_dev), C.c_void_p(detail_dev or None)))

    def exposure_stats(self, thumbs_dev, height, width, batch, shadow, highlight, stats_dev, brightness_dev):
        self.check(self.lib.ss_exposure_stats(self.handle, C.c_void_p(thumbs_dev), height, width, batch,
                                              float(shadow), float(highlight),
                                              C.c_void_p(stats_dev or None), C.c_void_p(brightness_dev)))

    def exposure_multipliers(self, brightness_dev, n, ref_index, first, count, mult_dev):
        self.check(self.lib.ss_exposure_multipliers(self.handle, C.c_void_p(brightness_dev), n,
                                                    -1 if ref_index is None else int(ref_index),
                                                    first, count, C.c_void_p(mult_dev)))

    def unpack_u16(self, src_dev, dst_dev, n_values):
        """float32(src) / 65535 on the device: the RAW decoder's 16-bit output to the chain's input (asynchronous)."""
        self.check(self.lib.ss_unpack_u16(self.handle, C.c_void_p(src_dev), C.c_void_p(dst_dev), int(n_values)))

    def is_lab_image(self, src_dev, n_pixels, out_dev, minmax_dev=None):
        self.check(self.lib.ss_is_lab_image(self.handle, C.c_void_p(src_dev), int(n_pixels),
                                            C.c_void_p(out_dev), C.c_void_p(minmax_dev or None)))

    def rgb_to_lab(self, rgb_dev, n_pixels, L_dev, a_dev, b_dev):
        """Interleaved sRGB-encoded float32 pixels -> planes L, a, b (asynchronous)."""
        self.check(self.lib.ss_rgb_to_lab(self.handle, C.c_void_p(rgb_dev), int(n_pixels), C.c_void_p(L_dev),
                                          C.c_void_p(a_dev), C.c_void_p(b_dev)))

    def lab_to_rgb(self, L_dev, a_dev, b_dev, n_pixels, rgb_dev):
        """Planes L, a, b -> interleaved sRGB-encoded float32 pixels clipped to [0, 1] (asynchronous)."""
        self.check(self.lib.ss_lab_to_rgb(self.handle, C.c_void_p(L_dev), C.c_void_p(a_dev), C.c_void_p(b_dev),
                                          int(n_pixels), C.c_void_p(rgb_dev)))

    def mask_regions(self, img_dev, height, width, batch, shadow_limit, highlight_limit, min_area=200, count_dev=None,
                     float32_luminance=False):
        """Zero, in place, the 8-connected regions of at least ``min_area`` pixels whose luminance is <= shadow_limit
        or >= highlight_limit (asynchronous).  ``count_dev``: optional device uint64[batch] of zeroed pixels.
        ``float32_luminance``: luminance and comparison in float32, as numpy < 2 forms them for a float32 image."""
        self.check(self.lib.ss_mask_regions_f32(self.handle, C.c_void_p(img_dev), int(height), int(width), int(batch),
                                                float(shadow_limit), float(highlight_limit), int(min_area),
                                                int(bool(float32_luminance)), C.c_void_p(count_dev or None)))

    def resize_area(self, src_dev, height, width, batch, dst_dev, dst_height, dst_width):
        """Area-average downscale of [batch][height][width][3] float32 on the device (asynchronous)."""
        self.check(self.lib.ss_resize_area(self.handle, C.c_void_p(src_dev), int(height), int(width), int(batch),
                                           C.c_void_p(dst_dev), int(dst_height), int(dst_width)))

    def resize_area_host(self, src, dst_width, dst_height):
        """``cv2.resize(src, (dst_width, dst_height), interpolation=cv2.INTER_AREA)`` for an (H, W, 3) array."""
        src = np.ascontiguousarray(src, dtype=np.float32)
        if src.ndim != 3 or src.shape[2] != 3:
            raise ValueError(f"expected an (H, W, 3) array, got {src.shape}")
        out = np.empty((int(dst_height), int(dst_width), 3), dtype=np.float32)
        self.check(self.lib.ss_resize_area_host(self.handle, src.ctypes.data_as(C.c_void_p), src.shape[0], src.shape[1],
                                                out.ctypes.data_as(C.c_void_p), int(dst_height), int(dst_width)))
        return out

    # -- entry point on host arrays
    def chain_run_host(self, desc, src, neg_auto=False, low=0.05, high=0.995, legacy_numpy=True, lab_auto=False,
                       out=None):
        """Run the chain on a C-contiguous float32 array of shape (H, W, 3) or (B, H, W, 3).
        Returns ``(out, lab_detected)``; ``out`` may be supplied (e.g. from ``pinned_empty``)."""
        src = np.ascontiguousarray(src, dtype=np.float32)
        if src.ndim not in (3, 4) or src.shape[-1] != 3 or src.size != desc.batch * desc.height * desc.width * 3:
            raise ValueError(f"array of shape {src.shape} does not hold the descriptor's {desc.batch} image(s) of "
                             f"{desc.height} x {desc.width} x 3")
        if out is None:
            out = np.empty(src.shape, dtype=OUT_DTYPES[desc.out_kind])
        elif (out.shape != src.shape or out.dtype != OUT_DTYPES[desc.out_kind] or not out.flags.c_contiguous
              or not out.flags.writeable):
            raise ValueError("out must be a writeable C-contiguous array of the input's shape and the output dtype")
        opts = HostOpts(int(bool(neg_auto)), int(bool(legacy_numpy)), float(low), float(high), int(bool(lab_auto)), 0)
        self.check(self.lib.ss_chain_run_host(self.handle, C.byref(desc), src.ctypes.data_as(C.c_void_p),
                                              out.ctypes.data_as(C.c_void_p), C.byref(opts)))
        return out, bool(opts.lab_detected)


_tls = threading.local()


def default_device():
    """The GPU a process uses when it names none: SCANSPACE_DEVICE, else LOCAL_RANK (one process per GPU), else 0."""
    return int(os.environ.get("SCANSPACE_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def default_context(device=None):
    """Per-thread context (the reference runs several worker threads at once)."""
    if device is None:
        device = default_device()
    ctxs = getattr(_tls, "ctxs", None)
    if ctxs is None:
        ctxs = _tls.ctxs = {}
    if device not in ctxs:
        ctxs[device] = Context(device)
    return ctxs[device]


def percentile_rank(n, fraction, legacy_numpy=True):
    """Host-only: order-statistic indices and weight of np.percentile(method='linear')."""
    lib = load_library()
    lo, hi, g = C.c_int64(), C.c_int64(), C.c_double()
    rc = lib.ss_percentile_rank(int(n), float(fraction), int(bool(legacy_numpy)), C.byref(lo), C.byref(hi), C.byref(g))
    if rc != SS_OK:
        raise ScanSpaceError(rc, "invalid percentile arguments")
    return lo.value, hi.value, g.value


def shard_range(n_items, rank, world):
    """Host-only: ``(first, count)`` of ``ss_shard_range``."""
    first, count = C.c_int64(), C.c_int64()
    rc = load_library().ss_shard_range(int(n_items), int(rank), int(world), C.byref(first), C.byref(count))
    if rc != SS_OK:
        raise ValueError(f"bad shard request: n={n_items} rank={rank} world={world}")
    return first.value, count.value


def comm_unique_id():
    """128 bytes that identify a new communicator: make them on one rank, give them to every rank."""
    lib = load_library()
    buf = C.create_string_buffer(128)
    rc = lib.ss_comm_unique_id(buf)
    if rc != SS_OK:
        raise ScanSpaceError(rc, lib.ss_last_error(None).decode())
    return buf.raw


def negative_sample_step(height, width):
    return int(load_library().ss_negative_sample_step(int(height), int(width)))
