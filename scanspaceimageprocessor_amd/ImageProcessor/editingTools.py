"""Pixel operators of the correction chain, same names and signatures as the
reference's ``ImageProcessor/editingTools.py``, executed by the gfx950 kernels of
libscanspace_hip.so.

Contract kept from the reference: numpy ``(H, W, 3)`` arrays in; a NEW array
out (the input is never modified); every function returns its input object
unchanged at its neutral setting.  Results are float32 (the reference returns
float64 as soon as colour-science touches the data).

Stages that run back to back (``apply_all_adjustments``) are fused into one
kernel pass: one upload, one launch, one download.
"""

from __future__ import annotations

import numpy as np

from .. import _lib
from ..chain import make_desc
from .._lib import ScanSpaceError, default_context

# np.percentile index arithmetic: True = numpy < 2 (the reference's pin, float64), False = numpy >= 2 (float32)
NUMPY_LEGACY_PERCENTILE = True


def _is_rgb_image(image) -> bool:
    return image is not None and getattr(image, "ndim", 0) == 3 and image.shape[2] == 3


def _run(image, *, lab_auto=False, **kw):
    h, w, _ = image.shape
    desc = make_desc(h, w, 1, **kw)
    out, lab = default_context().chain_run_host(desc, image, lab_auto=lab_auto)
    return out, lab


# ---------------------------------------------------------------------------------------------
# negative film (reference editingTools.py:296-365)
# ---------------------------------------------------------------------------------------------


def invert_negative_auto(image: np.ndarray, low_percentile: float = 0.05, high_percentile: float = 0.995) -> np.ndarray:
    """Per-channel percentile normalisation and inversion (reference editingTools.py:296-338).
    ``low_percentile=0.05`` is the 5th percentile.  Percentiles are exact order statistics of the
    strided sample, found on the GPU."""
    if not _is_rgb_image(image):
        return image
    h, w, _ = image.shape
    desc = make_desc(h, w, 1, neg=True, out="f32")
    out, _ = default_context().chain_run_host(desc, image, neg_auto=True, low=low_percentile, high=high_percentile,
                                              legacy_numpy=NUMPY_LEGACY_PERCENTILE)
    return out


def ensure_positive_if_negative(image: np.ndarray, enabled: bool = False, low_percentile: float = 0.05,
                                high_percentile: float = 0.995) -> np.ndarray:
    """Reference editingTools.py:340-365: pass-through unless enabled; data errors fall back to the
    original image.  A missing library or a failing GPU is NOT swallowed."""
    if not enabled:
        return image
    try:
        return invert_negative_auto(image, low_percentile, high_percentile)
    except ScanSpaceError as e:
        if e.code in (_lib.SS_ERR_HIP, _lib.SS_ERR_NOMEM):
            raise
        return image
    except (ValueError, TypeError):
        return image


# ---------------------------------------------------------------------------------------------
# Lab helpers (reference editingTools.py:367-432)
# ---------------------------------------------------------------------------------------------


def _rgb_to_lab(image: np.ndarray):
    """``(L, a, b)`` planes of an sRGB-encoded image (reference editingTools.py:367-386): sRGB decode, XYZ, CIE Lab
    under D65; L in 0..100.  float32 arrays of the image's leading shape."""
    arr = np.ascontiguousarray(image, dtype=np.float32)
    if arr.ndim < 1 or arr.shape[-1] != 3:
        raise ValueError(f"expected (..., 3) RGB data, got shape {arr.shape}")
    n = arr.size // 3
    planes = np.empty((3, n), dtype=np.float32)
    if n == 0:
        return tuple(planes[k].reshape(arr.shape[:-1]) for k in range(3))
    ctx = default_context()
    src = ctx.malloc(arr.nbytes)
    dst = ctx.malloc(planes.nbytes)
    try:
        ctx.upload(src, arr)
        ctx.rgb_to_lab(src, n, dst, dst + 4 * n, dst + 8 * n)
        ctx.download(planes, dst)
    finally:
        ctx.free(src)
        ctx.free(dst)
    return tuple(planes[k].reshape(arr.shape[:-1]) for k in range(3))


def _lab_to_rgb(L: np.ndarray, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """sRGB-encoded image, clipped to [0, 1], of three Lab planes (reference editingTools.py:389-408)."""
    L = np.asarray(L)
    planes = np.ascontiguousarray(np.stack([np.broadcast_to(np.asarray(p, dtype=np.float32), L.shape)
                                            for p in (L, a, b)]).reshape(3, -1))
    n = planes.shape[1]
    out = np.empty(L.shape + (3,), dtype=np.float32)
    if n == 0:
        return out
    ctx = default_context()
    src = ctx.malloc(planes.nbytes)
    dst = ctx.malloc(out.nbytes)
    try:
        ctx.upload(src, planes)
        ctx.lab_to_rgb(src, src + 4 * n, src + 8 * n, n, dst)
        ctx.download(out, dst)
    finally:
        ctx.free(src)
        ctx.free(dst)
    return out


def _is_lab_image(image: np.ndarray) -> bool:
    """max(ch0) > 1.5 and (min(ch1) < -1 or min(ch2) < -1), reduced on the GPU."""
    if image.shape[-1] != 3:
        return False
    ctx = default_context()
    arr = np.ascontiguousarray(image, dtype=np.float32)
    src = ctx.malloc(arr.nbytes)
    flag = ctx.malloc(4)
    try:
        ctx.upload(src, arr)
        ctx.is_lab_image(src, arr.size // 3, flag)
        out = np.zeros(1, dtype=np.int32)
        ctx.download(out, flag)
        return bool(out[0])
    finally:
        ctx.free(src)
        ctx.free(flag)


# ---------------------------------------------------------------------------------------------
# exposure / shadows / highlights / sharpen
# ---------------------------------------------------------------------------------------------


def adjust_exposure(image: np.ndarray, exposure_value: float) -> np.ndarray:
    """``clip(image * 2**ev, 0, 1)`` (reference editingTools.py:435-466)."""
    if exposure_value == 0:
        return image
    out, _ = _run(image, gains=[float(np.power(2.0, exposure_value))])
    return out


def adjust_shadows_highlights_combined(image: np.ndarray, shadow_amount: float, highlight_amount: float,
                                       falloff_power: float = 7.0) -> np.ndarray:
    """Exponential shadow/highlight curve on CIE L (reference editingTools.py:469-544)."""
    if shadow_amount == 0 and highlight_amount == 0:
        return image
    out, _ = _run(image, lab_auto=True, sh=(shadow_amount, highlight_amount, falloff_power))
    return out


def adjust_sharpen(image: np.ndarray, amount: float = 0.0, radius: float = 1.0, threshold: float = 0.0) -> np.ndarray:
    """Unsharp mask on CIE L with threshold and adaptive masks (reference editingTools.py:714-820)."""
    if amount <= 0:
        return image
    out, _ = _run(image, lab_auto=True, usm=(amount, radius, threshold))
    return out


# ---------------------------------------------------------------------------------------------
# white balance (reference editingTools.py:547-640)
# ---------------------------------------------------------------------------------------------


def _cct_to_xy_cie_d(cct: float):
    """``colour.temperature.CCT_to_xy`` with its default method, "CIE Illuminant D Series" (the daylight locus;
    colour evaluates it outside 4000 K .. 25000 K too, with a usage warning only).  Its docstring example:
    6504.38938305 K -> (0.3127077, 0.3291128)."""
    T = float(cct)
    if T <= 7000:
        x = -4.607 * 10 ** 9 / T ** 3 + 2.9678 * 10 ** 6 / T ** 2 + 0.09911 * 10 ** 3 / T + 0.244063
    else:
        x = -2.0064 * 10 ** 9 / T ** 3 + 1.9018 * 10 ** 6 / T ** 2 + 0.24748 * 10 ** 3 / T + 0.23704
    y = -3.000 * x ** 2 + 2.870 * x - 0.275
    return x, y


# ``colour.XYZ_to_RGB(XYZ, RGB_COLOURSPACES['sRGB'], apply_cctf_encoding=False)``: no chromatic adaptation without an
# illuminant argument, and the colourspace's own XYZ -> RGB matrix, which is the IEC 61966-2-1 four-decimal table
_M_XYZ_TO_SRGB = np.array([[3.2406, -1.5372, -0.4986], [-0.9689, 1.8758, 0.0415], [0.0557, -0.2040, 1.0570]])


def calculate_white_balance_multipliers(current_temp: float, target_temp: float, tint: float = 0.0):
    """RGB multipliers from two colour temperatures (reference editingTools.py:547-602).  Host arithmetic."""

    def rgb_of(cct):
        x, y = _cct_to_xy_cie_d(cct)
        XYZ = np.array([x / y, 1.0, (1.0 - x - y) / y])
        return _M_XYZ_TO_SRGB @ XYZ

    multipliers = rgb_of(target_temp) / (rgb_of(current_temp) + 1e-10)
    if tint != 0:
        multipliers[1] *= 1 + (tint / 100.0) * 0.1
    min_mult = np.min(multipliers)
    if min_mult > 0:
        multipliers = multipliers / min_mult
    return tuple(multipliers)


def adjust_white_balance(image: np.ndarray, current_temp: float, target_temp: float, tint: float = 0.0) -> np.ndarray:
    """Per-channel gains + clip (reference editingTools.py:605-640)."""
    if current_temp == target_temp and tint == 0:
        return image
    out, _ = _run(image, gains=[calculate_white_balance_multipliers(current_temp, target_temp, tint)])
    return out


def sample_white_balance_from_point(image: np.ndarray, sample_x: int, sample_y: int, current_temp: float,
                                    sample_radius: int = 5) -> float:
    """Colour temperature that neutralises the patch around a picked point (reference editingTools.py:643-711).

    Mean colour of the ``(2 r + 1)^2`` window clipped to the image; with ``q = (R/G) / (B/G)`` the new temperature
    is ``T (1 + 0.3 ln q)`` for a reddish patch (q > 1) and ``T / (1 + 0.3 ln(1/q))`` otherwise, limited to
    1000..12000 K.  A patch whose mean green is below 0.001 leaves ``current_temp`` as it is.  Host arithmetic:
    the window holds 121 pixels."""
    height, width = image.shape[:2]
    rows = slice(max(0, sample_y - sample_radius), min(height, sample_y + sample_radius + 1))
    cols = slice(max(0, sample_x - sample_radius), min(width, sample_x + sample_radius + 1))
    red, green, blue = np.mean(image[rows, cols], axis=(0, 1))[:3]
    if green < 0.001:
        return current_temp
    q = (red / green) / (blue / green + 1e-10)
    factor = 1.0 + 0.3 * np.log(q) if q > 1 else 1.0 / (1.0 + 0.3 * np.log(1.0 / q))
    return float(np.clip(current_temp * factor, 1000, 12000))


# ---------------------------------------------------------------------------------------------
# cache hooks of the reference's preview path (editingTools.py:206-254)
# ---------------------------------------------------------------------------------------------
# The reference keeps denoised previews in a memory-managed cache and lets the GUI steer it.  Denoise is not
# part of this package, nothing is cached, and a fused pass over a 4 MP preview takes well under a millisecond
# of GPU time; the hooks stay so that callers of the reference's module find them.

_CURRENT_IMAGE_ID = None


def clear_adjustment_cache() -> None:
    global _CURRENT_IMAGE_ID
    _CURRENT_IMAGE_ID = None


def set_current_image_id(image_id: str) -> None:
    global _CURRENT_IMAGE_ID
    _CURRENT_IMAGE_ID = image_id


def get_cache_stats() -> dict:
    """The reference's statistics record (editingTools.py:173-184) of an always-empty cache."""
    return {'entries': 0, 'memory_mb': 0.0, 'memory_percent': 0.0, 'hits': 0, 'misses': 0, 'evictions': 0,
            'hit_rate': 0}


def check_memory_usage() -> bool:
    """True when a cleanup was performed (reference editingTools.py:224-254): there is never anything to clean."""
    return False


# ---------------------------------------------------------------------------------------------
# the whole adjustment chain (reference editingTools.py:823-909)
# ---------------------------------------------------------------------------------------------


def adjustment_stages(exposure=0.0, shadows=0.0, highlights=0.0, current_wb=5500.0, target_wb=5500.0, wb_tint=0.0,
                      denoise_strength=0.0, sharpen_amount=0.0, sharpen_radius=1.0, sharpen_threshold=0.0):
    """Translate apply_all_adjustments' arguments into chain stages (keyword arguments of make_desc).
    Keeps the reference's pre-scaling ``shadows / 70`` and ``highlights * 1.3`` and its order
    white balance -> exposure -> shadows/highlights -> sharpen."""
    if denoise_strength > 0:
        raise NotImplementedError("denoise (8-bit cv2 median/bilateral filters) is outside the accelerated chain")
    shadows = shadows / 70
    highlights = highlights * 1.3
    kw = dict(gains=[])
    if current_wb != target_wb or wb_tint != 0:
        kw["gains"].append(calculate_white_balance_multipliers(current_wb, target_wb, wb_tint))
    if exposure != 0:
        kw["gains"].append(float(np.power(2.0, exposure)))
    if shadows != 0 or highlights != 0:
        kw["sh"] = (shadows, highlights, 7.0)
    if sharpen_amount > 0:
        kw["usm"] = (sharpen_amount, sharpen_radius, sharpen_threshold)
    return kw


def apply_all_adjustments(image: np.ndarray, exposure: float = 0.0, shadows: float = 0.0, highlights: float = 0.0,
                          current_wb: float = 5500.0, target_wb: float = 5500.0, wb_tint: float = 0.0,
                          denoise_strength: float = 0.0, sharpen_amount: float = 0.0, sharpen_radius: float = 1.0,
                          sharpen_threshold: float = 0.0) -> np.ndarray:
    """All adjustments in the reference's order, as ONE fused kernel pass."""
    kw = adjustment_stages(exposure, shadows, highlights, current_wb, target_wb, wb_tint, denoise_strength,
                           sharpen_amount, sharpen_radius, sharpen_threshold)
    if not kw["gains"] and "sh" not in kw and "usm" not in kw:
        return image.copy()
    # the LAB-in heuristic is evaluated on what reaches the first Lab stage; after a gain+clip stage data is in [0,1]
    lab_auto = not kw["gains"]
    out, _ = _run(image, lab_auto=lab_auto, **kw)
    return out


class GlobalAdjustments:
    """Parameter bag applied to the whole dataset (reference editingTools.py:913-976)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.exposure = 0.0
        self.shadows = 0.0
        self.highlights = 0.0
        self.white_balance = None
        self.wb_tint = 0.0
        self.denoise_strength = 0.0
        self.enabled = False

    def has_adjustments(self) -> bool:
        return self.enabled and (self.exposure != 0 or self.shadows != 0 or self.highlights != 0
                                 or self.white_balance is not None or self.denoise_strength > 0)

    def apply_to_image(self, image: np.ndarray, current_wb: float = 5500) -> np.ndarray:
        if not self.enabled or not self.has_adjustments():
            return image
        target_wb = self.white_balance if self.white_balance else current_wb
        return apply_all_adjustments(image, exposure=self.exposure, shadows=self.shadows, highlights=self.highlights,
                                     current_wb=current_wb, target_wb=target_wb, wb_tint=self.wb_tint,
                                     denoise_strength=self.denoise_strength)


global_adjustments = GlobalAdjustments()
