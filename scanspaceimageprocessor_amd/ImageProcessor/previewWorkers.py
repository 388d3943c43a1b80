"""Preview producers of the reference's main window that sit on the correction chain.

``SwatchPreviewWorker`` (image_space.py:98-129) and the chart-detection preview (image_space.py:4098-4100) both
form ``np.uint8(255 * cctf_encoding(clip(colour_correction(img, swatches, reference), 0, 1)))``; here that is one
fused GPU pass (3-term Cheung fit on the host; matrix, clip, sRGB encode, truncation to 8 bits on the device).
The detection overlays next to it (image_space.py:4107-4113) are plain ``np.uint8(255 * cctf_encoding(x))``:
``encode_preview_u8``.
"""

from __future__ import annotations

import numpy as np

from .._lib import default_context
from ..chain import make_desc
from ..colour_fit import matrix_colour_correction
from .imageProcessorWorker import _Signal


def corrected_preview_u8(img_fp, swatch_colours, reference_swatches) -> np.ndarray:
    """Colour-corrected, clipped, sRGB-encoded 8-bit image of a linear float image."""
    img = np.ascontiguousarray(img_fp, dtype=np.float32)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError(f"expected an (H, W, 3) image, got shape {img.shape}")
    ccm = matrix_colour_correction(np.asarray(swatch_colours), np.asarray(reference_swatches))
    desc = make_desc(img.shape[0], img.shape[1], 1, ccm=ccm, final_clip=True, out="u8")
    out, _ = default_context().chain_run_host(desc, img)
    return out


def encode_preview_u8(img_fp) -> np.ndarray:
    """``np.uint8(255 * cctf_encoding(img))`` for an image already in [0, 1]."""
    img = np.ascontiguousarray(img_fp, dtype=np.float32)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError(f"expected an (H, W, 3) image, got shape {img.shape}")
    desc = make_desc(img.shape[0], img.shape[1], 1, out="u8")
    out, _ = default_context().chain_run_host(desc, img)
    return out


class SwatchPreviewSignals:
    """``finished(uint8 array)`` and ``error(str)`` (image_space.py:93-96)."""

    def __init__(self):
        self.finished = _Signal()
        self.error = _Signal()


class SwatchPreviewWorker:
    """Preview of a manual swatch correction (image_space.py:98-129): same constructor, ``run`` emits the corrected
    uint8 array through ``signals.finished`` or the error text through ``signals.error``."""

    def __init__(self, img_fp, swatch_colours, reference_swatches):
        self.img_fp = img_fp
        self.swatch_colours = swatch_colours
        self.reference_swatches = reference_swatches
        self.signals = SwatchPreviewSignals()

    def run(self):
        try:
            self.signals.finished.emit(corrected_preview_u8(self.img_fp, self.swatch_colours, self.reference_swatches))
        except Exception as e:
            self.signals.error.emit(str(e))


FAST_DISPLAY_MAX_PIXELS = 500_000  # image_space.py:3052: previews this small are shown without the sRGB encode


def display_u8(img_array) -> np.ndarray:
    """The array half of ``MainWindow._array_to_pixmap`` (image_space.py:3039-3058): ``uint8(255 * clip(x, 0, 1))``
    for images of at most 0.5 MP, ``uint8(255 * cctf_encoding(clip(x, 0, 1)))`` above."""
    img = np.ascontiguousarray(img_array, dtype=np.float32)
    h, w = img.shape[:2]
    out_kind = "u8_linear" if h * w <= FAST_DISPLAY_MAX_PIXELS else "u8"
    out, _ = default_context().chain_run_host(make_desc(h, w, 1, final_clip=True, out=out_kind), img)
    return out


def adjusted_preview_u8(preview_array, **adjustments) -> np.ndarray:
    """``_update_preview_display`` (image_space.py:2988-3037) up to the pixmap: ``apply_all_adjustments`` with the
    given keyword arguments followed by ``_array_to_pixmap``'s conversion, as ONE fused pass from the float preview
    to display bytes (the float result never leaves the GPU)."""
    from .editingTools import adjustment_stages, apply_all_adjustments

    img = np.ascontiguousarray(preview_array, dtype=np.float32)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError(f"Invalid preview array shape: {img.shape}")
    kw = adjustment_stages(**adjustments)
    if not kw["gains"] and ("sh" in kw or "usm" in kw):
        # the Lab stages come first: they look at the data to tell LAB input from RGB, and LAB output is not a
        # display format, so this rare combination takes the two passes the reference takes
        return display_u8(apply_all_adjustments(img, **adjustments))
    h, w = img.shape[:2]
    out_kind = "u8_linear" if h * w <= FAST_DISPLAY_MAX_PIXELS else "u8"
    out, _ = default_context().chain_run_host(make_desc(h, w, 1, final_clip=True, out=out_kind, **kw), img)
    return out
