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
