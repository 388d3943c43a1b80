"""CPU restatement of ``ImageCorrectionWorker._apply_masking`` (imageProcessorWorker.py:979-1030).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The reference labels the candidate mask with ``cv2.connectedComponentsWithStats(mask, connectivity=8)`` and
keeps the components of at least 200 pixels with a Python loop over the labels.  OpenCV is not installed
here; the labelling below uses ``scipy.ndimage.label`` with the full 3 x 3 structure (8-connectivity).
The outcome - which pixels are zeroed - is a property of the mask (components and their areas), not of the
labelling routine or its numbering, so any correct 8-connected labelling pins it; the hand-made cases in
tests/test_oracle_masking.py check that this one is.
"""

from __future__ import annotations

import numpy as np


def candidate_mask(corrected, shadow_limit=0.05, highlight_limit=0.98, float32_luminance=False):
    """imageProcessorWorker.py:989-996.  By default the float64 luminance of the float64 image the reference holds
    there in chart mode or behind a Lab edit.  ``float32_luminance``: the image is still float32 (no chart, exposure
    and white balance only) and numpy < 2, the reference's pin, does not let Python scalars widen an array: the
    products, their sum and the limits it is compared with are float32 (spelled out here, since this numpy may
    promote differently)."""
    if float32_luminance:
        c = np.asarray(corrected, dtype=np.float32)
        f = np.float32
        lum = (f(0.2126) * c[:, :, 0] + f(0.7152) * c[:, :, 1]) + f(0.0722) * c[:, :, 2]
        assert lum.dtype == np.float32
        return (lum <= f(shadow_limit)) | (lum >= f(highlight_limit))
    c = np.asarray(corrected, dtype=np.float64)
    lum = 0.2126 * c[:, :, 0] + 0.7152 * c[:, :, 1] + 0.0722 * c[:, :, 2]
    return (lum <= shadow_limit) | (lum >= highlight_limit)


def clean_mask(mask, min_area=200):
    """imageProcessorWorker.py:1000-1013: candidates that belong to an 8-connected component of at least
    ``min_area`` pixels."""
    from scipy import ndimage

    labels, n = ndimage.label(mask, structure=np.ones((3, 3), dtype=bool))
    if n == 0:
        return np.zeros_like(mask, dtype=bool)
    areas = np.bincount(labels.ravel(), minlength=n + 1)
    keep = areas >= min_area
    keep[0] = False  # background
    return keep[labels]


def apply_masking(corrected, shadow_limit=0.05, highlight_limit=0.98, min_area=200, float32_luminance=False):
    """Returns ``(masked image (copy), number of masked pixels)`` (imageProcessorWorker.py:1019-1020)."""
    m = clean_mask(candidate_mask(corrected, shadow_limit, highlight_limit, float32_luminance), min_area)
    out = np.array(corrected, copy=True)
    out[m] = 0
    return out, int(m.sum())
