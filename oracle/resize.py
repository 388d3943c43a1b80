"""CPU restatement of ``cv2.resize(..., interpolation=cv2.INTER_AREA)`` for downscaling, the call behind the
reference's preview and thumbnail producers (imageLoader.py:94-121, imageLoader.py:298-302,
image_space.py:2959-2986).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

PARITY UNPINNED: OpenCV (``opencv-python>=4.8.0``, requirements.txt:6) is not installed in this build's
containers and the reference holds no fixture for these calls.  The algorithm below restates OpenCV 4.x's
published area resampler (``computeResizeAreaTab`` and ``ResizeArea_Invoker`` in imgproc/resize.cpp): every
destination pixel is the average of the source rectangle it covers, edge source pixels weighted by the
fraction covered.  It is evaluated in float64; OpenCV accumulates float32 images in float32, so agreement
with the real library is expected to ~1e-6, not to the bit.
"""

from __future__ import annotations

import math

import numpy as np


def area_table(ssize, dsize):
    """Per destination index: list of (source index, weight) along one axis.

    scale = 1 / (dsize / ssize) (>= 1).  Destination cell dx covers [dx*scale, (dx+1)*scale) clipped to the
    source; a partially covered source pixel enters only if more than 1e-3 of it is covered (OpenCV's
    threshold); weights are divided by the cell width min(scale, ssize - dx*scale)."""
    if dsize < 1 or ssize < dsize:
        raise ValueError("area resampling is defined here for 1 <= dsize <= ssize")
    scale = 1.0 / (dsize / ssize)  # cv::resize forms the inverse scale first
    table = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1 = math.ceil(fsx1)
        sx2 = math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        row = []
        if sx1 - fsx1 > 1e-3:
            row.append((sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            row.append((sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            row.append((sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        table.append(row)
    return table


def area_matrix(ssize, dsize):
    """The table as a scipy CSR matrix (dsize x ssize), float64 values of the float32 weights."""
    from scipy.sparse import csr_matrix

    rows, cols, vals = [], [], []
    for dx, row in enumerate(area_table(ssize, dsize)):
        for sx, a in row:
            rows.append(dx)
            cols.append(sx)
            vals.append(float(a))
    return csr_matrix((vals, (rows, cols)), shape=(dsize, ssize))


def resize_area(img, dsize_wh):
    """``cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_AREA)`` for (H, W, C) or (H, W) float input
    with new_w <= W and new_h <= H.  float64 result."""
    img = np.asarray(img, dtype=np.float64)
    new_w, new_h = int(dsize_wh[0]), int(dsize_wh[1])
    h, w = img.shape[:2]
    wy = area_matrix(h, new_h)
    wx = area_matrix(w, new_w)
    flat = img.reshape(h, -1)
    tmp = (wy @ flat).reshape((new_h, w) + img.shape[2:])  # rows first or columns first: linear, same result
    tmp = np.moveaxis(tmp, 1, 0).reshape(w, -1)
    out = (wx @ tmp).reshape((new_w, new_h) + img.shape[2:])
    return np.moveaxis(out, 0, 1)


# ---- the reference's size rules -----------------------------------------------------------------------


def preview_size(height, width, target_pixels):
    """imageLoader.py:94-111 (1 MP) and image_space.py:2969-2980 (4 MP): None when the image is small enough,
    else (new_w, new_h) with scale = sqrt(target / current), truncated, at least 1."""
    current = height * width
    if current <= target_pixels:
        return None
    s = np.sqrt(target_pixels / current)
    return max(1, int(width * s)), max(1, int(height * s))


def thumbnail_size(height, width, max_size=(512, 512)):
    """imageLoader.py:299-302: None when no shrink is needed, else (new_w, new_h).  Note the reference pairs
    max_size[0] with the HEIGHT and max_size[1] with the width."""
    scale = min(max_size[0] / height, max_size[1] / width, 1.0)
    if scale >= 1.0:
        return None
    return int(width * scale), int(height * scale)
