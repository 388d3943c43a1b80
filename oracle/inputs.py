"""Deterministic synthetic inputs shared by the golden-vector generator, the
parity tests and the benchmark's CPU baseline.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  numpy's ``default_rng``
(PCG64) streams are stable across numpy versions, so a (generator, seed, h, w)
tuple identifies an input exactly and fixtures only need to store outputs.
"""

from __future__ import annotations

import numpy as np


def linear_image(seed, h, w):
    """Linear scene-like RGB in [0, 0.9], skewed dark."""
    rng = np.random.default_rng(seed)
    return (rng.random((h, w, 3)) ** 2.2 * 0.9).astype(np.float32)


def negative_image(seed, h, w):
    """Camera-scanned colour negative look: inverted, orange-mask channel scales."""
    pos = linear_image(seed, h, w)
    scale = np.array([1.0, 0.75, 0.55], dtype=np.float32)
    return (np.float32(0.08) + np.float32(0.8) * (np.float32(1.0) - pos) * scale).astype(np.float32)


def wide_image(seed, h, w):
    """Values in [-0.3, 1.3] to exercise every clip."""
    rng = np.random.default_rng(seed)
    return (rng.random((h, w, 3)) * 1.6 - 0.3).astype(np.float32)


def lab_image(seed, h, w):
    """CIE Lab planes (L 0..100, a/b -80..80): triggers the reference's LAB-input branch."""
    rng = np.random.default_rng(seed)
    L = rng.random((h, w)) * 100.0
    a = rng.random((h, w)) * 160.0 - 80.0
    b = rng.random((h, w)) * 160.0 - 80.0
    return np.stack([L, a, b], axis=-1).astype(np.float32)


def smooth_image(seed, h, w):
    """Low-frequency content plus noise: a sharper test for the blur than white noise."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = 0.45 + 0.4 * np.sin(xx / 17.0 + seed) * np.cos(yy / 23.0)
    img = np.stack([base, np.roll(base, 5, axis=1), np.roll(base, 9, axis=0)], axis=-1)
    img = img + rng.normal(0.0, 0.03, size=img.shape)
    return np.clip(img, 0.0, 1.0).astype(np.float32)


GENERATORS = {
    "linear": linear_image,
    "negative": negative_image,
    "wide": wide_image,
    "lab": lab_image,
    "smooth": smooth_image,
}


def synthetic_swatches(seed_ref=7, seed_cam=8):
    """A well-conditioned (M_T, M_R) pair of 24x3 swatch tables:
    M_R fixed in [0.03, 0.9]; M_T = M_R @ A^T + N(0, 0.002), A = I + 0.15*U(-1,1)."""
    r = np.random.default_rng(seed_ref)
    M_R = 0.03 + 0.87 * r.random((24, 3))
    c = np.random.default_rng(seed_cam)
    A = np.eye(3) + 0.15 * (c.random((3, 3)) * 2.0 - 1.0)
    M_T = M_R @ A.T + c.normal(0.0, 0.002, size=(24, 3))
    return M_T.astype(np.float32), M_R.astype(np.float64)


def thumbnail_of(img, th=341, tw=512):
    """Box-average thumbnail (float32) used for the exposure statistics inputs."""
    h, w, _ = img.shape
    ys = (np.arange(th + 1) * h) // th
    xs = (np.arange(tw + 1) * w) // tw
    out = np.empty((th, tw, 3), dtype=np.float32)
    for i in range(th):
        rows = img[ys[i]:max(ys[i + 1], ys[i] + 1)]
        for j in range(tw):
            out[i, j] = rows[:, xs[j]:max(xs[j + 1], xs[j] + 1)].mean(axis=(0, 1))
    return out
