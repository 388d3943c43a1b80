"""Float64 restatements of the third-party leaves of the correction chain.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The arithmetic of the reference's hot path lives in two packages that are not
vendored under /root/reference and are not installed in this image:

* ``colour-science>=0.4.6``  (requirements.txt:14) -- call sites
  imageProcessorWorker.py:116-125,337,592,595; editingTools.py:378-379,405-406
* ``opencv-python>=4.8.0``   (requirements.txt:6)  -- editingTools.py:761-764
* ``numpy>=1.21,<2.0``       (requirements.txt:7)  -- ``np.percentile``
  editingTools.py:329-330

Every function below restates the *published* algorithm of the named leaf in
plain numpy float64.  PARITY UNPINNED: the reference holds no test, golden
vector or fixture for any of them, so the constants here are specification,
checked by the known-answer tests in tests/test_oracle_leaves.py (analytic
identities, published tables, the worked examples of colour-science's docstrings).
"""

from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# colour.algebra.spow
# --------------------------------------------------------------------------


def spow(a, p):
    """colour's safe power: ``sign(a) * |a| ** p`` (nan -> 0)."""
    a = np.asarray(a, dtype=np.float64)
    out = np.sign(a) * np.abs(a) ** p
    return np.where(np.isnan(out), 0.0, out)


# --------------------------------------------------------------------------
# colour.models.rgb.transfer_functions.srgb
# --------------------------------------------------------------------------


def eotf_inverse_sRGB(L):
    """``colour.cctf_encoding(x)`` with its default ``function='sRGB'``
    (imageProcessorWorker.py:592,595): linear -> sRGB-encoded."""
    L = np.asarray(L, dtype=np.float64)
    return np.where(L <= 0.0031308, L * 12.92, 1.055 * spow(L, 1 / 2.4) - 0.055)


def eotf_sRGB(V):
    """sRGB EOTF (decode), the ``apply_cctf_decoding=True`` default of
    ``colour.sRGB_to_XYZ`` (editingTools.py:378)."""
    V = np.asarray(V, dtype=np.float64)
    thr = 12.92 * 0.0031308  # == eotf_inverse_sRGB(0.0031308)
    return np.where(V <= thr, V / 12.92, spow((V + 0.055) / 1.055, 2.4))


cctf_encoding = eotf_inverse_sRGB
cctf_decoding = eotf_sRGB

# --------------------------------------------------------------------------
# sRGB <-> XYZ <-> Lab   (D65 on both sides => CAT02 adaptation is identity)
# --------------------------------------------------------------------------

MATRIX_sRGB_TO_XYZ = np.array(
    [[0.4124, 0.3576, 0.1805], [0.2126, 0.7152, 0.0722], [0.0193, 0.1192, 0.9505]],
    dtype=np.float64,
)
# IEC 61966-2-1:1999 gives BOTH matrices to four decimals and colour's sRGB colourspace carries both as given
# (``use_derived_matrix_XYZ_to_RGB`` is off by default), so the pair is not an exact inverse pair:
# XYZ_TO_sRGB . sRGB_TO_XYZ differs from the identity by up to 4e-5.  Pinned by colour's own docstring
# example, which only this table reproduces (tests/test_oracle_leaves.py):
#   XYZ_to_sRGB([0.20654008, 0.12197225, 0.05136952]) -> [0.70573936, 0.19248266, 0.22354169]
# (the numerical inverse gives [0.70574132, 0.19244650, 0.22353975]).
MATRIX_XYZ_TO_sRGB = np.array(
    [[3.2406, -1.5372, -0.4986], [-0.9689, 1.8758, 0.0415], [0.0557, -0.2040, 1.0570]],
    dtype=np.float64,
)

D65_xy = (0.3127, 0.3290)


def xy_to_XYZ(xy):
    x, y = float(xy[0]), float(xy[1])
    return np.array([x / y, 1.0, (1.0 - x - y) / y], dtype=np.float64)


D65_XYZ = xy_to_XYZ(D65_xy)


def sRGB_to_XYZ(RGB, apply_cctf_decoding=True):
    """``colour.sRGB_to_XYZ`` with defaults (editingTools.py:378)."""
    RGB = np.asarray(RGB, dtype=np.float64)
    if apply_cctf_decoding:
        RGB = eotf_sRGB(RGB)
    return RGB @ MATRIX_sRGB_TO_XYZ.T


def XYZ_to_sRGB(XYZ, apply_cctf_encoding=True):
    """``colour.XYZ_to_sRGB`` with defaults (editingTools.py:406)."""
    XYZ = np.asarray(XYZ, dtype=np.float64)
    RGB = XYZ @ MATRIX_XYZ_TO_sRGB.T
    if apply_cctf_encoding:
        RGB = eotf_inverse_sRGB(RGB)
    return RGB


def _lab_f(t):
    return np.where(t > (24.0 / 116.0) ** 3, spow(t, 1.0 / 3.0), (841.0 / 108.0) * t + 16.0 / 116.0)


def _lab_f_inv(f):
    return np.where(f > 24.0 / 116.0, f ** 3, (f - 16.0 / 116.0) * (108.0 / 841.0))


def XYZ_to_Lab(XYZ):
    """``colour.XYZ_to_Lab`` with the default D65 illuminant (editingTools.py:379)."""
    XYZ = np.asarray(XYZ, dtype=np.float64)
    f = _lab_f(XYZ / D65_XYZ)
    fx, fy, fz = f[..., 0], f[..., 1], f[..., 2]
    return np.stack([116.0 * fy - 16.0, 500.0 * (fx - fy), 200.0 * (fy - fz)], axis=-1)


def Lab_to_XYZ(Lab):
    """``colour.Lab_to_XYZ`` with the default D65 illuminant (editingTools.py:405)."""
    Lab = np.asarray(Lab, dtype=np.float64)
    L, a, b = Lab[..., 0], Lab[..., 1], Lab[..., 2]
    fy = (L + 16.0) / 116.0
    fx = a / 500.0 + fy
    fz = fy - b / 200.0
    return np.stack([_lab_f_inv(fx), _lab_f_inv(fy), _lab_f_inv(fz)], axis=-1) * D65_XYZ


# --------------------------------------------------------------------------
# colour.characterisation.colour_correction
# --------------------------------------------------------------------------

CHEUNG_TERMS = (3, 5, 7, 8, 10, 11, 14, 16, 17, 19, 20, 22)


def matrix_augmented_Cheung2004(RGB, terms=3):
    """``colour.characterisation.matrix_augmented_Cheung2004``: polynomial
    expansion of (N,3) RGB into (N,terms)."""
    RGB = np.asarray(RGB, dtype=np.float64)
    R, G, B = RGB[..., 0], RGB[..., 1], RGB[..., 2]
    one = np.ones_like(R)
    if terms == 3:
        cols = [R, G, B]
    elif terms == 5:
        cols = [R, G, B, R * G * B, one]
    elif terms == 7:
        cols = [R, G, B, R * G, R * B, G * B, one]
    elif terms == 8:
        cols = [R, G, B, R * G, R * B, G * B, R * G * B, one]
    elif terms == 10:
        cols = [R, G, B, R * G, R * B, G * B, R ** 2, G ** 2, B ** 2, one]
    elif terms == 11:
        cols = [R, G, B, R * G, R * B, G * B, R ** 2, G ** 2, B ** 2, R * G * B, one]
    elif terms == 14:
        cols = [R, G, B, R * G, R * B, G * B, R ** 2, G ** 2, B ** 2, R * G * B,
                R ** 3, G ** 3, B ** 3, one]
    elif terms == 16:
        cols = [R, G, B, R * G, R * B, G * B, R ** 2, G ** 2, B ** 2, R * G * B,
                R ** 2 * G, G ** 2 * B, B ** 2 * R, R ** 3, G ** 3, B ** 3]
    elif terms == 17:
        cols = [R, G, B, R * G, R * B, G * B, R ** 2, G ** 2, B ** 2, R * G * B,
                R ** 2 * G, G ** 2 * B, B ** 2 * R, R ** 3, G ** 3, B ** 3, one]
    elif terms == 19:
        cols = [R, G, B, R * G, R * B, G * B, R ** 2, G ** 2, B ** 2, R * G * B,
                R ** 2 * G, G ** 2 * B, B ** 2 * R, R ** 2 * B, G ** 2 * R, B ** 2 * G,
                R ** 3, G ** 3, B ** 3]
    elif terms == 20:
        cols = [R, G, B, R * G, R * B, G * B, R ** 2, G ** 2, B ** 2, R * G * B,
                R ** 2 * G, G ** 2 * B, B ** 2 * R, R ** 2 * B, G ** 2 * R, B ** 2 * G,
                R ** 3, G ** 3, B ** 3, one]
    elif terms == 22:
        cols = [R, G, B, R * G, R * B, G * B, R ** 2, G ** 2, B ** 2, R * G * B,
                R ** 2 * G, G ** 2 * B, B ** 2 * R, R ** 2 * B, G ** 2 * R, B ** 2 * G,
                R ** 3, G ** 3, B ** 3, R ** 2 * G * B, R * G ** 2 * B, R * G * B ** 2]
    else:
        raise ValueError(f'"Cheung et al. (2004)" method does not define an augmented matrix with {terms} terms')
    return np.stack(cols, axis=-1)


ROOTPOLY_TERMS = {1: 3, 2: 6, 3: 13, 4: 22}


def polynomial_expansion_Finlayson2015(RGB, degree=1, root_polynomial_expansion=True):
    """``colour.characterisation.polynomial_expansion_Finlayson2015``."""
    RGB = np.asarray(RGB, dtype=np.float64)
    R, G, B = RGB[..., 0], RGB[..., 1], RGB[..., 2]
    if degree == 1:
        cols = [R, G, B]
    elif degree == 2:
        if root_polynomial_expansion:
            cols = [R, G, B, spow(R * G, 1 / 2), spow(G * B, 1 / 2), spow(R * B, 1 / 2)]
        else:
            cols = [R, G, B, R ** 2, G ** 2, B ** 2, R * G, G * B, R * B]
    elif degree == 3:
        if root_polynomial_expansion:
            cols = [R, G, B, spow(R * G, 1 / 2), spow(G * B, 1 / 2), spow(R * B, 1 / 2),
                    spow(R * G ** 2, 1 / 3), spow(G * B ** 2, 1 / 3), spow(R * B ** 2, 1 / 3),
                    spow(G * R ** 2, 1 / 3), spow(B * G ** 2, 1 / 3), spow(B * R ** 2, 1 / 3),
                    spow(R * G * B, 1 / 3)]
        else:
            cols = [R, G, B, R ** 2, G ** 2, B ** 2, R * G, G * B, R * B,
                    R ** 3, G ** 3, B ** 3, R * G ** 2, G * B ** 2, R * B ** 2,
                    G * R ** 2, B * G ** 2, B * R ** 2, R * G * B]
    elif degree == 4:
        if root_polynomial_expansion:
            cols = [R, G, B, spow(R * G, 1 / 2), spow(G * B, 1 / 2), spow(R * B, 1 / 2),
                    spow(R * G ** 2, 1 / 3), spow(G * B ** 2, 1 / 3), spow(R * B ** 2, 1 / 3),
                    spow(G * R ** 2, 1 / 3), spow(B * G ** 2, 1 / 3), spow(B * R ** 2, 1 / 3),
                    spow(R * G * B, 1 / 3),
                    spow(R ** 3 * G, 1 / 4), spow(R ** 3 * B, 1 / 4), spow(G ** 3 * R, 1 / 4),
                    spow(G ** 3 * B, 1 / 4), spow(B ** 3 * R, 1 / 4), spow(B ** 3 * G, 1 / 4),
                    spow(R ** 2 * G * B, 1 / 4), spow(G ** 2 * R * B, 1 / 4), spow(B ** 2 * R * G, 1 / 4)]
        else:
            raise ValueError("plain polynomial degree 4 is outside the hot path")
    else:
        raise ValueError(f'"Finlayson et al. (2015)" method does not define an expansion for degree {degree}')
    return np.stack(cols, axis=-1)


def expand(RGB, method="Cheung 2004", terms=3, degree=1, root_polynomial_expansion=True):
    if method == "Cheung 2004":
        return matrix_augmented_Cheung2004(RGB, terms)
    if method == "Finlayson 2015":
        return polynomial_expansion_Finlayson2015(RGB, degree, root_polynomial_expansion)
    raise ValueError(f"unsupported colour correction method {method!r}")


def matrix_colour_correction(M_T, M_R, method="Cheung 2004", **kw):
    """``colour.matrix_colour_correction``: ``M_R^T . pinv(expand(M_T)^T)`` -> (3,T)."""
    M_T = np.asarray(M_T, dtype=np.float64)
    M_R = np.asarray(M_R, dtype=np.float64)
    return np.dot(np.transpose(M_R), np.linalg.pinv(np.transpose(expand(M_T, method, **kw))))


def colour_correction(RGB, M_T, M_R, method="Cheung 2004", **kw):
    """``colour.colour_correction`` (imageProcessorWorker.py:337): the reference
    always calls it with the defaults, i.e. Cheung 2004 with 3 terms."""
    RGB = np.asarray(RGB, dtype=np.float64)
    shape = RGB.shape
    RGB = np.reshape(RGB, (-1, 3))
    RGB_e = expand(RGB, method, **kw)
    CCM = matrix_colour_correction(M_T, M_R, method, **kw)
    return np.reshape(np.transpose(np.dot(CCM, np.transpose(RGB_e))), shape)


# --------------------------------------------------------------------------
# cv2.GaussianBlur(src_f32, (k,k), sigma, borderType=BORDER_REFLECT)
# --------------------------------------------------------------------------


def usm_kernel_size(sigma):
    """Kernel size rule of adjust_sharpen (editingTools.py:756-758)."""
    k = int(6 * sigma) + 1
    if k % 2 == 0:
        k += 1
    return k


def gaussian_kernel_f32(ksize, sigma):
    """``cv2.getGaussianKernel(ksize, sigma, CV_32F)`` for sigma > 0: taps are
    computed in double, normalised in double, then rounded to float32.
    (OpenCV substitutes a fixed table only for ksize<=7 with sigma<=0, which
    adjust_sharpen never requests.)"""
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    g = np.exp(-(i * i) / (2.0 * float(sigma) * float(sigma)))
    return (g / g.sum()).astype(np.float32)


def _reflect_pad(x, r, axis):
    """BORDER_REFLECT ``fedcba|abcdefgh|hgfedcb`` == numpy 'symmetric'."""
    pad = [(0, 0)] * x.ndim
    pad[axis] = (r, r)
    return np.pad(x, pad, mode="symmetric")


def _correlate1d_f32(x, taps, axis):
    k = taps.shape[0]
    r = k // 2
    xp = _reflect_pad(x, r, axis)
    n = x.shape[axis]
    acc = np.zeros_like(x, dtype=np.float32)
    for j in range(k):
        sl = [slice(None)] * x.ndim
        sl[axis] = slice(j, j + n)
        acc += taps[j] * xp[tuple(sl)]  # float32 multiply-accumulate
    return acc


def gaussian_blur_reflect(src, ksize, sigma):
    """``cv2.GaussianBlur(src.astype(f32), (k,k), sigma, borderType=BORDER_REFLECT)``
    on a single (H,W) plane: separable, horizontal pass -> float32 buffer ->
    vertical pass, float32 taps and float32 accumulation, sigmaY = sigmaX."""
    src = np.asarray(src, dtype=np.float32)
    taps = gaussian_kernel_f32(ksize, sigma)
    tmp = _correlate1d_f32(src, taps, axis=1)
    return _correlate1d_f32(tmp, taps, axis=0)


# --------------------------------------------------------------------------
# np.percentile(x, q, axis=0), default method 'linear'
# --------------------------------------------------------------------------


def percentile_ranks(n, q_percent, legacy=True):
    """Index arithmetic of numpy's linear-interpolation percentile.

    Returns ``(lo, hi, gamma)`` with ``lo``/``hi`` the two order-statistic
    indices and ``gamma`` the interpolation weight.

    ``legacy=True``  -> numpy < 2.0 (the reference's pin, requirements.txt:7):
                         q and the virtual index are float64.
    ``legacy=False`` -> numpy >= 2.0: q is cast to the data dtype (float32)
                         before the index arithmetic.
    """
    if legacy:
        q = np.float64(q_percent) / np.float64(100.0)
    else:
        q = np.float32(q_percent) / np.float32(100.0)
    # _QuantileMethods['linear']['get_virtual_index']: (n - 1) * quantiles
    v = (n - 1) * q
    if v >= n - 1:  # _get_indexes: above bounds -> last element twice
        return n - 1, n - 1, type(v)(0.0)
    if v < 0:  # below bounds -> first element twice
        return 0, 0, type(v)(0.0)
    lo = int(np.floor(v))
    gamma = v - type(v)(lo)
    return lo, lo + 1, gamma


def lerp(a, b, t):
    """numpy's ``_lerp``: ``a + (b-a)*t`` switching to ``b - (b-a)*(1-t)`` for t >= 0.5."""
    d = b - a
    return np.where(t >= 0.5, b - d * (1 - t), a + d * t)


def percentile_linear(x, q_percent, legacy=True):
    """Per-column linear percentile of (n, C) float32 data.

    Returns ``(value, x_lo, x_hi, lo_index, gamma)``; ``x_lo``/``x_hi`` are the
    exact float32 order statistics (the "bit-exact" part of the contract).
    legacy -> float64 interpolation (numpy<2 returns float64 for this call
    shape because gamma is a shape-(1,) float64 array); otherwise float32.
    """
    x = np.asarray(x, dtype=np.float32)
    n = x.shape[0]
    lo, hi, gamma = percentile_ranks(n, q_percent, legacy)
    part = np.partition(x, sorted({lo, hi}), axis=0)
    x_lo, x_hi = part[lo], part[hi]
    if legacy:
        # _lerp subtracts in the data dtype first (float32), then the float64
        # gamma array promotes the rest of the expression to float64.
        d = (x_hi - x_lo).astype(np.float64)
        t = np.float64(gamma)
        a64, b64 = x_lo.astype(np.float64), x_hi.astype(np.float64)
        val = np.where(t >= 0.5, b64 - d * (1 - t), a64 + d * t)
    else:
        val = lerp(x_lo, x_hi, np.float32(gamma)).astype(np.float32)
    return val, x_lo, x_hi, lo, gamma
