"""Host side of the colour-correction stage: the least-squares fit of the 3 x T
matrix from the 24 chart swatches (float64 numpy, 24 rows) and its hand-over
to the kernels.  Mirrors the parts of ``colour.colour_correction`` the reference
calls at imageProcessorWorker.py:337 (always the defaults: 'Cheung 2004',
terms=3) plus the other term counts and the 'Finlayson 2015' root-polynomial
expansion of the same API.
"""

from __future__ import annotations

import numpy as np

from . import _lib

CHEUNG_TERMS = (3, 5, 7, 8, 10, 11, 14, 16, 17, 19, 20, 22)
ROOTPOLY_TERMS = {1: 3, 2: 6, 3: 13, 4: 22}


def _sroot(a, n):
    return np.sign(a) * np.abs(a) ** (1.0 / n)


def expand_swatches(M, method="Cheung 2004", terms=3, degree=1):
    """Polynomial expansion of an (N, 3) swatch table -> (N, T), float64."""
    M = np.asarray(M, dtype=np.float64)
    R, G, B = M[:, 0], M[:, 1], M[:, 2]
    one = np.ones_like(R)
    if method == "Cheung 2004":
        if terms not in CHEUNG_TERMS:
            raise ValueError(f'"Cheung et al. (2004)" method does not define an augmented matrix with {terms} terms')
        base = [R, G, B]
        pairs = [R * G, R * B, G * B]
        squares = [R * R, G * G, B * B]
        rgb = [R * G * B]
        mixed3 = [R * R * G, G * G * B, B * B * R]
        mixed3b = [R * R * B, G * G * R, B * B * G]
        cubes = [R ** 3, G ** 3, B ** 3]
        table = {
            3: base,
            5: base + rgb + [one],
            7: base + pairs + [one],
            8: base + pairs + rgb + [one],
            10: base + pairs + squares + [one],
            11: base + pairs + squares + rgb + [one],
            14: base + pairs + squares + rgb + cubes + [one],
            16: base + pairs + squares + rgb + mixed3 + cubes,
            17: base + pairs + squares + rgb + mixed3 + cubes + [one],
            19: base + pairs + squares + rgb + mixed3 + mixed3b + cubes,
            20: base + pairs + squares + rgb + mixed3 + mixed3b + cubes + [one],
            22: base + pairs + squares + rgb + mixed3 + mixed3b + cubes
            + [R ** 2 * G * B, R * G ** 2 * B, R * G * B ** 2],
        }
        cols = table[terms]
    elif method == "Finlayson 2015":
        if degree not in ROOTPOLY_TERMS:
            raise ValueError(f'"Finlayson et al. (2015)" method does not define an expansion for degree {degree}')
        cols = [R, G, B]
        if degree >= 2:
            cols += [_sroot(R * G, 2), _sroot(G * B, 2), _sroot(R * B, 2)]
        if degree >= 3:
            # (the products in colour-science's order, squares first: with coefficients of 10^5 on an ill-conditioned
            # fit, a last-bit difference in a swatch's term shows in the fitted matrix)
            cols += [_sroot(R * G ** 2, 3), _sroot(G * B ** 2, 3), _sroot(R * B ** 2, 3),
                     _sroot(G * R ** 2, 3), _sroot(B * G ** 2, 3), _sroot(B * R ** 2, 3), _sroot(R * G * B, 3)]
        if degree >= 4:
            cols += [_sroot(R ** 3 * G, 4), _sroot(R ** 3 * B, 4), _sroot(G ** 3 * R, 4), _sroot(G ** 3 * B, 4),
                     _sroot(B ** 3 * R, 4), _sroot(B ** 3 * G, 4), _sroot(R ** 2 * G * B, 4),
                     _sroot(G ** 2 * R * B, 4), _sroot(B ** 2 * R * G, 4)]
    else:
        raise ValueError(f"unsupported colour correction method {method!r} (hot path: 'Cheung 2004', 'Finlayson 2015')")
    return np.stack(cols, axis=-1)


def matrix_colour_correction(M_T, M_R, method="Cheung 2004", terms=3, degree=1, root_polynomial_expansion=True):
    """``M_R^T . pinv(expand(M_T)^T)`` -> (3, T) float64 matrix plus the kernel's (kind, terms)."""
    if method == "Finlayson 2015" and not root_polynomial_expansion:
        raise NotImplementedError("plain-polynomial Finlayson 2015 is outside the accelerated path")
    M_T = np.asarray(M_T, dtype=np.float64)
    M_R = np.asarray(M_R, dtype=np.float64)
    if M_T.shape != M_R.shape or M_T.ndim != 2 or M_T.shape[1] != 3:
        raise ValueError(f"swatch tables must both be (N, 3), got {M_T.shape} and {M_R.shape}")
    E = expand_swatches(M_T, method, terms, degree)
    ccm = np.dot(np.transpose(M_R), np.linalg.pinv(np.transpose(E)))
    kind = _lib.SS_CCM_CHEUNG if method == "Cheung 2004" else _lib.SS_CCM_ROOTPOLY
    return ccm, kind, E.shape[1]
