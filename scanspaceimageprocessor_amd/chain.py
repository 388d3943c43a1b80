"""Builder for ``ss_chain_desc`` (include/scanspace.h): one fused pass over a batch."""

from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import ChainDesc

OUT_KINDS = {
    "f32": _lib.SS_OUT_F32_LINEAR, "f32_linear": _lib.SS_OUT_F32_LINEAR, "f32_srgb": _lib.SS_OUT_F32_SRGB,
    "u16": _lib.SS_OUT_U16_SRGB, "u16_srgb": _lib.SS_OUT_U16_SRGB,
    "u8": _lib.SS_OUT_U8_SRGB, "u8_srgb": _lib.SS_OUT_U8_SRGB, "u8_linear": _lib.SS_OUT_U8_LINEAR,
}


def make_desc(height, width, batch=1, *, neg=None, neg_params_dev=None, ccm=None, gains=(), gain0_dev=None,
              sh=None, usm=None, lab_io=False, final_clip=False, out="f32"):
    """
    neg            : None | True (parameters come from ``neg_params_dev`` / automatic percentiles) |
                     (p_low[3], inv_denom[3])
    ccm            : None | (matrix (3,T), kind, terms) as returned by colour_fit.matrix_colour_correction
    gains          : up to 3 entries, each a scalar or 3 per-channel multipliers; each stage is followed by clip[0,1]
    gain0_dev      : device pointer to per-image float32 factors multiplied into stage 0
    sh             : None | (shadow_amount, highlight_amount[, falloff_power])  (adjust_shadows_highlights_combined)
    usm            : None | (amount, radius[, threshold])                        (adjust_sharpen)
    """
    d = ChainDesc()
    d.height, d.width, d.batch = int(height), int(width), int(batch)
    if neg is not None and neg is not False:
        d.neg_enabled = 1
        if neg is not True:
            lo, inv = neg
            for k in range(3):
                d.neg_p_low[k] = float(lo[k])
                d.neg_inv_denom[k] = float(inv[k])
        d.neg_params_dev = neg_params_dev
    if ccm is not None:
        m, kind, terms = ccm
        m = np.asarray(m, dtype=np.float64)
        if m.shape != (3, terms):
            raise ValueError(f"colour correction matrix must be (3, {terms}), got {m.shape}")
        d.ccm_kind, d.ccm_terms = int(kind), int(terms)
        flat = m.astype(np.float32).ravel()
        low = (m.ravel() - flat.astype(np.float64)).astype(np.float32)  # ccm + ccm_lo = the float64 coefficient
        for i, v in enumerate(flat):
            d.ccm[i] = float(v)
            d.ccm_lo[i] = float(low[i])
    gains = list(gains)
    if len(gains) > _lib.SS_MAX_GAIN_STAGES:
        raise ValueError("at most 3 gain stages")
    d.n_gain_stages = len(gains)
    for s, g in enumerate(gains):
        g3 = np.broadcast_to(np.asarray(g, dtype=np.float64), (3,))
        for k in range(3):
            d.gain[s][k] = float(g3[k])
    d.gain0_dev = gain0_dev
    if sh is not None:
        d.sh_enabled = 1
        d.sh_shadow_amount = float(sh[0])
        d.sh_highlight_amount = float(sh[1])
        d.sh_falloff_power = float(sh[2]) if len(sh) > 2 else 7.0
    if usm is not None:
        d.usm_enabled = 1
        d.usm_amount = float(usm[0])
        d.usm_radius = float(usm[1])
        d.usm_threshold = float(usm[2]) if len(usm) > 2 else 0.0
    d.lab_io = int(bool(lab_io))
    d.final_clip = int(bool(final_clip))
    d.out_kind = OUT_KINDS[out] if isinstance(out, str) else int(out)
    return d
