"""Parity of the HIP path against the oracle and the reference's golden vectors.
All tests call libscanspace_hip.so through its C ABI and need a real MI355X.

Tolerances (from the project's north star): max-abs 1e-5 on float32 output, <= 1 LSB on
16-bit (and 8-bit) output, exact order statistics for the percentiles."""

import ctypes as C

import contextlib

import numpy as np
import pytest

from conftest import golden_input
from gpu_util import DeviceBuffer, fit, fuzz_seeds, run_chain_dev
from scanspaceimageprocessor_amd.chain import make_desc
from oracle import chain as oc
from oracle import inputs
from oracle import leaves as lv
from scanspaceimageprocessor_amd import _lib, default_context
from scanspaceimageprocessor_amd.ImageProcessor import editingTools as et

pytestmark = pytest.mark.gpu

F32_TOL = 1e-5


@pytest.fixture(scope="module")
def swatches():
    return inputs.synthetic_swatches()


# ---- config 1: CCM + sRGB encode -------------------------------------------------------------


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (64, 96), (257, 389)])
def test_config1_ccm_srgb_f32(swatches, shape):
    M_T, M_R = swatches
    img = inputs.linear_image(100 + shape[0], *shape)
    got = run_chain_dev(img, ccm=fit(M_T, M_R), gains=[1.0], out="f32_srgb")[0]
    want = oc.config1(img, M_T, M_R)
    assert got.dtype == np.float32
    assert np.max(np.abs(got - want)) <= F32_TOL


def test_ccm_linear_output_is_unclipped(swatches):
    M_T, M_R = swatches
    img = inputs.wide_image(7, 40, 56)
    got = run_chain_dev(img, ccm=fit(M_T, M_R), out="f32")[0]
    want = lv.colour_correction(img, M_T, M_R)
    assert want.min() < 0 and want.max() > 1
    assert np.max(np.abs(got - want)) <= F32_TOL


@pytest.mark.parametrize("out,scale,dtype", [("u16", 65535.0, np.uint16), ("u8", 255.0, np.uint8)])
def test_quantised_outputs_within_one_lsb(swatches, out, scale, dtype):
    M_T, M_R = swatches
    img = inputs.linear_image(5, 123, 77)  # odd pixel count: exercises the tail path
    got = run_chain_dev(img, ccm=fit(M_T, M_R), gains=[1.0], final_clip=True, out=out)[0]
    c = np.clip(lv.colour_correction(img, M_T, M_R), 0, 1)
    want = oc.encode_image(c, ".tiff", 16 if out == "u16" else 8)
    assert got.dtype == dtype and want.dtype == dtype
    diff = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert diff.max() <= 1
    assert (diff != 0).mean() < 0.02  # flips only where the float value sits on a truncation boundary


def test_u8_linear_preview(swatches):
    M_T, M_R = swatches
    img = inputs.linear_image(6, 50, 70)
    got = run_chain_dev(img, ccm=fit(M_T, M_R), gains=[1.0], out="u8_linear")[0]
    c = np.clip(lv.colour_correction(img, M_T, M_R), 0, 1)
    want = np.uint8(np.clip(c * 255.0, 0, 255))  # imageProcessorWorker.py:922
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1


@pytest.mark.parametrize("terms", lv.CHEUNG_TERMS)
def test_cheung_polynomial_terms(swatches, terms):
    M_T, M_R = swatches
    img = inputs.linear_image(terms, 48, 64)
    ccm = fit(M_T, M_R, method="Cheung 2004", terms=terms)
    got = run_chain_dev(img, ccm=ccm, out="f32")[0]
    want = lv.colour_correction(img, M_T, M_R, "Cheung 2004", terms=terms)
    # (where float32 cannot hold the sum of |coefficient * term| to the bar, the library evaluates the pixel in float64)
    assert np.max(np.abs(got - want)) <= F32_TOL


@pytest.mark.parametrize("degree", [1, 2, 3, 4])
def test_root_polynomial_degrees(swatches, degree):
    M_T, M_R = swatches
    img = inputs.linear_image(20 + degree, 48, 64)
    img[0, 0] = 0.0  # exact zeros go through the roots
    ccm = fit(M_T, M_R, method="Finlayson 2015", degree=degree)
    got = run_chain_dev(img, ccm=ccm, out="f32")[0]
    want = lv.colour_correction(img, M_T, M_R, "Finlayson 2015", degree=degree, root_polynomial_expansion=True)
    # (the degree-4 fit on 24 swatches extrapolates to values in the thousands where a channel is almost zero: a float32
    # output holds those to half a unit of its last place, 1.2e-4 at 2,000, and no closer)
    assert np.all(np.abs(got - want) <= np.maximum(F32_TOL, 0.5 * np.spacing(np.abs(want).astype(np.float32)).astype(np.float64) * 1.001))
    np.testing.assert_array_equal(got[0, 0], 0.0)


# ---- negative film mode ------------------------------------------------------------------------


def _negative_params(images, low, high, legacy):
    ctx = default_context()
    b, h, w, _ = images.shape
    src = DeviceBuffer.from_array(ctx, images)
    params = DeviceBuffer(ctx, b * _lib.SS_NEG_PARAM_STRIDE * 4)
    detail = DeviceBuffer(ctx, b * C.sizeof(_lib.PercentileDetail))
    try:
        ctx.negative_params(src.ptr, h, w, b, low, high, legacy, params.ptr, detail.ptr)
        ctx.sync()
        p = params.to_array((b, _lib.SS_NEG_PARAM_STRIDE), np.float32)
        raw = detail.to_array((b * C.sizeof(_lib.PercentileDetail),), np.uint8)
        det = [_lib.PercentileDetail.from_buffer_copy(raw[i * C.sizeof(_lib.PercentileDetail):].tobytes())
               for i in range(b)]
        return p, det
    finally:
        src.free()
        params.free()
        detail.free()


def test_negative_percentiles_known_answers_of_the_numpy1_lerp():
    """numpy < 2's _lerp on float32 neighbours with a float64 weight, hand-computed: a + (b - a) t below t = 0.5,
    b - (b - a)(1 - t) from 0.5 on, the difference taken in float32 first.  11 samples per channel: the values
    k / 10 (red), their squares (green) and a constant (blue)."""
    k = np.arange(11, dtype=np.float32)
    img = np.stack([k / np.float32(10), (k / np.float32(10)) ** 2, np.full(11, 0.4, np.float32)], axis=-1)[None, None]
    img = np.ascontiguousarray(img[:, :, ::-1][:, :, np.random.default_rng(0).permutation(11)])  # order must not matter
    p, det = _negative_params(img, 0.25, 0.77, True)   # virtual indices 2.5 and 7.7
    for c in range(3):
        v = np.sort(img[0, 0, :, c])
        a0, b0, a1, b1 = v[2], v[3], v[7], v[8]
        want_low = float(b0) - float(np.float32(b0 - a0)) * (1.0 - 0.5)                 # t = 0.5: the upper form
        t_hi = 10 * ((0.77 * 100.0) / 100.0) - 7
        want_high = float(b1) - float(np.float32(b1 - a1)) * (1.0 - t_hi)               # t = 0.7
        assert det[0].p_low[c] == want_low and det[0].p_high[c] == want_high
        assert p[0, c] == np.float32(want_low)
        assert p[0, 3 + c] == np.float32(1.0 / max(want_high - want_low, 1e-3))
    p, det = _negative_params(img, 0.12, 0.34, True)   # virtual indices 1.2 and 3.4: the lower form twice
    v = np.sort(img[0, 0, :, 1])
    assert det[0].p_low[1] == float(v[1]) + float(np.float32(v[2] - v[1])) * (10 * ((0.12 * 100.0) / 100.0) - 1)
    assert det[0].p_high[1] == float(v[3]) + float(np.float32(v[4] - v[3])) * (10 * ((0.34 * 100.0) / 100.0) - 3)


@pytest.mark.parametrize("shape,low,high", [((96, 128), 0.05, 0.995), ((1100, 1300), 0.05, 0.995),
                                            ((700, 1000), 0.005, 0.995), ((64, 64), 0.0, 1.0), ((1, 1), 0.05, 0.995),
                                            # sample counts 3, 6, 35, 5063: no multiple of the 4 keys a lane loads at once
                                            ((1, 3), 0.05, 0.995), ((2, 3), 0.3, 0.7), ((5, 7), 0.05, 0.995),
                                            ((61, 83), 0.01, 0.9), ((1001, 1003), 0.05, 0.995)])
@pytest.mark.parametrize("legacy", [True, False])
def test_percentiles_are_exact_order_statistics(shape, low, high, legacy):
    imgs = np.stack([inputs.negative_image(31, *shape), inputs.wide_image(32, *shape)])
    params, det = _negative_params(imgs, low, high, legacy)
    for i in range(imgs.shape[0]):
        d = {}
        oc.invert_negative_auto(imgs[i], low, high, legacy=legacy, details=d)
        (lo_a, lo_b, lo_idx, _), (hi_a, hi_b, hi_idx, _) = d["low_stats"], d["high_stats"]
        assert det[i].step == d["step"] and det[i].n_samples == d["n"]
        assert det[i].k_low == lo_idx and det[i].k_high == hi_idx
        # bit-exact order statistics
        np.testing.assert_array_equal(np.array(det[i].x_low_lo[:], np.float32), lo_a)
        np.testing.assert_array_equal(np.array(det[i].x_low_hi[:], np.float32), lo_b)
        np.testing.assert_array_equal(np.array(det[i].x_high_lo[:], np.float32), hi_a)
        np.testing.assert_array_equal(np.array(det[i].x_high_hi[:], np.float32), hi_b)
        np.testing.assert_allclose(np.array(det[i].p_low[:]), d["p_low"], rtol=0, atol=1e-12 if legacy else 0)
        np.testing.assert_allclose(np.array(det[i].p_high[:]), d["p_high"], rtol=0, atol=1e-12 if legacy else 0)
        np.testing.assert_allclose(params[i, :3], d["p_low"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(params[i, 3:6], 1.0 / d["denom"], rtol=2e-7)


def test_percentiles_of_constant_and_two_valued_images():
    a = np.full((200, 300, 3), 0.25, np.float32)
    b = np.zeros((200, 300, 3), np.float32)
    b[::2] = 1.0
    params, det = _negative_params(np.stack([a, b]), 0.05, 0.995, True)
    np.testing.assert_array_equal(np.array(det[0].x_low_lo[:]), 0.25)
    np.testing.assert_array_equal(np.array(det[0].x_high_hi[:]), 0.25)
    np.testing.assert_allclose(params[0, 3:6], 1000.0, rtol=1e-6)  # denom floor 1e-3
    np.testing.assert_array_equal(np.array(det[1].x_low_lo[:]), 0.0)
    np.testing.assert_array_equal(np.array(det[1].x_high_hi[:]), 1.0)


@pytest.mark.parametrize("name", ["invert_small_default", "invert_small_q", "invert_wide", "invert_large_step2",
                                  "ensure_enabled"])
def test_negative_inversion_against_reference_vectors(golden, name, monkeypatch):
    meta, data = golden
    m = meta[name]
    img = golden_input(m)
    kw = {k: v for k, v in m["kwargs"].items() if k != "enabled"}
    for legacy in (False, True):  # the vectors were made under numpy>=2; numpy<2 differs below 1e-6
        monkeypatch.setattr(et, "NUMPY_LEGACY_PERCENTILE", legacy)
        got = et.invert_negative_auto(img, **kw)
        if m["stride"]:
            got = got[:: m["stride"][0], :: m["stride"][1]]
        assert got.dtype == np.float32
        assert np.max(np.abs(got - data[name])) <= 2e-6


def test_config3_negative_then_ccm_batch(swatches):
    M_T, M_R = swatches
    imgs = np.stack([inputs.negative_image(40 + i, 600, 900) for i in range(3)])
    ctx = default_context()
    src = DeviceBuffer.from_array(ctx, imgs)
    params = DeviceBuffer(ctx, 3 * 32)
    dst = DeviceBuffer(ctx, imgs.nbytes)
    try:
        ctx.negative_params(src.ptr, 600, 900, 3, 0.005, 0.995, True, params.ptr, None)
        desc = make_desc(600, 900, 3, neg=True, neg_params_dev=params.ptr, ccm=fit(M_T, M_R), gains=[1.0], out="f32")
        ctx.chain_run(desc, src.ptr, dst.ptr)
        ctx.sync()
        got = dst.to_array(imgs.shape, np.float32)
    finally:
        src.free(); params.free(); dst.free()
    for i in range(3):
        want = oc.config3(imgs[i], M_T, M_R, 0.005, 0.995, legacy=True)
        assert np.max(np.abs(got[i] - want)) <= F32_TOL


# ---- reference vectors through the drop-in Python entry points ---------------------------------

MIRROR = {
    "exposure": lambda img, kw: et.adjust_exposure(img, **kw),
    "sh": lambda img, kw: et.adjust_shadows_highlights_combined(img, **kw),
    "usm": lambda img, kw: et.adjust_sharpen(img, **kw),
    "all": lambda img, kw: et.apply_all_adjustments(img, **kw),
}


@pytest.mark.parametrize("name", ["exposure_up", "exposure_down", "sh_basic", "sh_shadow_only", "sh_chroma",
                                  "sh_falloff", "usm_r2_a50", "usm_r2_a1", "usm_r1_thr", "usm_r15_a80",
                                  "all_basic", "all_exposure_only", "all_chart_mode_like"])
def test_editing_tools_against_reference_vectors(golden, name):
    meta, data = golden
    m = meta[name]
    img = golden_input(m)
    before = img.copy()
    got = MIRROR[name.split("_")[0]](img, m["kwargs"])
    np.testing.assert_array_equal(img, before)  # input never modified
    assert got is not img and got.dtype == np.float32 and got.shape == img.shape
    assert np.max(np.abs(got - data[name])) <= F32_TOL


@pytest.mark.parametrize("name", ["sh_lab_input", "usm_lab_input"])
def test_lab_input_branch_against_reference_vectors(golden, name):
    meta, data = golden
    m = meta[name]
    img = golden_input(m)
    assert et._is_lab_image(img)
    got = MIRROR[name.split("_")[0]](img, m["kwargs"])
    # Lab units (L 0..100, a/b +-127): 1e-5 of full scale
    assert np.max(np.abs(got - data[name])) <= 1e-3


def test_neutral_settings_return_the_input_object():
    img = inputs.linear_image(1, 8, 8)
    assert et.adjust_exposure(img, 0) is img
    assert et.adjust_shadows_highlights_combined(img, 0, 0) is img
    assert et.adjust_sharpen(img, 0) is img
    assert et.adjust_white_balance(img, 5500, 5500) is img
    assert et.ensure_positive_if_negative(img, False) is img
    flat = np.zeros((8, 8), np.float32)
    assert et.invert_negative_auto(flat) is flat  # not (H,W,3): returned unchanged
    out = et.apply_all_adjustments(img)
    assert out is not img and np.array_equal(out, img)
    assert not et._is_lab_image(img)


# ---- config 2 / config 4 ------------------------------------------------------------------------


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (2, 2), (3, 5), (33, 129), (200, 300), (257, 389)])
@pytest.mark.parametrize("amount", [1.0, 50.0])
def test_config2_ccm_gain_usm_u16(swatches, shape, amount):
    M_T, M_R = swatches
    img = inputs.smooth_image(60, *shape)
    mult = 1.17
    got = run_chain_dev(img, ccm=fit(M_T, M_R), gains=[mult], usm=(amount, 2.0, 0.0), final_clip=True, out="u16")[0]
    want = oc.config2(img, M_T, M_R, mult, amount, 2.0, 0.0)
    diff = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert diff.max() <= 1


@pytest.mark.parametrize("max_wg,seg_tiles", [(1, 8), (3, 3), (7, 16), (512, 2)])
@pytest.mark.parametrize("radius", [1.0, 2.0])
def test_usm_rows_carried_between_tiles(swatches, monkeypatch, max_wg, seg_tiles, radius):
    """Consecutive tiles of a vertical segment hand their lower halo rows to the next tile instead of
    recomputing them, and workgroups draw further segments from a ticket counter.  SS_USM_SEG_TILES sets the
    segment length and SS_USM_MAX_WG limits the workgroups, so that on this small batch (2 images of 4 x 10
    tiles of 128 x 24 pixels) every workgroup walks several multi-tile segments."""
    M_T, M_R = swatches
    monkeypatch.setenv("SS_USM_MAX_WG", str(max_wg))
    monkeypatch.setenv("SS_USM_SEG_TILES", str(seg_tiles))
    img = inputs.smooth_image(62, 230, 400)
    imgs = np.stack([img, img[::-1].copy()])
    got = run_chain_dev(imgs, ccm=fit(M_T, M_R), gains=[1.05], usm=(50.0, radius, 0.0), final_clip=True, out="f32")
    for i in range(2):
        c = np.clip(lv.colour_correction(imgs[i], M_T, M_R) * 1.05, 0, 1)
        want = np.clip(oc.adjust_sharpen(c, 50.0, radius, 0.0), 0, 1)
        assert np.max(np.abs(got[i] - want)) <= F32_TOL


@pytest.mark.parametrize("seed", fuzz_seeds(16))
def test_sharpen_of_random_shapes_and_work_queues(swatches, monkeypatch, seed):
    """Ragged tile grids of several hundred pixels a side, 1-3 images, random segment length and workgroup
    limit (tile carries, ticket draws and steals in every mixture), against the float64 oracle."""
    M_T, M_R = swatches
    rng = np.random.default_rng(7000 + seed)
    h, w, b = int(rng.integers(1, 420)), int(rng.integers(1, 760)), int(rng.integers(1, 4))
    monkeypatch.setenv("SS_USM_MAX_WG", str(int(rng.choice([1, 2, 5, 13, 64, 4096]))))
    monkeypatch.setenv("SS_USM_SEG_TILES", str(int(rng.choice([1, 2, 3, 5, 8, 64]))))
    radius = float(rng.choice([0.5, 1.0, 1.5, 2.0, 2.5, 3.0]))
    amount, threshold = float(rng.choice([1.0, 40.0, 80.0])), float(rng.choice([0.0, 2.0]))
    with_ccm = bool(rng.random() < 0.5)
    out = ["f32", "u16"][int(rng.integers(0, 2))]
    imgs = np.stack([inputs.smooth_image(900 + 5 * seed + i, h, w) for i in range(b)])
    kw = dict(gains=[1.1], usm=(amount, radius, threshold), final_clip=True, out=out)
    if with_ccm:
        kw["ccm"] = fit(M_T, M_R)
    got = run_chain_dev(imgs, **kw)
    for i in range(b):
        x = lv.colour_correction(imgs[i], M_T, M_R) if with_ccm else imgs[i].astype(np.float64)
        pre = np.clip(x * 1.1, 0, 1)

        def sharpened():
            return np.clip(oc.adjust_sharpen(pre, amount, radius, threshold), 0, 1)

        c = sharpened()
        if out == "f32":
            assert np.all(np.abs(got[i] - c) <= F32_TOL + _blur_allowance(sharpened)), (seed, h, w, b, radius)
        else:
            want = np.clip(lv.eotf_inverse_sRGB(c) * 65535.0, 0, 65535.0).astype(np.uint16)
            diff = np.abs(got[i].astype(np.int64) - want.astype(np.int64))
            extra = _blur_allowance(lambda: lv.eotf_inverse_sRGB(sharpened()) * 65535.0)
            assert np.all(diff <= 1 + np.floor(F32_TOL * 65535.0 + extra)), (seed, h, w, b, radius, int(diff.max()))


@pytest.mark.parametrize("radius,threshold", [(0.5, 0.0), (1.0, 3.0), (1.5, 0.0), (3.0, 1.0), (5.0, 0.0)])
def test_usm_all_radii_f32(radius, threshold):
    img = inputs.smooth_image(61, 150, 210)
    got = run_chain_dev(img, usm=(60.0, radius, threshold), out="f32")[0]
    want = oc.adjust_sharpen(img, 60.0, radius, threshold)
    assert np.max(np.abs(got - want)) <= F32_TOL


@pytest.mark.parametrize("radius", [1.0, 2.0, 2.5])
@pytest.mark.parametrize("out", ["f32", "u16", "u8"])
def test_exposure_and_sharpen_without_a_chart(radius, out):
    """The worker's no-chart path with exposure and sharpen only (its own kernel specialisation)."""
    imgs = np.stack([inputs.linear_image(88 + i, 97, 301) for i in range(2)])
    got = run_chain_dev(imgs, gains=[2.0 ** 0.4], usm=(35.0, radius, 0.0), final_clip=True, out=out)
    for i in range(2):
        c = np.clip(oc.adjust_sharpen(np.clip(imgs[i].astype(np.float64) * 2.0 ** 0.4, 0, 1), 35.0, radius, 0.0), 0, 1)
        if out == "f32":
            assert np.max(np.abs(got[i] - c)) <= F32_TOL
        else:
            scale = 65535.0 if out == "u16" else 255.0
            want = np.clip(lv.eotf_inverse_sRGB(c) * scale, 0, scale).astype(got.dtype)
            assert np.max(np.abs(got[i].astype(np.int64) - want.astype(np.int64))) <= 1


def test_usm_of_constant_image_is_the_bare_round_trip():
    img = np.full((70, 140, 3), 0.5, np.float32)
    got = run_chain_dev(img, usm=(80.0, 2.0, 0.0), out="f32")[0]
    # nothing to sharpen: what is left is the reference's own round trip sRGB -> Lab -> sRGB, which is not the
    # identity (the two IEC sRGB matrices are rounded independently): about 1e-5 on mid grey
    want = oc.adjust_sharpen(img, 80.0, 2.0, 0.0)
    assert 3e-6 < np.max(np.abs(want - 0.5)) < 3e-5
    assert np.max(np.abs(got - want)) <= 2e-6


@pytest.mark.parametrize("stage_bytes", [None, 1, 2 * 150 * 220 * 12])
@pytest.mark.parametrize("method,kw,negative", [("Cheung 2004", {"terms": 5}, True), ("Cheung 2004", {"terms": 10}, True),
                                                ("Finlayson 2015", {"degree": 3}, False)])
def test_usm_behind_larger_colour_corrections(swatches, monkeypatch, stage_bytes, method, kw, negative):
    """Corrections of more than 3 terms run as a pointwise pass into a staging buffer in front of the tile
    kernel; SS_CHAIN_STAGE_BYTES sizes that buffer so that the 3 images go through in 1, 2 or 3 groups.
    Per-image inversion parameters and gain factors (device tables) must follow the groups.  (The root
    polynomial is fed positives: its roots have unbounded slope at the zeros an inversion produces, where
    float32 and float64 inputs cannot agree to 1e-5.)"""
    M_T, M_R = swatches
    if stage_bytes is not None:
        monkeypatch.setenv("SS_CHAIN_STAGE_BYTES", str(stage_bytes))
    h, w = 150, 220
    make = inputs.negative_image if negative else inputs.linear_image
    imgs = np.stack([make(50 + i, h, w) for i in range(3)])
    factors = np.array([0.9, 1.0, 1.25], np.float32)
    ctx = default_context()
    src = DeviceBuffer.from_array(ctx, imgs)
    params = DeviceBuffer(ctx, 3 * 32)
    gains = DeviceBuffer.from_array(ctx, factors)
    dst = DeviceBuffer(ctx, imgs.nbytes)
    try:
        if negative:
            ctx.negative_params(src.ptr, h, w, 3, 0.05, 0.995, True, params.ptr, None)
        desc = make_desc(h, w, 3, neg=True if negative else None, neg_params_dev=params.ptr if negative else None,
                         ccm=fit(M_T, M_R, method=method, **kw), gains=[1.0], gain0_dev=gains.ptr,
                         usm=(60.0, 2.0, 0.0), final_clip=True, out="f32")
        ctx.chain_run(desc, src.ptr, dst.ptr)
        ctx.sync()
        got = dst.to_array(imgs.shape, np.float32)
    finally:
        src.free(); params.free(); gains.free(); dst.free()
    for i in range(3):
        pos = oc.invert_negative_auto(imgs[i], 0.05, 0.995, legacy=True) if negative else imgs[i]
        c = np.clip(lv.colour_correction(pos, M_T, M_R, method=method, **kw) * float(factors[i]), 0, 1)
        want = np.clip(oc.adjust_sharpen(c, 60.0, 2.0, 0.0), 0, 1)
        assert np.max(np.abs(got[i] - want)) <= F32_TOL


@pytest.mark.parametrize("out", ["f32", "f32_srgb", "u16", "u8"])
@pytest.mark.parametrize("method,kw", [("Cheung 2004", {"terms": 22}), ("Finlayson 2015", {"degree": 3}),
                                       ("Finlayson 2015", {"degree": 4})])
def test_float64_colour_correction_finishes_simple_chains_itself(swatches, method, kw, out):
    """With only gains, the final clip and the encoding behind it, the float64 colour-correction pass stores the
    chain's output itself (one launch, no staging image).  Its tail is the pointwise kernel's arithmetic, so the
    result is, bit for bit, the bare correction (float32, unclipped) followed by a chain without a correction."""
    M_T, M_R = swatches
    imgs = np.stack([inputs.linear_image(310 + i, 37, 1031) for i in range(2)])  # 1031: a ragged last trip
    ccm = fit(M_T, M_R, method=method, **kw)
    gains = [np.array([1.1, 0.9, 1.2]), 0.8]
    one = run_chain_dev(imgs, ccm=ccm, gains=gains, final_clip=True, out=out)
    bare = run_chain_dev(imgs, ccm=ccm, gains=[], out="f32")
    two = run_chain_dev(bare, gains=gains, final_clip=True, out=out)
    assert np.array_equal(one.view(np.uint8), two.view(np.uint8))
    assert one.any()


@pytest.mark.parametrize("radius", [1.0, 2.0])
def test_no_chart_path_all_adjustments(radius):
    """The worker's no-chart path: exposure, shadows/highlights and sharpen with no colour correction."""
    img = inputs.linear_image(77, 130, 260)
    got = run_chain_dev(img, gains=[2.0 ** 0.3], sh=(0.4 / 70, -0.3 * 1.3, 7.0), usm=(45.0, radius, 0.0),
                        final_clip=True, out="f32")[0]
    want = oc.apply_all_adjustments(img, exposure=0.3, shadows=0.4, highlights=-0.3, sharpen_amount=45.0,
                                    sharpen_radius=radius)
    assert np.max(np.abs(got - np.clip(want, 0, 1))) <= F32_TOL


@pytest.mark.parametrize("degree", [2, 3])
def test_config4_rootpoly_shadows_highlights(swatches, degree):
    M_T, M_R = swatches
    img = inputs.linear_image(70 + degree, 160, 240)
    ccm = fit(M_T, M_R, method="Finlayson 2015", degree=degree)
    got = run_chain_dev(img, ccm=ccm, gains=[1.0], sh=(0.5, -0.5, 7.0), final_clip=True, out="f32")[0]
    want = oc.config4(img, M_T, M_R, degree, 0.5, -0.5)
    assert np.max(np.abs(got - want)) <= F32_TOL


def test_worker_chart_path_with_all_stages(swatches):
    M_T, M_R = swatches
    img = inputs.negative_image(80, 120, 180)
    d = {}
    oc.invert_negative_auto(img, 0.05, 0.995, legacy=True, details=d)
    neg = (d["p_low"], 1.0 / d["denom"])
    wb = et.calculate_white_balance_multipliers(5200.0, 6100.0, 0.0)
    got = run_chain_dev(img, neg=neg, ccm=fit(M_T, M_R), gains=[1.3, wb, 2.0 ** 0.4],
                        sh=(30.0 / 70, 0.2 * 1.3, 7.0), final_clip=True, out="f32")[0]
    want = oc.worker_chart_path(img, M_T, M_R, multiplier=1.3, negative_mode=True, exposure_adj=0.4,
                                shadow_adj=30.0, highlight_adj=0.2, wb_gains=wb)
    assert np.max(np.abs(got - want)) <= F32_TOL


# ---- dataset exposure normalisation ---------------------------------------------------------------


def test_exposure_stats_and_multipliers():
    ctx = default_context()
    rng = np.random.default_rng(9)
    n, h, w = 7, 341, 512
    thumbs = np.stack([inputs.linear_image(90 + i, h, w) * np.float32(2.0 ** rng.uniform(-0.5, 0.5)) for i in range(n)])
    thumbs[3] = 0.99  # mask empty -> falls back to the plain mean
    thumbs[5] = 0.0   # brightness 0 -> multiplier 1.0
    src = DeviceBuffer.from_array(ctx, thumbs)
    stats = DeviceBuffer(ctx, n * 4 * 8)
    bright = DeviceBuffer(ctx, n * 4)
    mult = DeviceBuffer(ctx, n * 4)
    try:
        ctx.exposure_stats(src.ptr, h, w, n, 0.06, 0.96, stats.ptr, bright.ptr)
        ctx.exposure_multipliers(bright.ptr, n, None, 0, n, mult.ptr)
        ctx.sync()
        b = bright.to_array((n,), np.float32)
        s = stats.to_array((n, 4), np.float64)
        m_mean = mult.to_array((n,), np.float32)
        ctx.exposure_multipliers(bright.ptr, n, 2, 1, 4, mult.ptr)
        ctx.sync()
        m_chart = mult.to_array((4,), np.float32)
    finally:
        src.free(); stats.free(); bright.free(); mult.free()
    want_b = [oc.image_brightness(t, 0.06, 0.96) for t in thumbs]
    np.testing.assert_allclose(b, want_b, rtol=1e-6, atol=0)
    for i, t in enumerate(thumbs):
        lum = 0.2126 * t[:, :, 0] + 0.7152 * t[:, :, 1] + 0.0722 * t[:, :, 2]
        mask = (lum > np.float32(0.06)) & (lum < np.float32(0.96))
        assert s[i, 1] == mask.sum() and s[i, 3] == lum.size  # exact counts: the mask is bit-exact
    np.testing.assert_allclose(m_mean, oc.exposure_multipliers(want_b, None), rtol=1e-6)
    np.testing.assert_allclose(m_chart, oc.exposure_multipliers(want_b, 2)[1:5], rtol=1e-6)
    assert m_mean[5] == 1.0


# ---- errors ------------------------------------------------------------------------------------------


def test_bad_descriptors_are_rejected():
    img = inputs.linear_image(1, 8, 8)
    with pytest.raises(_lib.ScanSpaceError) as e:
        run_chain_dev(img, ccm=(np.zeros((3, 4)), _lib.SS_CCM_CHEUNG, 4))
    assert e.value.code == _lib.SS_ERR_INVALID
    with pytest.raises(_lib.ScanSpaceError):
        run_chain_dev(img, usm=(10.0, 9.0, 0.0))  # 55-tap kernel: unsupported
    for radius in (float("inf"), float("nan"), 3.0e9, -1.0):
        with pytest.raises(_lib.ScanSpaceError):
            run_chain_dev(img, usm=(10.0, radius, 0.0))
    with pytest.raises(_lib.ScanSpaceError):
        run_chain_dev(img, sh=(1.0, 0.0, 7.0), lab_io=True, final_clip=True)


def test_other_entry_points_reject_bad_arguments():
    ctx = default_context()
    buf = DeviceBuffer(ctx, 4096)
    try:
        with pytest.raises(_lib.ScanSpaceError) as e:
            ctx.unpack_u16(buf.ptr + 2, buf.ptr, 16)            # source not 16-byte aligned
        assert e.value.code == _lib.SS_ERR_INVALID
        with pytest.raises(_lib.ScanSpaceError):
            ctx.unpack_u16(0, buf.ptr, 16)
        ctx.unpack_u16(buf.ptr, buf.ptr + 2048, 0)                # nothing to do is not an error
        with pytest.raises(_lib.ScanSpaceError):
            ctx.mask_regions(buf.ptr, 1, 70000, 1, 0.05, 0.98)    # columns are stored in 16 bits
        with pytest.raises(_lib.ScanSpaceError):
            ctx.mask_regions(buf.ptr, 4, 4, 1, float("nan"), 0.98)
        with pytest.raises(_lib.ScanSpaceError):
            ctx.resize_area(buf.ptr, 4, 4, 1, buf.ptr + 2048, 8, 8)   # area resampling only shrinks
        with pytest.raises(_lib.ScanSpaceError):
            ctx.negative_params(buf.ptr, 4, 4, 1, 0.5, 1.5, True, buf.ptr + 2048)   # fraction outside [0, 1]
        ctx.sync()
    finally:
        buf.free()
    assert ctx.pinned_empty((0, 3), np.float32).shape == (0, 3)


# ---- re-entrancy: one context per worker thread (image_space.py:3483 runs N workers at once) --------


def test_worker_threads_with_their_own_contexts_agree_with_serial_runs(swatches):
    import threading

    M_T, M_R = swatches
    imgs = [inputs.smooth_image(300 + i, 260, 300 + 16 * i) for i in range(6)]

    def work(i):
        # through the drop-in Python entry points: each thread gets its own context (own stream, own
        # workspace, own work-queue counter)
        out = et.adjust_sharpen(et.adjust_shadows_highlights_combined(imgs[i], 0.3, -0.2), 40.0 + i, 2.0, 0.0)
        return out

    serial = [work(i) for i in range(len(imgs))]
    results = [None] * len(imgs)
    errors = []

    def runner(i):
        try:
            for _ in range(3):
                results[i] = work(i)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=runner, args=(i,)) for i in range(len(imgs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for a, b in zip(serial, results):
        np.testing.assert_array_equal(a, b)


# ---- random stage combinations ------------------------------------------------------------------


def _oracle_chain(img, M_T, M_R, neg, ccm_kw, gains, sh, usm, final_clip, out):
    """The stages of a descriptor composed from the oracle's functions, in the kernel's stage order."""
    x = img.astype(np.float64)
    if neg is not None:
        # Stage 1 as the descriptor defines it (include/scanspace.h: float32 parameters, float32 subtraction and
        # product, the reference's invert_negative_auto under numpy >= 2 up to its division): every operation is a
        # single IEEE rounding, so these are the kernel's inverted values bit for bit, and everything behind them
        # is held to float64.  (What the inversion itself may differ by from the reference's two arithmetics is
        # the subject of the negative-mode tests, not of this one: a chain that cancels a channel amplifies one
        # unit in the last place of an inverted value a thousandfold.)
        lo, inv = (np.asarray(v, np.float32) for v in neg)
        q = np.clip((np.clip(img.astype(np.float32), np.float32(0), np.float32(1)) - lo) * inv, np.float32(0), np.float32(1))
        x = (np.float32(1.0) - q).astype(np.float64)
    if ccm_kw is not None:
        x = lv.colour_correction(x, M_T, M_R, **ccm_kw)
    for g in gains:
        x = np.clip(x * np.asarray(g, np.float64), 0, 1)
    if sh is not None:
        x = oc.adjust_shadows_highlights_combined(x, sh[0], sh[1], sh[2])
    if usm is not None:
        x = oc.adjust_sharpen(x, usm[0], usm[1], usm[2])
    if final_clip:
        x = np.clip(x, 0, 1)
    if out == "f32":
        return x
    enc = lv.eotf_inverse_sRGB(x)
    if out == "f32_srgb":
        return enc
    scale = 65535.0 if out == "u16" else 255.0
    return np.clip(enc * scale, 0, scale).astype(np.uint16 if out == "u16" else np.uint8)


def fuzz_case(seed, M_T, M_R):
    """One random descriptor: returns (imgs, kw, ingredients of the oracle chain)."""
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 300))
    b = int(rng.integers(1, 3))
    imgs = np.stack([inputs.linear_image(500 + seed * 3 + i, h, w) for i in range(b)])
    neg = None
    if rng.random() < 0.3:
        lo = rng.uniform(0.0, 0.2, 3)
        neg = (lo.astype(np.float32), (1.0 / rng.uniform(0.4, 0.9, 3)).astype(np.float32))
    ccm_kw = [None, {}, {"method": "Cheung 2004", "terms": 5}, {"method": "Cheung 2004", "terms": 8},
              {"method": "Finlayson 2015", "degree": 2}, {"method": "Cheung 2004", "terms": 14},
              {"method": "Finlayson 2015", "degree": 3}][int(rng.integers(0, 7))]
    gains = [rng.uniform(0.6, 1.6, 3).astype(np.float32).astype(np.float64) if rng.random() < 0.5
             else float(np.float32(rng.uniform(0.6, 1.6))) for _ in range(int(rng.integers(0, 4)))]
    sh = usm = None
    if rng.random() < 0.5:
        # the GUI's range (amounts far above it saturate everything; the chroma branch they switch on is pinned by
        # the reference vectors)
        # (values a float32 holds exactly: the C ABI takes the amounts as floats)
        sh = (float(np.float32(rng.uniform(-1, 1))), float(np.float32(rng.uniform(-1, 1))), 7.0)
    if rng.random() < 0.6:
        usm = (float(rng.choice([0.5, 20.0, 45.0, 80.0])), float(rng.choice([0.5, 1.0, 2.0, 2.5])),
               float(rng.choice([0.0, 2.0])))
    if (sh is not None or usm is not None) and not gains:
        gains = [1.0]  # the reference clips to [0,1] before these stages
    final_clip = bool(rng.random() < 0.7) or sh is not None or usm is not None
    out = ["f32", "f32_srgb", "u16", "u8"][int(rng.integers(0, 4))]
    if out != "f32" and not final_clip and ccm_kw is not None:
        final_clip = True  # encoded outputs of unclipped data are not a reference path
    kw = dict(gains=gains, final_clip=final_clip, out=out)
    if neg is not None:
        kw["neg"] = neg
    if ccm_kw is not None:
        kw["ccm"] = fit(M_T, M_R, **ccm_kw)
    if sh is not None:
        kw["sh"] = sh
    if usm is not None:
        kw["usm"] = usm
    return imgs, kw, (neg, ccm_kw, gains, sh, usm, final_clip, out)


BLUR_LAST_BITS = 1.5e-7  # what the float32 blurred lightness of two float32 blurs may differ by (about 2 ulp at 0.5)
_BLUR_PROBE = 2.0 ** -20  # a multiple of every float32 ulp below 1: adding it to the blurred plane is exact


@contextlib.contextmanager
def _blur_shifted(delta):
    """The oracle's float32 blur with `delta` added to every blurred value."""
    plain = lv.gaussian_blur_reflect
    lv.gaussian_blur_reflect = lambda src, ksize, sigma: plain(src, ksize, sigma) + np.float32(delta)
    try:
        yield
    finally:
        lv.gaussian_blur_reflect = plain


def _blur_allowance(evaluate):
    """What the last bits of the float32 blur may add to the error of sharpened values.

    The reference blurs L/100 in float32 (cv2.GaussianBlur on a float32 plane, editingTools.py:761-764), so the
    last bits of the blurred value depend on the order and fusing of 13 x 2 multiply-adds: OpenCV's, the oracle's
    and the kernel's all differ there, legitimately.  Everything behind the blur is pointwise, so what a value may
    move by is BLUR_LAST_BITS times the oracle's own response to its blurred lightness: `evaluate()` (the float64
    oracle chain up to the continuous value in front of any quantiser) is run with the blurred plane shifted up and
    down by 2^-20 and the larger one-sided slope taken (clips and the threshold mask have kinks).  For most pixels
    that is ~0.7 x amount / 50 x the slopes of the sRGB encodes behind the stage; a dark channel of a saturated
    pixel responds three times as strongly (2.6e-4 per unit of blur at amount 80: one ulp of the blurred value moves
    a twice-encoded output by 8e-6).  Everything else of the chain is held to the plain bars: where float32 cannot
    follow the float64 reference - Lab edits that cancel a channel to almost zero - the library re-evaluates the
    pixel in float64."""
    centre = evaluate()
    with _blur_shifted(+_BLUR_PROBE):
        up = evaluate()
    with _blur_shifted(-_BLUR_PROBE):
        down = evaluate()
    return BLUR_LAST_BITS * np.maximum(np.abs(up - centre), np.abs(down - centre)) / _BLUR_PROBE


@pytest.mark.parametrize("seed", fuzz_seeds(200))
def test_random_stage_combinations(swatches, seed):
    M_T, M_R = swatches
    imgs, kw, recipe = fuzz_case(seed, M_T, M_R)
    neg, ccm_kw, gains, sh, usm, final_clip, out = recipe
    got = run_chain_dev(imgs, **kw)
    for i in range(imgs.shape[0]):
        want = _oracle_chain(imgs[i], M_T, M_R, *recipe)
        if out in ("u16", "u8"):
            # 1 step: an error below one step that crosses a truncation boundary
            scale = 65535.0 if out == "u16" else 255.0
            diff = np.abs(got[i].astype(np.int64) - want.astype(np.int64))
            extra = 0.0
            if usm is not None:
                extra = _blur_allowance(lambda: _oracle_chain(imgs[i], M_T, M_R, *recipe[:-1], "f32_srgb") * scale)
            assert np.all(diff <= 1 + np.floor(F32_TOL * scale + extra)), (seed, list(kw.keys()), int(diff.max()))
        else:
            tol = F32_TOL
            err = np.abs(got[i] - want)
            extra = _blur_allowance(lambda: _oracle_chain(imgs[i], M_T, M_R, *recipe)) if usm is not None else 0.0
            assert np.all(err <= tol + extra), (seed, out, ccm_kw, sh, usm, len(gains), float(err.max()))
