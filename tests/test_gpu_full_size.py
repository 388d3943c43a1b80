"""Full-size runs (24 MP and 45 MP, the benchmark's image sizes) checked through properties that do not
need the oracle to process a whole image:

* pointwise chains: the oracle evaluated on a random sample of pixels;
* the unsharp mask: locality - the oracle evaluated on windows (corners, borders, tile and segment seams,
  the ragged last tile column) cut out with a margin wider than the blur radius;
* percentiles: the oracle's order statistics on the same strided sample (that part is cheap at any size);
* scheduling independence: the same image twice in one batch gives bit-identical outputs, equal to a
  single-image run.

All through the C ABI, same tolerances as the small-size parity tests."""

import numpy as np
import pytest

from gpu_util import DeviceBuffer, fit, run_chain_dev
from oracle import chain as oc
from oracle import inputs
from oracle import leaves as lv
from scanspaceimageprocessor_amd import default_context
from scanspaceimageprocessor_amd.chain import make_desc

pytestmark = pytest.mark.gpu

F32_TOL = 1e-5
H24, W24 = 4000, 6000
H45, W45 = 5504, 8256


@pytest.fixture(scope="module")
def swatches():
    return inputs.synthetic_swatches()


@pytest.fixture(scope="module")
def textured_24mp():
    """Low-frequency structure + noise, built in float32 row blocks to stay light on host memory."""
    rng = np.random.default_rng(2024)
    img = np.empty((H24, W24, 3), np.float32)
    xx = np.arange(W24, dtype=np.float32)[None, :]
    for y0 in range(0, H24, 500):
        yy = np.arange(y0, y0 + 500, dtype=np.float32)[:, None]
        base = 0.45 + 0.4 * np.sin(xx / 17.0) * np.cos(yy / 23.0)
        blk = np.stack([base, np.roll(base, 5, axis=1), 0.9 - base], axis=-1)
        blk += rng.normal(0.0, 0.03, size=blk.shape).astype(np.float32)
        img[y0:y0 + 500] = np.clip(blk, 0.0, 1.0)
    return img


def _sample(h, w, n, seed):
    rng = np.random.default_rng(seed)
    ys = rng.integers(0, h, n)
    xs = rng.integers(0, w, n)
    # always include the four corners and the last pixel of the first row
    ys[:5] = [0, 0, h - 1, h - 1, 0]
    xs[:5] = [0, w - 1, 0, w - 1, w - 2]
    return ys, xs


def test_config1_24mp_sampled_pixels(swatches):
    M_T, M_R = swatches
    img = inputs.linear_image(1234, H24, W24)
    got = run_chain_dev(img, ccm=fit(M_T, M_R), gains=[1.0], out="f32_srgb")[0]
    ys, xs = _sample(H24, W24, 200_000, 1)
    want = oc.config1(img[ys, xs][None], M_T, M_R)[0]
    assert np.max(np.abs(got[ys, xs] - want)) <= F32_TOL


def test_config4_45mp_sampled_pixels(swatches):
    M_T, M_R = swatches
    img = inputs.linear_image(4321, H45, W45)
    ccm = fit(M_T, M_R, method="Finlayson 2015", degree=2)
    got = run_chain_dev(img, ccm=ccm, gains=[1.0], sh=(0.5, -0.5, 7.0), final_clip=True, out="f32")[0]
    ys, xs = _sample(H45, W45, 100_000, 2)
    want = oc.config4(img[ys, xs][None], M_T, M_R, 2, 0.5, -0.5)[0]
    assert np.max(np.abs(got[ys, xs] - want)) <= F32_TOL


def test_config3_24mp_percentiles_and_sampled_pixels(swatches):
    M_T, M_R = swatches
    img = inputs.negative_image(99, H24, W24)
    ctx = default_context()
    src = DeviceBuffer.from_array(ctx, img)
    params = DeviceBuffer(ctx, 32)
    dst = DeviceBuffer(ctx, img.nbytes)
    try:
        ctx.negative_params(src.ptr, H24, W24, 1, 0.005, 0.995, True, params.ptr, None)
        desc = make_desc(H24, W24, 1, neg=True, neg_params_dev=params.ptr, ccm=fit(M_T, M_R), out="f32")
        ctx.chain_run(desc, src.ptr, dst.ptr)
        ctx.sync()
        p = params.to_array((8,), np.float32)
        got = dst.to_array(img.shape, np.float32)
    finally:
        src.free(); params.free(); dst.free()
    assert oc.negative_sample_step(H24, W24) == 9
    d = {}
    pos = oc.invert_negative_auto(img, 0.005, 0.995, legacy=True, details=d)  # whole image: ~3 s of numpy
    np.testing.assert_array_equal(p[:3], d["p_low"].astype(np.float32))
    assert np.max(np.abs(p[3:6] * d["denom"] - 1.0)) <= 1e-6
    ys, xs = _sample(H24, W24, 200_000, 3)
    want = lv.colour_correction(pos[ys, xs][None], M_T, M_R)[0]
    assert np.max(np.abs(got[ys, xs] - want)) <= F32_TOL


def _windows(h, w):
    """(y0, y1, x0, x1): corners, border middles, tile seams (every 24 rows / 128 columns), segment seams (every
    8 tiles = 192 rows; 4000 rows leave a ragged last tile of 16), the ragged last tile column
    (6000 = 46 x 128 + 112) and the centre."""
    return [
        (0, 80, 0, 150), (0, 80, w - 150, w), (h - 80, h, 0, 150), (h - 80, h, w - 150, w),
        (0, 70, w // 2 - 70, w // 2 + 70), (h - 70, h, 2990, 3130), (1990, 2060, 0, 140), (1990, 2060, w - 200, w),
        (160, 224, 100, 280), (2080, 2144, 2990, 3130), (3808, 3872, 5850, w), (h // 2 - 40, h // 2 + 40, w // 2 - 70, w // 2 + 70),
    ]


@pytest.mark.parametrize("amount", [1.0, 50.0])
def test_config2_24mp_windows_and_batch_invariance(swatches, textured_24mp, amount):
    M_T, M_R = swatches
    img = textured_24mp
    mult, radius, margin = 1.1, 2.0, 16
    batch = np.stack([img, img])
    got = run_chain_dev(batch, ccm=fit(M_T, M_R), gains=[mult], usm=(amount, radius, 0.0), final_clip=True, out="u16")
    del batch
    # scheduling independence: workgroups draw their segments in a different order for the two copies
    assert np.array_equal(got[0], got[1])
    worst = 0
    for (y0, y1, x0, x1) in _windows(H24, W24):
        ey0, ey1 = max(0, y0 - margin), min(H24, y1 + margin)
        ex0, ex1 = max(0, x0 - margin), min(W24, x1 + margin)
        want = oc.config2(img[ey0:ey1, ex0:ex1], M_T, M_R, mult, amount, radius, 0.0)
        want = want[y0 - ey0:y1 - ey0, x0 - ex0:x1 - ex0]
        diff = np.abs(got[0, y0:y1, x0:x1].astype(np.int64) - want.astype(np.int64))
        worst = max(worst, int(diff.max()))
    assert worst <= 1
    if amount == 1.0:
        single = run_chain_dev(img, ccm=fit(M_T, M_R), gains=[mult], usm=(amount, radius, 0.0), final_clip=True, out="u16")[0]
        assert np.array_equal(single, got[0])


def test_chart_path_with_adjustments_24mp_sampled_pixels_u16(swatches):
    """The worker's chart path with exposure and shadows/highlights, 16-bit sRGB output: the Lab edit followed by
    the encode is where float32 is tightest (near-black channels of saturated colours)."""
    M_T, M_R = swatches
    img = inputs.linear_image(555, H24, W24)
    got = run_chain_dev(img, ccm=fit(M_T, M_R), gains=[1.1, 2.0 ** 0.25], sh=(25.0 / 70, -0.3 * 1.3, 7.0),
                        final_clip=True, out="u16")[0]
    ys, xs = _sample(H24, W24, 300_000, 4)
    c = oc.worker_chart_path(img[ys, xs][None], M_T, M_R, multiplier=1.1, exposure_adj=0.25, shadow_adj=25.0,
                             highlight_adj=-0.3)
    want = oc.encode_image(c, ".tiff", 16)[0]
    diff = np.abs(got[ys, xs].astype(np.int64) - want.astype(np.int64))
    assert diff.max() <= 1


def test_no_chart_path_24mp_windows_u16(textured_24mp):
    """The worker's no-chart path (exposure, shadows/highlights, sharpen radius 1) at 24 MP, on windows."""
    img = textured_24mp
    margin = 12
    got = run_chain_dev(img, gains=[2.0 ** 0.2], sh=(20.0 / 70, -0.25 * 1.3, 7.0), usm=(45.0, 1.0, 0.0),
                        final_clip=True, out="u16")[0]
    worst = 0
    for (y0, y1, x0, x1) in _windows(H24, W24):
        ey0, ey1 = max(0, y0 - margin), min(H24, y1 + margin)
        ex0, ex1 = max(0, x0 - margin), min(W24, x1 + margin)
        c = oc.worker_nochart_path(img[ey0:ey1, ex0:ex1], exposure_adj=0.2, shadow_adj=20.0, highlight_adj=-0.25,
                                   sharpen_amount=45.0)
        want = oc.encode_image(c, ".tiff", 16)[y0 - ey0:y1 - ey0, x0 - ex0:x1 - ex0]
        diff = np.abs(got[y0:y1, x0:x1].astype(np.int64) - want.astype(np.int64))
        worst = max(worst, int(diff.max()))
    assert worst <= 1
