"""ImageLoader.load_thumbnail_array (reference imageLoader.py:185-261): host behaviour on the CPU, the corrected
and inverted thumbnails on the GPU."""

import numpy as np
import pytest

from oracle import chain as oc
from oracle import inputs
from oracle import leaves as lv
from scanspaceimageprocessor_amd.ImageProcessor import imageLoader as il
from scanspaceimageprocessor_amd.ImageProcessor import imageWriters as iw


def _png(tmp_path, name, h=300, w=420, seed=3):
    img = inputs.linear_image(seed, h, w)
    arr8 = np.uint8(np.clip(lv.eotf_inverse_sRGB(img) * 255.0, 0, 255))
    path = str(tmp_path / name)
    iw.write_png(path, arr8)
    return path, arr8


def _pil_thumbnail(path, max_size):
    from PIL import Image

    with Image.open(path) as im:
        im.thumbnail(max_size)
        return np.asarray(im, dtype=np.float32) / 255.0


def test_thumbnail_of_a_rendered_file_and_the_cache_rules(tmp_path):
    path, _ = _png(tmp_path, "a.png")
    want = _pil_thumbnail(path, (128, 128))
    got = il.ImageLoader.load_thumbnail_array(path, max_size=(128, 128))
    assert got.dtype == np.float32 and max(got.shape[:2]) == 128
    np.testing.assert_array_equal(got, want)
    # the reference tests `if cache:`: an empty dict is never filled, one that holds something is
    empty = {}
    il.ImageLoader.load_thumbnail_array(path, max_size=(128, 128), cache=empty)
    assert empty == {}
    cache = {"other": {}}
    il.ImageLoader.load_thumbnail_array(path, max_size=(128, 128), cache=cache)
    np.testing.assert_array_equal(cache[path]["array"], want)
    marker = np.zeros((1, 1, 3), np.float32)
    cache[path]["array"] = marker
    assert il.ImageLoader.load_thumbnail_array(path, cache=cache) is marker   # a hit short-circuits everything


def test_unreadable_files_are_logged_and_yield_none(tmp_path):
    logs = []
    assert il.ImageLoader.load_thumbnail_array(str(tmp_path / "missing.png"), log_callback=logs.append) is None
    assert len(logs) == 1 and logs[0].startswith("[Thumb] Could not load thumbnail")
    bad = tmp_path / "broken.nef"
    bad.write_bytes(b"not a raw file")
    logs.clear()
    assert il.ImageLoader.load_thumbnail_array(str(bad), log_callback=logs.append) is None
    assert logs[0].startswith("[Thumb] RAW fallback failed") and logs[1].startswith("[Thumb] Could not load thumbnail")


def test_a_failing_correction_is_logged_and_the_plain_thumbnail_returned(tmp_path):
    from scanspaceimageprocessor_amd.ImageProcessor import imageProcessorWorker as ipw

    path, _ = _png(tmp_path, "b.png", 64, 80)
    logs = []
    saved = ipw.reference_swatches
    ipw.reference_swatches = None
    try:
        got = il.ImageLoader.load_thumbnail_array(path, chart_swatches=np.ones((24, 3)), correct_thumbnails=True,
                                                  log_callback=logs.append)
    finally:
        ipw.reference_swatches = saved
    assert got.dtype == np.float32 and logs and logs[0].startswith("[Thumb] Colour correction on thumbnail failed")


@pytest.mark.gpu
def test_corrected_thumbnail_matches_the_reference_arithmetic(tmp_path):
    from scanspaceimageprocessor_amd.ImageProcessor import imageProcessorWorker as ipw

    M_T, M_R = inputs.synthetic_swatches()
    path, _ = _png(tmp_path, "c.png")
    thumb = _pil_thumbnail(path, (256, 256))
    logs = []
    saved = ipw.reference_swatches
    ipw.set_reference_swatches(M_R)
    try:
        got = il.ImageLoader.load_thumbnail_array(path, max_size=(256, 256), chart_swatches=M_T, correct_thumbnails=True,
                                                  log_callback=logs.append)
    finally:
        ipw.reference_swatches = saved
    assert logs == ["[Thumb] Applied colour correction to thumbnail."]
    want = np.uint8(255 * lv.eotf_inverse_sRGB(np.clip(lv.colour_correction(thumb, M_T, M_R), 0, 1)))
    assert got.dtype == np.uint8 and got.shape == want.shape
    assert np.max(np.abs(got.astype(np.int32) - want.astype(np.int32))) <= 1


@pytest.mark.gpu
def test_negative_mode_inverts_the_thumbnail(tmp_path):
    neg = inputs.GENERATORS["negative"](9, 240, 360)
    arr8 = np.uint8(np.clip(neg * 255.0, 0, 255))
    path = str(tmp_path / "n.png")
    iw.write_png(path, arr8)
    thumb = _pil_thumbnail(path, (200, 200))
    got = il.ImageLoader.load_thumbnail_array(path, max_size=(200, 200), negative_mode=True)
    want = oc.invert_negative_auto(thumb, legacy=True)
    assert got.dtype == np.float32
    assert np.max(np.abs(got - want)) <= 1e-5


@pytest.mark.gpu
def test_swatch_preview_worker_and_overlay_encoding():
    """image_space.py:98-129 and 4098-4113: corrected 8-bit previews and plain 8-bit encodes."""
    from scanspaceimageprocessor_amd.ImageProcessor import previewWorkers as pw

    M_T, M_R = inputs.synthetic_swatches()
    img = inputs.linear_image(12, 211, 307)
    got, errors = [], []
    w = pw.SwatchPreviewWorker(img, M_T, M_R)
    w.signals.finished.connect(got.append)
    w.signals.error.connect(errors.append)
    w.run()
    assert not errors and len(got) == 1
    want = np.uint8(255 * lv.eotf_inverse_sRGB(np.clip(lv.colour_correction(img, M_T, M_R), 0, 1)))
    assert got[0].dtype == np.uint8 and np.max(np.abs(got[0].astype(np.int32) - want.astype(np.int32))) <= 1
    plain = pw.encode_preview_u8(img)
    want_plain = np.uint8(255 * lv.eotf_inverse_sRGB(img.astype(np.float64)))
    assert np.max(np.abs(plain.astype(np.int32) - want_plain.astype(np.int32))) <= 1
    bad = pw.SwatchPreviewWorker(img[..., 0], M_T, M_R)   # not an RGB image: reported, not raised
    bad.signals.error.connect(errors.append)
    bad.run()
    assert len(errors) == 1 and "expected an (H, W, 3) image" in errors[0]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(400, 600), (800, 1000)])   # below and above the 0.5 MP display rule
def test_adjusted_preview_bytes_match_adjust_then_convert(shape):
    """image_space.py:2988-3058: apply_all_adjustments, then uint8(255 * clip) or uint8(255 * encode(clip))."""
    from scanspaceimageprocessor_amd.ImageProcessor import previewWorkers as pw

    img = inputs.smooth_image(31, *shape)
    small = shape[0] * shape[1] <= 500_000

    def to_bytes(x):
        x = np.clip(x, 0, 1)
        return np.uint8(255 * (x if small else lv.eotf_inverse_sRGB(x)))

    for adj, wb in ((dict(exposure=0.4, shadows=20.0, highlights=-0.5, sharpen_amount=35.0), None),
                    (dict(shadows=30.0, sharpen_amount=50.0), None),         # Lab stages first: the two-pass route
                    (dict(), None)):
        got = pw.adjusted_preview_u8(img, **adj)
        want = to_bytes(oc.apply_all_adjustments(img, **adj))
        assert got.dtype == np.uint8 and got.shape == img.shape
        diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert diff.max() <= 1, (adj, int(diff.max()))
    np.testing.assert_array_equal(pw.display_u8(img), pw.adjusted_preview_u8(img))
