"""Area-average downscale (the reference's cv2.INTER_AREA preview / thumbnail producers) through the C ABI,
against the float64 restatement in oracle/resize.py.  Parity with OpenCV itself is unpinned (library absent);
tolerance 2e-6 absolute on [0,1] data: float32 accumulation against float64."""

import numpy as np
import pytest

from gpu_util import DeviceBuffer, fuzz_seeds
from oracle import inputs
from oracle import resize as orz
from scanspaceimageprocessor_amd import default_context
from scanspaceimageprocessor_amd.ImageProcessor import imageLoader as il

pytestmark = pytest.mark.gpu

TOL = 2e-6


@pytest.mark.parametrize("shape,dsize", [
    ((12, 20), (5, 4)),          # integer scales
    ((37, 53), (53, 37)),        # same size
    ((37, 53), (1, 1)),          # everything into one pixel
    ((5, 5), (2, 2)),            # the hand-computed fractional case
    ((257, 389), (100, 99)),     # ragged, non-integer
    ((300, 1000), (999, 299)),   # scale just above 1: most cells touch two source pixels
    ((1000, 700), (3, 700)),     # one axis only
    ((600, 900), (183, 122)),
])
def test_resize_area_matches_restatement(shape, dsize):
    img = inputs.linear_image(sum(shape) + dsize[0], *shape)
    got = il.resize_area(img, dsize)
    want = orz.resize_area(img, dsize)
    assert got.shape == (dsize[1], dsize[0], 3) and got.dtype == np.float32
    assert np.max(np.abs(got - want)) <= TOL


def test_batched_device_call_and_reference_size_rules():
    ctx = default_context()
    h, w, b = 341 * 3 + 2, 512 * 3 + 1, 3
    imgs = np.stack([inputs.smooth_image(10 + i, h, w) for i in range(b)])
    new_w, new_h = il.thumbnail_size(h, w, (512, 512))
    assert (new_w, new_h) == orz.thumbnail_size(h, w, (512, 512))
    src = DeviceBuffer.from_array(ctx, imgs)
    dst = DeviceBuffer(ctx, b * new_h * new_w * 12)
    try:
        ctx.resize_area(src.ptr, h, w, b, dst.ptr, new_h, new_w)
        ctx.sync()
        got = dst.to_array((b, new_h, new_w, 3), np.float32)
    finally:
        src.free(); dst.free()
    for i in range(b):
        assert np.max(np.abs(got[i] - orz.resize_area(imgs[i], (new_w, new_h)))) <= TOL


def test_full_size_previews_24mp_and_45mp():
    for (h, w, target) in [(4000, 6000, il.RAW_PREVIEW_PIXELS), (5504, 8256, il.EDIT_PREVIEW_PIXELS)]:
        img = inputs.linear_image(h, h, w)
        size = il.preview_size(h, w, target)
        assert size == orz.preview_size(h, w, target)
        got = il.raw_preview(img, target) if target == il.RAW_PREVIEW_PIXELS else il.resize_for_preview(img, target)
        want = orz.resize_area(img, size)
        assert got.shape == (size[1], size[0], 3)
        assert np.max(np.abs(got - want)) <= TOL


def test_small_images_pass_through_and_bad_input_is_rejected():
    img = inputs.linear_image(1, 100, 120)
    assert il.resize_for_preview(img) is img
    assert il.raw_preview(img) is img
    assert il.thumbnail_from_array(img) is img
    with pytest.raises(ValueError):
        il.resize_area(img[:, :, :2], (10, 10))
    big = inputs.linear_image(2, 1100, 1000)
    big[5, 5, 1] = np.nan
    with pytest.raises(ValueError):
        il.raw_preview(big)
    with pytest.raises(RuntimeError):
        il.resize_area(img, (200, 50))  # wider than the source: the library only shrinks


def test_uint16_decoder_output_gives_the_same_previews():
    rng = np.random.default_rng(21)
    raw16 = (rng.random((1300, 1900, 3)) ** 2.2 * 65535).astype(np.uint16)
    as_float = np.array(raw16, dtype=np.float32) / 65535.0   # imageLoader.py:92
    np.testing.assert_array_equal(il.raw_preview(raw16), il.raw_preview(as_float))
    np.testing.assert_array_equal(il.thumbnail_from_array(raw16), il.thumbnail_from_array(as_float))
    small16 = raw16[:100, :150]
    np.testing.assert_array_equal(il.raw_preview(small16), as_float[:100, :150])
    np.testing.assert_array_equal(il.thumbnail_from_array(small16), as_float[:100, :150])


@pytest.mark.parametrize("seed", fuzz_seeds(40))
def test_random_shapes_and_scales(seed):
    # scales below 2, between 2 and 16 (the strip kernels: 64 / 128 / 256 destination pixels per workgroup) and above,
    # independent per axis, ragged sizes
    rng = np.random.default_rng(9000 + seed)
    h, w = int(rng.integers(1, 400)), int(rng.integers(1, 1200))
    sx = float(rng.choice([1.0, 1.3, 2.0, 2.7, 4.0, 5.5, 8.0, 11.9, 16.0, 23.0]))
    sy = float(rng.choice([1.0, 1.7, 3.0, 6.2, 9.0, 20.0]))
    dsize = (max(1, int(w / sx)), max(1, int(h / sy)))
    img = inputs.linear_image(seed, h, w)
    got = il.resize_area(img, dsize)
    want = orz.resize_area(img, dsize)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= TOL, (h, w, dsize)
