"""ss_chain_run2: the encoded image and the 8-bit preview from one pass are bit for bit what two passes give
(the reference makes the preview from the array it also encodes: imageProcessorWorker.py:907-922)."""

import numpy as np
import pytest

from gpu_util import DeviceBuffer, fit
from oracle import inputs
from scanspaceimageprocessor_amd import ScanSpaceError, _lib, default_context
from scanspaceimageprocessor_amd.chain import make_desc

pytestmark = pytest.mark.gpu

M_T, M_R = inputs.synthetic_swatches()


def _chains():
    lin3 = fit(M_T, M_R)
    root3 = fit(M_T, M_R, method="Finlayson 2015", degree=3)
    return {
        "ccm": dict(ccm=lin3, gains=[1.1], final_clip=True),
        "ccm_noclip": dict(ccm=lin3, gains=[]),
        "sh": dict(ccm=lin3, gains=[1.05], sh=(0.5, -0.4, 7.0), final_clip=True),
        "usm_plain": dict(ccm=lin3, gains=[1.1], usm=(1.0, 2.0, 0.0), final_clip=True),
        "usm_r1": dict(gains=[0.9, 1.2], usm=(60.0, 1.0, 3.0), final_clip=True),
        "sh_usm": dict(ccm=lin3, gains=[1.1], sh=(-0.3, 0.6, 5.0), usm=(45.0, 2.0, 1.0), final_clip=True),
        "usm_runtime_radius": dict(ccm=lin3, gains=[1.0], usm=(30.0, 4.0, 0.0), final_clip=True),
        "staged": dict(ccm=root3, gains=[1.1], usm=(80.0, 2.0, 0.0), final_clip=True),
        "dense_f64": dict(ccm=root3, gains=[1.1, 0.95], final_clip=True),  # the float64 pass finishes the chain itself
        "dense_f64_sh": dict(ccm=root3, gains=[1.1], sh=(0.4, -0.5, 7.0), final_clip=True),  # ... and stages it here
        "neg": dict(neg=([0.05, 0.04, 0.03], [1.2, 1.3, 1.4]), ccm=lin3, final_clip=True),
    }


@pytest.mark.parametrize("name", list(_chains()))
@pytest.mark.parametrize("out", ["u16", "f32_srgb", "f32"])
@pytest.mark.parametrize("second", ["u8_linear", "u8_srgb"])
def test_two_outputs_equal_two_passes(name, out, second):
    ctx = default_context()
    kw = _chains()[name]
    b, h, w = 2, 61, 149  # odd width, partial tiles, a remainder for every lane grouping
    imgs = np.stack([inputs.linear_image(500 + i, h, w) for i in range(b)]).astype(np.float32)
    imgs[0, :5] *= 1.3    # values the final clip has to catch
    src = DeviceBuffer.from_array(ctx, imgs)
    dt = _lib.OUT_DTYPES[make_desc(h, w, b, out=out).out_kind]
    one = DeviceBuffer(ctx, imgs.size * np.dtype(dt).itemsize)
    two = DeviceBuffer(ctx, imgs.size)
    ref1 = DeviceBuffer(ctx, imgs.size * np.dtype(dt).itemsize)
    ref2 = DeviceBuffer(ctx, imgs.size)
    try:
        ctx.chain_run2(make_desc(h, w, b, out=out, **kw), src.ptr, one.ptr, make_desc(h, w, b, out=second).out_kind, two.ptr)
        ctx.chain_run(make_desc(h, w, b, out=out, **kw), src.ptr, ref1.ptr)
        ctx.chain_run(make_desc(h, w, b, out=second, **kw), src.ptr, ref2.ptr)
        ctx.sync()
        got1, got2 = one.to_array(imgs.shape, dt), two.to_array(imgs.shape, np.uint8)
        want1, want2 = ref1.to_array(imgs.shape, dt), ref2.to_array(imgs.shape, np.uint8)
    finally:
        for x in (src, one, two, ref1, ref2):
            x.free()
    assert np.array_equal(got1.view(np.uint8), want1.view(np.uint8)), "first output differs from ss_chain_run's"
    assert np.array_equal(got2, want2), "second output differs from the 8-bit pass"
    assert got2.max() > 0


def test_two_outputs_on_a_24mp_image():
    """The worker's case: one 4000 x 6000 image, 16-bit sRGB plus the linear preview."""
    ctx = default_context()
    h, w = 4000, 6000
    img = inputs.linear_image(77, h, w).astype(np.float32)[None]
    kw = dict(ccm=fit(M_T, M_R), gains=[1.1], usm=(1.0, 2.0, 0.0), final_clip=True)
    src = DeviceBuffer.from_array(ctx, img)
    bufs = [DeviceBuffer(ctx, img.size * 2), DeviceBuffer(ctx, img.size), DeviceBuffer(ctx, img.size * 2), DeviceBuffer(ctx, img.size)]
    try:
        ctx.chain_run2(make_desc(h, w, 1, out="u16", **kw), src.ptr, bufs[0].ptr, _lib.SS_OUT_U8_LINEAR, bufs[1].ptr)
        ctx.chain_run(make_desc(h, w, 1, out="u16", **kw), src.ptr, bufs[2].ptr)
        ctx.chain_run(make_desc(h, w, 1, out="u8_linear", **kw), src.ptr, bufs[3].ptr)
        ctx.sync()
        assert np.array_equal(bufs[0].to_array(img.shape, np.uint16), bufs[2].to_array(img.shape, np.uint16))
        assert np.array_equal(bufs[1].to_array(img.shape, np.uint8), bufs[3].to_array(img.shape, np.uint8))
    finally:
        src.free()
        for x in bufs:
            x.free()


def test_second_output_arguments_are_checked():
    ctx = default_context()
    d = make_desc(8, 8, 1, out="u16")
    a, b, c = (DeviceBuffer(ctx, 8 * 8 * 3 * 4) for _ in range(3))
    try:
        with pytest.raises(ScanSpaceError, match="8-bit"):
            ctx.chain_run2(d, a.ptr, b.ptr, _lib.SS_OUT_U16_SRGB, c.ptr)
        with pytest.raises(ScanSpaceError, match="dst2_dev"):
            ctx.chain_run2(d, a.ptr, b.ptr, _lib.SS_OUT_U8_LINEAR, b.ptr)
        with pytest.raises(ScanSpaceError, match="dst2_dev"):
            ctx.chain_run2(d, a.ptr, b.ptr, _lib.SS_OUT_U8_LINEAR, 0)
        with pytest.raises(ScanSpaceError, match="Lab data"):
            ctx.chain_run2(make_desc(8, 8, 1, sh=(0.3, 0.0, 7.0), lab_io=True, out="f32"), a.ptr, b.ptr,
                           _lib.SS_OUT_U8_LINEAR, c.ptr)
    finally:
        for x in (a, b, c):
            x.free()


@pytest.mark.parametrize("shape,out", [((25, 41), "u16"),      # 1025 px: one pixel behind 512 full groups of two
                                       ((1, 1), "u16"),        # fewer pixels than one group
                                       ((13, 79), "u8_srgb"),  # 1027 px: three pixels behind 256 full groups of four
                                       ((1, 3), "u8_srgb"),
                                       ((32, 32), "u16"), ((25, 41), "f32")])
@pytest.mark.parametrize("name", ["ccm", "sh"])
def test_second_output_of_the_pixels_behind_the_last_full_group(shape, out, name):
    """A lane stores 12 bytes = 2 (16-bit) or 4 (8-bit) pixels; when H * W leaves a remainder and the full groups fill
    their last 1024-pixel chunk exactly, the remaining pixels get a trip of their own: their second output and their
    place in the record of pixels to evaluate again must not depend on that."""
    ctx = default_context()
    kw = _chains()[name]
    h, w = shape
    b = 2
    imgs = np.stack([inputs.linear_image(900 + i, h, w) for i in range(b)]).astype(np.float32)
    src = DeviceBuffer.from_array(ctx, imgs)
    dt = _lib.OUT_DTYPES[make_desc(h, w, b, out=out).out_kind]
    one = DeviceBuffer(ctx, imgs.size * np.dtype(dt).itemsize)
    two = DeviceBuffer.from_array(ctx, np.full(imgs.size, 0xA5, np.uint8))  # stale bytes must not survive
    ref1 = DeviceBuffer(ctx, imgs.size * np.dtype(dt).itemsize)
    ref2 = DeviceBuffer(ctx, imgs.size)
    try:
        ctx.chain_run2(make_desc(h, w, b, out=out, **kw), src.ptr, one.ptr, _lib.SS_OUT_U8_LINEAR, two.ptr)
        ctx.chain_run(make_desc(h, w, b, out=out, **kw), src.ptr, ref1.ptr)
        ctx.chain_run(make_desc(h, w, b, out="u8_linear", **kw), src.ptr, ref2.ptr)
        ctx.sync()
        got1, got2 = one.to_array(imgs.shape, dt), two.to_array(imgs.shape, np.uint8)
        want1, want2 = ref1.to_array(imgs.shape, dt), ref2.to_array(imgs.shape, np.uint8)
    finally:
        for x in (src, one, two, ref1, ref2):
            x.free()
    assert np.array_equal(got1.view(np.uint8), want1.view(np.uint8))
    assert np.array_equal(got2, want2), f"second output differs at {np.argwhere(got2 != want2)[:4].tolist()}"


def test_recorded_pixels_behind_the_last_full_group_are_evaluated_again():
    """The same tail pixels with an edit float32 cannot follow: they must reach the float64 kernel like any other
    pixel.  As a batch of 1 x 1 images every pixel is such a tail pixel (16-bit output: groups of two); the same
    pixels as one image give the count and the values to expect."""
    ctx = default_context()
    h, w = 160, 250
    img = inputs.linear_image(41, h, w).astype(np.float32)
    kw = dict(ccm=fit(M_T, M_R), gains=[1.0], sh=(0.5, -0.5, 7.0), final_clip=True)
    src = DeviceBuffer.from_array(ctx, img)
    whole, single = DeviceBuffer(ctx, img.size * 2), DeviceBuffer(ctx, img.size * 2)
    try:
        ctx.chain_run(make_desc(h, w, 1, out="u16", **kw), src.ptr, whole.ptr)
        n_whole = ctx.chain_requeued()
        ctx.chain_run(make_desc(1, 1, h * w, out="u16", **kw), src.ptr, single.ptr)
        n_single = ctx.chain_requeued()
        a, b = whole.to_array(img.shape, np.uint16), single.to_array(img.shape, np.uint16)
    finally:
        for x in (src, whole, single):
            x.free()
    assert n_whole > 0, "the case is meant to record pixels"
    assert n_single == n_whole
    assert np.array_equal(a, b)


def test_in_place_rules():
    """src == dst: fine for a float32 output of a chain without a Lab edit, refused wherever the result would depend
    on the order lanes run in or on the source being intact for the float64 re-evaluation."""
    ctx = default_context()
    h, w = 40, 50
    img = inputs.linear_image(3, h, w).astype(np.float32)[None]
    lin3 = fit(M_T, M_R)
    plain = dict(ccm=lin3, gains=[1.1], final_clip=True)
    a = DeviceBuffer.from_array(ctx, img)
    b = DeviceBuffer(ctx, img.nbytes)
    try:
        ctx.chain_run(make_desc(h, w, 1, out="f32_srgb", **plain), a.ptr, b.ptr)
        ctx.chain_run(make_desc(h, w, 1, out="f32_srgb", **plain), a.ptr, a.ptr)
        ctx.sync()
        assert np.array_equal(a.to_array(img.shape, np.float32), b.to_array(img.shape, np.float32))
        ctx.upload(a.ptr, img)
        for bad in (dict(sh=(0.5, -0.5, 7.0), **plain), dict(usm=(20.0, 1.0, 0.0), **plain)):
            for run in (ctx.chain_run, ctx.chain_run_static):
                with pytest.raises(ScanSpaceError, match="in-place"):
                    run(make_desc(h, w, 1, out="f32_srgb", **bad), a.ptr, a.ptr)
        with pytest.raises(ScanSpaceError, match="in-place"):
            ctx.chain_run(make_desc(h, w, 1, out="u16", **plain), a.ptr, a.ptr)
        # an 8-bit output of a shadows/highlights chain is never re-evaluated, but is narrower than the input: refused too
        with pytest.raises(ScanSpaceError, match="in-place"):
            ctx.chain_run(make_desc(h, w, 1, out="u8_srgb", sh=(0.5, -0.5, 7.0), **plain), a.ptr, a.ptr)
        ctx.sync()
        assert np.array_equal(a.to_array(img.shape, np.float32), img), "a refused call must not touch the image"
    finally:
        a.free()
        b.free()
