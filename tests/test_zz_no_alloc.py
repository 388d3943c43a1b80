"""ss_chain_run_static (the allocation-free entry point) against ss_chain_run, through the C ABI."""

import numpy as np
import pytest

from scanspaceimageprocessor_amd import _lib, default_context
from scanspaceimageprocessor_amd.chain import make_desc

from gpu_util import DeviceBuffer, fit

pytestmark = pytest.mark.gpu


def _swatches():
    r = np.random.default_rng(7)
    M_R = 0.03 + 0.87 * r.random((24, 3))
    c = np.random.default_rng(8)
    A = np.eye(3) + 0.15 * (c.random((3, 3)) * 2.0 - 1.0)
    return (M_R @ A.T + c.normal(0.0, 0.002, size=(24, 3))).astype(np.float32), M_R


def _both(ctx, images, **kw):
    b, h, w, _ = images.shape
    desc = make_desc(h, w, b, **kw)
    dt = _lib.OUT_DTYPES[desc.out_kind]
    src = DeviceBuffer.from_array(ctx, images)
    d0 = DeviceBuffer(ctx, images.size * np.dtype(dt).itemsize)
    d1 = DeviceBuffer(ctx, images.size * np.dtype(dt).itemsize)
    try:
        ctx.chain_run(desc, src.ptr, d0.ptr)
        ctx.chain_run_static(desc, src.ptr, d1.ptr)
        ctx.sync()
        return d0.to_array(images.shape, dt), d1.to_array(images.shape, dt)
    finally:
        src.free()
        d0.free()
        d1.free()


def test_static_entry_point_matches_chain_run_twice_in_a_row():
    """Two calls in a row on one GPU, the second with a smaller input: the shared static counters must be
    cleared per call.  Same kernels, so the outputs are bit-identical."""
    ctx = default_context()
    M_T, M_R = _swatches()
    ccm = fit(M_T, M_R)
    rng = np.random.default_rng(5)
    big = (rng.random((3, 301, 517, 3)) ** 2.2 * 0.9).astype(np.float32)
    small = (rng.random((1, 97, 150, 3)) ** 2.2 * 0.9).astype(np.float32)
    for images in (big, small):
        for kw in (dict(ccm=ccm, gains=[1.1], usm=(1.0, 2.0, 0.0), final_clip=True, out="u16"),
                   dict(ccm=ccm, gains=[0.9], sh=(0.5, -0.5, 7.0), usm=(50.0, 1.0, 2.0), final_clip=True, out="f32_srgb"),
                   dict(ccm=ccm, gains=[1.0], final_clip=True, out="f32_srgb")):
            ref, got = _both(ctx, images, **kw)
            assert np.array_equal(ref, got)
            assert ref.any()


def test_static_entry_point_refuses_what_needs_a_staging_buffer():
    ctx = default_context()
    M_T, M_R = _swatches()
    ccm = fit(M_T, M_R, method="Finlayson 2015", degree=3)
    img = np.full((1, 32, 40, 3), 0.25, np.float32)
    desc = make_desc(32, 40, 1, ccm=ccm, usm=(10.0, 1.0, 0.0), out="u16")
    src = DeviceBuffer.from_array(ctx, img)
    dst = DeviceBuffer(ctx, img.size * 2)
    try:
        with pytest.raises(_lib.ScanSpaceError) as e:
            ctx.chain_run_static(desc, src.ptr, dst.ptr)
        assert e.value.code == _lib.SS_ERR_UNSUPPORTED and "ss_chain_run" in str(e.value)
    finally:
        src.free()
        dst.free()
