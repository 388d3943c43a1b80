"""Helpers for the -m gpu parity tests: everything goes through the C ABI (ctypes)."""


import os

import numpy as np

from scanspaceimageprocessor_amd import _lib, default_context
from scanspaceimageprocessor_amd.chain import make_desc
from scanspaceimageprocessor_amd.colour_fit import matrix_colour_correction


class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self.ptr = ctx.malloc(max(self.nbytes, 16))

    @classmethod
    def from_array(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(ctx, arr.nbytes)
        ctx.upload(b.ptr, arr)
        return b

    def to_array(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx.download(out, self.ptr)
        return out

    def free(self):
        if self.ptr:
            self.ctx.free(self.ptr)
            self.ptr = None


def run_chain_dev(images, **kw):
    """images: (B,H,W,3) float32 -> output through ss_chain_run on device buffers."""
    ctx = default_context()
    images = np.ascontiguousarray(images, dtype=np.float32)
    if images.ndim == 3:
        images = images[None]
    b, h, w, _ = images.shape
    desc = make_desc(h, w, b, **kw)
    src = DeviceBuffer.from_array(ctx, images)
    dt = _lib.OUT_DTYPES[desc.out_kind]
    dst = DeviceBuffer(ctx, images.size * np.dtype(dt).itemsize)
    try:
        ctx.chain_run(desc, src.ptr, dst.ptr)
        ctx.sync()
        return dst.to_array(images.shape, dt)
    finally:
        src.free()
        dst.free()


def fit(M_T, M_R, **kw):
    return matrix_colour_correction(M_T, M_R, **kw)


def fuzz_seeds(default_count):
    """Seeds of a randomised test: ``default_count`` of them from 0, or the range the environment asks for
    (SS_FUZZ_FIRST, SS_FUZZ_COUNT) when a longer soak run is wanted."""
    first = int(os.environ.get("SS_FUZZ_FIRST", "0"))
    count = int(os.environ.get("SS_FUZZ_COUNT", str(default_count)))
    return range(first, first + count)
