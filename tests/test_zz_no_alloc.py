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
    root3 = fit(M_T, M_R, method="Finlayson 2015", degree=3)  # evaluated in float64 for every pixel
    rng = np.random.default_rng(5)
    big = (rng.random((3, 301, 517, 3)) ** 2.2 * 0.9).astype(np.float32)
    small = (rng.random((1, 97, 150, 3)) ** 2.2 * 0.9).astype(np.float32)
    for images in (big, small):
        for kw in (dict(ccm=ccm, gains=[1.1], usm=(1.0, 2.0, 0.0), final_clip=True, out="u16"),
                   dict(ccm=ccm, gains=[0.9], sh=(0.5, -0.5, 7.0), usm=(50.0, 1.0, 2.0), final_clip=True, out="f32_srgb"),
                   dict(ccm=ccm, gains=[1.0], final_clip=True, out="f32_srgb"),
                   dict(ccm=root3, gains=[1.05], final_clip=True, out="u16")):
            ref, got = _both(ctx, images, **kw)
            assert np.array_equal(ref, got)
            assert ref.any()


@pytest.mark.parametrize("behind", [dict(usm=(10.0, 1.0, 0.0)), dict(gains=[1.0], sh=(0.5, -0.5, 7.0))])
def test_static_entry_point_refuses_what_needs_a_staging_buffer(behind):
    ctx = default_context()
    M_T, M_R = _swatches()
    ccm = fit(M_T, M_R, method="Finlayson 2015", degree=3)
    img = np.full((1, 32, 40, 3), 0.25, np.float32)
    desc = make_desc(32, 40, 1, ccm=ccm, out="u16", **behind)
    src = DeviceBuffer.from_array(ctx, img)
    dst = DeviceBuffer(ctx, img.size * 2)
    try:
        with pytest.raises(_lib.ScanSpaceError) as e:
            ctx.chain_run_static(desc, src.ptr, dst.ptr)
        assert e.value.code == _lib.SS_ERR_UNSUPPORTED and "ss_chain_run" in str(e.value)
    finally:
        src.free()
        dst.free()


# ---- the property itself: nothing in the call allocates -------------------------------------------------------------
# A stream under capture in global mode refuses hipMalloc / hipFree / synchronisation from the capturing thread, and
# what the call queues becomes a graph instead of running.  ss_chain_run_static is captured, the graph replayed twice,
# and the results compared with ss_chain_run; ss_chain_run itself, on a context whose workspace still has to grow,
# fails under the same capture (it calls hipMalloc): that contrast is the test.

def _hip():
    import ctypes as C

    for name in ("libamdhip64.so.7", "libamdhip64.so"):  # the runtime already in the process answers to its soname
        try:
            return C.CDLL(name)
        except OSError:
            continue
    pytest.skip("no HIP runtime to drive the capture with")


class _Capture:
    """hipStreamBeginCapture(global) ... hipStreamEndCapture on a context's stream."""

    def __init__(self, hip, stream):
        import ctypes as C

        self.C, self.hip, self.stream = C, hip, C.c_void_p(stream)
        self.graph = C.c_void_p()
        self.status = None

    def __enter__(self):
        rc = self.hip.hipStreamBeginCapture(self.stream, 0)  # hipStreamCaptureModeGlobal
        assert rc == 0, f"hipStreamBeginCapture: {rc}"
        return self

    def __exit__(self, *exc):
        self.status = self.hip.hipStreamEndCapture(self.stream, self.C.byref(self.graph))
        return False


def test_static_entry_point_runs_under_stream_capture_and_replays():
    import ctypes as C

    hip = _hip()
    ctx = default_context()
    M_T, M_R = _swatches()
    ccm = fit(M_T, M_R)
    rng = np.random.default_rng(11)
    images = (rng.random((2, 211, 333, 3)) ** 2.2 * 0.9).astype(np.float32)
    b, h, w, _ = images.shape
    src = DeviceBuffer.from_array(ctx, images)
    try:
        for kw in (dict(ccm=ccm, gains=[1.1], usm=(1.0, 2.0, 0.0), final_clip=True, out="u16"),
                   dict(ccm=ccm, gains=[0.9], sh=(0.5, -0.5, 7.0), usm=(50.0, 1.0, 2.0), final_clip=True, out="f32_srgb"),
                   dict(ccm=ccm, gains=[1.0], final_clip=True, out="f32_srgb")):
            desc = make_desc(h, w, b, **kw)
            dt = _lib.OUT_DTYPES[desc.out_kind]
            nbytes = images.size * np.dtype(dt).itemsize
            ref, out = DeviceBuffer(ctx, nbytes), DeviceBuffer(ctx, nbytes)
            try:
                ctx.chain_run(desc, src.ptr, ref.ptr)
                ctx.chain_run_static(desc, src.ptr, out.ptr)  # (first launches load code: not part of the claim)
                ctx.sync()
                want = ref.to_array(images.shape, dt)
                zeros = np.zeros(images.shape, dt)
                ctx.upload(out.ptr, zeros)
                with _Capture(hip, ctx.stream) as cap:
                    ctx.chain_run_static(desc, src.ptr, out.ptr)
                assert cap.status == 0 and cap.graph.value, f"capture did not survive the call: {cap.status}"
                ctx.sync()
                assert not out.to_array(images.shape, dt).any(), "a captured call must not have run"
                exe = C.c_void_p()
                assert hip.hipGraphInstantiate(C.byref(exe), cap.graph, None, None, C.c_size_t(0)) == 0
                try:
                    for _ in range(2):
                        ctx.upload(out.ptr, zeros)
                        assert hip.hipGraphLaunch(exe, C.c_void_p(ctx.stream)) == 0
                        ctx.sync()
                        assert np.array_equal(out.to_array(images.shape, dt), want)
                finally:
                    hip.hipGraphExecDestroy(exe)
                    hip.hipGraphDestroy(cap.graph)
            finally:
                ref.free()
                out.free()
    finally:
        src.free()


def test_allocating_entry_point_fails_under_the_same_capture():
    import ctypes as C

    hip = _hip()
    ctx = _lib.Context(0)  # a fresh context: the record of its first chain with a Lab edit still has to be allocated
    # The refused allocation invalidates the capture, and this HIP runtime leaves a stream whose capture was invalidated
    # unusable ("operation failed due to a previous error during capture"): the failing call gets a stream of its own
    # that is thrown away afterwards.
    doomed = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(doomed), 1) == 0  # hipStreamNonBlocking
    try:
        M_T, M_R = _swatches()
        ccm = fit(M_T, M_R)
        images = (np.random.default_rng(12).random((1, 96, 160, 3)) ** 2.2 * 0.9).astype(np.float32)
        desc = make_desc(96, 160, 1, ccm=ccm, gains=[1.1], usm=(50.0, 2.0, 0.0), final_clip=True, out="u16")
        src = DeviceBuffer.from_array(ctx, images)
        dst = DeviceBuffer(ctx, images.size * 2)
        try:
            default_context().chain_run(desc, src.ptr, dst.ptr)  # (code loaded by another context: only the workspace is new)
            default_context().sync()
            ctx.set_stream(doomed.value)
            try:
                with _Capture(hip, doomed.value) as cap:
                    with pytest.raises(_lib.ScanSpaceError) as e:
                        ctx.chain_run(desc, src.ptr, dst.ptr)
            finally:
                ctx.set_stream(None)
            assert e.value.code in (_lib.SS_ERR_HIP, _lib.SS_ERR_NOMEM) and "hipMalloc" in str(e.value)
            # no graph comes out of an invalidated capture
            assert cap.status != 0 or not cap.graph.value
            if cap.graph.value:
                hip.hipGraphDestroy(cap.graph)
            hip.hipStreamDestroy(doomed)
            doomed = None
            # on its own stream the context works as before, and the static entry point captures where ss_chain_run could not
            ctx.chain_run(desc, src.ptr, dst.ptr)
            ctx.sync()
            want = dst.to_array(images.shape, np.uint16)
            assert want.any()
            ctx.upload(dst.ptr, np.zeros_like(want))
            with _Capture(hip, ctx.stream) as cap2:
                ctx.chain_run_static(desc, src.ptr, dst.ptr)
            assert cap2.status == 0 and cap2.graph.value
            exe = C.c_void_p()
            assert hip.hipGraphInstantiate(C.byref(exe), cap2.graph, None, None, C.c_size_t(0)) == 0
            assert hip.hipGraphLaunch(exe, C.c_void_p(ctx.stream)) == 0
            ctx.sync()
            hip.hipGraphExecDestroy(exe)
            hip.hipGraphDestroy(cap2.graph)
            assert np.array_equal(dst.to_array(images.shape, np.uint16), want)
        finally:
            src.free()
            dst.free()
    finally:
        if doomed is not None:
            hip.hipStreamDestroy(doomed)
        ctx.close()
