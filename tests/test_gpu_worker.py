"""The worker mirror end to end on the GPU: ``process_array`` / ``run`` against the oracle's restatement of
ImageCorrectionWorker.run steps 3-7 (imageProcessorWorker.py:839-923), masking included."""

import numpy as np
import pytest

from oracle import chain as oc
from oracle import inputs
from oracle import masking as om
from scanspaceimageprocessor_amd.ImageProcessor import imageProcessorWorker as ipw

pytestmark = pytest.mark.gpu


def scene(seed, h, w):
    """Bright object with a dark surround and a blown patch: gives the masking something to do."""
    rng = np.random.default_rng(seed)
    img = (0.004 * rng.random((h, w, 3))).astype(np.float32)
    yy, xx = np.ogrid[:h, :w]
    obj = ((yy - h / 2) / (0.35 * h)) ** 2 + ((xx - w / 2) / (0.3 * w)) ** 2 <= 1.0
    img[obj] = (0.1 + 0.6 * rng.random((int(obj.sum()), 3))).astype(np.float32)
    img[h // 2 - 8:h // 2 + 8, w // 2 - 10:w // 2 + 10] = 2.0   # clips to 1 after the correction
    img[5:8, 5:8] = 0.5                                          # small bright island in the dark surround
    return img


@pytest.mark.parametrize("fmt,depth", [(".tiff", 16), (".jpg", 8), (".exr", 8)])
@pytest.mark.parametrize("masked", [False, True])
def test_chart_mode_process_array(fmt, depth, masked):
    M_T, M_R = inputs.synthetic_swatches()
    ipw.set_reference_swatches(M_R)
    logs = []
    sig = ipw.WorkerSignals()
    sig.log.connect(logs.append)
    w = ipw.ImageCorrectionWorker(["a.nef"], M_T, "/tmp", sig, output_format=fmt, tiff_bitdepth=depth,
                                  export_masked=masked, exposure_adj=0.2, shadow_adj=10.0)
    w.image_metadata_map = {"a.nef": {"average_exposure": "1.1"}}
    # (limits well away from any pixel's luminance: the lifted surround sits at 0.0600000x, where float32 and
    # float64 data fall on different sides of a limit of 0.06)
    w.shadow_limit, w.highlight_limit = 0.065, 0.96
    img = scene(3, 150, 210)
    encoded, preview = w.process_array(img, "a.nef")

    c = oc.worker_chart_path(img, M_T, M_R, multiplier=1.1, exposure_adj=0.2, shadow_adj=10.0)
    if masked:
        c, n = om.apply_masking(c, 0.065, 0.96, 200)
        assert n > 200
        assert any(f"Masked {n} pixels (min area=200) with limits shadow<=0.07, highlight>=0.96" in m for m in logs)
    want = oc.encode_image(c, fmt, depth)
    if fmt == ".exr":
        assert encoded.dtype == np.float32 and np.max(np.abs(encoded - want)) <= 1e-5
    else:
        assert encoded.dtype == want.dtype
        assert np.max(np.abs(encoded.astype(np.int64) - want.astype(np.int64))) <= 1
    want_preview = want if want.dtype == np.uint8 else np.uint8(np.clip(c * 255.0, 0, 255))
    assert np.max(np.abs(preview.astype(np.int64) - want_preview.astype(np.int64))) <= 1
    if masked:  # masked pixels are exactly zero, in the output and in the preview
        m = (c == 0).all(axis=2)
        assert (encoded[m] == 0).all() and (preview[m] == 0).all()


def test_run_in_no_chart_negative_mode_writes_and_signals(tmp_path):
    M_T, M_R = inputs.synthetic_swatches()
    neg = inputs.negative_image(9, 140, 200)
    written, previews, states = {}, [], []
    sig = ipw.WorkerSignals()
    sig.preview.connect(lambda d: previews.append(d))
    sig.status.connect(lambda *a: states.append(a[1] if len(a) > 1 else a[0]))
    w = ipw.ImageCorrectionWorker(["n1.nef", "n2.nef"], M_T, str(tmp_path), sig, output_format=".tiff", tiff_bitdepth=16,
                                  use_chart=False, negative_mode=True, sharpen_amount=35.0, highlight_adj=-0.2)
    w.raw_loader = lambda path: neg if path == "n1.nef" else neg[::-1].copy()
    w.image_writer = lambda out_path, arr, ext: written.__setitem__(out_path, arr.copy())
    w.run()
    # two images deep: image 2 is being decoded while image 1 is on the GPU
    assert states == ["started", "started", "finished", "finished"]
    assert len(written) == 2 and len(previews) == 2
    for path, src in (("n1", neg), ("n2", neg[::-1])):
        got = written[str(tmp_path / (path + ".tiff"))]
        c = oc.worker_nochart_path(src, negative_mode=True, highlight_adj=-0.2, sharpen_amount=35.0)
        want = oc.encode_image(c, ".tiff", 16)
        assert np.max(np.abs(got.astype(np.int64) - want.astype(np.int64))) <= 1


def test_host_array_masking_entry_point():
    M_T, _ = inputs.synthetic_swatches()
    w = ipw.ImageCorrectionWorker([], M_T, "/tmp", None)
    img = scene(4, 120, 160)
    img = np.clip(img, 0, 1)
    want, n = om.apply_masking(img)
    out = w._apply_masking(img)
    assert out is img  # float32 arrays are modified in place, as the reference does with its array
    np.testing.assert_array_equal(out, want)


def test_run_writes_files_with_the_built_in_containers(tmp_path):
    from scanspaceimageprocessor_amd.ImageProcessor import imageWriters as iw

    M_T, M_R = inputs.synthetic_swatches()
    ipw.set_reference_swatches(M_R)
    img = inputs.linear_image(12, 90, 130)
    for fmt, depth in ((".tiff", 16), (".tiff", 8), (".png", 8)):
        out_dir = tmp_path / f"{fmt[1:]}{depth}"
        out_dir.mkdir()
        w = ipw.ImageCorrectionWorker(["shot.nef"], M_T, str(out_dir), None, output_format=fmt, tiff_bitdepth=depth)
        w.raw_loader = lambda path: img
        states = []
        w.signals.status.connect(lambda *a: states.append(a[1]))
        w.run()
        assert states == ["started", "finished"]
        want = oc.encode_image(oc.worker_chart_path(img, M_T, M_R), fmt, depth)
        path = str(out_dir / ("shot" + fmt))
        if fmt == ".tiff":
            got = iw.read_tiff(path)
        else:
            Image = pytest.importorskip("PIL.Image")
            with Image.open(path) as im:
                got = np.asarray(im)
        assert got.dtype == want.dtype
        assert np.max(np.abs(got.astype(np.int64) - want.astype(np.int64))) <= 1
    # EXR: the float image itself (linear, clipped), tagged sRGB; another target colour space is refused loudly
    out_dir = tmp_path / "exr"
    out_dir.mkdir()
    w = ipw.ImageCorrectionWorker(["shot.nef"], M_T, str(out_dir), None, output_format=".exr", exr_colorspace="sRGB")
    w.raw_loader = lambda path: img
    w.run()
    got, attrs = iw.read_exr(str(out_dir / "shot.exr"))
    assert attrs["oiio:ColorSpace"] == "sRGB"
    assert np.max(np.abs(got - oc.worker_chart_path(img, M_T, M_R))) <= 1e-5
    w = ipw.ImageCorrectionWorker(["shot.nef"], M_T, str(out_dir), None, output_format=".exr", exr_colorspace="ACEScg")
    w.raw_loader = lambda path: img
    states = []
    w.signals.status.connect(lambda *a: states.append(a[1]))
    w.run()
    assert states == ["started", "error"]


def test_unpack_u16_is_numpys_division_bit_for_bit():
    from gpu_util import DeviceBuffer
    from scanspaceimageprocessor_amd import default_context

    ctx = default_context()
    for n in (65536, 65536 + 13, 5):
        vals = np.arange(n, dtype=np.uint64).astype(np.uint16) if n >= 65536 else np.array([0, 1, 65535, 32768, 7], np.uint16)
        src = DeviceBuffer.from_array(ctx, vals)
        dst = DeviceBuffer(ctx, 4 * n)
        try:
            ctx.unpack_u16(src.ptr, dst.ptr, n)
            ctx.sync()
            got = dst.to_array((n,), np.float32)
        finally:
            src.free(); dst.free()
        want = np.array(vals, dtype=np.float32) / 65535.0   # imageProcessorWorker.py:313
        assert want.dtype == np.float32
        np.testing.assert_array_equal(got, want)


def test_worker_takes_the_decoders_uint16_output():
    from oracle import inputs
    from scanspaceimageprocessor_amd.ImageProcessor import imageProcessorWorker as ipw

    M_T, M_R = inputs.synthetic_swatches()
    ipw.set_reference_swatches(M_R)
    rng = np.random.default_rng(5)
    raw16 = (rng.random((96, 130, 3)) ** 2.2 * 60000).astype(np.uint16)
    as_float = np.array(raw16, dtype=np.float32) / 65535.0
    for fmt, depth in ((".tiff", 16), (".jpg", 8)):
        w = ipw.ImageCorrectionWorker(["a.nef"], M_T, "/tmp", None, output_format=fmt, tiff_bitdepth=depth,
                                      sharpen_amount=40.0)
        e16, p16 = w.process_array(raw16, "a.nef")
        ef, pf = w.process_array(as_float, "a.nef")
        np.testing.assert_array_equal(e16, ef)
        np.testing.assert_array_equal(p16, pf)


def test_page_locked_host_buffers_are_copied_directly_with_the_same_result():
    from gpu_util import DeviceBuffer
    from scanspaceimageprocessor_amd import default_context
    from scanspaceimageprocessor_amd.chain import make_desc

    ctx = default_context()
    rng = np.random.default_rng(3)
    data = rng.random((600, 800, 3), dtype=np.float32)          # 5.8 MB: above the direct-copy threshold
    pinned_in = ctx.pinned_empty(data.shape, np.float32)
    pinned_in[...] = data
    pinned_out = ctx.pinned_empty(data.shape, np.float32)
    buf = DeviceBuffer(ctx, data.nbytes)
    try:
        ctx.upload(buf.ptr, pinned_in)
        ctx.download(pinned_out, buf.ptr)
        np.testing.assert_array_equal(pinned_out, data)
        back = np.empty_like(data)
        ctx.download(back, buf.ptr)                              # pageable destination: staged path
        np.testing.assert_array_equal(back, data)
    finally:
        buf.free()
    # the host entry point of the chain, pinned source against pageable source
    desc = make_desc(600, 800, 1, ccm=(np.eye(3) * 0.9 + 0.03, 1, 3), gains=[1.1], usm=(40.0, 1.0, 0.0), final_clip=True,
                     out="u16")
    a, _ = ctx.chain_run_host(desc, pinned_in)
    b, _ = ctx.chain_run_host(desc, data)
    np.testing.assert_array_equal(a, b)
    del pinned_in, pinned_out


def test_saved_job_runs_end_to_end(tmp_path):
    from scanspaceimageprocessor_amd.ImageProcessor import imageWriters as iw
    from scanspaceimageprocessor_amd.ImageProcessor.jobs import ProcessingJob, worker_from_job

    M_T, M_R = inputs.synthetic_swatches()
    ipw.set_reference_swatches(M_R)
    raw16 = (inputs.linear_image(31, 80, 120) * 65535.0).astype(np.uint16)
    out_path = str(tmp_path / "out" / "a.tiff")
    (tmp_path / "out").mkdir()
    text = ProcessingJob("job-1", "/data/a.nef", out_path, M_T.tolist(),
                         {"output_format": ".tiff", "tiff_bitdepth": 16, "exposure_adj": 0.25}).to_json()
    states = []
    sig = ipw.WorkerSignals()
    sig.status.connect(lambda *a: states.append(a))
    w = worker_from_job(ProcessingJob.from_json(text), sig)
    w.raw_loader = lambda path: raw16                       # the decoder's 16-bit image
    w.run()
    assert states[0] == ("started",) and states[-1][1] == "finished" and states[-1][3] == out_path
    img = np.array(raw16, dtype=np.float32) / 65535.0
    want = oc.encode_image(oc.worker_chart_path(img, M_T, M_R, exposure_adj=0.25), ".tiff", 16)
    got = iw.read_tiff(out_path)
    assert np.max(np.abs(got.astype(np.int64) - want.astype(np.int64))) <= 1


def test_reused_output_buffers_give_the_same_images():
    M_T, M_R = inputs.synthetic_swatches()
    ipw.set_reference_swatches(M_R)
    imgs = [inputs.linear_image(40 + i, 70, 110) for i in range(3)]
    plain = ipw.ImageCorrectionWorker([], M_T, "/tmp", None, output_format=".tiff", tiff_bitdepth=16)
    pooled = ipw.ImageCorrectionWorker([], M_T, "/tmp", None, output_format=".tiff", tiff_bitdepth=16)
    pooled.reuse_output_buffers = True
    seen = set()
    for img in imgs:
        e0, p0 = plain.process_array(img, "a.nef")
        e1, p1 = pooled.process_array(img, "a.nef")
        np.testing.assert_array_equal(e0, e1)
        np.testing.assert_array_equal(p0, p1)
        seen.add(e1.__array_interface__["data"][0])
    assert len(seen) == 1      # the same page-locked memory every time
    big = inputs.linear_image(50, 90, 130)   # a larger image grows the buffers
    e1, p1 = pooled.process_array(big, "a.nef")
    e0, p0 = plain.process_array(big, "a.nef")
    np.testing.assert_array_equal(e0, e1)
    np.testing.assert_array_equal(p0, p1)


def test_batches_in_page_locked_memory_are_pipelined_image_by_image():
    """A batch handed over in page-locked buffers goes through upload / kernels / download image by image on
    three streams; the result must be bit for bit what one call per image gives (pageable buffers, one image each:
    the unpipelined path), automatic negative percentiles and sharpening included."""
    from scanspaceimageprocessor_amd import default_context
    from scanspaceimageprocessor_amd.chain import make_desc

    ctx = default_context()
    b, h, w = 5, 520, 700   # 4.4 MB per image
    imgs = np.stack([inputs.GENERATORS["negative"](200 + i, h, w) for i in range(b)])
    M_T, M_R = inputs.synthetic_swatches()
    from scanspaceimageprocessor_amd.colour_fit import matrix_colour_correction

    ccm = matrix_colour_correction(M_T, M_R)
    for kw, out_kind, dtype in ((dict(neg=True, ccm=ccm, gains=[1.1], usm=(40.0, 2.0, 0.0), final_clip=True), "u16", np.uint16),
                                (dict(ccm=ccm, gains=[0.9], sh=(0.3, -0.4, 7.0), final_clip=True), "f32", np.float32)):
        neg_auto = bool(kw.get("neg"))
        pin_in = ctx.pinned_empty(imgs.shape, np.float32)
        pin_in[...] = imgs
        pin_out = ctx.pinned_empty(imgs.shape, dtype)
        pin_out[...] = 0
        got, _ = ctx.chain_run_host(make_desc(h, w, b, out=out_kind, **kw), pin_in, neg_auto=neg_auto, out=pin_out)
        assert got is pin_out
        for i in range(b):
            want, _ = ctx.chain_run_host(make_desc(h, w, 1, out=out_kind, **kw), imgs[i], neg_auto=neg_auto)
            np.testing.assert_array_equal(got[i], want)
        # and the same batch from pageable memory (whole-batch upload, one launch, whole-batch download)
        whole, _ = ctx.chain_run_host(make_desc(h, w, b, out=out_kind, **kw), imgs, neg_auto=neg_auto)
        np.testing.assert_array_equal(whole, got)


def test_asynchronous_copies_of_page_locked_buffers_run_in_stream_order():
    """upload_async -> chain_run -> download_async -> sync: nothing stalls the host in between, the copies are
    ordered with the kernels on the context's stream; pageable memory is refused."""
    from scanspaceimageprocessor_amd import _lib, default_context
    from scanspaceimageprocessor_amd.chain import make_desc

    ctx = default_context()
    h, w = 300, 500
    img = inputs.linear_image(321, h, w)
    pin_in = ctx.pinned_empty(img.shape, np.float32)
    pin_in[...] = img
    pin_out = ctx.pinned_empty(img.shape, np.uint16)
    pin_out[...] = 0
    src, dst = ctx.malloc(img.nbytes), ctx.malloc(pin_out.nbytes)
    try:
        desc = make_desc(h, w, 1, gains=[1.2], final_clip=True, out="u16")
        ctx.upload_async(src, pin_in)
        ctx.chain_run(desc, src, dst)
        ctx.download_async(pin_out, dst)
        ctx.sync()
        want, _ = ctx.chain_run_host(desc, img)
        np.testing.assert_array_equal(pin_out, want)
        with pytest.raises(_lib.ScanSpaceError) as e:
            ctx.upload_async(src, img)            # numpy's own (pageable) memory
        assert e.value.code == _lib.SS_ERR_INVALID
        with pytest.raises(_lib.ScanSpaceError):
            ctx.download_async(np.empty_like(want), dst)
        ctx.sync()
    finally:
        ctx.free(src)
        ctx.free(dst)


def _run_worker(depth, images, loader, **kw):
    M_T, M_R = inputs.synthetic_swatches()
    ipw.set_reference_swatches(M_R)
    written, previews, events, logs = {}, {}, [], []
    sig = ipw.WorkerSignals()
    sig.preview.connect(lambda d: previews.__setitem__(d[1], d[0].copy()))
    sig.status.connect(lambda *a: events.append((a[0], a[1])))
    sig.log.connect(logs.append)
    w = ipw.ImageCorrectionWorker(images, M_T, "/tmp/out", sig, output_format=".tiff", tiff_bitdepth=16, **kw)
    w.pipeline_depth = depth
    w.raw_loader = loader
    w.image_writer = lambda out_path, arr, ext: written.__setitem__(out_path, arr.copy())
    w.image_metadata_map = {p: {"average_exposure": str(1.0 + 0.05 * i)} for i, p in enumerate(images)}
    w.run()
    return written, previews, events, logs


def test_pipelined_run_is_the_sequential_run_bit_for_bit():
    """Five images of two sizes (uint16 from the 'decoder' and float32), two deep against one after the other: the
    same files and previews, every image's own signals in order, 'finished' in image order."""
    rng = np.random.default_rng(21)
    data = {}
    for i in range(5):
        h, w = (120, 176) if i % 2 == 0 else (97, 150)
        img = inputs.linear_image(30 + i, h, w)
        data[f"img{i}.nef"] = (img * 65535.0).astype(np.uint16) if i != 3 else img
    images = list(data)
    loader = lambda path: data[path]
    seq = _run_worker(1, images, loader, shadow_adj=10.0, exposure_adj=0.1)
    pipe = _run_worker(2, images, loader, shadow_adj=10.0, exposure_adj=0.1)
    assert list(seq[0]) == list(pipe[0]) == [f"/tmp/out/img{i}.tiff" for i in range(5)]
    for path in seq[0]:
        np.testing.assert_array_equal(seq[0][path], pipe[0][path])
        np.testing.assert_array_equal(seq[1][path], pipe[1][path])
    assert [e for e in seq[2]] == [(p, s) for p in images for s in ("started", "finished")]
    for p in images:  # each image: started, then finished
        own = [s for q, s in pipe[2] if q == p]
        assert own == ["started", "finished"]
    assert [q for q, s in pipe[2] if s == "finished"] == images
    # the loop really was two deep: the second image started before the first finished
    assert pipe[2].index((images[1], "started")) < pipe[2].index((images[0], "finished"))


def test_pipelined_run_survives_a_bad_image_in_the_middle():
    """imageProcessorWorker.py:950-970: an image that cannot be loaded, and one that cannot be processed, are
    reported as 'error' and the rest of the batch finishes."""
    good = (inputs.linear_image(40, 80, 120) * 65535.0).astype(np.uint16)

    def loader(path):
        if path == "broken.nef":
            raise IOError("truncated file")
        if path == "flat.nef":
            return np.zeros((80, 120), np.uint16)  # not (H, W, 3): the upload refuses it
        return good

    images = ["a.nef", "broken.nef", "b.nef", "flat.nef", "c.nef"]
    written, previews, events, logs = _run_worker(2, images, loader)
    assert sorted(written) == ["/tmp/out/a.tiff", "/tmp/out/b.tiff", "/tmp/out/c.tiff"]
    final = {p: [s for q, s in events if q == p][-1] for p in images}
    assert final == {"a.nef": "finished", "broken.nef": "error", "b.nef": "finished", "flat.nef": "error",
                     "c.nef": "finished"}
    assert any("[Image Load Error] broken.nef: truncated file" in m for m in logs)
    # the same batch one after the other: the same outcome, the same files
    written1, _, events1, _ = _run_worker(1, images, loader)
    assert {p: [s for q, s in events1 if q == p][-1] for p in images} == final
    assert sorted(written1) == sorted(written)
    for path in written:
        np.testing.assert_array_equal(written[path], written1[path])
