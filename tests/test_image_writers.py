"""File containers written without the reference's I/O libraries, and the calibration file format."""

import importlib.util

import numpy as np
import pytest

from scanspaceimageprocessor_amd.ImageProcessor import imageWriters as iw


@pytest.mark.parametrize("dtype,shape", [(np.uint8, (37, 53, 3)), (np.uint16, (37, 53, 3)), (np.uint16, (5, 7)), (np.uint8, (1, 1, 3))])
def test_tiff_round_trip_and_pillow_agrees(tmp_path, dtype, shape):
    rng = np.random.default_rng(0)
    arr = rng.integers(0, np.iinfo(dtype).max + 1, size=shape).astype(dtype)
    path = str(tmp_path / "a.tiff")
    iw.write_tiff(path, arr)
    np.testing.assert_array_equal(iw.read_tiff(path), arr)
    try:
        from PIL import Image
    except ImportError:
        return
    with Image.open(path) as im:  # an independent reader accepts the container
        assert im.size == (shape[1], shape[0])
        if dtype == np.uint8 and len(shape) == 3:
            np.testing.assert_array_equal(np.asarray(im), arr)
        if dtype == np.uint16 and len(shape) == 2:
            np.testing.assert_array_equal(np.asarray(im), arr)


def test_png_is_read_back_by_pillow(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(1)
    for shape in [(40, 61, 3), (9, 4)]:
        arr = rng.integers(0, 256, size=shape).astype(np.uint8)
        path = str(tmp_path / "a.png")
        iw.write_png(path, arr)
        with Image.open(path) as im:
            np.testing.assert_array_equal(np.asarray(im), arr)


def test_write_image_dispatch_and_rejections(tmp_path):
    arr = np.zeros((4, 5, 3), np.uint8)
    iw.write_image(str(tmp_path / "x.tiff"), arr, ".tiff")
    iw.write_image(str(tmp_path / "x.png"), arr, ".png")
    iw.write_image(str(tmp_path / "x.exr"), arr.astype(np.float32), ".exr")
    with pytest.raises(NotImplementedError):   # another target colour space needs OpenImageIO's colour management
        iw.write_image(str(tmp_path / "z.exr"), arr.astype(np.float32), ".exr", exr_colorspace="ACEScg")
    with pytest.raises(NotImplementedError):
        iw.write_image(str(tmp_path / "x.bmp"), arr, ".bmp")
    with pytest.raises(ValueError):
        iw.write_tiff(str(tmp_path / "y.tiff"), arr.astype(np.float32))
    with pytest.raises(ValueError):
        iw.write_png(str(tmp_path / "y.png"), arr.astype(np.uint16))


def test_calibration_files(tmp_path):
    table = np.random.default_rng(2).random((24, 3)).astype(np.float32)
    path = str(tmp_path / "chart.npy")
    iw.save_calibration(path, table)
    back = iw.load_calibration(path)
    assert back.dtype == np.float32
    np.testing.assert_array_equal(back, table)
    np.testing.assert_array_equal(np.load(path), table)  # a plain np.save file, as the reference reads it
    with pytest.raises(ValueError):
        iw.save_calibration(path, table[:, :2])
    np.save(str(tmp_path / "bad.npy"), np.zeros((3, 4)))
    with pytest.raises(ValueError):
        iw.load_calibration(str(tmp_path / "bad.npy"))
    np.save(str(tmp_path / "obj.npy"), np.array([{"a": 1}], dtype=object), allow_pickle=True)
    with pytest.raises(ValueError):
        iw.load_calibration(str(tmp_path / "obj.npy"))


def test_exr_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(4)
    arr = (rng.random((7, 11, 3)) * 3.0 - 0.5).astype(np.float32)   # EXR holds unclipped linear data
    arr[0, 0] = [np.float32(0.0), np.float32(1.0), np.float32(65504.0)]
    path = str(tmp_path / "a.exr")
    iw.write_exr(path, arr)
    back, attrs = iw.read_exr(path)
    np.testing.assert_array_equal(back, arr)
    assert attrs["oiio:ColorSpace"] == "sRGB"
    raw = open(path, "rb").read()
    assert raw[:8] == bytes([0x76, 0x2F, 0x31, 0x01, 2, 0, 0, 0])          # magic, version 2, no flags
    # header, one offset per scanline, one block per scanline: y, byte count, B row, G row, R row
    assert len(raw) == raw.index(b"screenWindowWidth") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1 + 8 * 7 + 7 * (8 + 132)
    first = int(np.frombuffer(raw, "<u8", 1, raw.index(b"screenWindowWidth") + 33)[0])
    y, n = np.frombuffer(raw, "<i4", 2, first)
    assert (y, n) == (0, 132)
    np.testing.assert_array_equal(np.frombuffer(raw, "<f4", 11, first + 8), arr[0, :, 2])   # B first
    if importlib.util.find_spec("OpenImageIO"):
        import OpenImageIO as oiio

        np.testing.assert_array_equal(np.asarray(oiio.ImageBuf(path).get_pixels(oiio.FLOAT)), arr)


def test_worker_export_methods_write_the_named_formats(tmp_path):
    """export_image_tiff / png / exr of the worker (reference imageProcessorWorker.py:604-662) by their names."""
    from scanspaceimageprocessor_amd.ImageProcessor import imageProcessorWorker as ipw

    w = ipw.ImageCorrectionWorker(["a.nef"], None, str(tmp_path), None, output_format=".tiff", tiff_bitdepth=16)
    rng = np.random.default_rng(6)
    u16 = rng.integers(0, 65536, size=(9, 13, 3), dtype=np.uint16)
    w.export_image_tiff(u16, str(tmp_path / "a.tiff"))
    np.testing.assert_array_equal(iw.read_tiff(str(tmp_path / "a.tiff")), u16)
    u8 = rng.integers(0, 256, size=(9, 13, 3), dtype=np.uint8)
    w.export_image_png(u8, str(tmp_path / "a.png"))
    assert (tmp_path / "a.png").stat().st_size > 0
    lin = rng.random((9, 13, 3)).astype(np.float32)
    buf, preview = w.export_image_exr(lin, str(tmp_path / "a.exr"))
    assert buf is None and preview.dtype == np.uint8 and preview.shape == lin.shape
    np.testing.assert_array_equal(preview, np.uint8(np.clip(lin * 255.0, 0, 255)))
    back, attrs = iw.read_exr(str(tmp_path / "a.exr"))
    np.testing.assert_array_equal(back, lin)


def test_jpeg_export_needs_nothing_the_image_lacks(tmp_path):
    """'.jpg' through write_image and the worker's export_image_jpg (imageProcessorWorker.py:604-606) with the
    encoder that is installed here (Pillow; imageio is preferred when present, as in the reference): the file decodes
    to the right size, close to the input at quality 100, and a lower quality makes a smaller file."""
    Image = pytest.importorskip("PIL.Image")
    from scanspaceimageprocessor_amd.ImageProcessor import imageProcessorWorker as ipw

    y, x = np.mgrid[0:96, 0:128]
    arr = np.stack([(x * 2) % 256, (y * 2 + x) % 256, (x + y) % 256], axis=-1).astype(np.uint8)
    iw.write_image(str(tmp_path / "a.jpg"), arr, ".jpg", jpeg_quality=100)
    with Image.open(tmp_path / "a.jpg") as im:
        back = np.asarray(im.convert("RGB"))
    assert back.shape == arr.shape
    assert np.mean(np.abs(back.astype(int) - arr.astype(int))) < 6.0
    w = ipw.ImageCorrectionWorker(["a.nef"], None, str(tmp_path), None, output_format=".jpg", jpeg_quality=40)
    w.export_image_jpg(arr, str(tmp_path / "b.jpg"))
    assert 0 < (tmp_path / "b.jpg").stat().st_size < (tmp_path / "a.jpg").stat().st_size
