"""File output of the worker and the calibration file format, usable without the reference's I/O libraries.

The reference writes its results with imageio (JPEG, PNG; imageProcessorWorker.py:604-610), tifffile
(imageProcessorWorker.py:612-615) and OpenImageIO (EXR, imageProcessorWorker.py:617-662), and exchanges chart
calibrations as ``np.save`` files holding an (N, 3) swatch table (image_space.py:1103-1127, 1254).  This module
uses those libraries when they can be imported and otherwise writes the two lossless containers itself:

* TIFF: baseline TIFF 6.0, little-endian, one uncompressed strip, 8 or 16 bits per sample, chunky RGB -
  the layout ``tifffile.TiffWriter.write`` produces for such arrays (pixel payload identical; tags are the
  baseline set only);
* PNG: 8-bit RGB (or grey), zlib-compressed scanlines with filter type 0.

JPEG needs a real encoder (imageio or Pillow); EXR in a target colour space needs OpenImageIO: without them
those formats raise, they are not approximated.
"""

from __future__ import annotations

import struct
import zlib

import numpy as np


def _as_image(arr, dtypes):
    arr = np.asarray(arr)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    if arr.ndim != 3 or arr.shape[2] not in (1, 3) or arr.dtype not in dtypes:
        raise ValueError(f"unsupported image for this container: shape {arr.shape}, dtype {arr.dtype}")
    return np.ascontiguousarray(arr)


def write_tiff(path, arr):
    """Uncompressed little-endian baseline TIFF of a (H, W, 3) or (H, W) uint8 / uint16 array."""
    img = _as_image(arr, (np.uint8, np.uint16))
    h, w, c = img.shape
    bits = img.dtype.itemsize * 8
    data = img.astype(img.dtype.newbyteorder("<"), copy=False).tobytes()
    entries = []  # (tag, type, count, value or offset)

    def entry(tag, typ, count, value):
        entries.append((tag, typ, count, value))

    n_entries = 11
    ifd_offset = 8
    extra_offset = ifd_offset + 2 + 12 * n_entries + 4
    extra = b""
    if c == 3:
        bits_offset = extra_offset + len(extra)
        extra += struct.pack("<3H", bits, bits, bits)
    xres_offset = extra_offset + len(extra)
    extra += struct.pack("<2I", 72, 1)
    yres_offset = extra_offset + len(extra)
    extra += struct.pack("<2I", 72, 1)
    data_offset = extra_offset + len(extra)
    SHORT, LONG, RATIONAL = 3, 4, 5
    entry(256, LONG, 1, w)                               # ImageWidth
    entry(257, LONG, 1, h)                               # ImageLength
    entry(258, SHORT, c, bits_offset if c == 3 else bits)  # BitsPerSample
    entry(259, SHORT, 1, 1)                              # Compression: none
    entry(262, SHORT, 1, 2 if c == 3 else 1)             # PhotometricInterpretation: RGB / BlackIsZero
    entry(273, LONG, 1, data_offset)                     # StripOffsets
    entry(277, SHORT, 1, c)                              # SamplesPerPixel
    entry(278, LONG, 1, h)                               # RowsPerStrip
    entry(279, LONG, 1, len(data))                       # StripByteCounts
    entry(282, RATIONAL, 1, xres_offset)                 # XResolution
    entry(283, RATIONAL, 1, yres_offset)                 # YResolution
    assert len(entries) == n_entries
    if data_offset + len(data) >= 2 ** 32:
        raise ValueError("image too large for classic TIFF")
    out = [b"II", struct.pack("<HI", 42, ifd_offset), struct.pack("<H", n_entries)]
    for tag, typ, count, value in entries:
        if typ == SHORT and count == 1:
            out.append(struct.pack("<HHIHH", tag, typ, count, value, 0))
        else:
            out.append(struct.pack("<HHII", tag, typ, count, value))
    out.append(struct.pack("<I", 0))
    out.append(extra)
    with open(path, "wb") as f:
        f.write(b"".join(out))
        f.write(data)


def read_tiff(path):
    """Reader for the files ``write_tiff`` makes (single uncompressed strip); used to check them."""
    raw = open(path, "rb").read()
    if raw[:4] != b"II*\x00":
        raise ValueError("not a little-endian classic TIFF")
    (ifd,) = struct.unpack_from("<I", raw, 4)
    (n,) = struct.unpack_from("<H", raw, ifd)
    tags = {}
    for i in range(n):
        tag, typ, count, value = struct.unpack_from("<HHII", raw, ifd + 2 + 12 * i)
        if typ == 3 and count == 1:
            value &= 0xFFFF
        tags[tag] = (typ, count, value)
    w, h, c = tags[256][2], tags[257][2], tags[277][2]
    bits = struct.unpack_from("<H", raw, tags[258][2])[0] if c == 3 else tags[258][2]
    if tags[259][2] != 1:
        raise ValueError("compressed TIFF")
    dt = np.dtype("<u2") if bits == 16 else np.dtype("u1")
    img = np.frombuffer(raw, dtype=dt, count=h * w * c, offset=tags[273][2]).reshape(h, w, c)
    return img[:, :, 0] if c == 1 else img


def write_png(path, arr, compress_level=6):
    """8-bit PNG of a (H, W, 3) or (H, W) uint8 array."""
    img = _as_image(arr, (np.uint8,))
    h, w, c = img.shape

    def chunk(kind, payload):
        body = kind + payload
        return struct.pack(">I", len(payload)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    rows = np.empty((h, 1 + w * c), dtype=np.uint8)
    rows[:, 0] = 0  # filter type 0 on every scanline
    rows[:, 1:] = img.reshape(h, w * c)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 0, 0, 0, 0)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", ihdr))
        f.write(chunk(b"IDAT", zlib.compress(rows.tobytes(), compress_level)))
        f.write(chunk(b"IEND", b""))


def write_image(path, arr, ext, jpeg_quality=100):
    """The worker's export step (imageProcessorWorker.py:915-920) for '.jpg' / '.jpeg' / '.png' / '.tiff'."""
    ext = ext.lower()
    if ext in (".jpg", ".jpeg"):
        try:
            import imageio

            imageio.imwrite(path, arr, quality=jpeg_quality)
            return
        except ImportError:
            pass
        try:
            from PIL import Image
        except ImportError as e:
            raise RuntimeError("JPEG export needs imageio or Pillow") from e
        Image.fromarray(np.asarray(arr)).save(path, format="JPEG", quality=int(jpeg_quality))
    elif ext == ".png":
        try:
            import imageio

            imageio.imwrite(path, arr)
        except ImportError:
            write_png(path, arr)
    elif ext in (".tiff", ".tif"):
        try:
            from tifffile import tifffile

            with tifffile.TiffWriter(path) as tw:
                tw.write(arr)
        except ImportError:
            write_tiff(path, arr)
    else:
        raise NotImplementedError(f"no writer for {ext}: EXR export goes through OpenImageIO in the reference")


# ---- chart calibration files -----------------------------------------------------------------------------


def save_calibration(path, swatches):
    """``np.save(file_path, self.chart_swatches)`` (image_space.py:1254): an (N, 3) table, dtype kept."""
    table = np.asarray(swatches)
    if table.ndim != 2 or table.shape[1] != 3:
        raise ValueError(f"Invalid chart data shape: {table.shape}. Expected (N, 3)")
    with open(path, "wb") as f:
        np.save(f, table, allow_pickle=False)


def load_calibration(path):
    """``load_default_color_chart`` (image_space.py:1103-1127): the table, validated to be (N, 3).  Pickled
    payloads are refused."""
    table = np.load(path, allow_pickle=False)
    if table.ndim != 2 or table.shape[1] != 3:
        raise ValueError(f"Invalid chart data shape: {table.shape}. Expected (N, 3)")
    return table
