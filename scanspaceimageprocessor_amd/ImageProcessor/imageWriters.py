"""File output of the worker and the calibration file format, usable without the reference's I/O libraries.

The reference writes its results with imageio (JPEG, PNG; imageProcessorWorker.py:604-610), tifffile
(imageProcessorWorker.py:612-615) and OpenImageIO (EXR, imageProcessorWorker.py:617-662), and exchanges chart
calibrations as ``np.save`` files holding an (N, 3) swatch table (image_space.py:1103-1127, 1254).  This module
uses those libraries when they can be imported and otherwise writes the two lossless containers itself:

* TIFF: baseline TIFF 6.0, little-endian, one uncompressed strip, 8 or 16 bits per sample, chunky RGB -
  the layout ``tifffile.TiffWriter.write`` produces for such arrays (pixel payload identical; tags are the
  baseline set only);
* PNG: 8-bit RGB (or grey), zlib-compressed scanlines with filter type 0.

* OpenEXR: single-part scanline file, 32-bit float channels B, G, R, no compression: what the reference's
  ``ImageSpec(w, h, 3, TypeFloat)`` image holds when the target colour space is the source one ("sRGB", the
  default of the network mode, networkProcessor.py:2755: ``colorconvert("sRGB", "sRGB")`` is the identity).

JPEG needs a real encoder (imageio or Pillow); EXR in any other target colour space needs OpenImageIO's
colour management: without them those cases raise, they are not approximated.
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


def _exr_attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def write_exr(path, arr, colorspace="sRGB"):
    """Uncompressed scanline OpenEXR (version 2) of a (H, W, 3) float image: FLOAT channels B, G, R."""
    img = np.asarray(arr)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError(f"unsupported image for this container: shape {img.shape}")
    img = np.ascontiguousarray(img, dtype="<f4")
    h, w, _ = img.shape
    chlist = b"".join(n + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = b"".join([
        _exr_attr("channels", "chlist", chlist),
        _exr_attr("compression", "compression", b"\0"),
        _exr_attr("dataWindow", "box2i", box),
        _exr_attr("displayWindow", "box2i", box),
        _exr_attr("lineOrder", "lineOrder", b"\0"),
        _exr_attr("oiio:ColorSpace", "string", colorspace.encode()),
        _exr_attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)),
        _exr_attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)),
        _exr_attr("screenWindowWidth", "float", struct.pack("<f", 1.0)),
        b"\0",
    ])
    line_bytes = 3 * w * 4
    first_block = 8 + len(header) + 8 * h
    offsets = first_block + np.arange(h, dtype="<u8") * (8 + line_bytes)
    blocks = np.empty((h, 8 + line_bytes), dtype=np.uint8)
    blocks[:, 0:4] = np.arange(h, dtype="<i4").view(np.uint8).reshape(h, 4)
    blocks[:, 4:8] = np.frombuffer(struct.pack("<i", line_bytes), dtype=np.uint8)
    planar = np.ascontiguousarray(img[:, :, ::-1].transpose(0, 2, 1))  # (H, [B, G, R], W)
    blocks[:, 8:] = planar.view(np.uint8).reshape(h, line_bytes)
    with open(path, "wb") as f:
        f.write(struct.pack("<Ii", 20000630, 2))
        f.write(header)
        f.write(offsets.tobytes())
        f.write(blocks.tobytes())


def read_exr(path):
    """Reader for the files ``write_exr`` makes (uncompressed FLOAT scanlines); used to check them.
    Returns ``(array (H, W, 3) float32 in R, G, B order, attributes)``."""
    raw = open(path, "rb").read()
    magic, version = struct.unpack_from("<Ii", raw, 0)
    if magic != 20000630 or version & 0xFF != 2 or version & ~0xFF:
        raise ValueError("not a single-part scanline OpenEXR file")
    pos, attrs = 8, {}
    while raw[pos] != 0:
        end = raw.index(b"\0", pos)
        name = raw[pos:end].decode()
        pos = end + 1
        end = raw.index(b"\0", pos)
        typ = raw[pos:end].decode()
        pos = end + 1
        (size,) = struct.unpack_from("<i", raw, pos)
        attrs[name] = (typ, raw[pos + 4:pos + 4 + size])
        pos += 4 + size
    pos += 1
    if attrs["compression"][1] != b"\0":
        raise ValueError("compressed OpenEXR")
    names, p, ch = [], 0, attrs["channels"][1]
    while ch[p] != 0:
        end = ch.index(b"\0", p)
        names.append(ch[p:end].decode())
        (ptype,) = struct.unpack_from("<i", ch, end + 1)
        if ptype != 2:
            raise ValueError("only FLOAT channels")
        p = end + 1 + 16
    xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = xmax - xmin + 1, ymax - ymin + 1
    offsets = np.frombuffer(raw, dtype="<u8", count=h, offset=pos)
    out = np.empty((h, w, 3), dtype=np.float32)
    for off in offsets:
        y, size = struct.unpack_from("<ii", raw, int(off))
        line = np.frombuffer(raw, dtype="<f4", count=len(names) * w, offset=int(off) + 8).reshape(len(names), w)
        for k, n in enumerate(names):
            out[y - ymin, :, "RGB".index(n)] = line[k]
    return out, {k: v[1].decode() for k, v in attrs.items() if v[0] == "string"}


def write_image(path, arr, ext, jpeg_quality=100, exr_colorspace=None):
    """The worker's export step (imageProcessorWorker.py:915-920) for '.jpg' / '.jpeg' / '.png' / '.tiff' / '.exr'."""
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
    elif ext == ".exr":
        # export_image_exr (imageProcessorWorker.py:617-662): float image tagged sRGB, converted to the target space
        target = exr_colorspace or "sRGB"
        if target != "sRGB":
            raise NotImplementedError(f"EXR export to colour space {target!r} needs OpenImageIO's colorconvert")
        write_exr(path, arr, "sRGB")
    else:
        raise NotImplementedError(f"no writer for {ext}")


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
