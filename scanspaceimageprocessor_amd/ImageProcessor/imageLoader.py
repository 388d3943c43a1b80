"""Preview and thumbnail producers of the reference's loader, on the GPU.

The reference shrinks decoded images with ``cv2.resize(..., interpolation=cv2.INTER_AREA)`` in three places:

* ``RawLoadWorker.run`` (imageLoader.py:94-121): the float32 RAW preview, at most 1 MP;
* ``ImageLoader.load_thumbnail_array_from_raw`` fallback (imageLoader.py:298-302): thumbnails that fit
  ``max_size``, the arrays ``calculate_average_exposure`` (image_space.py:4323-4398) reads;
* ``MainWindow._resize_for_preview`` (image_space.py:2959-2986): the edit preview, at most 4 MP.

The functions below keep those size rules and hand the resampling to ``ss_resize_area``.  RAW decoding itself
(rawpy) is not part of this package: callers pass the decoded image, either as the float32 array the reference
forms (``rgb.astype(np.float32) / 65535.0``, imageLoader.py:92) or as the decoder's uint16 array itself, which is
then uploaded as it is (half the bytes) and divided by 65535 on the device, with the same result.
"""

from __future__ import annotations

import numpy as np

from .._lib import default_context

RAW_PREVIEW_PIXELS = 1_000_000   # imageLoader.py:97
EDIT_PREVIEW_PIXELS = 4_000_000  # image_space.py:2971


def preview_size(height, width, target_pixels):
    """None when the image already has at most ``target_pixels`` pixels, else ``(new_width, new_height)``:
    scale = sqrt(target / current), both sides truncated, at least 1 (imageLoader.py:104-111,
    image_space.py:2977-2980)."""
    current = int(height) * int(width)
    if current <= target_pixels:
        return None
    scale = np.sqrt(target_pixels / current)
    return max(1, int(width * scale)), max(1, int(height * scale))


def thumbnail_size(height, width, max_size=(512, 512)):
    """None when no shrink is needed, else ``(new_width, new_height)`` (imageLoader.py:299-302; the reference
    pairs ``max_size[0]`` with the height and ``max_size[1]`` with the width)."""
    scale = min(max_size[0] / height, max_size[1] / width, 1.0)
    if scale >= 1.0:
        return None
    return int(width * scale), int(height * scale)


def _check_image(arr):
    arr = np.asarray(arr)
    if arr.ndim != 3 or arr.shape[2] != 3:
        raise ValueError(f"Invalid RGB array shape: {arr.shape}, expected (H, W, 3)")  # imageLoader.py:88-89
    return arr


def resize_area(arr, dsize, ctx=None):
    """``cv2.resize(arr, dsize, interpolation=cv2.INTER_AREA)`` for float RGB, ``dsize = (new_width, new_height)``,
    shrinking only.  Returns a new float32 array."""
    arr = _check_image(arr)
    new_w, new_h = int(dsize[0]), int(dsize[1])
    if new_w < 1 or new_h < 1:
        raise ValueError(f"Invalid image dimensions: {new_w}x{new_h}")  # cv2 raises on an empty size, too
    ctx = ctx or default_context()
    return ctx.resize_area_host(arr, new_w, new_h)


def _to_float(arr16):
    return np.array(arr16, dtype=np.float32) / np.float32(65535.0)  # imageLoader.py:92


def _resize_area_u16(arr16, dsize, ctx=None):
    """``resize_area(float32(arr16) / 65535, dsize)`` with the conversion on the device."""
    new_w, new_h = int(dsize[0]), int(dsize[1])
    if new_w < 1 or new_h < 1:
        raise ValueError(f"Invalid image dimensions: {new_w}x{new_h}")
    ctx = ctx or default_context()
    arr16 = np.ascontiguousarray(arr16)
    h, w, _ = arr16.shape
    raw = ctx.workspace("loader.raw16", arr16.nbytes)
    full = ctx.workspace("loader.f32", arr16.size * 4)
    small = ctx.workspace("loader.small", new_h * new_w * 12)
    ctx.upload(raw, arr16)
    ctx.unpack_u16(raw, full, arr16.size)
    ctx.resize_area(full, h, w, 1, small, new_h, new_w)
    out = np.empty((new_h, new_w, 3), dtype=np.float32)
    ctx.download(out, small)
    return out


def raw_preview(full_fp, target_pixels=RAW_PREVIEW_PIXELS, ctx=None):
    """``RawLoadWorker.run`` after the decode (imageLoader.py:91-124): float32 (or the decoder's uint16) in,
    shrunk to ``target_pixels`` when larger, NaN / infinity rejected (imageLoader.py:117-118)."""
    full_fp = _check_image(full_fp)
    size = preview_size(full_fp.shape[0], full_fp.shape[1], target_pixels)
    if full_fp.dtype == np.uint16:
        return _to_float(full_fp) if size is None else _resize_area_u16(full_fp, size, ctx)
    full_fp = full_fp.astype(np.float32, copy=False)
    if size is None:
        return full_fp
    if not np.isfinite(full_fp).all():
        raise ValueError("Array contains NaN or infinite values")
    return resize_area(full_fp, size, ctx)


def thumbnail_from_array(arr, max_size=(512, 512), ctx=None):
    """The fallback branch of ``load_thumbnail_array_from_raw`` after the decode (imageLoader.py:296-306)."""
    arr = _check_image(arr)
    size = thumbnail_size(arr.shape[0], arr.shape[1], max_size)
    if arr.dtype == np.uint16:
        return _to_float(arr) if size is None else _resize_area_u16(arr, size, ctx)
    arr = arr.astype(np.float32, copy=False)
    if size is None:
        return arr
    return resize_area(arr, size, ctx)


def resize_for_preview(img_array, max_preview_pixels=EDIT_PREVIEW_PIXELS, ctx=None):
    """``MainWindow._resize_for_preview`` (image_space.py:2959-2986): returns the input object when it is small
    enough."""
    h, w = img_array.shape[:2]
    size = preview_size(h, w, max_preview_pixels)
    if size is None:
        return img_array
    return resize_area(img_array, size, ctx)


RAW_EXTENSIONS = ('.nef', '.cr2', '.cr3', '.dng', '.arw', '.raw')


class ImageLoader:
    """The array-producing half of the reference's ``ImageLoader`` (imageLoader.py:130-306).  File decoding stays
    on the host (Pillow for rendered formats, rawpy for RAW, both looked up when called); negative inversion, colour
    correction, encoding and the area downscale run on the GPU.  Pixmap conversion is Qt's business."""

    @staticmethod
    def load_thumbnail_array(path, max_size=(512, 512), cache=None, chart_swatches=None, correct_thumbnails=False,
                             log_callback=None, negative_mode: bool = False):
        """Thumbnail of an image file as an RGB array, or None (reference imageLoader.py:185-261).

        float32 in 0..1; with ``chart_swatches`` and ``correct_thumbnails`` the colour-corrected, sRGB-encoded
        uint8 array the reference returns in that case.  Kept from the reference: a cache hit short-circuits
        everything; results are stored only in a cache that already holds something (``if cache:``); a failing
        correction is logged and the uncorrected array returned; RAW files Pillow cannot open take
        ``_load_raw_thumbnail``; anything else is logged and yields None."""
        import os

        from .editingTools import ensure_positive_if_negative

        if cache:
            cached = cache.get(path)
            if cached and 'array' in cached and cached['array'] is not None:
                return cached['array']

        def log(msg):
            if log_callback:
                log_callback(msg)

        def remember(arr):
            if cache:
                cache[path] = cache.get(path, {})
                cache[path]['array'] = arr
            return arr

        try:
            from PIL import Image

            with Image.open(path) as img:
                img.thumbnail(max_size)
                arr = np.asarray(img, dtype=np.float32) / 255.0
        except Exception as e:
            if os.path.splitext(path)[1].lower() in RAW_EXTENSIONS:
                try:
                    arr = ImageLoader._load_raw_thumbnail(path, max_size, log_callback)
                    if arr is not None and negative_mode:
                        arr = ensure_positive_if_negative(arr, enabled=True)
                    if arr is not None:
                        remember(arr)
                    return arr
                except Exception as e2:
                    log(f"[Thumb] RAW fallback failed: {e2}")
            log(f"[Thumb] Could not load thumbnail: {e}")
            return None
        if negative_mode:
            arr = ensure_positive_if_negative(arr, enabled=True)
        if chart_swatches is not None and correct_thumbnails:
            try:
                arr = ImageLoader.correct_thumbnail(arr, chart_swatches)
                log("[Thumb] Applied colour correction to thumbnail.")
            except Exception as e:
                log(f"[Thumb] Colour correction on thumbnail failed: {e}")
        return remember(arr)

    @staticmethod
    def correct_thumbnail(arr, chart_swatches, reference=None):
        """``np.uint8(255 * cctf_encoding(clip(colour_correction(arr, chart_swatches, reference), 0, 1)))``
        (imageLoader.py:226-228) as one fused pass: 3-term Cheung fit on the host, matrix, clip, sRGB encode and
        truncation to 8 bits on the device."""
        from ..chain import make_desc
        from ..colour_fit import matrix_colour_correction
        from . import imageProcessorWorker as ipw

        reference = ipw.reference_swatches if reference is None else reference
        if reference is None:
            raise RuntimeError("no reference swatch table: call set_reference_swatches first")
        arr = _check_image(arr)
        ccm = matrix_colour_correction(np.asarray(chart_swatches), np.asarray(reference))
        h, w, _ = arr.shape
        desc = make_desc(h, w, 1, ccm=ccm, final_clip=True, out="u8")
        out, _ = default_context().chain_run_host(desc, np.ascontiguousarray(arr, dtype=np.float32))
        return out

    @staticmethod
    def _load_raw_thumbnail(path, max_size, log_callback=None):
        """Reference imageLoader.py:264-306: the embedded JPEG when the RAW file has one, else an 8-bit decode shrunk
        (area average, on the GPU) to fit ``max_size``."""
        import io

        import rawpy  # host decode stays on LibRaw
        from PIL import Image

        def log(msg):
            if log_callback:
                log_callback(msg)

        with rawpy.imread(path) as raw:
            try:
                thumb = raw.extract_thumb()
                if thumb.format == rawpy.ThumbFormat.JPEG:
                    pil = Image.open(io.BytesIO(thumb.data))
                    pil.thumbnail(max_size)
                    log("[Thumb] Loaded embedded JPEG thumbnail from RAW.")
                    return np.asarray(pil, dtype=np.float32) / 255.0
                log("[Thumb] No embedded JPEG, using raw postprocess.")
            except Exception:
                log("[Thumb] No embedded thumbnail, using raw postprocess.")
            rgb = raw.postprocess(output_bps=8, no_auto_bright=True)
        arr = thumbnail_from_array(rgb.astype(np.float32) / 255.0, max_size)
        log("[Thumb] Loaded thumbnail from rawpy postprocess.")
        return arr
