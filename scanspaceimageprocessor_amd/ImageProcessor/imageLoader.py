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
