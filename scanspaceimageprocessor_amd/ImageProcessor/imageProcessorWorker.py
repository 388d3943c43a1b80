"""Per-image worker of the correction chain: the counterpart of the reference's
``ImageProcessor/imageProcessorWorker.py`` (``ImageCorrectionWorker``, lines 203-977) with the
pixel work done by the gfx950 kernels of libscanspace_hip.so.

Kept from the reference: constructor arguments and defaults, the method names ``load_image``,
``apply_colour_correction``, ``apply_image_adjustments``, ``encode_image``, ``run``; the order
load -> [negative inversion] -> CCM -> exposure multiplier -> clip -> adjustments -> clip -> encode;
the three signals ``log(str)``, ``status(path, state, seconds, out_path)`` (one-argument
``'started'`` form in network mode) and ``preview([uint8_array, out_path])``; and the rule that a
failing image is reported as ``'error'`` and the loop continues.

Different by design: ``run`` keeps each image on the GPU between its steps (one upload, one fused
kernel pass per output, downloads of the encoded result and of the 8-bit preview only).  RAW
decoding (rawpy) and file encoding (imageio / tifffile) stay on the host and are looked up at
call time; metadata embedding and path schemas are outside this package.
"""

from __future__ import annotations

import os
import time

import numpy as np

from .. import _lib
from ..chain import make_desc
from ..colour_fit import matrix_colour_correction
from .._lib import default_context
from . import editingTools
from .editingTools import apply_all_adjustments, ensure_positive_if_negative

RAW_EXTENSIONS = ('.nef', '.cr2', '.cr3', '.dng', '.arw', '.raw')
OUTPUT_FORMATS = ('.jpg', '.png', '.tiff', '.exr')


def _default_reference_swatches():
    """Linear-sRGB values of the 'ColorChecker24 - After November 2014' chart, CAT02-adapted to D65
    (reference imageProcessorWorker.py:116-125).  The table lives in colour-science; without that
    package the caller supplies it through ``set_reference_swatches``."""
    try:
        import colour  # type: ignore
        from colour import RGB_COLOURSPACES  # type: ignore
    except Exception:
        return None
    chart = colour.CCS_COLOURCHECKERS['ColorChecker24 - After November 2014']
    patches_xyz = colour.xyY_to_XYZ(np.array([xyY for xyY in chart.data.values()], dtype=np.float64))
    linear_srgb = colour.XYZ_to_RGB(patches_xyz, RGB_COLOURSPACES['sRGB'], illuminant=chart.illuminant,
                                    chromatic_adaptation_transform="CAT02", apply_cctf_encoding=False)
    return np.asarray(linear_srgb, dtype=np.float64)


reference_swatches = _default_reference_swatches()


def set_reference_swatches(swatches) -> None:
    """Install the 24x3 reference (target) swatch table used by ``apply_colour_correction``."""
    global reference_swatches
    swatches = np.asarray(swatches, dtype=np.float64)
    if swatches.ndim != 2 or swatches.shape[1] != 3:
        raise ValueError(f"reference swatches must be (N, 3), got {swatches.shape}")
    reference_swatches = swatches


class _Signal:
    """Minimal stand-in for a Qt signal: ``connect(fn)`` / ``emit(*args)``."""

    def __init__(self):
        self._slots = []

    def connect(self, fn):
        self._slots.append(fn)

    def emit(self, *args):
        for fn in list(self._slots):
            fn(*args)


class WorkerSignals:
    """Qt-free signals object (the reference ships one for its network mode, networkProcessor.py:3471)."""

    def __init__(self):
        self.log = _Signal()
        self.status = _Signal()
        self.preview = _Signal()


class ImageCorrectionWorker:
    def __init__(self, images, swatches, output_folder, signals, jpeg_quality=100,
                 rename_map=None, name_base='', padding=0, export_masked=False,
                 output_format: str = '.jpg', tiff_bitdepth=8, exr_colorspace=None,
                 export_schema: str = "", use_export_schema: bool = True,
                 custom_name: str = "", root_folder: str = "", from_network=False,
                 network_output_path: str = "", group_name: str = "",
                 use_chart: bool = True, exposure_adj: float = 0.0,
                 shadow_adj: float = 0.0, highlight_adj: float = 0.0,
                 white_balance_adj: int = 5500, denoise_strength: float = 0.0,
                 sharpen_amount: float = 0.0,
                 negative_mode: bool = False,
                 negative_low_pct: float = 0.05,
                 negative_high_pct: float = 0.995):
        self.images = images
        self.swatches = swatches
        self.output_folder = output_folder
        self.signals = signals if signals is not None else WorkerSignals()
        self.jpeg_quality = jpeg_quality
        self.output_format = output_format
        self.tiff_bitdepth = tiff_bitdepth
        self.exr_colorspace = exr_colorspace
        self.rename_map = rename_map or {}
        self.name_base = name_base
        self.padding = padding
        self.export_masked = export_masked
        self.cancelled = False
        self.export_schema = export_schema
        self.use_export_schema = use_export_schema
        self.custom_name = custom_name
        self.root_folder = root_folder
        self.from_network = from_network
        self.network_output_path = network_output_path
        self.group_name = group_name
        self.use_chart = use_chart
        self.exposure_adj = exposure_adj
        self.shadow_adj = shadow_adj
        self.highlight_adj = highlight_adj
        self.white_balance_adj = white_balance_adj
        self.denoise_strength = denoise_strength
        self.sharpen_amount = sharpen_amount
        self.negative_mode = negative_mode
        self.negative_low_pct = negative_low_pct
        self.negative_high_pct = negative_high_pct
        # host-side hooks (defaults look the libraries up when first needed)
        self.raw_loader = None     # callable(path) -> (H,W,3) float32 in [0,1]
        self.image_writer = None   # callable(path, array, fmt)
        # True: `process_array` returns views of two page-locked arrays that the thread's context keeps and
        # overwrites with the next image (no allocation, no staging copy: 11 -> 7 ms per 24 MP image).  For
        # headless runs whose writer and preview consumer are done with an image before the next one starts.
        self.reuse_output_buffers = False
        # Images in flight in `run`: 2 = while image i is on the GPU, a helper thread decodes and uploads image i + 1 and
        # another writes image i - 1 (two contexts, i.e. two streams: the upload of one image and the download of another
        # share the link, one direction each).  1 = the reference's strictly sequential loop.
        self.pipeline_depth = 2
        self._ccm = None

    def cancel(self):
        self.cancelled = True

    # ------------------------------------------------------------------ reference method surface

    def load_image(self, img_path: str) -> np.ndarray:
        """RAW -> float32 linear sRGB in [0,1] on the host (rawpy/LibRaw), then the optional
        negative inversion (reference imageProcessorWorker.py:291-325)."""
        raw_img_float = self._decode_raw(img_path)
        if raw_img_float.dtype == np.uint16:
            raw_img_float = np.array(raw_img_float, dtype=np.float32) / np.float32(65535.0)
        return ensure_positive_if_negative(raw_img_float, enabled=self.negative_mode,
                                           low_percentile=self.negative_low_pct,
                                           high_percentile=self.negative_high_pct)

    def apply_colour_correction(self, img_array: np.ndarray) -> np.ndarray:
        """``colour.colour_correction(img, swatches, reference_swatches)`` with the defaults
        (reference imageProcessorWorker.py:327-338): unclipped float output."""
        h, w, _ = img_array.shape
        desc = make_desc(h, w, 1, ccm=self._fit(), out="f32")
        out, _ = default_context().chain_run_host(desc, img_array)
        return out

    def apply_image_adjustments(self, img_array: np.ndarray, img_path: str) -> np.ndarray:
        """No-chart mode (reference imageProcessorWorker.py:340-370): sharpen radius 1.0, threshold 0."""
        current_wb = self._extract_white_balance_from_exif(img_path)
        return apply_all_adjustments(
            img_array, exposure=self.exposure_adj, shadows=self.shadow_adj, highlights=self.highlight_adj,
            current_wb=current_wb, target_wb=self.white_balance_adj, wb_tint=0.0,
            denoise_strength=self.denoise_strength, sharpen_amount=self.sharpen_amount,
            sharpen_radius=1.0, sharpen_threshold=0.0)

    def encode_image(self, img_array: np.ndarray, output_format: str) -> np.ndarray:
        """sRGB encoding + truncating quantisation (reference imageProcessorWorker.py:579-602)."""
        kind = self._out_kind(output_format)
        if kind == _lib.SS_OUT_F32_LINEAR:
            return img_array
        h, w, _ = img_array.shape
        out, _ = default_context().chain_run_host(make_desc(h, w, 1, out=kind), img_array)
        return out

    def _has_non_default_adjustments(self) -> bool:
        return (self.exposure_adj != 0.0 or self.shadow_adj != 0.0 or self.highlight_adj != 0.0
                or self.white_balance_adj != 5500 or self.denoise_strength != 0.0 or self.sharpen_amount != 0.0)

    RAW_EXTENSIONS = ('.nef', '.cr2', '.cr3', '.dng', '.arw', '.raw')  # imageProcessorWorker.py:411

    def _extract_white_balance_from_exif(self, img_path: str) -> float:
        """As-shot colour temperature of an image (imageProcessorWorker.py:397-427): the camera's white-balance
        multipliers through LibRaw for RAW files, then an EXIF temperature through OpenImageIO, then daylight -
        each outcome logged as the reference logs it.  A caller that knows the value already puts it in
        ``image_metadata_map[path]['current_wb']`` and no file is opened."""
        meta = getattr(self, "image_metadata_map", {}).get(img_path) or {}
        if meta.get('current_wb') is not None:
            try:
                wb = float(meta['current_wb'])
                self.signals.log.emit(f"[WB] Using white balance from the image metadata map: {wb}K")
                return wb
            except (TypeError, ValueError):
                self.signals.log.emit(f"[WB] Ignoring unusable current_wb {meta['current_wb']!r} for {img_path}")
        cache = self.__dict__.setdefault("_wb_by_path", {})  # one look per file, however often the stages are asked for
        if img_path not in cache:
            cache[img_path] = self._read_white_balance(img_path)
        return cache[img_path]

    def _read_white_balance(self, img_path: str) -> float:
        default_wb = 5500.0
        try:
            if img_path.lower().endswith(self.RAW_EXTENSIONS):
                wb = self._get_wb_from_rawpy(img_path)
                if wb is not None:
                    self.signals.log.emit(f"[WB] Found white balance from rawpy: {wb}K")
                    return wb
            wb = self._get_wb_from_oiio_exif(img_path)
            if wb is not None:
                self.signals.log.emit(f"[WB] Found white balance from EXIF: {wb}K")
                return wb
        except Exception as e:
            self.signals.log.emit(f"[WB] Error reading white balance from {img_path}: {e}")
        # (a white-balance target other than 5500 K now shifts the image relative to a daylight source: say so)
        self.signals.log.emit(f"[WB] Using default white balance: {default_wb}K")
        return default_wb

    @staticmethod
    def wb_temperature_from_multipliers(wb_coeffs):
        """The reference's estimate of a colour temperature from camera white-balance multipliers
        (imageProcessorWorker.py:443-473): red/blue ratio after normalising to green, three linear pieces, clamped
        to 2000-12000 K.  None when the multipliers are unusable."""
        if wb_coeffs is None or len(wb_coeffs) < 3:
            return None
        r_mult, g_mult, b_mult = wb_coeffs[0], wb_coeffs[1], wb_coeffs[2]
        if not g_mult > 0:
            return None
        rb_ratio = (r_mult / g_mult) / (b_mult / g_mult + 1e-10)
        if rb_ratio > 1.5:
            temp = 2000 + (rb_ratio - 1.5) * 1000
        elif rb_ratio > 1.0:
            temp = 3000 + (rb_ratio - 1.0) * 4000
        else:
            temp = 5000 + (1.0 - rb_ratio) * 5000
        return float(max(2000, min(12000, temp)))

    def _get_wb_from_rawpy(self, img_path: str):
        """``raw.camera_whitebalance`` -> temperature (imageProcessorWorker.py:429-478); None when unavailable."""
        try:
            import rawpy

            with rawpy.imread(img_path) as raw:
                return self.wb_temperature_from_multipliers(raw.camera_whitebalance)
        except Exception as e:
            self.signals.log.emit(f"[WB] Error reading rawpy white balance: {e}")
        return None

    def _get_wb_from_oiio_exif(self, img_path: str):
        """A colour temperature among the EXIF attributes OpenImageIO reports (imageProcessorWorker.py:480-534)."""
        try:
            from OpenImageIO import ImageInput

            img_input = ImageInput.open(img_path)
            if not img_input:
                return None
            spec = img_input.spec()
            img_input.close()
            for field in ('Exif:ColorTemperature', 'Exif:WhiteBalance', 'EXIF:ColorTemperature', 'EXIF:WhiteBalance',
                          'ColorTemperature', 'WhiteBalance'):
                try:
                    value = spec.getattribute(field)
                except Exception:
                    continue
                if isinstance(value, str):
                    try:
                        value = float(value.replace('K', '').replace('k', '').strip())
                    except ValueError:
                        continue
                if isinstance(value, (int, float)) and 1000 <= float(value) <= 12000:
                    return float(value)
        except Exception as e:
            self.signals.log.emit(f"[WB] Error reading OIIO EXIF white balance: {e}")
        return None

    # ------------------------------------------------------------------ fused device path

    def _fit(self):
        if self._ccm is None:
            if reference_swatches is None:
                raise RuntimeError("reference_swatches is not set: colour-science is not installed; call "
                                   "imageProcessorWorker.set_reference_swatches(table_24x3)")
            self._ccm = matrix_colour_correction(self.swatches, reference_swatches)
        return self._ccm

    def _out_kind(self, output_format: str) -> int:
        if output_format in ('.png', '.jpg', '.jpeg'):
            return _lib.SS_OUT_U8_SRGB
        if output_format == '.tiff':
            return _lib.SS_OUT_U16_SRGB if self.tiff_bitdepth == 16 else _lib.SS_OUT_U8_SRGB
        return _lib.SS_OUT_F32_LINEAR

    def _multiplier(self, img_path: str) -> float:
        """Exposure-normalisation multiplier from the metadata map (reference imageProcessorWorker.py:863-873)."""
        multiplier = 1.0
        if hasattr(self, "image_metadata_map"):
            meta = self.image_metadata_map.get(img_path)
            if meta is not None:
                try:
                    multiplier = float(meta.get('average_exposure'))
                except (TypeError, ValueError):
                    multiplier = 1.0
        return multiplier

    def chain_stages(self, img_path: str) -> dict:
        """The reference's steps 5 of ``run`` as chain stages (keyword arguments of ``make_desc``)."""
        if self.use_chart:
            kw = dict(ccm=self._fit(), gains=[self._multiplier(img_path)])  # gain + clip always runs
            if self._has_non_default_adjustments():
                current_wb = self._extract_white_balance_from_exif(img_path)
                target_wb = self.white_balance_adj if self.white_balance_adj != 5500 else current_wb
                adj = editingTools.adjustment_stages(
                    exposure=self.exposure_adj, shadows=self.shadow_adj, highlights=self.highlight_adj,
                    current_wb=current_wb, target_wb=target_wb, wb_tint=0.0,
                    denoise_strength=self.denoise_strength)  # chart mode never passes the sharpen arguments
                kw["gains"] += adj.pop("gains")
                kw.update(adj)
        else:
            current_wb = self._extract_white_balance_from_exif(img_path)
            kw = editingTools.adjustment_stages(
                exposure=self.exposure_adj, shadows=self.shadow_adj, highlights=self.highlight_adj,
                current_wb=current_wb, target_wb=self.white_balance_adj, wb_tint=0.0,
                denoise_strength=self.denoise_strength, sharpen_amount=self.sharpen_amount,
                sharpen_radius=1.0, sharpen_threshold=0.0)
        kw["final_clip"] = True
        return kw

    def _upload_image(self, ctx, raw_img):
        """Upload step of ``process_array``: the decoded image into the context's "worker.src" buffer (a uint16 image
        goes through the output buffer and is divided by 65535 on the device).  Returns ``(h, w)``."""
        raw16 = isinstance(raw_img, np.ndarray) and raw_img.dtype == np.uint16
        img = np.ascontiguousarray(raw_img, dtype=np.uint16 if raw16 else np.float32)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError(f"expected an (H, W, 3) image, got shape {img.shape}")
        h, w, _ = img.shape
        # buffers stay with the context from image to image
        f32_bytes = h * w * 3 * 4
        src = ctx.workspace("worker.src", f32_bytes)
        dst = ctx.workspace("worker.dst", f32_bytes)
        if raw16:
            ctx.upload(dst, img)                       # staged in the output buffer, which is free until the chain runs
            ctx.unpack_u16(dst, src, h * w * 3)
        else:
            ctx.upload(src, img)
        return h, w

    def _device_chain(self, ctx, h, w, img_path):
        """Steps 5-7 on the image in "worker.src": queues the kernels on the context's stream and returns
        ``(kind, encoded_dev, preview_dev)`` - the device buffers that will hold the encoded image and the 8-bit
        preview (``None``: the encoded image is its own preview).  Nothing here waits for the GPU except the masked
        export's pixel count."""
        kind = self._out_kind(self.output_format)
        kw = self.chain_stages(img_path)
        f32_bytes = h * w * 3 * 4
        src = ctx.workspace("worker.src", f32_bytes)
        dst = ctx.workspace("worker.dst", f32_bytes)
        if self.negative_mode:
            neg = ctx.workspace("worker.neg", _lib.SS_NEG_PARAM_STRIDE * 4)
            ctx.negative_params(src, h, w, 1, self.negative_low_pct, self.negative_high_pct,
                                editingTools.NUMPY_LEGACY_PERCENTILE, neg)
            kw.update(neg=True, neg_params_dev=neg)
        eight_bit = np.dtype(_lib.OUT_DTYPES[kind]) == np.uint8
        if getattr(self, "export_masked", False):
            # reference order (imageProcessorWorker.py:902-912): chain -> _apply_masking -> encode.  The chain
            # leaves the clipped linear image in `dst`, the masking zeroes regions in place, and the encode
            # (a chain with no stage) writes into `src`, whose contents are no longer needed.
            ctx.chain_run(make_desc(h, w, 1, out="f32", **kw), src, dst)
            self._apply_masking_device(ctx, dst, h, w)
            # The preview is made from the array that is encoded (imageProcessorWorker.py:907-922): both outputs
            # of one pass over it.
            if eight_bit:
                ctx.chain_run(make_desc(h, w, 1, out=kind), dst, src)
                return kind, src, None
            pre = ctx.workspace("worker.preview", h * w * 3)
            ctx.chain_run2(make_desc(h, w, 1, out=kind), dst, src, _lib.SS_OUT_U8_LINEAR, pre)
            return kind, src, pre
        if eight_bit:
            ctx.chain_run(make_desc(h, w, 1, out=kind, **kw), src, dst)
            return kind, dst, None
        # one pass over the image, two outputs: the encoded image and the 8-bit linear preview of the same pixels
        pre = ctx.workspace("worker.preview", h * w * 3)
        ctx.chain_run2(make_desc(h, w, 1, out=kind, **kw), src, dst, _lib.SS_OUT_U8_LINEAR, pre)
        return kind, dst, pre

    def process_array(self, raw_img_float: np.ndarray, img_path: str = ""):
        """Steps 5-7 of the reference's ``run`` on one decoded image, resident on the GPU:
        returns ``(encoded, preview_uint8)``.  ``raw_img_float`` is the output of the RAW decode
        BEFORE negative inversion (the inversion runs on the device here).  A uint16 array is taken as the
        RAW decoder's 16-bit output: it is uploaded as it is and divided by 65535 on the device (the
        reference's ``np.array(rgb, dtype=np.float32) / 65535.0``, imageProcessorWorker.py:313)."""
        ctx = default_context()
        h, w = self._upload_image(ctx, raw_img_float)
        kind, enc_dev, pre_dev = self._device_chain(ctx, h, w, img_path)

        def host_array(tag, dtype):
            if self.reuse_output_buffers:
                return ctx.host_workspace(tag, (h, w, 3), dtype)
            return np.empty((h, w, 3), dtype=dtype)

        encoded = host_array("worker.encoded", _lib.OUT_DTYPES[kind])
        ctx.download(encoded, enc_dev)
        if pre_dev is None:
            return encoded, encoded
        preview = host_array("worker.preview", np.uint8)
        ctx.download(preview, pre_dev)
        return encoded, preview

    # ------------------------------------------------------------------ masking

    MASK_MIN_AREA = 200  # imageProcessorWorker.py:1005

    def _apply_masking_device(self, ctx, img_dev, h, w):
        """``_apply_masking`` (imageProcessorWorker.py:979-1030) on an image resident on the GPU, in place."""
        shadow_limit = getattr(self, "shadow_limit", 0.05)
        highlight_limit = getattr(self, "highlight_limit", 0.98)
        count_dev = ctx.workspace("worker.mask_count", 8)
        ctx.mask_regions(img_dev, h, w, 1, shadow_limit, highlight_limit, self.MASK_MIN_AREA, count_dev,
                         float32_luminance=self._reference_image_is_float32())
        count = np.zeros(1, dtype=np.uint64)
        ctx.download(count, count_dev)
        self.signals.log.emit(
            f"[Export Masked] Masked {int(count[0])} pixels (min area={self.MASK_MIN_AREA}) "
            f"with limits shadow<={shadow_limit:.2f}, highlight>={highlight_limit:.2f}"
        )

    def _reference_image_is_float32(self) -> bool:
        """Whether the array the reference masks is still float32 there: no chart (colour-science returns float64)
        and no Lab edit (shadows / highlights, sharpen: float64 as well); exposure and white balance keep float32
        (editingTools.py:435-466, 606-640).  Under the reference's numpy < 2 the luminance test is then float32."""
        return (not self.use_chart and self.shadow_adj == 0.0 and self.highlight_adj == 0.0
                and self.sharpen_amount == 0.0)

    def _apply_masking(self, corrected: np.ndarray) -> np.ndarray:
        """Host-array form with the reference's signature: zeroes the regions in ``corrected`` (float32 images are
        modified in place like the reference's array; other dtypes come back as a float32 copy) and returns it."""
        ctx = default_context()
        arr = np.ascontiguousarray(corrected, dtype=np.float32)
        if arr.ndim != 3 or arr.shape[2] != 3:
            raise ValueError(f"expected an (H, W, 3) image, got shape {arr.shape}")
        h, w, _ = arr.shape
        buf = ctx.workspace("worker.dst", arr.nbytes)
        ctx.upload(buf, arr)
        self._apply_masking_device(ctx, buf, h, w)
        ctx.download(arr, buf)
        if arr is not corrected and isinstance(corrected, np.ndarray) and corrected.dtype == np.float32 \
                and corrected.flags.writeable:
            corrected[...] = arr
            return corrected
        return arr

    # ------------------------------------------------------------------ host I/O hooks

    def _decode_raw(self, img_path: str) -> np.ndarray:
        """The RAW decode of ``load_image`` (imageProcessorWorker.py:304-313).  Returns the decoder's uint16 image
        (``process_array`` divides by 65535 on the device); a ``raw_loader`` hook may return uint16 or float data."""
        logs = []
        try:
            return self._decode_raw_quiet(img_path, logs)
        finally:
            for msg in logs:
                self.signals.log.emit(msg)

    def _write(self, out_path: str, encoded: np.ndarray, ext: str) -> None:
        if self.image_writer is not None:
            self.image_writer(out_path, encoded, ext)
            return
        from .imageWriters import write_image

        write_image(out_path, encoded, ext, self.jpeg_quality, exr_colorspace=self.exr_colorspace)

    # the reference's per-format writers (imageProcessorWorker.py:604-662), same names; `encode_image` made the array
    def export_image_jpg(self, img_array: np.ndarray, output_path: str):
        self._write(output_path, img_array, '.jpg')

    def export_image_png(self, img_array: np.ndarray, output_path: str):
        self._write(output_path, img_array, '.png')

    def export_image_tiff(self, img_array: np.ndarray, output_path: str):
        self._write(output_path, img_array, '.tiff')

    def export_image_exr(self, img_array: np.ndarray, output_path: str):
        """Writes the float image as scene-linear EXR in ``exr_colorspace`` and returns ``(None, preview_uint8)``:
        the reference returns its OpenImageIO buffer first, which only its metadata step reads."""
        self._write(output_path, img_array, '.exr')
        return None, np.uint8(np.clip(img_array * 255.0, 0, 255))

    def construct_export_path(self, img_path: str) -> str:
        """Reference imageProcessorWorker.py:664-746: the server's path in network mode, else the naming schema
        (when the host application supplies ``fileNamingSchema.apply_naming_schema``; any failure falls back to
        the default name, as in the reference), else the legacy rules (original file name, or
        ``name_base_<sequence>`` from ``rename_map``)."""
        default_fn = os.path.splitext(os.path.basename(img_path))[0] + self.output_format
        if self.from_network:
            if self.network_output_path:
                return self.network_output_path
            if getattr(self, "use_original_filenames", False) or not self.name_base:
                return os.path.join(self.output_folder, default_fn)
            seq = self.rename_map.get(img_path, 1) if self.rename_map else 1
            return os.path.join(self.output_folder, f"{self.name_base}_{seq:0{self.padding}d}{self.output_format}")
        if self.use_export_schema and self.export_schema:
            try:
                from fileNamingSchema import apply_naming_schema  # the host application's module

                out_path = apply_naming_schema(schema=self.export_schema, input_path=img_path,
                                               output_base_dir=self.output_folder, custom_name=self.custom_name or "",
                                               image_number=self.rename_map.get(img_path, 1),
                                               output_extension=self.output_format, root_folder=self.root_folder or "",
                                               group_name=self.group_name or "")
                out_dir = os.path.dirname(out_path)
                if out_dir and out_dir != self.output_folder:
                    os.makedirs(out_dir, exist_ok=True)
                return out_path
            except Exception as e:
                self.signals.log.emit(f"[Schema Error] Failed to apply naming schema for {img_path}: {e}")
                return os.path.join(self.output_folder, default_fn)
        if not getattr(self, "use_original_filenames", False):
            seq = self.rename_map.get(img_path)
            if seq is not None and self.name_base:
                return os.path.join(self.output_folder, f"{self.name_base}_{seq:0{self.padding}d}{self.output_format}")
        return os.path.join(self.output_folder, default_fn)

    # ------------------------------------------------------------------ main loop

    def run(self):
        """Reference imageProcessorWorker.py:815-977 without metadata embedding.  With ``pipeline_depth`` 2 (the default)
        and more than one image the loop is two images deep (``_run_pipelined``); the results are those of the
        sequential loop bit for bit, and so is each image's own sequence of signals - only the signals of neighbouring
        images interleave ('started' of image i + 1 comes before 'finished' of image i)."""
        if self.pipeline_depth >= 2 and len(self.images) > 1:
            return self._run_pipelined()
        return self._run_sequential()

    def _default_out_path(self, img_path):
        return os.path.join(self.output_folder, os.path.splitext(os.path.basename(img_path))[0] + self.output_format)

    def _run_sequential(self):
        ext = self.output_format
        for img_path in self.images:
            t0 = time.time()
            out_path = None
            try:
                out_path = self.construct_export_path(img_path)
                if self.from_network:
                    self.signals.status.emit('started')
                else:
                    self.signals.status.emit(img_path, 'started', time.time() - t0, out_path)
                self.signals.log.emit(f"[Worker] Starting {os.path.basename(img_path)} -> {os.path.basename(out_path)}")
                try:
                    raw_img_float = self._decode_raw(img_path)
                except Exception as e:
                    self.signals.log.emit(f"[Image Load Error] {img_path}: {e}")
                    self.signals.status.emit(img_path, 'error', time.time() - t0, out_path)
                    continue
                if self.cancelled:
                    self.signals.log.emit("[Worker] Cancelled by user. Exiting thread.")
                    self.signals.status.emit(img_path, 'cancelled', time.time() - t0, out_path)
                    return
                encoded, preview = self.process_array(raw_img_float, img_path)
                del raw_img_float
                self._write(out_path, encoded, ext)
                self.signals.log.emit(f"[Saved] {out_path}")
                self.signals.preview.emit([preview, out_path])
                self.signals.status.emit(img_path, 'finished', time.time() - t0, out_path)
            except Exception as e:
                self.signals.log.emit(f"[Processing Error] {img_path}: {e}")
                if out_path is None:
                    out_path = self._default_out_path(img_path)
                self.signals.status.emit(img_path, 'error', time.time() - t0, out_path)

    def _run_pipelined(self):
        """The loop of ``run`` two images deep.  Image i lives in slot i % 2: a context of its own (its stream, its
        device buffers) and two page-locked output arrays.

            helper thread   decode image i + 1, upload it (and divide by 65535) on its slot's context
            this thread     queue the kernels and the downloads of image i; then finish image i - 1: wait for its
                            stream, hand it to the writer thread; signals
            writer thread   write image i - 1

        The error contract is the reference's (imageProcessorWorker.py:950-970): whatever goes wrong with one image is
        logged, reported as status 'error', and the loop goes on with the next.  All signals are emitted by this
        thread."""
        from concurrent.futures import ThreadPoolExecutor

        ext = self.output_format
        # One context per slot (its stream, device buffers and page-locked arrays), kept by the worker from run to run.
        # They are made when the first image has been decoded: a batch whose files cannot be read is reported image by
        # image without a GPU ever being asked for.
        slots = [dict(k=0), dict(k=1)]

        def slot_ctx(slot):
            from .. import Context

            ctxs = self.__dict__.setdefault("_pipeline_ctxs", {})
            ctx = ctxs.get(slot["k"])
            if ctx is None or ctx.handle is None:
                ctx = ctxs[slot["k"]] = Context(_lib.default_device())
            slot["ctx"] = ctx
            return ctx

        loader = ThreadPoolExecutor(max_workers=1, thread_name_prefix="scanspace-load")
        writer = ThreadPoolExecutor(max_workers=1, thread_name_prefix="scanspace-write")

        def load(img_path, slot):  # (helper thread) -> (h, w, log lines); raises what the decode or the upload raises
            logs = []  # the decoder's log lines: emitted by the loop's thread, not from here
            raw = self._decode_raw_quiet(img_path, logs)
            ctx = slot_ctx(slot)
            h, w = self._upload_image(ctx, raw)
            ctx.sync()  # (the slot's context goes back to the loop's thread idle)
            return h, w, logs

        def start(i):  # image i enters the pipeline: 'started', then its decode on the helper thread
            img_path = self.images[i]
            job = dict(path=img_path, t0=time.time(), out_path=None, slot=slots[i % 2], state="loading")
            try:
                job["out_path"] = self.construct_export_path(img_path)
                if self.from_network:
                    self.signals.status.emit('started')
                else:
                    self.signals.status.emit(img_path, 'started', 0.0, job["out_path"])
                self.signals.log.emit(f"[Worker] Starting {os.path.basename(img_path)} -> {os.path.basename(job['out_path'])}")
                job["load"] = loader.submit(load, img_path, job["slot"])
            except Exception as e:
                fail(job, e)
            return job

        def fail(job, e, kind="Processing Error"):
            self.signals.log.emit(f"[{kind}] {job['path']}: {e}")
            out_path = job["out_path"] if job["out_path"] is not None else self._default_out_path(job["path"])
            self.signals.status.emit(job["path"], 'error', time.time() - job["t0"], out_path)
            job["state"] = "done"

        def launch(job):  # decoded and uploaded -> kernels and downloads queued on the slot's stream
            h, w, logs = job["load"].result()
            job["state"] = "loaded"
            ctx = job["slot"]["ctx"]
            for msg in logs:
                self.signals.log.emit(msg)
            kind, enc_dev, pre_dev = self._device_chain(ctx, h, w, job["path"])
            job["encoded"] = ctx.host_workspace("worker.encoded", (h, w, 3), _lib.OUT_DTYPES[kind])
            ctx.download_async(job["encoded"], enc_dev)
            job["preview"] = job["encoded"]
            if pre_dev is not None:
                job["preview"] = ctx.host_workspace("worker.preview", (h, w, 3), np.uint8)
                ctx.download_async(job["preview"], pre_dev)
            job["state"] = "on the GPU"

        def write(job):  # (writer thread)
            self._write(job["out_path"], job["encoded"], ext)
            # The slot's page-locked arrays are overwritten two images later: a listener that keeps the preview gets
            # an array of its own, unless the caller has agreed to reused buffers.
            return job["preview"] if self.reuse_output_buffers else job["preview"].copy()

        def collect(job):  # wait for the slot's stream, hand the image to the writer thread
            job["slot"]["ctx"].sync()
            job["write"] = writer.submit(write, job)
            job["state"] = "writing"

        def finish(job):  # written: the image's last signals; its slot's output arrays are free again
            preview = job["write"].result()
            self.signals.log.emit(f"[Saved] {job['out_path']}")
            self.signals.preview.emit([preview, job["out_path"]])
            self.signals.status.emit(job["path"], 'finished', time.time() - job["t0"], job["out_path"])
            job["state"] = "done"

        def advance(job, step):
            if job is None or job["state"] == "done":
                return
            try:
                step(job)
            except Exception as e:
                fail(job, e, "Image Load Error" if job["state"] == "loading" else "Processing Error")

        try:
            n = len(self.images)
            jobs = [None] * n
            jobs[0] = start(0)
            for i in range(n):
                cur, prev = jobs[i], jobs[i - 1] if i > 0 else None
                if cur["state"] == "loading":
                    try:
                        cur["load"].result()
                    except Exception:
                        pass  # (reported by `launch` below)
                if self.cancelled:
                    for job in (prev, ):
                        if job is not None and job["state"] == "on the GPU":
                            advance(job, collect)
                        if job is not None and job["state"] == "writing":
                            advance(job, finish)
                    self.signals.log.emit("[Worker] Cancelled by user. Exiting thread.")
                    if cur["state"] != "done":
                        self.signals.status.emit(cur["path"], 'cancelled', time.time() - cur["t0"], cur["out_path"])
                    return
                # the slot's output arrays were last used by image i - 2: its write must be over before new downloads
                if i >= 2 and jobs[i - 2]["state"] == "writing":
                    advance(jobs[i - 2], finish)
                if cur["state"] == "loading":
                    advance(cur, launch)
                    if cur["state"] == "loaded":  # (launch stopped half-way without raising: cannot happen)
                        cur["state"] = "done"
                if prev is not None and prev["state"] == "on the GPU":
                    advance(prev, collect)  # (waits for image i - 1 only: image i runs on the other stream)
                if i + 1 < n:
                    jobs[i + 1] = start(i + 1)  # its slot is the one image i - 1 has just left
                if prev is not None and prev["state"] == "writing":
                    advance(prev, finish)
            last = jobs[n - 1]
            if last["state"] == "on the GPU":
                advance(last, collect)
            if last["state"] == "writing":
                advance(last, finish)
        finally:
            loader.shutdown(wait=True)
            writer.shutdown(wait=True)

    def _decode_raw_quiet(self, img_path, logs):
        """``_decode_raw`` for the helper thread: its log lines go to ``logs`` (the loop's thread emits them)."""
        if self.raw_loader is not None:
            arr = np.asarray(self.raw_loader(img_path))
            return arr if arr.dtype == np.uint16 else np.asarray(arr, dtype=np.float32)
        logs.append(f"[RAW] Using original path for rawpy: {img_path}")
        import rawpy  # host decode stays on LibRaw

        with rawpy.imread(img_path) as raw:
            rgb = raw.postprocess(output_bps=16, gamma=(1, 1), no_auto_bright=True, use_camera_wb=True,
                                  output_color=rawpy.ColorSpace.sRGB)
            return np.ascontiguousarray(rgb, dtype=np.uint16)
