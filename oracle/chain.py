"""CPU restatement of the reference's correction chain (numpy, float64).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Each function follows the reference function cited in its docstring
operation for operation (same order, same temporaries, same clips), with the
third-party leaves taken from ``oracle.leaves``.  The functions that restate
``ImageProcessor/editingTools.py`` are pinned by ``tests/golden/*.npz``, which
``oracle/make_golden.py`` produced by executing the reference file itself.

dtype policy: the reference runs its chain in float64 as soon as a
colour-science function touches the data (``as_float_array``).  This oracle is
float64 everywhere except where the reference forces float32 explicitly
(``image.astype(np.float32)`` before the negative inversion and before
``cv2.GaussianBlur``).  Where numpy>=2 (present here) and numpy<2 (the
reference's pin, requirements.txt:7) promote differently, ``legacy=True``
selects the pinned numpy<2 behaviour.
"""

from __future__ import annotations

import numpy as np

from . import leaves as lv

# --------------------------------------------------------------------------
# editingTools.py
# --------------------------------------------------------------------------


def negative_sample_step(h, w):
    """editingTools.py:325."""
    return max(1, int(np.sqrt((h * w) / 250_000)))


def invert_negative_auto(image, low_percentile=0.05, high_percentile=0.995, legacy=True, details=None):
    """editingTools.py:296-338.  ``low_percentile=0.05`` means the 5th
    percentile (the argument is multiplied by 100 before np.percentile).

    legacy=True : numpy<2 -> percentiles and the whole normalisation in float64.
    legacy=False: numpy>=2 -> float32 throughout (what executes in this image).
    """
    if image is None or image.ndim != 3 or image.shape[2] != 3:
        return image
    img = np.clip(image.astype(np.float32), 0.0, 1.0)
    h, w, _ = img.shape
    step = negative_sample_step(h, w)
    sample = img[::step, ::step, :].reshape(-1, 3)
    p_low, lo_a, lo_b, lo_idx, lo_g = lv.percentile_linear(sample, low_percentile * 100.0, legacy)
    p_high, hi_a, hi_b, hi_idx, hi_g = lv.percentile_linear(sample, high_percentile * 100.0, legacy)
    denom = np.maximum(p_high - p_low, 1e-3)
    if details is not None:
        details.update(step=step, n=sample.shape[0], p_low=p_low, p_high=p_high, denom=denom,
                       low_stats=(lo_a, lo_b, lo_idx, lo_g), high_stats=(hi_a, hi_b, hi_idx, hi_g))
    out = (img - p_low[None, None, :]) / denom[None, None, :]
    out = 1.0 - np.clip(out, 0.0, 1.0)
    return out


def ensure_positive_if_negative(image, enabled=False, low_percentile=0.05, high_percentile=0.995, legacy=True):
    """editingTools.py:340-365."""
    if not enabled:
        return image
    try:
        return invert_negative_auto(image, low_percentile, high_percentile, legacy)
    except Exception:
        return image


def rgb_to_lab(image):
    """editingTools.py:367-386 (note: the data is treated as sRGB-ENCODED)."""
    lab = lv.XYZ_to_Lab(lv.sRGB_to_XYZ(image))
    return lab[..., 0], lab[..., 1], lab[..., 2]


def lab_to_rgb(L, a, b):
    """editingTools.py:389-408."""
    rgb = lv.XYZ_to_sRGB(lv.Lab_to_XYZ(np.stack([L, a, b], axis=-1)))
    return np.clip(rgb, 0, 1)


def is_lab_image(image):
    """editingTools.py:411-432."""
    if image.shape[-1] != 3:
        return False
    return bool(np.max(image[..., 0]) > 1.5 and (np.min(image[..., 1]) < -1 or np.min(image[..., 2]) < -1))


def adjust_exposure(image, exposure_value):
    """editingTools.py:435-466."""
    if exposure_value == 0:
        return image
    multiplier = np.power(2.0, exposure_value)
    return np.clip(np.asarray(image, dtype=np.float64) * multiplier, 0, 1)


def adjust_shadows_highlights_combined(image, shadow_amount, highlight_amount, falloff_power=7.0):
    """editingTools.py:469-544 (purely pointwise on L)."""
    if shadow_amount == 0 and highlight_amount == 0:
        return image
    is_lab_input = is_lab_image(image)
    if not is_lab_input:
        L, a, b = rgb_to_lab(image)
    else:
        image = np.asarray(image, dtype=np.float64)
        L, a, b = image[..., 0], image[..., 1], image[..., 2]
    L_norm = L / 100.0
    shadows_scaled = shadow_amount * 0.05
    highlights_scaled = highlight_amount * 0.05
    highlight_adj = highlights_scaled * (np.power(falloff_power, L_norm) - 1.0) if highlights_scaled != 0 else 0
    shadow_adj = shadows_scaled * (np.power(falloff_power, 1.0 - L_norm) - 1.0) if shadows_scaled != 0 else 0
    total_adjustment = (highlight_adj + shadow_adj) * 100.0
    L_adjusted = np.clip(L + total_adjustment, 0, 100)
    if abs(shadow_amount) > 50 or abs(highlight_amount) > 50:
        adjustment_strength = np.abs(total_adjustment) / 20.0
        chroma_scale = 1.0 - np.clip(adjustment_strength * 0.05, 0, 0.2)
        a_adjusted = a * chroma_scale
        b_adjusted = b * chroma_scale
    else:
        a_adjusted, b_adjusted = a, b
    if not is_lab_input:
        return np.clip(lab_to_rgb(L_adjusted, a_adjusted, b_adjusted), 0, 1)
    return np.stack([L_adjusted, a_adjusted, b_adjusted], axis=-1)


def adjust_sharpen(image, amount=0.0, radius=1.0, threshold=0.0):
    """editingTools.py:714-820."""
    if amount <= 0:
        return image
    is_lab_input = is_lab_image(image)
    if not is_lab_input:
        L, a, b = rgb_to_lab(image)
    else:
        image = np.asarray(image, dtype=np.float64)
        L, a, b = image[..., 0], image[..., 1], image[..., 2]
    L_norm = L / 100.0
    amount_scaled = amount / 50.0
    threshold_scaled = threshold / 100.0
    sigma = radius
    kernel_size = lv.usm_kernel_size(sigma)
    L_blurred = lv.gaussian_blur_reflect(L_norm.astype(np.float32), kernel_size, sigma)
    unsharp_mask = L_norm - L_blurred
    if threshold_scaled > 0:
        edge_strength = np.abs(unsharp_mask)
        edge_mask = np.where(edge_strength > threshold_scaled, 1.0, edge_strength / threshold_scaled)
        unsharp_mask = unsharp_mask * edge_mask
    highlight_mask = np.where(L_norm > 0.9, 1.0 - (L_norm - 0.9) * 10.0, 1.0)
    shadow_mask = np.where(L_norm < 0.1, L_norm * 10.0, 1.0)
    adaptive_mask = np.clip(highlight_mask * shadow_mask, 0.1, 1.0)
    sharpened_L = L_norm + (unsharp_mask * amount_scaled * adaptive_mask)
    sharpened_L = np.clip(sharpened_L, 0, 1) * 100.0
    if amount > 30:
        edge_boost = np.abs(unsharp_mask) * (amount / 100.0) * 0.1
        saturation_boost = 1.0 + edge_boost
        a_enh = np.clip(a * saturation_boost, -127, 127)
        b_enh = np.clip(b * saturation_boost, -127, 127)
    else:
        a_enh, b_enh = a, b
    if not is_lab_input:
        return np.clip(lab_to_rgb(sharpened_L, a_enh, b_enh), 0, 1)
    return np.stack([sharpened_L, a_enh, b_enh], axis=-1)


def adjust_white_balance_gains(image, gains):
    """editingTools.py:634-640 with the RGB multipliers already known."""
    adjusted = np.array(image, dtype=np.float64, copy=True)
    adjusted[..., 0] *= gains[0]
    adjusted[..., 1] *= gains[1]
    adjusted[..., 2] *= gains[2]
    return np.clip(adjusted, 0, 1)


def sample_white_balance_from_point(image, sample_x, sample_y, current_temp, sample_radius=5):
    """editingTools.py:643-711: window mean -> (R/G)/(B/G) -> logarithmic temperature shift, clamped to 1000..12000 K."""
    h, w = image.shape[:2]
    region = image[max(0, sample_y - sample_radius):min(h, sample_y + sample_radius + 1),
                   max(0, sample_x - sample_radius):min(w, sample_x + sample_radius + 1)]
    avg = np.mean(region, axis=(0, 1))
    if avg[1] < 0.001:
        return current_temp
    rb = (avg[0] / avg[1]) / (avg[2] / avg[1] + 1e-10)
    mult = 1 + np.log(rb) * 0.3 if rb > 1 else 1 / (1 + np.log(1 / rb) * 0.3)
    return float(np.clip(current_temp * mult, 1000, 12000))


def apply_all_adjustments(image, exposure=0.0, shadows=0.0, highlights=0.0, wb_gains=None,
                          sharpen_amount=0.0, sharpen_radius=1.0, sharpen_threshold=0.0):
    """editingTools.py:823-909 without the out-of-scope denoise stage.  White
    balance enters as the already computed RGB multipliers (``wb_gains=None``
    is the neutral ``current_wb == target_wb and tint == 0`` case)."""
    shadows = shadows / 70
    highlights = highlights * 1.3
    result = image.copy()
    if wb_gains is not None:
        result = adjust_white_balance_gains(result, wb_gains)
    if exposure != 0:
        result = adjust_exposure(result, exposure)
    if shadows != 0 or highlights != 0:
        result = adjust_shadows_highlights_combined(result, shadows, highlights)
    if sharpen_amount > 0:
        result = adjust_sharpen(result, sharpen_amount, sharpen_radius, sharpen_threshold)
    return result


# --------------------------------------------------------------------------
# imageProcessorWorker.py
# --------------------------------------------------------------------------


def encode_image(img_array, output_format, tiff_bitdepth=8):
    """imageProcessorWorker.py:579-602 (truncating quantisers)."""
    if output_format in (".png", ".jpg", ".jpeg"):
        return np.uint8(255 * lv.cctf_encoding(img_array))
    if output_format == ".tiff":
        enc = lv.cctf_encoding(img_array)
        if tiff_bitdepth == 16:
            return np.uint16(np.clip(enc * 65535.0, 0, 65535))
        return np.uint8(np.clip(enc * 255.0, 0, 255))
    return img_array


def worker_chart_path(raw_img_float, swatches, reference_swatches, multiplier=1.0,
                      negative_mode=False, negative_low_pct=0.05, negative_high_pct=0.995,
                      exposure_adj=0.0, shadow_adj=0.0, highlight_adj=0.0, wb_gains=None,
                      ccm_method="Cheung 2004", ccm_kwargs=None, legacy=True):
    """ImageCorrectionWorker.run steps 3-5, chart mode
    (imageProcessorWorker.py:319-324, 858-894): [invert] -> CCM -> gain -> clip
    -> apply_all_adjustments WITHOUT sharpen arguments -> clip."""
    img = ensure_positive_if_negative(raw_img_float, negative_mode, negative_low_pct, negative_high_pct, legacy)
    corrected = lv.colour_correction(img, swatches, reference_swatches, ccm_method, **(ccm_kwargs or {}))
    corrected = corrected * float(multiplier)
    corrected = np.clip(corrected, 0, 1)
    if exposure_adj != 0.0 or shadow_adj != 0.0 or highlight_adj != 0.0 or wb_gains is not None:
        corrected = apply_all_adjustments(corrected, exposure=exposure_adj, shadows=shadow_adj,
                                          highlights=highlight_adj, wb_gains=wb_gains)
        corrected = np.clip(corrected, 0, 1)
    return corrected


def worker_nochart_path(raw_img_float, negative_mode=False, negative_low_pct=0.05, negative_high_pct=0.995,
                        exposure_adj=0.0, shadow_adj=0.0, highlight_adj=0.0, wb_gains=None,
                        sharpen_amount=0.0, legacy=True):
    """ImageCorrectionWorker.run no-chart mode (imageProcessorWorker.py:356-368,
    895-900): sharpen radius 1.0 / threshold 0.0 are hard-wired."""
    img = ensure_positive_if_negative(raw_img_float, negative_mode, negative_low_pct, negative_high_pct, legacy)
    adjusted = apply_all_adjustments(img, exposure=exposure_adj, shadows=shadow_adj, highlights=highlight_adj,
                                     wb_gains=wb_gains, sharpen_amount=sharpen_amount,
                                     sharpen_radius=1.0, sharpen_threshold=0.0)
    return np.clip(adjusted, 0, 1)


# --------------------------------------------------------------------------
# image_space.py: dataset exposure normalisation
# --------------------------------------------------------------------------


def image_brightness(arr, shadow_threshold, highlight_threshold):
    """image_space.py:4360-4365 on one thumbnail (float32 arithmetic, strict mask)."""
    arr = np.asarray(arr, dtype=np.float32)
    lum = 0.2126 * arr[:, :, 0] + 0.7152 * arr[:, :, 1] + 0.0722 * arr[:, :, 2]
    mask = (lum > shadow_threshold) & (lum < highlight_threshold)
    valid = lum[mask]
    return valid.mean() if valid.size > 0 else lum.mean()


def exposure_multipliers(brightness_list, chart_index=None):
    """image_space.py:4376-4394: reference = flagged chart image's brightness,
    else the mean over all; multiplier = ref / img (1.0 where img <= 0)."""
    ref = brightness_list[chart_index] if chart_index is not None else np.mean(brightness_list)
    return [float(ref / b) if b > 0 else 1.0 for b in brightness_list]


# --------------------------------------------------------------------------
# BASELINE.json configurations, end to end (used by tests and bench cpu_baseline)
# --------------------------------------------------------------------------


def config1(img, swatches, reference_swatches):
    """3x3 Cheung-2004 CCM -> clip -> sRGB encode, float output."""
    c = np.clip(lv.colour_correction(img, swatches, reference_swatches), 0, 1)
    return lv.cctf_encoding(c)


def config2(img, swatches, reference_swatches, multiplier, amount, radius=2.0, threshold=0.0):
    """CCM -> exposure gain -> clip -> unsharp mask -> clip -> 16-bit sRGB quantise."""
    c = np.clip(lv.colour_correction(img, swatches, reference_swatches) * float(multiplier), 0, 1)
    c = np.clip(adjust_sharpen(c, amount, radius, threshold), 0, 1)
    return encode_image(c, ".tiff", 16)


def config3(img, swatches, reference_swatches, low_percentile, high_percentile, legacy=True):
    """Negative mode: percentile inversion -> CCM -> clip, float linear output."""
    pos = invert_negative_auto(img, low_percentile, high_percentile, legacy)
    return np.clip(lv.colour_correction(pos, swatches, reference_swatches), 0, 1)


def config4(img, swatches, reference_swatches, degree, shadow_amount, highlight_amount):
    """Finlayson root-polynomial CCM -> clip -> Lab shadow/highlight, float linear output."""
    c = np.clip(lv.colour_correction(img, swatches, reference_swatches, "Finlayson 2015",
                                     degree=degree, root_polynomial_expansion=True), 0, 1)
    return np.clip(adjust_shadows_highlights_combined(c, shadow_amount, highlight_amount), 0, 1)
