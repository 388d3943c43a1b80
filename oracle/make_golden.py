"""Generate tests/golden/*.npz by executing the REFERENCE's own code.

TEST INFRASTRUCTURE ONLY.  Run in the build container, where /root/reference
is mounted:  ``python oracle/make_golden.py``.  The reference never travels;
only the vectors written here (inputs are re-creatable from the seeds, outputs
are data) are committed.

How: ``/root/reference/ImageProcessor/editingTools.py`` is loaded by file path
(its package ``__init__`` pulls Qt and is not imported).  The two third-party
modules it needs and that are not installed (``colour``, ``cv2``) are
replaced by stub modules whose few attributes are the float64 restatements of
``oracle.leaves``.  Everything else that executes -- inversion, exposure gain,
shadow/highlight curve, unsharp-mask mask logic, the order and pre-scaling in
``apply_all_adjustments`` -- is the reference's verbatim code, which is what
these vectors pin.  The leaves themselves stay "parity unpinned".

The vectors are produced under the numpy installed here (2.x); the reference
pins numpy<2.  The only place that matters is ``invert_negative_auto``
(float32 percentiles under numpy 2, float64 under numpy<2): the vectors pin
the ``legacy=False`` branch of the oracle.
"""

from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import leaves as lv  # noqa: E402

REF_FILE = "/root/reference/ImageProcessor/editingTools.py"
OUT_DIR = os.path.join(ROOT, "tests", "golden")


def load_reference_editing_tools():
    colour = types.ModuleType("colour")
    colour.sRGB_to_XYZ = lv.sRGB_to_XYZ
    colour.XYZ_to_Lab = lv.XYZ_to_Lab
    colour.Lab_to_XYZ = lv.Lab_to_XYZ
    colour.XYZ_to_sRGB = lv.XYZ_to_sRGB
    cv2 = types.ModuleType("cv2")
    cv2.BORDER_REFLECT = 2

    def GaussianBlur(src, ksize, sigmaX, borderType=None):
        assert ksize[0] == ksize[1] and borderType == cv2.BORDER_REFLECT
        return lv.gaussian_blur_reflect(src, ksize[0], sigmaX)

    cv2.GaussianBlur = GaussianBlur
    saved = {k: sys.modules.get(k) for k in ("colour", "cv2")}
    sys.modules["colour"] = colour
    sys.modules["cv2"] = cv2
    try:
        spec = importlib.util.spec_from_file_location("_ref_editingTools", REF_FILE)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


from oracle.inputs import GENERATORS  # noqa: E402


def main():
    ref = load_reference_editing_tools()
    os.makedirs(OUT_DIR, exist_ok=True)
    cases = {}

    def add(name, gen, seed, h, w, fn, kwargs, stride=None):
        img = GENERATORS[gen](seed, h, w)
        out = fn(img, **kwargs)
        out = np.asarray(out)
        if stride is not None:
            out = out[:: stride[0], :: stride[1]]
        cases[name] = dict(gen=gen, seed=seed, h=h, w=w, kwargs=kwargs, stride=stride, out=out)

    # a1 invert_negative_auto  (editingTools.py:296)
    add("invert_small_default", "negative", 11, 96, 128, ref.invert_negative_auto, {})
    add("invert_small_q", "negative", 12, 61, 83, ref.invert_negative_auto,
        dict(low_percentile=0.005, high_percentile=0.995))
    add("invert_wide", "wide", 13, 64, 96, ref.invert_negative_auto, {})
    add("invert_large_step2", "negative", 14, 1100, 1300, ref.invert_negative_auto, {}, stride=(37, 41))
    # a2 ensure_positive_if_negative (editingTools.py:340)
    add("ensure_enabled", "negative", 15, 48, 64, ref.ensure_positive_if_negative, dict(enabled=True))
    add("ensure_disabled", "negative", 15, 48, 64, ref.ensure_positive_if_negative, dict(enabled=False))
    # a6 adjust_exposure (editingTools.py:435)
    add("exposure_up", "linear", 21, 48, 64, ref.adjust_exposure, dict(exposure_value=0.7))
    add("exposure_down", "wide", 22, 48, 64, ref.adjust_exposure, dict(exposure_value=-1.3))
    # a8 adjust_shadows_highlights_combined (editingTools.py:469)
    add("sh_basic", "linear", 31, 96, 128, ref.adjust_shadows_highlights_combined,
        dict(shadow_amount=0.5, highlight_amount=-0.5))
    add("sh_shadow_only", "linear", 32, 61, 83, ref.adjust_shadows_highlights_combined,
        dict(shadow_amount=1.2, highlight_amount=0.0))
    add("sh_chroma", "linear", 33, 61, 83, ref.adjust_shadows_highlights_combined,
        dict(shadow_amount=60.0, highlight_amount=-20.0))
    add("sh_falloff", "wide", 34, 48, 64, ref.adjust_shadows_highlights_combined,
        dict(shadow_amount=-0.8, highlight_amount=0.9, falloff_power=4.0))
    add("sh_lab_input", "lab", 35, 48, 64, ref.adjust_shadows_highlights_combined,
        dict(shadow_amount=0.5, highlight_amount=-0.5))
    # a9 adjust_sharpen (editingTools.py:714)
    add("usm_r2_a50", "linear", 41, 96, 128, ref.adjust_sharpen, dict(amount=50.0, radius=2.0, threshold=0.0))
    add("usm_r2_a1", "linear", 42, 96, 128, ref.adjust_sharpen, dict(amount=1.0, radius=2.0, threshold=0.0))
    add("usm_r1_thr", "linear", 43, 61, 83, ref.adjust_sharpen, dict(amount=20.0, radius=1.0, threshold=5.0))
    add("usm_r15_a80", "wide", 44, 61, 83, ref.adjust_sharpen, dict(amount=80.0, radius=1.5, threshold=2.0))
    add("usm_lab_input", "lab", 45, 48, 64, ref.adjust_sharpen, dict(amount=40.0, radius=1.0, threshold=0.0))
    # a10 apply_all_adjustments (editingTools.py:823), WB neutral, denoise off
    add("all_basic", "linear", 51, 96, 128, ref.apply_all_adjustments,
        dict(exposure=0.3, shadows=28.0, highlights=-0.6, sharpen_amount=40.0, sharpen_radius=1.0))
    add("all_exposure_only", "linear", 52, 48, 64, ref.apply_all_adjustments, dict(exposure=-0.5))
    add("all_chart_mode_like", "linear", 53, 61, 83, ref.apply_all_adjustments,
        dict(exposure=0.25, shadows=35.0, highlights=0.4))

    # a7 _rgb_to_lab / _lab_to_rgb (editingTools.py:367, 389)
    add("rgb_to_lab_linear", "linear", 61, 48, 64, ref._rgb_to_lab, {})
    add("rgb_to_lab_wide", "wide", 62, 33, 47, ref._rgb_to_lab, {})
    add("lab_to_rgb", "lab", 63, 48, 64, lambda lab: ref._lab_to_rgb(lab[..., 0], lab[..., 1], lab[..., 2]), {})
    # white balance picked from a point (editingTools.py:643)
    add("wb_sample_centre", "linear", 64, 48, 64, ref.sample_white_balance_from_point,
        dict(sample_x=30, sample_y=20, current_temp=5500.0))
    add("wb_sample_corner", "negative", 65, 48, 64, ref.sample_white_balance_from_point,
        dict(sample_x=0, sample_y=47, current_temp=4200.0, sample_radius=8))
    add("wb_sample_wide", "wide", 66, 48, 64, ref.sample_white_balance_from_point,
        dict(sample_x=63, sample_y=3, current_temp=9000.0, sample_radius=2))

    meta = {}
    arrays = {}
    for name, c in cases.items():
        arrays[name] = c["out"]
        meta[name] = dict(gen=c["gen"], seed=c["seed"], h=c["h"], w=c["w"], kwargs=c["kwargs"],
                          stride=c["stride"], dtype=str(c["out"].dtype))
    import json

    np.savez_compressed(os.path.join(OUT_DIR, "editing_tools.npz"), **arrays)
    with open(os.path.join(OUT_DIR, "editing_tools.json"), "w") as f:
        json.dump(dict(numpy=np.__version__, reference_file="ImageProcessor/editingTools.py", cases=meta), f, indent=1)
    print(f"wrote {len(cases)} cases to {OUT_DIR}")
    for name, c in cases.items():
        print(f"  {name:24s} {c['out'].dtype} {c['out'].shape}")


if __name__ == "__main__":
    main()
