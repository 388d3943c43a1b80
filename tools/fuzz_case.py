"""One seed of tests/test_gpu_parity.py::test_random_stage_combinations outside pytest, with the worst value's
surroundings printed: recipe, position, got / want, the allowance there, the sharpen stage's intermediates.
Usage: python tools/fuzz_case.py SEED [SEED ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import test_gpu_parity as tp  # noqa: E402
from gpu_util import run_chain_dev  # noqa: E402
from oracle import chain as oc  # noqa: E402
from oracle import inputs  # noqa: E402
from oracle import leaves as lv  # noqa: E402
from scanspaceimageprocessor_amd import default_context  # noqa: E402


def one(seed):
    M_T, M_R = inputs.synthetic_swatches()
    imgs, kw, recipe = tp.fuzz_case(seed, M_T, M_R)
    neg, ccm_kw, gains, sh, usm, final_clip, out = recipe
    print(f"seed {seed}: shape {imgs.shape} neg {neg is not None} ccm {ccm_kw} gains {gains} sh {sh} usm {usm} "
          f"final_clip {final_clip} out {out}")
    got = run_chain_dev(imgs, **kw)
    print("  re-evaluated pixels:", default_context().chain_requeued())
    for i in range(imgs.shape[0]):
        want = tp._oracle_chain(imgs[i], M_T, M_R, *recipe)
        quant = out in ("u16", "u8")
        scale = (65535.0 if out == "u16" else 255.0) if quant else 1.0
        err = np.abs(got[i].astype(np.float64) - want.astype(np.float64))
        extra = np.zeros_like(err)
        if usm is not None:
            cont = recipe[:-1] + ("f32_srgb",) if quant else recipe
            extra = tp._blur_allowance(lambda: tp._oracle_chain(imgs[i], M_T, M_R, *cont) * scale)
        bar = (1 + np.floor(tp.F32_TOL * scale + extra)) if quant else tp.F32_TOL + extra
        over = err - bar
        y, x, c = np.unravel_index(np.argmax(over), over.shape)
        print(f"  image {i}: worst excess {over[y, x, c]:.3g} at (y {y}, x {x}, c {c}): got {got[i][y, x]} want {want[y, x]} "
              f"err {err[y, x]} bar {bar[y, x] if np.ndim(bar) else bar}; values over the bar: {int((over > 0).sum())}")
        if usm is not None and over[y, x, c] > 0:
            pre = imgs[i].astype(np.float64)
            if neg is not None:
                pre = 1.0 - np.clip((np.clip(pre, 0, 1) - np.asarray(neg[0], np.float64)) * np.asarray(neg[1], np.float64), 0, 1)
            if ccm_kw is not None:
                pre = lv.colour_correction(pre, M_T, M_R, **ccm_kw)
            for g in gains:
                pre = np.clip(pre * np.asarray(g, np.float64), 0, 1)
            if sh is not None:
                pre = oc.adjust_shadows_highlights_combined(pre, *sh)
            L, a, b = oc.rgb_to_lab(pre)
            Ln = L / 100.0
            blur = lv.gaussian_blur_reflect(Ln.astype(np.float32), lv.usm_kernel_size(usm[1]), usm[1])
            print(f"    sharpen input {pre[y, x]} L/100 {Ln[y, x]:.9f} blurred {blur[y, x]:.9f} mask {Ln[y, x] - blur[y, x]:.3e} "
                  f"a {a[y, x]:.4f} b {b[y, x]:.4f}")
            sharp = oc.adjust_sharpen(pre, *usm)
            print(f"    sharpened linear {sharp[y, x]}")


if __name__ == "__main__":
    for s in sys.argv[1:]:
        one(int(s))
