"""Run a range of seeds of tests/test_gpu_parity.py::test_random_stage_combinations outside pytest and tally the
failures by kind of chain (a survey for tuning the criteria of the float64 re-evaluation).
Usage: python tools/fuzz_survey.py FIRST COUNT"""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import test_gpu_parity as tp  # noqa: E402
from gpu_util import run_chain_dev  # noqa: E402
from oracle import inputs  # noqa: E402
from scanspaceimageprocessor_amd import default_context  # noqa: E402


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    M_T, M_R = inputs.synthetic_swatches()
    fails = collections.Counter()
    total = collections.Counter()
    worst = {}
    requeued = collections.defaultdict(list)
    for seed in range(first, first + count):
        imgs, kw, recipe = tp.fuzz_case(seed, M_T, M_R)
        neg, ccm_kw, gains, sh, usm, final_clip, out = recipe
        got = run_chain_dev(imgs, **kw)
        key = (str(ccm_kw), out, sh is not None, usm is not None)
        total[key] += 1
        requeued[key].append(default_context().chain_requeued() / imgs[..., 0].size)
        bad = 0.0
        for i in range(imgs.shape[0]):
            want = tp._oracle_chain(imgs[i], M_T, M_R, *recipe)
            if out in ("u16", "u8"):
                scale = 65535.0 if out == "u16" else 255.0
                diff = np.abs(got[i].astype(np.int64) - want.astype(np.int64))
                extra = 0.0
                if usm is not None:
                    extra = tp._blur_allowance(lambda: tp._oracle_chain(imgs[i], M_T, M_R, *recipe[:-1], "f32_srgb") * scale)
                over = diff - (1 + np.floor(tp.F32_TOL * scale + extra))
            else:
                err = np.abs(got[i] - want)
                extra = tp._blur_allowance(lambda: tp._oracle_chain(imgs[i], M_T, M_R, *recipe)) if usm is not None else 0.0
                over = (err - (tp.F32_TOL + extra)) / tp.F32_TOL
            bad = max(bad, float(np.max(over)))
        if bad > 0:
            fails[key] += 1
            worst[key] = max(worst.get(key, 0.0), bad)
    for key in sorted(total):
        q = requeued[key]
        print(f"{key}: {fails[key]} of {total[key]} fail, worst excess {worst.get(key, 0.0):.3g}, "
              f"re-evaluated share mean {np.mean(q):.4f} max {np.max(q):.4f}")
    print("failing cases:", sum(fails.values()), "of", sum(total.values()))


if __name__ == "__main__":
    main()
