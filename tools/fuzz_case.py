"""Replay one seed of tests/test_gpu_parity.py::test_random_stage_combinations on the GPU and print where and
by how much the device result departs from the float64 oracle.  Usage: python tools/fuzz_case.py SEED [SEED ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import test_gpu_parity as tp  # noqa: E402
from gpu_util import run_chain_dev  # noqa: E402
from oracle import inputs  # noqa: E402


def main():
    M_T, M_R = inputs.synthetic_swatches()
    for seed in map(int, sys.argv[1:]):
        imgs, kw, recipe = tp.fuzz_case(seed, M_T, M_R)
        neg, ccm_kw, gains, sh, usm, final_clip, out = recipe
        got = run_chain_dev(imgs, **kw)
        print(f"seed {seed}: shape {imgs.shape} neg={neg is not None} ccm={ccm_kw} gains={len(gains)} sh={sh} usm={usm} "
              f"clip={final_clip} out={out}")
        for i in range(imgs.shape[0]):
            want = tp._oracle_chain(imgs[i], M_T, M_R, *recipe)
            d = np.abs(got[i].astype(np.float64) - want.astype(np.float64))
            idx = np.unravel_index(np.argmax(d), d.shape)
            print(f"  image {i}: max |diff| {d.max():.6g} at {idx}: got {got[i][idx]!r} want {want[idx]!r}; "
                  f"input pixel {imgs[i][idx[0], idx[1]]}, want pixel {want[idx[0], idx[1]]}")
            print(f"   pixel as computed: {got[i][idx[0], idx[1]]}")
            worst = np.argsort(d.ravel())[::-1][:5]
            print("   next worst:", [(np.unravel_index(k, d.shape), float(d.ravel()[k]), float(want.ravel()[k])) for k in worst])


if __name__ == "__main__":
    main()
