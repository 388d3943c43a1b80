"""Replay one fuzz seed and print one pixel of the device result: python tools/fuzz_pixel.py SEED IMG Y X
(with SS_CAREFUL=0 a recorded pixel of a sharpening chain shows the stashed blur value in its first channel)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import test_gpu_parity as tp  # noqa: E402
from gpu_util import run_chain_dev  # noqa: E402
from oracle import inputs  # noqa: E402
from scanspaceimageprocessor_amd import default_context  # noqa: E402

seed, i, y, x = map(int, sys.argv[1:5])
M_T, M_R = inputs.synthetic_swatches()
imgs, kw, recipe = tp.fuzz_case(seed, M_T, M_R)
got = run_chain_dev(imgs, **kw)
want = tp._oracle_chain(imgs[i], M_T, M_R, *recipe)
print("requeued", default_context().chain_requeued(), "of", imgs.size // 3)
print("got ", got[i][y, x], "\nwant", want[y, x])
