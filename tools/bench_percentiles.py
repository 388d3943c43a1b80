#!/usr/bin/env python3
"""Time ss_negative_params alone: tools/bench_percentiles.py [batch ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from scanspaceimageprocessor_amd import Context, _lib

H, W = 4000, 6000
ctx = Context(0)
for b in [int(x) for x in sys.argv[1:]] or [1, 8, 64]:
    img = torch.rand((b, H, W, 3), device="cuda")
    params = torch.empty((b, _lib.SS_NEG_PARAM_STRIDE), device="cuda")
    torch.cuda.synchronize()
    run = lambda: ctx.negative_params(img.data_ptr(), H, W, b, 0.005, 0.995, True, params.data_ptr())
    for _ in range(5):
        run()
    ctx.sync()
    ts = []
    for _ in range(30):
        ctx.timer_start()
        run()
        ts.append(ctx.timer_stop())
    print(f"batch {b:3d}: {np.median(ts) * 1e3:8.1f} us median, {min(ts) * 1e3:8.1f} us min", flush=True)
    del img
