#!/usr/bin/env python3
"""Run a chain twice into two buffers (one stream, a synchronisation after each run) and compare the outputs byte for
byte: python tools/determinism_check.py [config ...]   (configs of tools/ab_inproc.py: 2, 2b, ...)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from scanspaceimageprocessor_amd import Context  # noqa: E402
from scanspaceimageprocessor_amd.chain import make_desc  # noqa: E402
from scanspaceimageprocessor_amd.colour_fit import matrix_colour_correction  # noqa: E402
from tools.bench_kernels import swatches  # noqa: E402

dev = torch.device("cuda", 0)
ctx = Context(0)
M_T, M_R = swatches()
lin3 = matrix_colour_correction(M_T, M_R)
H, W, B = 4000, 6000, 8
g = torch.Generator(device=dev)
g.manual_seed(1234)
img = torch.rand((B, H, W, 3), generator=g, device=dev, dtype=torch.float32).pow_(2.2).mul_(0.9).contiguous()
for cfg in sys.argv[1:] or ["2", "2b"]:
    amount = 1.0 if cfg == "2" else 50.0
    d = make_desc(H, W, B, ccm=lin3, gains=[1.1], usm=(amount, 2.0, 0.0), final_clip=True, out="u16")
    outs = []
    for rep in range(3):
        o = torch.zeros((B, H, W, 3), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()  # (the library's stream does not wait for torch's: the fill must be over first)
        ctx.chain_run(d, img.data_ptr(), o.data_ptr())
        ctx.sync()
        torch.cuda.synchronize()
        outs.append(o)
    for rep in (1, 2):
        a, b = outs[0].to(torch.int32) & 0xFFFF, outs[rep].to(torch.int32) & 0xFFFF
        nd = int((a != b).sum())
        print(f"config {cfg}: run {rep} against run 0: {nd} of {a.numel()} values differ, largest {int((a - b).abs().max())}", flush=True)
    print("  re-evaluated pixels of the last run:", ctx.chain_requeued())
