#!/usr/bin/env python3
"""Wall time per image of ss_chain_run_host on a batch of 24 MP images in page-locked memory: upload, kernels and
download pipelined image by image on three streams (SS_HOST_PIPELINE=0 in the environment switches back to
whole-batch upload, one launch, whole-batch download).  PCIe bound; never the benchmark's headline value."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanspaceimageprocessor_amd import default_context
from scanspaceimageprocessor_amd.chain import make_desc

H, W, B = 4000, 6000, 8
ctx = default_context()
rng = np.random.default_rng(0)
one = (rng.random((H, W, 3), dtype=np.float32) ** 2.2 * 0.9).astype(np.float32)
pin_in = ctx.pinned_empty((B, H, W, 3), np.float32)
for i in range(B):
    pin_in[i] = one
ccm = (np.eye(3) * 0.9 + 0.03, 1, 3)
for name, kw, out, dtype in (("ccm+gain+usm -> u16", dict(ccm=ccm, gains=[1.1], usm=(40.0, 1.0, 0.0), final_clip=True), "u16", np.uint16),
                             ("ccm+gain -> f32", dict(ccm=ccm, gains=[1.1], final_clip=True), "f32", np.float32)):
    pin_out = ctx.pinned_empty((B, H, W, 3), dtype)
    d = make_desc(H, W, B, out=out, **kw)
    ctx.chain_run_host(d, pin_in, out=pin_out)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.chain_run_host(d, pin_in, out=pin_out)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts)) / B
    mode = "whole batch at once" if os.environ.get("SS_HOST_PIPELINE", "1")[0] == "0" else "pipelined"
    print(f"{B} x 24 MP page-locked, {name:22s} {mode:20s} {t * 1e3:7.2f} ms per image  {H * W / 1e6 / t:7.0f} MP/s", flush=True)
    del pin_out
