#!/usr/bin/env python3
"""Per-workgroup timeline of the unsharp-mask kernel (needs a library built with -DSS_USM_TRACE,
selected with SCANSPACE_LIB).  Prints start / end skew and the busy fraction of the workgroup slots."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scanspaceimageprocessor_amd import Context, _lib
from scanspaceimageprocessor_amd.chain import make_desc

H, W, B = 4000, 6000, 8
dev = torch.device("cuda", 0)
img = torch.rand((B, H, W, 3), device=dev)
out = torch.empty((B, H, W, 3), dtype=torch.int16, device=dev)
ctx = Context(0)
ccm = (np.eye(3) * 0.9 + 0.03, 1, 3)
desc = make_desc(H, W, B, ccm=ccm, gains=[1.1], usm=(1.0, 2.0, 0.0), final_clip=True, out="u16")
for _ in range(3):
    ctx.chain_run(desc, img.data_ptr(), out.data_ptr())
ctx.sync()
buf = np.zeros((1024, 4), np.uint64)
lib = _lib.load_library()
rc = lib.ss_debug_usm_trace(ctypes.c_void_p(buf.ctypes.data))
assert rc == 0, rc
n = int((buf[:, 1] > 0).sum())
t0 = buf[:n, 0].astype(np.int64)
t1 = buf[:n, 1].astype(np.int64)
base = t0.min()
t0 -= base
t1 -= base
tiles = buf[:n, 2].astype(np.int64)
xcc = (buf[:n, 3] >> np.uint64(32)).astype(np.int64)
hw = (buf[:n, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
tick = 1e-8  # wall_clock64 runs at 100 MHz
print(f"workgroups {n}, tiles/wg {tiles.min()}..{tiles.max()}")
print(f"start  : min {t0.min()*tick*1e3:.3f} ms  p50 {np.median(t0)*tick*1e3:.3f}  max {t0.max()*tick*1e3:.3f}")
print(f"end    : min {t1.min()*tick*1e3:.3f} ms  p10 {np.percentile(t1,10)*tick*1e3:.3f} p50 {np.median(t1)*tick*1e3:.3f} "
      f"p90 {np.percentile(t1,90)*tick*1e3:.3f} max {t1.max()*tick*1e3:.3f}")
dur = (t1 - t0) * tick * 1e3
print(f"length : min {dur.min():.3f} ms  p50 {np.median(dur):.3f}  max {dur.max():.3f}; slot busy fraction {dur.sum()/(n*t1.max()*tick*1e3):.3f}")
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    cu = (hw[m] >> 8) & 0xF
    se = (hw[m] >> 13) & 0x7
    print(f"xcc {x}: wgs {m.sum()}, mean length {dur[m].mean():.3f} ms, end max {t1[m].max()*tick*1e3:.3f}, distinct (se,cu) {len(set(zip(se.tolist(), cu.tolist())))}")
