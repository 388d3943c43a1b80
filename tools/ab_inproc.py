#!/usr/bin/env python3
"""Same-process A/B of library builds: the launches of the builds alternate (A B C A B C ...), so clock drift and
box-to-box differences hit all of them alike.  Each build is a separate dlopen of its own .so.

  python tools/ab_inproc.py [--config 2] [--batch 64] [--rounds 60] LIB_A.so LIB_B.so [...]

Prints median / mean ms per launch and the paired difference of every build against the first.
Configs: 2 = BASELINE configs[1] chain (amount 1), 2b = amount 50, 4 = root-polynomial degree 2 + shadows/highlights
(45 MP), 1 = pointwise CCM + sRGB float32.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from scanspaceimageprocessor_amd import _lib  # noqa: E402
from scanspaceimageprocessor_amd.chain import make_desc  # noqa: E402
from scanspaceimageprocessor_amd.colour_fit import matrix_colour_correction  # noqa: E402
from tools.bench_kernels import swatches  # noqa: E402


class Build:
    def __init__(self, path):
        self.path = path
        self.lib = C.CDLL(os.path.abspath(path))
        for name, (res, args) in _lib.SIGNATURES.items():
            if hasattr(self.lib, name):
                fn = getattr(self.lib, name)
                fn.restype, fn.argtypes = res, args
        self.ctx = C.c_void_p()
        assert self.lib.ss_ctx_create(0, C.byref(self.ctx)) == 0, path

    def run(self, desc, src, dst):
        rc = self.lib.ss_chain_run(self.ctx, C.byref(desc), C.c_void_p(src), C.c_void_p(dst))
        assert rc == 0, (self.path, rc, self.lib.ss_last_error(self.ctx))

    def timed(self, desc, src, dst):
        self.lib.ss_timer_start(self.ctx)
        self.run(desc, src, dst)
        ms = C.c_float()
        self.lib.ss_timer_stop(self.ctx, C.byref(ms))
        return ms.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=60)
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    builds = [Build(p) for p in a.libs]
    M_T, M_R = swatches()
    lin3 = matrix_colour_correction(M_T, M_R)
    H, W, B = 4000, 6000, a.batch
    if a.config in ("4", "4b"):
        H, W, B = 5504, 8256, max(1, a.batch // 4)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    img = torch.rand((B, H, W, 3), generator=g, device=dev, dtype=torch.float32).pow_(2.2).mul_(0.9).contiguous()
    outs = [torch.empty((B, H, W, 3), dtype=torch.float32, device=dev) for _ in builds]
    if a.config == "2":
        d = make_desc(H, W, B, ccm=lin3, gains=[1.1], usm=(1.0, 2.0, 0.0), final_clip=True, out="u16")
    elif a.config == "2b":
        d = make_desc(H, W, B, ccm=lin3, gains=[1.1], usm=(50.0, 2.0, 0.0), final_clip=True, out="u16")
    elif a.config == "1":
        d = make_desc(H, W, B, ccm=lin3, gains=[1.0], out="f32_srgb")
    elif a.config in ("4", "4b"):
        ccm = matrix_colour_correction(M_T, M_R, method="Finlayson 2015", degree=2 if a.config == "4" else 3)
        d = make_desc(H, W, B, ccm=ccm, gains=[1.0], sh=(0.5, -0.5, 7.0), final_clip=True, out="f32")
    else:
        raise SystemExit("unknown config")
    torch.cuda.synchronize()  # (the builds' streams do not wait for torch's: the input must be complete first)
    for _ in range(3):
        for b, o in zip(builds, outs):
            b.run(d, img.data_ptr(), o.data_ptr())
            b.lib.ss_ctx_sync(b.ctx)
    torch.cuda.synchronize()
    # results of the builds against the first (bytes that differ)
    for b, o in zip(builds[1:], outs[1:]):
        n = B * H * W * 3 * (2 if a.config.startswith("2") else 4)
        x = outs[0].view(torch.uint8).flatten()[:n]
        y = o.view(torch.uint8).flatten()[:n]
        step = 1 << 30  # (comparisons of more than 2^32 elements at once came back wrong on this torch build)
        diff = sum(int((x[i:i + step] != y[i:i + step]).sum()) for i in range(0, n, step))
        line = f"{os.path.basename(b.path)}: {diff} of {n} output bytes differ from {os.path.basename(builds[0].path)}"
        if a.config.startswith("2") and diff:
            m = 0
            for i in range(0, n // 2, step):
                xa = outs[0].view(torch.int16).flatten()[i:i + step].to(torch.int32) & 0xFFFF
                ya = o.view(torch.int16).flatten()[i:i + step].to(torch.int32) & 0xFFFF
                m = max(m, int((xa - ya).abs().max()))
            line += f" (largest difference {m} steps of 16 bits)"
        print(line)
    ts = [[] for _ in builds]
    for r in range(a.rounds):
        order = list(range(len(builds)))
        if r & 1:
            order.reverse()
        for i in order:
            ts[i].append(builds[i].timed(d, img.data_ptr(), outs[i].data_ptr()))
    ts = np.array(ts)
    for i, b in enumerate(builds):
        line = f"{os.path.basename(b.path):44s} median {np.median(ts[i]):8.4f} ms  mean {ts[i].mean():8.4f}  min {ts[i].min():8.4f}"
        if i:
            rel = (ts[i] - ts[0]) / ts[0]
            line += f"   vs first: {100 * np.median(rel):+.2f} % (mean {100 * rel.mean():+.2f} %, sd {100 * rel.std() / np.sqrt(len(rel)):.2f})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
