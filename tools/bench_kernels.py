#!/usr/bin/env python3
"""Kernel-level timings of the BASELINE.json configurations on one GPU (HIP events on the library's
stream, inputs resident in HBM).  Prints one line per configuration: ms per launch, MP/s and the
algorithmic-bytes rate against the 8 TB/s HBM peak.

  python tools/bench_kernels.py [--batch 8] [--reps 5] [--configs 1,2,2b,3,4,4b,5s]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from scanspaceimageprocessor_amd import Context, _lib  # noqa: E402
from scanspaceimageprocessor_amd.chain import make_desc  # noqa: E402
from scanspaceimageprocessor_amd.colour_fit import matrix_colour_correction  # noqa: E402


def swatches():
    r = np.random.default_rng(7)
    M_R = 0.03 + 0.87 * r.random((24, 3))
    c = np.random.default_rng(8)
    A = np.eye(3) + 0.15 * (c.random((3, 3)) * 2.0 - 1.0)
    return (M_R @ A.T + c.normal(0.0, 0.002, size=(24, 3))).astype(np.float32), M_R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--requeued", action="store_true", help="print how many pixels the last chain call evaluated again in float64")
    ap.add_argument("--configs", default="1,2,2b,3,4,4b")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    M_T, M_R = swatches()
    lin3 = matrix_colour_correction(M_T, M_R)

    def images(h, w, b, negative=False):
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        x = torch.rand((b, h, w, 3), generator=g, device=dev, dtype=torch.float32).pow_(2.2).mul_(0.9)
        if negative:
            x = 0.08 + 0.8 * (1.0 - x) * torch.tensor([1.0, 0.75, 0.55], device=dev)
        return x.contiguous()

    def timeit(fn):
        fn()
        ctx.sync()
        ts = []
        for _ in range(args.reps):
            ctx.timer_start()
            fn()
            ts.append(ctx.timer_stop())
        return float(np.median(ts)), float(np.min(ts))

    def report(name, h, w, b, bytes_px, ms, ms_min):
        mp = b * h * w / 1e6
        gbs = bytes_px * b * h * w / (ms * 1e-3) / 1e9
        print(f"{name:34s} {b}x{h}x{w}  {ms:8.3f} ms (min {ms_min:8.3f})  {mp / (ms * 1e-3):10.0f} MP/s  "
              f"{gbs:7.0f} GB/s algorithmic = {gbs / 8000:.3f} of HBM peak", flush=True)
        if args.requeued:
            q = ctx.chain_requeued()
            print(f"     evaluated again in float64 by the last chain call: {q} pixels = {100.0 * q / (b * h * w):.2f} %", flush=True)

    todo = args.configs.split(",")
    H, W, B = 4000, 6000, args.batch
    if any(c in todo for c in ("1", "2", "2b", "2r1", "3", "1u16", "1u8", "1c", "5", "5p", "5h", "5b", "5c", "t", "m", "l")):
        img = images(H, W, B)
        out = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev)
    if "1" in todo:
        d = make_desc(H, W, B, ccm=lin3, gains=[1.0], out="f32_srgb")
        report("1  ccm3+clip+srgb f32", H, W, B, 24, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
    if "1u16" in todo:
        d = make_desc(H, W, B, ccm=lin3, gains=[1.0], out="u16")
        report("1u16 ccm3+clip+srgb u16", H, W, B, 18, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
    if "1u8" in todo:
        d = make_desc(H, W, B, ccm=lin3, gains=[1.0], out="u8")
        report("1u8 ccm3+clip+srgb u8", H, W, B, 15, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
    if "2" in todo:
        d = make_desc(H, W, B, ccm=lin3, gains=[1.1], usm=(1.0, 2.0, 0.0), final_clip=True, out="u16")
        report("2  ccm3+gain+usm(r2,a1)+u16", H, W, B, 18, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
    if "2b" in todo:
        d = make_desc(H, W, B, ccm=lin3, gains=[1.1], usm=(50.0, 2.0, 0.0), final_clip=True, out="u16")
        report("2b ccm3+gain+usm(r2,a50)+u16", H, W, B, 18, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
    if "2r1" in todo:
        d = make_desc(H, W, B, ccm=lin3, gains=[1.1], usm=(50.0, 1.0, 0.0), final_clip=True, out="u16")
        report("2r1 ccm3+gain+usm(r1,a50)+u16", H, W, B, 18, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
    if "5p" in todo:  # the worker's no-chart path with exposure and sharpen only
        d = make_desc(H, W, B, gains=[1.2], usm=(40.0, 1.0, 0.0), final_clip=True, out="u16")
        report("5p gain+usm(r1,a40)+u16", H, W, B, 18, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
    if "5" in todo:  # the worker's no-chart path with every adjustment on: exposure, S/H, sharpen radius 1
        d = make_desc(H, W, B, gains=[1.2], sh=(0.5 / 70, -0.5 * 1.3, 7.0), usm=(40.0, 1.0, 0.0), final_clip=True, out="u16")
        report("5  gain+S/H+usm(r1,a40)+u16", H, W, B, 18, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
    if "5h" in todo:  # the expensive case of the float64 re-evaluation: strong highlight compression + sharpen on a
        # photograph-like input (smooth saturated colour fields, edges, 1 % noise): ~12 % of the pixels are recorded
        yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
        smooth = torch.stack([0.5 + 0.5 * torch.sin(7 * xx + 3 * yy), 0.5 + 0.5 * torch.sin(5 * yy - 2 * xx + 1.0),
                              0.5 + 0.5 * torch.cos(9 * xx * yy + 0.5)], dim=-1)
        smooth = (smooth * (0.15 + 0.85 * ((xx * 8).floor() % 2 == (yy * 6).floor() % 2).float().unsqueeze(-1))).pow(2.2) * 0.9
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        smooth = (smooth * (1.0 + 0.01 * torch.randn((H, W, 3), generator=g, device=dev))).clamp_(0, 1)
        for i in range(B):
            img[i].copy_(smooth if i % 2 == 0 else smooth.flip(0))
        del smooth, yy, xx
        d = make_desc(H, W, B, gains=[1.2], sh=(0.4 / 70, -0.3 * 1.3, 7.0), usm=(40.0, 1.0, 0.0), final_clip=True, out="u16")
        report("5h gain+S/H+usm(r1,a40)+u16, photo-like", H, W, B, 18, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
        img.copy_(images(H, W, B))
    if "5b" in todo:  # chart path + S/H + sharpen: 3-term CCM, not the plain specialisation
        d = make_desc(H, W, B, ccm=lin3, gains=[1.1], sh=(0.5 / 70, -0.5 * 1.3, 7.0), usm=(40.0, 2.0, 1.0), final_clip=True, out="u16")
        report("5b ccm3+gain+S/H+usm(r2,a40,t1)+u16", H, W, B, 18, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
    if "5c" in todo:  # table-driven colour correction in front of the unsharp mask
        ccm = matrix_colour_correction(M_T, M_R, method="Finlayson 2015", degree=3)
        d = make_desc(H, W, B, ccm=ccm, gains=[1.1], usm=(50.0, 2.0, 0.0), final_clip=True, out="u16")
        report("5c rootpoly(d3)+gain+usm(r2,a50)+u16", H, W, B, 18, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
    if "1c" in todo:  # colour corrections with long term lists (coefficient table in LDS), pointwise
        for name, kw in (("cheung(22)", dict(method="Cheung 2004", terms=22)),
                         ("rootpoly(d4)", dict(method="Finlayson 2015", degree=4))):
            ccm = matrix_colour_correction(M_T, M_R, **kw)
            d = make_desc(H, W, B, ccm=ccm, gains=[1.0], out="f32_srgb")
            report(f"1c {name}+clip+srgb f32", H, W, B, 24, *timeit(lambda: ctx.chain_run(d, img.data_ptr(), out.data_ptr())))
    if "l" in todo:  # the Lab helpers as operators of their own: interleaved RGB <-> planes L, a, b
        n = B * H * W
        report("l  _rgb_to_lab", H, W, B, 24, *timeit(lambda: ctx.rgb_to_lab(
            img.data_ptr(), n, out.data_ptr(), out.data_ptr() + 4 * n, out.data_ptr() + 8 * n)))
        back = torch.empty_like(img)
        report("l  _lab_to_rgb", H, W, B, 24, *timeit(lambda: ctx.lab_to_rgb(
            out.data_ptr(), out.data_ptr() + 4 * n, out.data_ptr() + 8 * n, n, back.data_ptr())))
        del back
    if "t" in todo:  # preview / thumbnail producers: 24 MP -> 512 x 341 and -> 1 MP
        for name, (dw, dh) in (("512px thumbnail", (512, 341)), ("1 MP preview", (1224, 816)), ("4 MP preview", (2449, 1632))):
            small = torch.empty((B, dh, dw, 3), dtype=torch.float32, device=dev)
            report(f"t  area resize -> {name}", H, W, B, 12 + 12 * dw * dh / (H * W),
                   *timeit(lambda: ctx.resize_area(img.data_ptr(), H, W, B, small.data_ptr(), dh, dw)))
    if "m" in todo:  # region masking: an object on a dark backdrop, and white noise (many tiny components)
        yy = torch.arange(H, device=dev, dtype=torch.float32)[:, None]
        xx = torch.arange(W, device=dev, dtype=torch.float32)[None, :]
        inside = (((yy - H / 2) / (0.38 * H)) ** 2 + ((xx - W / 2) / (0.36 * W)) ** 2 <= 1.0)[None, :, :, None]
        scene = torch.where(inside, 0.08 + 0.8 * img, 0.01 * img).contiguous()
        work = torch.empty_like(scene)
        cnt = torch.zeros(B, dtype=torch.int64, device=dev)
        for name, data, lims in (("m  mask: object on dark backdrop", scene, (0.05, 0.98)), ("m  mask: noise", img, (0.05, 0.6))):
            work.copy_(data)
            torch.cuda.synchronize()
            # in place and repeatable: zeroed pixels stay candidates, so every repetition labels the same mask
            report(name, H, W, B, 12, *timeit(lambda: ctx.mask_regions(work.data_ptr(), H, W, B, lims[0], lims[1], 200,
                                                                        cnt.data_ptr())))
            ctx.sync()
            print(f"     masked pixels per image: {cnt.tolist()[:3]} ...")
        del scene, work
    if "3" in todo:
        neg = images(H, W, B, negative=True)
        params = torch.empty((B, _lib.SS_NEG_PARAM_STRIDE), dtype=torch.float32, device=dev)
        d = make_desc(H, W, B, neg=True, neg_params_dev=params.data_ptr(), ccm=lin3, out="f32")

        def run3():
            ctx.negative_params(neg.data_ptr(), H, W, B, 0.005, 0.995, True, params.data_ptr())
            ctx.chain_run(d, neg.data_ptr(), out.data_ptr())

        report("3  percentiles+invert+ccm3 f32", H, W, B, 24.15, *timeit(run3))
        report("3p percentiles only", H, W, B, 12.0 / 81,
               *timeit(lambda: ctx.negative_params(neg.data_ptr(), H, W, B, 0.005, 0.995, True, params.data_ptr())))
        del neg
    if "4" in todo or "4b" in todo:
        H4, W4, B4 = 5504, 8256, max(1, args.batch // 2)
        img4 = images(H4, W4, B4)
        out4 = torch.empty_like(img4)
        for name, deg in (("4", 2), ("4b", 3)):
            if name in todo:
                ccm = matrix_colour_correction(M_T, M_R, method="Finlayson 2015", degree=deg)
                d = make_desc(H4, W4, B4, ccm=ccm, gains=[1.0], sh=(0.5, -0.5, 7.0), final_clip=True, out="f32")
                report(f"{name:2s} rootpoly(d{deg})+clip+S/H f32", H4, W4, B4, 24,
                       *timeit(lambda: ctx.chain_run(d, img4.data_ptr(), out4.data_ptr())))


if __name__ == "__main__":
    main()
