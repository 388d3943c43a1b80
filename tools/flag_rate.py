# flag rates of the float64 re-evaluation on the benchmark-style configs (diagnostic)
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from scanspaceimageprocessor_amd import Context
from scanspaceimageprocessor_amd.chain import make_desc
from scanspaceimageprocessor_amd.colour_fit import matrix_colour_correction
r = np.random.default_rng(7); M_R = 0.03 + 0.87 * r.random((24, 3)); c = np.random.default_rng(8)
A = np.eye(3) + 0.15 * (c.random((3, 3)) * 2.0 - 1.0); M_T = (M_R @ A.T + c.normal(0.0, 0.002, size=(24, 3))).astype(np.float32)
lin3 = matrix_colour_correction(M_T, M_R)
dev = torch.device("cuda", 0); ctx = Context(0)
H, W, B = 4000, 6000, 2
g = torch.Generator(device=dev); g.manual_seed(1234)
img = torch.rand((B, H, W, 3), generator=g, device=dev).pow_(2.2).mul_(0.9).contiguous()
out = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev)
# a photograph-like input beside the noise: smooth colour fields (some saturated, some dark), edges, 1 % sensor noise
yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
smooth = torch.stack([0.5 + 0.5 * torch.sin(7 * xx + 3 * yy), 0.5 + 0.5 * torch.sin(5 * yy - 2 * xx + 1.0),
                      0.5 + 0.5 * torch.cos(9 * xx * yy + 0.5)], dim=-1)
smooth = (smooth * (0.15 + 0.85 * ((xx * 8).floor() % 2 == (yy * 6).floor() % 2).float().unsqueeze(-1))).pow(2.2) * 0.9
smooth = (smooth * (1.0 + 0.01 * torch.randn((H, W, 3), generator=g, device=dev))).clamp_(0, 1)
photo = torch.stack([smooth, smooth.flip(0)]).contiguous()
del smooth, yy, xx
inputs = {"noise": img, "photo-like": photo}
for name, kw in [("usm a1 u16", dict(ccm=lin3, gains=[1.1], usm=(1.0, 2.0, 0.0), final_clip=True, out="u16")),
                 ("usm a50 u16", dict(ccm=lin3, gains=[1.1], usm=(50.0, 2.0, 0.0), final_clip=True, out="u16")),
                 ("usm a80 f32", dict(ccm=lin3, gains=[1.1], usm=(80.0, 2.0, 0.0), final_clip=True, out="f32")),
                 ("sh .5/-.5 f32", dict(ccm=lin3, gains=[1.0], sh=(0.5, -0.5, 7.0), final_clip=True, out="f32")),
                 ("sh .5/-.5 u16", dict(ccm=lin3, gains=[1.0], sh=(0.5, -0.5, 7.0), final_clip=True, out="u16")),
                 ("sh+usm40 u16", dict(gains=[1.2], sh=(0.4/70, -0.3*1.3, 7.0), usm=(40.0, 1.0, 0.0), final_clip=True, out="u16"))]:
    d = make_desc(H, W, B, **kw)
    for label, data in inputs.items():
        ctx.chain_run(d, data.data_ptr(), out.data_ptr()); ctx.sync()
        ts = []
        for _ in range(5):
            ctx.timer_start(); ctx.chain_run(d, data.data_ptr(), out.data_ptr()); ts.append(ctx.timer_stop())
        q = ctx.chain_requeued()
        print(f"{name:16s} {label:10s} queued {q:9d} ({q / (B*H*W):.2e} of pixels)  {min(ts):.3f} ms per {B}x24MP", flush=True)
