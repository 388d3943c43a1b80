import sys, numpy as np
sys.path.insert(0, ".") ; sys.path.insert(0, "tests")
from gpu_util import fit, run_chain_dev
from oracle import chain as oc, inputs, leaves as lv
H24, W24 = 4000, 6000
def textured():
    """Low-frequency structure + noise, built in float32 row blocks to stay light on host memory."""
    rng = np.random.default_rng(2024)
    img = np.empty((H24, W24, 3), np.float32)
    xx = np.arange(W24, dtype=np.float32)[None, :]
    for y0 in range(0, H24, 500):
        yy = np.arange(y0, y0 + 500, dtype=np.float32)[:, None]
        base = 0.45 + 0.4 * np.sin(xx / 17.0) * np.cos(yy / 23.0)
        blk = np.stack([base, np.roll(base, 5, axis=1), 0.9 - base], axis=-1)
        blk += rng.normal(0.0, 0.03, size=blk.shape).astype(np.float32)
        img[y0:y0 + 500] = np.clip(blk, 0.0, 1.0)
    return img



M_T, M_R = inputs.synthetic_swatches()
img = textured()
amount, mult, radius, margin = 50.0, 1.1, 2.0, 16
got = run_chain_dev(img, ccm=fit(M_T, M_R), gains=[mult], usm=(amount, radius, 0.0), final_clip=True, out="u16")[0]
gotf = run_chain_dev(img, ccm=fit(M_T, M_R), gains=[mult], usm=(amount, radius, 0.0), final_clip=True, out="f32_srgb")[0]
h, w = H24, W24
wins = [(0, 80, 0, 150), (0, 80, w - 150, w), (h - 80, h, 0, 150), (h - 80, h, w - 150, w),
        (0, 70, w // 2 - 70, w // 2 + 70), (h - 70, h, 2990, 3130), (1990, 2060, 0, 140), (1990, 2060, w - 200, w),
        (226, 290, 100, 280), (2016, 2080, 2990, 3130), (3808, 3872, 5850, w), (h // 2 - 40, h // 2 + 40, w // 2 - 70, w // 2 + 70)]
for (y0, y1, x0, x1) in wins:
    ey0, ey1 = max(0, y0 - margin), min(H24, y1 + margin)
    ex0, ex1 = max(0, x0 - margin), min(W24, x1 + margin)
    crop = img[ey0:ey1, ex0:ex1]
    c = np.clip(lv.colour_correction(crop, M_T, M_R) * mult, 0, 1)
    sharp = np.clip(oc.adjust_sharpen(c, amount, radius, 0.0), 0, 1)
    enc = lv.cctf_encoding(sharp) if hasattr(lv, "cctf_encoding") else None
    want = oc.config2(crop, M_T, M_R, mult, amount, radius, 0.0)[y0 - ey0:y1 - ey0, x0 - ex0:x1 - ex0]
    g = got[y0:y1, x0:x1].astype(np.int64)
    diff = np.abs(g - want.astype(np.int64))
    k = np.unravel_index(np.argmax(diff), diff.shape)
    yy, xx, ch = k
    print((y0, y1, x0, x1), "max", diff.max(), "n>1", int((diff > 1).sum()), "at", (y0 + yy, x0 + xx, ch), "got", g[k], "want", want[k],
          "sharp(lin) oracle", sharp[yy + y0 - ey0, xx + x0 - ex0], "c", c[yy + y0 - ey0, xx + x0 - ex0], "gpu f32 srgb", gotf[y0 + yy, x0 + xx])
