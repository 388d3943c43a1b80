"""BASELINE configs[4] (dataset-wide exposure normalisation in front of the chain) at full image size, one GPU's
shard: 24 MP images -> thumbnails made on the device (ss_resize_area) -> masked-luminance statistics
(ss_exposure_stats) -> multipliers (ss_exposure_multipliers) -> CCM + per-image gain + clip + sRGB, float32
(ss_chain_run with gain0_dev).  Reference: image_space.py:4323-4398 (statistics and multipliers),
imageProcessorWorker.py:863-876 (the gain), through the C ABI.

Run twice: as the whole dataset on one rank, and as rank 1 of 2 (the second half of the images; the first
half's brightness values arrive through the gather), which exercises the (first, count) arithmetic of
ExposureNormaliser and of ss_exposure_multipliers."""

import numpy as np
import pytest

from gpu_util import DeviceBuffer, fit
from oracle import chain as oc
from oracle import inputs
from scanspaceimageprocessor_amd import default_context
from scanspaceimageprocessor_amd.chain import make_desc
from scanspaceimageprocessor_amd.dataset import ExposureNormaliser, shard_range

pytestmark = pytest.mark.gpu

H, W = 4000, 6000
TH, TW = 341, 512
N_IMAGES = 8
LO, HI = 0.06, 0.96
F32_TOL = 1e-5


@pytest.fixture(scope="module")
def dataset():
    """Eight 24 MP images (the benchmark's generator, seeds 1234 + i) with per-image exposure scales 2**U(-0.5, 0.5)
    resident on the GPU, their device-made thumbnails, and the oracle's statistics of those thumbnails."""
    ctx = default_context()
    scales = 2.0 ** np.random.default_rng(9).uniform(-0.5, 0.5, size=N_IMAGES)
    img_bytes = H * W * 12
    images = DeviceBuffer(ctx, N_IMAGES * img_bytes)
    rng = np.random.default_rng(77)
    ys = rng.integers(0, H, 25_000)
    xs = rng.integers(0, W, 25_000)
    ys[:4], xs[:4] = [0, 0, H - 1, H - 1], [0, W - 1, 0, W - 1]
    samples = []
    for i in range(N_IMAGES):
        img = np.clip(inputs.linear_image(1234 + i, H, W) * np.float32(scales[i]), 0.0, 1.0).astype(np.float32)
        ctx.upload(images.ptr + i * img_bytes, img)
        samples.append(img[ys, xs].copy())
        del img
    thumbs = DeviceBuffer(ctx, N_IMAGES * TH * TW * 12)
    ctx.resize_area(images.ptr, H, W, N_IMAGES, thumbs.ptr, TH, TW)
    ctx.sync()
    thumbs_host = thumbs.to_array((N_IMAGES, TH, TW, 3), np.float32)
    want_b = [oc.image_brightness(t, LO, HI) for t in thumbs_host]
    yield dict(ctx=ctx, images=images, thumbs=thumbs, thumbs_host=thumbs_host, want_b=want_b, ys=ys, xs=xs,
               samples=samples, img_bytes=img_bytes)
    images.free()
    thumbs.free()


def test_device_thumbnails_carry_the_exposure_differences(dataset):
    b = np.array(dataset["want_b"], np.float64)
    assert b.min() > 0.05 and b.max() / b.min() > 1.3  # the normalisation has work to do


@pytest.mark.parametrize("rank,world", [(0, 1), (1, 2)])
def test_config5_shard_at_24mp(dataset, rank, world):
    import torch

    ctx = dataset["ctx"]
    M_T, M_R = inputs.synthetic_swatches()
    want_m = oc.exposure_multipliers(dataset["want_b"], None)
    first, count = shard_range(N_IMAGES, rank, world)
    dev = torch.device("cuda", ctx.device)

    def gather(local, n_images):
        # the other shard's values as its rank would have computed them: the same kernel on its thumbnails
        other_first, other_count = shard_range(N_IMAGES, 1 - rank, world)
        other = torch.empty((other_count,), dtype=torch.float32, device=dev)
        ctx.exposure_stats(dataset["thumbs"].ptr + other_first * TH * TW * 12, TH, TW, other_count, LO, HI, None,
                           other.data_ptr())
        parts = [other, local] if other_first < first else [local, other]
        return torch.cat(parts)

    norm = ExposureNormaliser(ctx, N_IMAGES, rank, world, LO, HI, gather=gather)
    assert (norm.first, norm.count) == (first, count)
    gains = norm.run(dataset["thumbs"].ptr + first * TH * TW * 12, TH, TW)
    out = DeviceBuffer(ctx, count * dataset["img_bytes"])
    try:
        desc = make_desc(H, W, count, ccm=fit(M_T, M_R), gains=[1.0], gain0_dev=gains.data_ptr(), final_clip=True,
                         out="f32_srgb")
        ctx.chain_run(desc, dataset["images"].ptr + first * dataset["img_bytes"], out.ptr)
        ctx.sync()
        np.testing.assert_allclose(norm.brightness_all.cpu().numpy(), np.array(dataset["want_b"], np.float32), rtol=1e-6)
        np.testing.assert_allclose(gains.cpu().numpy(), np.array(want_m[first:first + count]), rtol=1e-6)
        got = out.to_array((count, H, W, 3), np.float32)
        ys, xs = dataset["ys"], dataset["xs"]
        worst = 0.0
        for k in range(count):
            px = dataset["samples"][first + k][None].astype(np.float32)
            # imageProcessorWorker.py:858-876: correction, multiplier, clip; then the float sRGB encoding
            c = np.clip(oc.lv.colour_correction(px, M_T, M_R) * float(want_m[first + k]), 0, 1)
            want = oc.lv.cctf_encoding(c)[0]
            worst = max(worst, float(np.max(np.abs(got[k][ys, xs] - want))))
        assert worst <= F32_TOL, worst
        assert count * len(ys) >= 100_000
    finally:
        out.free()
