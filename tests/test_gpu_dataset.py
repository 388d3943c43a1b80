"""Dataset exposure normalisation on the device with the collective in the loop: a one-rank RCCL group on the
one GPU of the test box exercises the same code the multi-GPU runs use (statistics kernel -> all-gather issued
on the context's own stream through an ExternalStream -> multiplier kernel)."""

import os

import numpy as np
import pytest

from oracle import chain as oc
from oracle import inputs
from scanspaceimageprocessor_amd import default_context
from scanspaceimageprocessor_amd.dataset import ExposureNormaliser

pytestmark = pytest.mark.gpu


def test_exposure_normaliser_with_an_rccl_all_gather_on_the_context_stream():
    import torch
    import torch.distributed as dist

    ctx = default_context()
    dev = torch.device("cuda", ctx.device)
    assert ctx.stream != 0
    n, h, w = 6, 96, 128
    rng = np.random.default_rng(3)
    thumbs = np.stack([inputs.linear_image(70 + i, h, w) * np.float32(2.0 ** rng.uniform(-0.5, 0.5)) for i in range(n)])
    t = torch.from_numpy(thumbs).to(dev)
    want_b = [oc.image_brightness(x, 0.06, 0.96) for x in thumbs]
    want_m = oc.exposure_multipliers(want_b, None)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29531", rank=0, world_size=1, device_id=dev)
    try:
        norm = ExposureNormaliser(ctx, n, 0, 1, 0.06, 0.96)
        norm.world = 2  # take the collective branch; the group itself has one rank, so the gather is an identity
        import scanspaceimageprocessor_amd.dataset as ds

        real = ds.gather_brightness

        def gather_one_rank(local, n_images, group=None):
            out = torch.empty_like(local)
            dist.all_gather_into_tensor(out, local, group=group)  # issued on the current stream = ctx stream
            assert torch.cuda.current_stream(dev).cuda_stream == ctx.stream
            return out

        ds.gather_brightness = gather_one_rank
        try:
            for _ in range(3):
                gains = norm.run(t.data_ptr(), h, w)
        finally:
            ds.gather_brightness = real
        ctx.sync()
        np.testing.assert_allclose(gains.cpu().numpy(), want_m, rtol=1e-6)
        np.testing.assert_allclose(norm.brightness_all.cpu().numpy(), want_b, rtol=1e-6)
    finally:
        dist.destroy_process_group()
