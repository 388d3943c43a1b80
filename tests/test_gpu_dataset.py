"""Dataset exposure normalisation on the device with the collective in the loop: a one-rank RCCL group on the
one GPU of the test box exercises the same code the multi-GPU runs use (statistics kernel -> all-gather issued
on the context's own stream through an ExternalStream -> multiplier kernel)."""

import os
import socket

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
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        # Two ranks' worth of the class on the one GPU: "rank 1 of 2" computes its own shard and receives the other
        # shard's brightness through the gather it is given, which runs a real RCCL all-gather (one-rank group) on
        # the context's stream for the local part.
        other = ExposureNormaliser(ctx, n, 0, 2, 0.06, 0.96, gather=lambda local, n_images: None)
        other.ops.stats(t.data_ptr(), h, w, other.count, 0.06, 0.96, other.brightness)

        def gather(local, n_images):
            assert torch.cuda.current_stream(dev).cuda_stream == ctx.stream
            out = torch.empty_like(local)
            dist.all_gather_into_tensor(out, local)  # issued on the current stream = the context's stream
            return torch.cat([other.brightness, out])

        norm = ExposureNormaliser(ctx, n, 1, 2, 0.06, 0.96, gather=gather)
        assert (norm.first, norm.count) == (3, 3)
        shard = t[norm.first:norm.first + norm.count]
        for _ in range(3):
            gains = norm.run(shard.data_ptr(), h, w)
        ctx.sync()
        np.testing.assert_allclose(gains.cpu().numpy(), want_m[norm.first:], rtol=1e-6)
        np.testing.assert_allclose(norm.brightness_all.cpu().numpy(), want_b, rtol=1e-6)
    finally:
        dist.destroy_process_group()


def test_calculate_average_exposure_over_mixed_thumbnail_shapes():
    """The main window's calculate_average_exposure (image_space.py:4323-4398) on a dict of thumbnails: landscape
    and portrait shapes, one thumbnail missing, reference = mean of all or the chart image."""
    from scanspaceimageprocessor_amd.dataset import calculate_average_exposure

    rng = np.random.default_rng(11)
    thumbs = {}
    for i in range(7):
        h, w = (341, 512) if i % 3 else (512, 341)
        thumbs[f"/d/img_{i}.nef"] = inputs.linear_image(40 + i, h, w) * np.float32(2.0 ** rng.uniform(-0.6, 0.6))
    thumbs["/d/img_missing.nef"] = None
    thumbs["/d/img_black.nef"] = np.zeros((341, 512, 3), np.float32)
    logs = []
    got = calculate_average_exposure(thumbs, 0.06, 0.96, log=logs.append)
    usable = [p for p, a in thumbs.items() if a is not None]
    assert list(got) == usable
    want_b = [oc.image_brightness(thumbs[p], 0.06, 0.96) for p in usable]
    want = oc.exposure_multipliers(want_b, None)
    np.testing.assert_allclose([got[p] for p in usable], want, rtol=1e-6)
    assert got["/d/img_black.nef"] == 1.0                                  # brightness 0: multiplier 1
    assert logs[0] == "[Exposure Calc] Failed on /d/img_missing.nef: could not load thumbnail"
    assert any(m.startswith("[Exposure Calc] Using average image brightness as reference") for m in logs)
    chart = "/d/img_3.nef"
    got_c = calculate_average_exposure(thumbs, 0.06, 0.96, average_source=chart)
    want_c = oc.exposure_multipliers(want_b, usable.index(chart))
    np.testing.assert_allclose([got_c[p] for p in usable], want_c, rtol=1e-6)
    assert got_c[chart] == 1.0
    assert calculate_average_exposure({"/d/x.nef": None}, log=logs.append) == {}


def test_all_gather_through_the_c_abi_alone(monkeypatch):
    """ss_comm_unique_id -> ss_comm_create -> ss_exposure_stats -> ss_allgather_stats -> ss_exposure_multipliers:
    the sequence of a host without torch.distributed, on a one-rank communicator (the box has one GPU, and RCCL
    refuses two ranks on one device).  Both forms of the collective: one all-gather, and the group of broadcasts
    that uneven shards use."""
    from gpu_util import DeviceBuffer
    from scanspaceimageprocessor_amd import _lib

    ctx = default_context()
    n, h, w = 5, 96, 128
    rng = np.random.default_rng(5)
    thumbs = np.stack([inputs.linear_image(90 + i, h, w) * np.float32(2.0 ** rng.uniform(-0.5, 0.5)) for i in range(n)])
    want_b = [oc.image_brightness(x, 0.06, 0.96) for x in thumbs]
    want_m = oc.exposure_multipliers(want_b, 2)
    comm = ctx.comm_create(_lib.comm_unique_id(), 0, 1)
    t = DeviceBuffer.from_array(ctx, thumbs)
    local, everyone, mult = (DeviceBuffer(ctx, 4 * n) for _ in range(3))
    try:
        for form in ("0", "1"):
            monkeypatch.setenv("SS_ALLGATHER_BROADCASTS", form)
            ctx.upload(everyone.ptr, np.zeros(n, np.float32))
            ctx.exposure_stats(t.ptr, h, w, n, 0.06, 0.96, None, local.ptr)
            ctx.allgather_stats(comm, 0, 1, n, local.ptr, everyone.ptr)
            ctx.exposure_multipliers(everyone.ptr, n, 2, 0, n, mult.ptr)
            ctx.sync()
            np.testing.assert_allclose(everyone.to_array((n,), np.float32), want_b, rtol=1e-6)
            np.testing.assert_allclose(mult.to_array((n,), np.float32), want_m, rtol=1e-6)
    finally:
        ctx.comm_destroy(comm)
        for b in (t, local, everyone, mult):
            b.free()


def test_abi_gather_serves_the_normaliser_class():
    """dataset.AbiGather as the ``gather`` of an ExposureNormaliser: "rank 1 of 2" as in the test above, the local
    part going through ss_allgather_stats on a one-rank communicator."""
    import torch

    from scanspaceimageprocessor_amd import _lib
    from scanspaceimageprocessor_amd.dataset import AbiGather

    ctx = default_context()
    dev = torch.device("cuda", ctx.device)
    n, h, w = 6, 96, 128
    thumbs = np.stack([inputs.linear_image(170 + i, h, w) for i in range(n)])
    t = torch.from_numpy(thumbs).to(dev)
    want_b = [oc.image_brightness(x, 0.06, 0.96) for x in thumbs]
    want_m = oc.exposure_multipliers(want_b, None)
    comm = ctx.comm_create(_lib.comm_unique_id(), 0, 1)
    try:
        other = ExposureNormaliser(ctx, n, 0, 2, 0.06, 0.96, gather=lambda local, n_images: None)
        other.ops.stats(t.data_ptr(), h, w, other.count, 0.06, 0.96, other.brightness)
        one_rank = AbiGather(ctx, comm, 0, 1)

        def gather(local, n_images):
            mine = one_rank(local, local.numel())
            ctx.sync()
            return torch.cat([other.brightness, mine])

        norm = ExposureNormaliser(ctx, n, 1, 2, 0.06, 0.96, gather=gather)
        gains = norm.run(t[norm.first:].data_ptr(), h, w)
        ctx.sync()
        np.testing.assert_allclose(gains.cpu().numpy(), want_m[norm.first:], rtol=1e-6)
    finally:
        ctx.comm_destroy(comm)
