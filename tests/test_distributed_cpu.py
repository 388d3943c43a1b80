"""The multi-rank exchange of dataset exposure normalisation on CPU tensors over gloo
(world_size 2 and 3, even and uneven shards).  Ranks build their shard's brightness values with
the oracle and all-gather them exactly as the GPU path does over RCCL."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import chain as oc
from oracle import inputs
from scanspaceimageprocessor_amd.dataset import ExposureNormaliser, gather_brightness, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _thumb(i):
    return inputs.linear_image(500 + i, 24, 32) * np.float32(0.5 + 0.1 * i)


def _worker(rank, world, port, n_images, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = shard_range(n_images, rank, world)
        local = torch.tensor([float(oc.image_brightness(_thumb(first + i), 0.06, 0.96)) for i in range(count)],
                             dtype=torch.float32)
        full = gather_brightness(local, n_images)
        mult = oc.exposure_multipliers([np.float32(v) for v in full.tolist()], None)
        ret[rank] = (full.tolist(), mult[first:first + count])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_images", [(2, 8), (2, 7), (3, 5), (3, 2)])  # (3, 2): one rank holds no image
def test_gather_and_multipliers_match_single_process(world, n_images):
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, n_images, ret), nprocs=world, join=True)
        results = dict(ret)
    want_b = [np.float32(oc.image_brightness(_thumb(i), 0.06, 0.96)) for i in range(n_images)]
    want_m = oc.exposure_multipliers(want_b, None)
    got_m = []
    for r in range(world):
        full, mult = results[r]
        np.testing.assert_array_equal(np.array(full, np.float32), np.array(want_b, np.float32))
        got_m += list(mult)
    np.testing.assert_allclose(got_m, want_m, rtol=0, atol=0)


class OracleExposureOps:
    """Host stand-in for the two HIP steps of ExposureNormaliser (test infrastructure): the oracle's arithmetic on
    CPU tensors, so that the class's shard arithmetic, step order and gather run unchanged over gloo."""

    def __init__(self):
        self.calls = []

    def empty(self, count):
        return torch.empty((count,), dtype=torch.float32)

    def stats(self, thumbs, thumb_h, thumb_w, count, shadow_limit, highlight_limit, brightness):
        assert thumbs.shape == (count, thumb_h, thumb_w, 3)
        for i in range(count):
            brightness[i] = float(oc.image_brightness(thumbs[i], shadow_limit, highlight_limit))
        self.calls.append("stats")

    def multipliers(self, brightness_all, n_images, chart_index, first, count, gains):
        m = oc.exposure_multipliers([np.float32(v) for v in brightness_all.tolist()], chart_index)
        gains[:] = torch.tensor(m[first:first + count], dtype=torch.float32)
        self.calls.append("multipliers")

    def stream_scope(self):
        import contextlib

        return contextlib.nullcontext()


def _normaliser_worker(rank, world, port, n_images, chart_index, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ops = OracleExposureOps()
        norm = ExposureNormaliser(None, n_images, rank, world, 0.06, 0.96, chart_index=chart_index, ops=ops)
        thumbs = np.stack([_thumb(norm.first + i) for i in range(norm.count)]) if norm.count else np.zeros((0, 24, 32, 3))
        gains = norm.run(thumbs, 24, 32)
        ret[rank] = (norm.first, norm.count, norm.brightness_all.tolist(), gains.tolist(), ops.calls)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_images,chart_index", [(2, 8, None), (2, 7, 5), (3, 5, 0), (3, 2, 1)])
def test_exposure_normaliser_class_over_gloo(world, n_images, chart_index):
    """The product's ExposureNormaliser itself (rank/offset arithmetic, statistics -> gather -> multipliers order,
    the world > 1 branch) on every rank of a gloo group; only the two compute steps are host stand-ins."""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_normaliser_worker, args=(world, port, n_images, chart_index, ret), nprocs=world, join=True)
        results = dict(ret)
    want_b = [np.float32(oc.image_brightness(_thumb(i), 0.06, 0.96)) for i in range(n_images)]
    want_m = np.array(oc.exposure_multipliers(want_b, chart_index), np.float32)
    covered = []
    for r in range(world):
        first, count, full, gains, calls = results[r]
        assert (first, count) == shard_range(n_images, r, world)
        assert calls == ["stats", "multipliers"]
        np.testing.assert_array_equal(np.array(full, np.float32), np.array(want_b, np.float32))
        np.testing.assert_array_equal(np.array(gains, np.float32), want_m[first:first + count])
        covered += list(range(first, first + count))
    assert covered == list(range(n_images))


def test_single_process_gather_is_identity():
    t = torch.arange(5, dtype=torch.float32)
    assert gather_brightness(t, 5) is t
    with pytest.raises(ValueError):
        gather_brightness(t, 6)
