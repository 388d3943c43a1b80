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
from scanspaceimageprocessor_amd.dataset import gather_brightness, shard_range


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


@pytest.mark.parametrize("world,n_images", [(2, 8), (2, 7), (3, 5)])
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


def test_single_process_gather_is_identity():
    t = torch.arange(5, dtype=torch.float32)
    assert gather_brightness(t, 5) is t
    with pytest.raises(ValueError):
        gather_brightness(t, 6)
