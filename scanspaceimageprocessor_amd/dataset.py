"""Batches of images sharded over the GPUs of one node, one process per GPU.

Images are independent, so a dataset is split into contiguous index ranges (the reference
chunks its image list the same way over worker threads, image_space.py:3423-3425) and no pixel
ever crosses xGMI.  The one exchange step is dataset exposure normalisation
(``calculate_average_exposure``, image_space.py:4323-4398): every rank needs the brightness of
ALL images to form the reference value, so the per-image brightness scalars are all-gathered
(RCCL through ``torch.distributed``; 4 bytes per image, latency-bound).
"""

from __future__ import annotations

from typing import Optional, Tuple


def _torch():
    """``import torch`` with the ordering rule of the two HIP runtimes checked first (``_lib._share_torch_hip_runtime``):
    once the library has been loaded against the system runtime, a first import of torch in the same process brings a
    second runtime that finds no GPU.  Said here, where it can still be explained."""
    import sys

    if "torch" not in sys.modules:
        from . import _lib

        if _lib.hip_runtime_choice() == "system":
            raise ImportError(
                "torch is being imported after libscanspace_hip.so was loaded against the system HIP runtime: two HIP "
                "runtimes in one process.  Import torch before the first scanspaceimageprocessor_amd call, or set "
                "SCANSPACE_SHARE_TORCH_HIP=1 (load_library(share_torch_hip=True)) so that the library shares torch's.")
    import torch

    return torch


def shard_range(n_images: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced shard ``(first, count)`` of ``n_images`` for ``rank`` of ``world``."""
    if world < 1 or not (0 <= rank < world) or n_images < 0:
        raise ValueError(f"bad shard request: n={n_images} rank={rank} world={world}")
    base, extra = divmod(n_images, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def gather_brightness(local, n_images: int, group=None):
    """All-gather the per-image brightness of every rank's shard into one tensor in image order.

    ``local`` is this rank's 1-D float32 tensor (``shard_range`` order), on the GPU (backend nccl =
    RCCL) or on the CPU (backend gloo).  Uneven shards are padded to the largest shard for the
    collective and compacted afterwards.  With a single process the input is returned as is.
    """
    torch = _torch()
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if local.numel() != n_images:
            raise ValueError("single-process gather needs the brightness of every image")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    first, count = shard_range(n_images, rank, world)
    if local.numel() != count:
        raise ValueError(f"rank {rank} holds {local.numel()} values, its shard has {count}")
    longest = shard_range(n_images, 0, world)[1]
    if n_images % world == 0:
        out = torch.empty((n_images,), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((longest,), dtype=local.dtype, device=local.device)
    padded[:count] = local
    buf = torch.empty((world * longest,), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    pieces = [buf[r * longest: r * longest + shard_range(n_images, r, world)[1]] for r in range(world)]
    return torch.cat(pieces)


class AbiGather:
    """``gather`` of an ``ExposureNormaliser`` through the C ABI alone (``ss_allgather_stats`` on the context's
    stream, RCCL resolved by the library): what a host without torch.distributed calls.  ``comm`` comes from
    ``ctx.comm_create(comm_id, rank, world)`` with the bytes of ``_lib.comm_unique_id()`` made on one rank."""

    def __init__(self, ctx, comm, rank: int, world: int):
        self.ctx, self.comm, self.rank, self.world = ctx, comm, rank, world
        self.out = None

    def __call__(self, local, n_images: int):
        torch = _torch()

        if self.out is None or self.out.numel() != n_images:
            self.out = torch.empty((n_images,), dtype=torch.float32, device=local.device)
        self.ctx.allgather_stats(self.comm, self.rank, self.world, n_images, local.data_ptr(), self.out.data_ptr())
        return self.out


class HipExposureOps:
    """The two device steps of dataset exposure normalisation on one context's GPU and stream
    (``ss_exposure_stats``, ``ss_exposure_multipliers``) over torch CUDA tensors."""

    def __init__(self, ctx):
        torch = _torch()

        self.ctx = ctx
        self.device = torch.device("cuda", ctx.device)

    def empty(self, count):
        torch = _torch()

        return torch.empty((count,), dtype=torch.float32, device=self.device)

    def stats(self, thumbs_ptr, thumb_h, thumb_w, count, shadow_limit, highlight_limit, brightness):
        self.ctx.exposure_stats(thumbs_ptr, thumb_h, thumb_w, count, shadow_limit, highlight_limit, None,
                                brightness.data_ptr())

    def multipliers(self, brightness_all, n_images, chart_index, first, count, gains):
        self.ctx.exposure_multipliers(brightness_all.data_ptr(), n_images, chart_index, first, count, gains.data_ptr())

    def stream_scope(self):
        """Context manager that makes the context's stream torch's current stream: the collective must follow the
        statistics kernel and precede the multiplier kernel, whatever torch's current stream is."""
        torch = _torch()

        return torch.cuda.stream(torch.cuda.ExternalStream(self.ctx.stream, device=self.device))


class ExposureNormaliser:
    """Dataset exposure normalisation for one rank's shard.

    ``run`` = per-thumbnail masked-luminance statistics -> all-gather (if more than one rank) -> multipliers
    for the local shard, written to ``self.gains`` (one float32 per local image) which
    ``ss_chain_desc.gain0_dev`` consumes.

    ``ops`` carries the two compute steps and the tensors (default: ``HipExposureOps(ctx)``, the HIP kernels on
    the context's stream); ``gather(local, n_images)`` returns the brightness of all images in image order
    (default: ``gather_brightness`` over ``group``, RCCL on GPU tensors).  The shard arithmetic, the order of the
    steps and the ranges handed to them are the same whatever the two are, so multi-process CPU tests drive this
    very class with host-side ``ops`` over gloo.
    """

    def __init__(self, ctx, n_images: int, rank: int = 0, world: int = 1, shadow_limit: float = 0.06,
                 highlight_limit: float = 0.96, chart_index: Optional[int] = None, group=None, gather=None, ops=None):
        self.ctx, self.n_images, self.rank, self.world, self.group = ctx, n_images, rank, world, group
        self.first, self.count = shard_range(n_images, rank, world)
        self.shadow_limit, self.highlight_limit, self.chart_index = shadow_limit, highlight_limit, chart_index
        if chart_index is not None and not (0 <= chart_index < n_images):
            raise ValueError(f"chart_index {chart_index} outside the dataset of {n_images} images")
        self.ops = ops if ops is not None else HipExposureOps(ctx)
        self.gather = gather if gather is not None else (lambda local, n: gather_brightness(local, n, self.group))
        self.brightness = self.ops.empty(self.count)
        self.gains = self.ops.empty(self.count)
        self.brightness_all = self.brightness

    def run(self, thumbs_ptr, thumb_h: int, thumb_w: int):
        """``thumbs_ptr``: this rank's ``count`` thumbnails, ``[count][thumb_h][thumb_w][3]`` float32 (a device
        address for the HIP ops)."""
        self.ops.stats(thumbs_ptr, thumb_h, thumb_w, self.count, self.shadow_limit, self.highlight_limit,
                       self.brightness)
        if self.world > 1:
            with self.ops.stream_scope():
                self.brightness_all = self.gather(self.brightness, self.n_images)
            if self.brightness_all.numel() != self.n_images:
                raise ValueError(f"gather returned {self.brightness_all.numel()} values for {self.n_images} images")
        else:
            self.brightness_all = self.brightness
        self.ops.multipliers(self.brightness_all, self.n_images, self.chart_index, self.first, self.count, self.gains)
        return self.gains


def calculate_average_exposure(thumbnails, shadow_limit: float = 0.06, highlight_limit: float = 0.96,
                               average_source=None, ctx=None, log=None) -> dict:
    """Exposure multipliers of a dataset from its thumbnails: the arithmetic of the main window's
    ``calculate_average_exposure`` (image_space.py:4323-4398) without the widget plumbing.

    ``thumbnails`` maps image path -> ``(H, W, 3)`` array (the thumbnail cache's ``'array'`` entries) in list
    order; a ``None`` entry is reported and skipped, as the reference does for a thumbnail it could not load.
    ``shadow_limit`` / ``highlight_limit`` are the spin-box values divided by 100.  ``average_source``: path of
    the chart image whose brightness is the reference, or None for the mean over all images.
    Returns path -> multiplier (what the reference stores as ``meta['average_exposure']`` and the worker reads
    through ``image_metadata_map``); empty when no thumbnail was usable.

    Masked luminance means run on the GPU, one launch per thumbnail shape (a dataset has one or two), and the
    multipliers in one more launch."""
    import numpy as np

    from ._lib import default_context

    say = log or (lambda msg: None)
    ctx = ctx or default_context()
    paths, arrays = [], []
    for path, arr in thumbnails.items():
        if arr is None:
            say(f"[Exposure Calc] Failed on {path}: could not load thumbnail")
            continue
        arr = np.asarray(arr, dtype=np.float32)
        if arr.ndim != 3 or arr.shape[2] < 3:
            say(f"[Exposure Calc] Failed on {path}: not an RGB thumbnail")
            continue
        paths.append(path)
        arrays.append(np.ascontiguousarray(arr[:, :, :3]))
    if not paths:
        say("[Exposure Calc] No valid images for exposure normalization.")
        return {}
    n = len(paths)
    by_shape = {}
    for i, arr in enumerate(arrays):
        by_shape.setdefault(arr.shape, []).append(i)
    brightness = np.empty(n, dtype=np.float32)
    for (h, w, _), members in by_shape.items():
        batch = np.stack([arrays[i] for i in members])
        src = ctx.workspace("exposure.thumbs", batch.nbytes)
        out = ctx.workspace("exposure.brightness", 4 * len(members))
        ctx.upload(src, batch)
        ctx.exposure_stats(src, h, w, len(members), shadow_limit, highlight_limit, None, out)
        values = np.empty(len(members), dtype=np.float32)
        ctx.download(values, out)
        brightness[members] = values
    ref_index = paths.index(average_source) if average_source in paths else None
    all_dev = ctx.workspace("exposure.all", 4 * n)
    mult_dev = ctx.workspace("exposure.mult", 4 * n)
    ctx.upload(all_dev, brightness)
    ctx.exposure_multipliers(all_dev, n, ref_index, 0, n, mult_dev)
    mult = np.empty(n, dtype=np.float32)
    ctx.download(mult, mult_dev)
    if ref_index is None:
        say(f"[Exposure Calc] Using average image brightness as reference ({float(np.mean(brightness)):.3f})")
    for path, m in zip(paths, mult):
        say(f"[Exposure Calc] {path} multiplier: {float(m):.3f}")
    say("[Exposure Calc] Average exposure multipliers calculated and stored.")
    return {path: float(m) for path, m in zip(paths, mult)}
