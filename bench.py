#!/usr/bin/env python3
"""Headline benchmark: megapixels/s of the full 24 MP correction chain.

Workload (BASELINE.json configs[1], per GPU): a batch of 64 synthetic 24 MP (4000 x 6000) float32
linear RGB images resident in HBM ->  dataset exposure normalisation (per-image thumbnail
statistics, [RCCL all-gather when N > 1], multipliers) -> ColorChecker CCM (Cheung 2004, 3 terms)
-> exposure gain -> clip -> unsharp mask (radius 2, amount 1.0) -> clip -> sRGB encode ->
16-bit truncating quantise.  One "step" = one pass over the batch.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python bench.py --gpus N --steps K --warmup W       (starts its own N ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` is whole-job MP/s (all ranks) with inputs resident in HBM;
`roofline` prices the fused unsharp-mask kernel against the 8 TB/s HBM peak using its ALGORITHMIC
bytes (12 B/px float32 in + 6 B/px uint16 out = 18 B/px) and the kernel time measured here with
events on the library's stream.  The kernel is bound by the vector ALU, not by HBM: `bound` says
so, and `valu_frac` prices the instruction count of the committed counter passes (the file is
named with its SHA-256; tools/pmc_bench.sh makes it from this very command) at this run's pixel
rate against one wave-instruction per SIMD per 4 cycles.  `others` times the kernels of the other
single-GPU configurations in the same process; `cpu_baseline` times the numpy restatement of the
reference's path (oracle/, a port: the reference itself cannot be imported on the GPU box) on
three 24 MP images with one thread, then on 4 and on all (at most 16) threads.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 4000, 6000
THUMB_H, THUMB_W = 341, 512
SHADOW_LIMIT, HIGHLIGHT_LIMIT = 0.06, 0.96
HBM_PEAK_GBS = 8000.0
ALGO_BYTES_PER_PX = 18.0  # 12 in (float32 RGB) + 6 out (uint16 RGB)
# Counter-derived figures of the dominant kernel (VALU instructions and HBM bytes per pixel, clock) cannot be collected
# inside this process; they come from the rocprofv3 PMC passes summarised in this file (made by tools/pmc_summary.py
# from the passes' CSVs, committed beside them), which the bench line names together with its SHA-256.  No file, no
# figure: nothing is hard-coded.
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r03_usm_pmc.json")
N_SIMD = 1024
# Issue cost of a wave-instruction on one SIMD in cycles, by class, measured on this GPU with four waves per SIMD
# (tools/micro/valu_rate3.hip; ns there, divided by the 2.05 GHz the chip ran at): a plain float32 / integer operation
# goes through the 32-lane float pipe in 2 cycles, a packed one (two results) in 4, compares / selects / min / max /
# med3 / conversions in about 3.6, 64-bit integer multiply-adds in 8, v_exp_f32 / v_log_f32 in 6.9.
VALU_CYCLES = {"plain": 2.05, "packed_extra": 2.05, "slow": 3.6, "int64": 8.0, "trans": 6.9}


def priced_valu_cycles(pmc):
    """Vector-ALU cycles per 64 pixels the kernel's dynamic instruction mix costs at the prices above (None without
    the mix pass).  Classes as the counters give them: float add / mul / fma, transcendental, 32- and 64-bit integer,
    conversions, and the rest (compares, selects, min / max / med3, moves, lane reads), priced as the slow class.
    A packed instruction counts once among add / mul / fma and costs twice: the flops counter gives
    packed add + packed mul + 2 packed fma; the same packed share is assumed for the three."""
    m = pmc.get("valu_mix")
    if not m:
        return None
    c = VALU_CYCLES
    floats = m["add_f32"] + m["mul_f32"] + m["fma_f32"]
    packed = max(m.get("packed_extra_flops", 0.0), 0.0) * floats / (m["add_f32"] + m["mul_f32"] + 2.0 * m["fma_f32"])
    return ((floats + m["int32"]) * c["plain"] + packed * c["packed_extra"] + m["trans_f32"] * c["trans"] +
            m["int64"] * c["int64"] + (m["cvt"] + m["other"]) * c["slow"])


def pmc_summary():
    import hashlib

    try:
        raw = open(PMC_SUMMARY, "rb").read()
        return json.loads(raw), hashlib.sha256(raw).hexdigest()
    except (OSError, ValueError):
        return None, None


def synthetic_swatches():
    """Same recipe as oracle.inputs.synthetic_swatches (kept local: the product path never imports oracle/)."""
    r = np.random.default_rng(7)
    M_R = 0.03 + 0.87 * r.random((24, 3))
    c = np.random.default_rng(8)
    A = np.eye(3) + 0.15 * (c.random((3, 3)) * 2.0 - 1.0)
    M_T = M_R @ A.T + c.normal(0.0, 0.002, size=(24, 3))
    return M_T.astype(np.float32), M_R.astype(np.float64)


def cpu_baseline(amount, n_images=3, check=None):
    """Time the numpy restatement of the reference path on a bounded sample of the workload: `n_images` of the
    64 images single-threaded (about 4.5 s each on the GPU box's host; this is the reported baseline), then
    one image on each of 4 threads, the reference's default worker count (`threads4`).  Inputs are generated
    outside the timed regions.  `check(image, gain)`: the library's 16-bit result for the same image; the first
    image's restatement output is compared with it (untimed) and the comparison reported as `parity_check`."""
    from oracle import chain as oc
    from oracle import inputs

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    M_T, M_R = inputs.synthetic_swatches()
    imgs = [inputs.linear_image(1234 + i, H, W) for i in range(n_images)]

    first_result = []

    def run_all():
        for img in imgs:
            r = oc.config2(img, M_T, M_R, 1.1, amount, 2.0, 0.0)
            if not first_result:
                first_result.append(r)

    t0 = time.perf_counter()
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            run_all()
    else:
        run_all()
    dt = time.perf_counter() - t0
    # the reference's own default: 4 worker threads (image_space.py:238), one image each
    threads = 4
    from concurrent.futures import ThreadPoolExecutor

    more = imgs + [inputs.linear_image(1234 + n_images, H, W)] if n_images < threads else imgs
    more = more[:threads]

    def run_threads():
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(lambda im: oc.config2(im, M_T, M_R, 1.1, amount, 2.0, 0.0), more))

    t1 = time.perf_counter()
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            run_threads()
    else:
        run_threads()
    dt_mt = time.perf_counter() - t1
    # every core this process may use (16 on the one-GPU box; os.cpu_count() there shows the whole machine)
    try:
        n_all = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n_all = os.cpu_count() or 1
    n_all = max(1, min(n_all, 16))
    every = [more[i % len(more)] for i in range(n_all)]

    def run_all_cores():
        with ThreadPoolExecutor(n_all) as pool:
            list(pool.map(lambda im: oc.config2(im, M_T, M_R, 1.1, amount, 2.0, 0.0), every))

    t2 = time.perf_counter()
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            run_all_cores()
    else:
        run_all_cores()
    dt_all = time.perf_counter() - t2
    parity = None
    if check is not None:
        got = check(imgs[0], 1.1).astype(np.int32)
        diff = np.abs(got - first_result[0].astype(np.int32))
        parity = {"input": f"oracle.inputs.linear_image(1234), {H}x{W}, gain 1.1, sharpen amount {amount:g}: every value of "
                           "the library's 16-bit output against the restatement's",
                  "values": int(diff.size), "differing_values": int(np.count_nonzero(diff)),
                  "max_abs_diff_lsb16": int(diff.max())}
    return {
        "parity_check": parity,
        "value": round(n_images * H * W / 1e6 / dt, 3), "unit": "MP/s", "cores": 1, "kind": "port",
        "sample": f"{n_images} of the 64 images of {H}x{W} (24 MP) through oracle.chain.config2 (float64 numpy restatement "
                  f"of the reference CPU path, BLAS limited to 1 thread), {dt:.1f} s; host has {os.cpu_count()} logical CPUs",
        "threads4": {"value": round(len(more) * H * W / 1e6 / dt_mt, 3), "unit": "MP/s", "cores": threads,
                     "sample": f"{len(more)} images on {threads} Python threads (the reference's default worker count), "
                               f"{dt_mt:.1f} s"},
        "threads_all": {"value": round(n_all * H * W / 1e6 / dt_all, 3), "unit": "MP/s", "cores": n_all,
                        "sample": f"{n_all} images on {n_all} Python threads (every core this process may use, at most 16), "
                                  f"{dt_all:.1f} s"},
    }


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: start one child process per rank and relay rank 0's line.

    Runs before this process has imported torch or touched HIP: the parent only probes a free port, starts the
    children (this same file, or the program named by SS_BENCH_RANK_PROGRAM, which the launcher's CPU test sets to a
    stub) with RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment,
    waits for all of them and returns the first non-zero exit status.  A failed child is not started again; when
    one fails the others are terminated (they would wait in a collective for it)."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    program = os.environ.get("SS_BENCH_RANK_PROGRAM", os.path.abspath(__file__))
    children = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        # rank 0's stdout is this process's stdout (its one JSON line); the other ranks print nothing there
        children.append(subprocess.Popen([sys.executable, "-u", program, *argv], env=env,
                                         stdout=None if r == 0 else subprocess.DEVNULL))
    status = 0
    stopped_at = None
    pending = dict(enumerate(children))
    while pending:
        for r, p in list(pending.items()):
            rc = p.poll()
            if rc is None:
                continue
            del pending[r]
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 128 - rc
                print(f"bench.py: rank {r} exited with status {rc}; stopping the other ranks", file=sys.stderr)
                for q in pending.values():
                    q.terminate()
                stopped_at = time.monotonic()
        if pending:
            if stopped_at is not None and time.monotonic() - stopped_at > 10.0:
                for q in pending.values():  # a rank stuck in a device call does not see SIGTERM
                    q.kill()
            time.sleep(0.05)
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (the GPU's clocks need a few 10 ms launches to settle: the first two steps run 5-15 % slower)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default: about 2 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--amount", type=float, default=1.0, help="unsharp-mask amount (BASELINE: 1.0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the kernel timings of the other BASELINE configurations")
    ap.add_argument("--collective", choices=("torch", "abi"), default="torch",
                    help="N > 1: the all-gather through torch.distributed (default) or through the library's own "
                         "ss_allgather_stats (RCCL resolved by the library; torch.distributed only carries the "
                         "communicator's identifier to the ranks)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 on a box with one GPU: every rank uses device 0 and the ranks talk over gloo (RCCL refuses "
                         "two ranks on one device); exercises the multi-rank plumbing, its numbers mean nothing")
    ap.add_argument("--workload", choices=("config2", "config5"), default="config2",
                    help="config2 = BASELINE.json configs[1] (default, the headline); config5 = the dataset-normalisation "
                         "scaling case: thumbnail statistics -> all-gather -> multipliers -> CCM + gain + sRGB, float32 "
                         "out, 128 images per GPU unless --batch is given")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (nothing here has touched the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    from scanspaceimageprocessor_amd import Context
    from scanspaceimageprocessor_amd.chain import make_desc
    from scanspaceimageprocessor_amd.colour_fit import matrix_colour_correction
    from scanspaceimageprocessor_amd.dataset import ExposureNormaliser, shard_range

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank} needs GPU {local_rank} but this node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    config5 = args.workload == "config5"
    if config5 and args.batch == 64:
        args.batch = 128
    B = args.batch
    n_total = B * world
    first, count = shard_range(n_total, rank, world)
    assert count == B

    # ---- synthetic dataset, generated on the device (not timed) ----
    scales = 2.0 ** np.random.default_rng(9).uniform(-0.5, 0.5, size=n_total)
    images = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev)
    thumbs = torch.empty((B, THUMB_H, THUMB_W, 3), dtype=torch.float32, device=dev)
    for i in range(B):
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + first + i)
        img = torch.rand((H, W, 3), generator=g, device=dev, dtype=torch.float32)
        img.pow_(2.2).mul_(0.9 * float(scales[first + i])).clamp_(0.0, 1.0)
        images[i] = img
        del img
    out = torch.empty((B, H, W, 3), dtype=torch.float32 if config5 else torch.int16, device=dev)  # int16 = uint16 payload

    ctx = Context(local_rank)
    # the 512-pixel thumbnails the exposure statistics read: the loader's area-average downscale, on the device
    torch.cuda.synchronize(dev)
    ctx.resize_area(images.data_ptr(), H, W, B, thumbs.data_ptr(), THUMB_H, THUMB_W)
    ctx.sync()
    # one dedicated stream carries the library's kernels, the collective and the timing events
    stream = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)
    gather = None
    if world > 1 and args.rehearse_on_one_gpu:
        from scanspaceimageprocessor_amd.dataset import gather_brightness

        def gather(local, n_images):  # gloo moves host tensors
            torch.cuda.current_stream(dev).synchronize()
            return gather_brightness(local.cpu(), n_images).to(dev)
    elif world > 1 and args.collective == "abi":
        from scanspaceimageprocessor_amd import _lib
        from scanspaceimageprocessor_amd.dataset import AbiGather

        ident = [_lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        gather = AbiGather(ctx, ctx.comm_create(ident[0], rank, world), rank, world)
    norm = ExposureNormaliser(ctx, n_total, rank, world, SHADOW_LIMIT, HIGHLIGHT_LIMIT, gather=gather)
    gains = norm.gains

    M_T, M_R = synthetic_swatches()
    ccm = matrix_colour_correction(M_T, M_R)  # Cheung 2004, 3 terms: the reference's only call form
    if config5:
        desc = make_desc(H, W, B, ccm=ccm, gains=[1.0], gain0_dev=gains.data_ptr(), final_clip=True, out="f32_srgb")
    else:
        desc = make_desc(H, W, B, ccm=ccm, gains=[1.0], gain0_dev=gains.data_ptr(), usm=(args.amount, 2.0, 0.0),
                         final_clip=True, out="u16")

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(k=None):
        norm.run(thumbs.data_ptr(), THUMB_H, THUMB_W)  # stats -> [all-gather] -> multipliers into `gains`
        if k is not None:
            ev[k][0].record(stream)
        ctx.chain_run(desc, images.data_ptr(), out.data_ptr())
        if k is not None:
            ev[k][1].record(stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    def others():
        """Kernel timings of the other BASELINE configurations in this process (HIP events on the library's stream,
        inputs resident): configs[0] (pointwise CCM), configs[2] (negative mode with its percentile search) and
        configs[3] (45 MP, root-polynomial CCM + shadows/highlights), each against its own algorithmic bytes."""
        from scanspaceimageprocessor_amd import _lib

        def timed(fn, reps=10, warm=3):
            for _ in range(warm):
                fn()
            ctx.sync()
            ts = []
            for _ in range(reps):
                ctx.timer_start()
                fn()
                ts.append(ctx.timer_stop())
            return float(np.mean(ts))

        def entry(config, kernel, bytes_px, n_px, ms):
            gbs = bytes_px * n_px / (ms * 1e-3) / 1e9
            return {"config": config, "kernel": kernel, "kernel_ms": round(ms, 4), "MP_per_s": round(n_px / 1e6 / (ms * 1e-3), 1),
                    "algorithmic_bytes_per_px": bytes_px, "frac": round(gbs / HBM_PEAK_GBS, 4)}

        res = []
        flat = images.view(-1)
        out32 = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev)
        n_px = float(B) * H * W
        # configs[1] at the slider's amount 50 (scale 1.0; above 30 the sharpen also boosts chroma): on this noise
        # 3 % of the pixels are evaluated again in float64
        d2b = make_desc(H, W, B, ccm=ccm, gains=[1.0], gain0_dev=gains.data_ptr(), usm=(50.0, 2.0, 0.0), final_clip=True,
                        out="u16")
        res.append(entry(f"configs[1] x{B} at sharpen amount 50 (chroma branch, float64 re-evaluation of recorded pixels)",
                         "usm_kernel<6, CCM_LIN3, U16_SRGB> + careful_kernel", 18.0, n_px,
                         timed(lambda: ctx.chain_run(d2b, images.data_ptr(), out.data_ptr()), reps=5, warm=2)))
        d1 = make_desc(H, W, B, ccm=ccm, gains=[1.0], out="f32_srgb")
        res.append(entry(f"configs[0] x{B}: CCM 3 terms + clip + sRGB, float32 out", "chain_kernel<CCM_LIN3, F32_SRGB>", 24.0, n_px,
                         timed(lambda: ctx.chain_run(d1, images.data_ptr(), out32.data_ptr()))))
        # configs[3]: 45 MP images in the same storage, fresh linear data
        H4, W4 = 5504, 8256
        B4 = min(16, int(flat.numel() // (H4 * W4 * 3)))
        img4 = flat[: B4 * H4 * W4 * 3].view(B4, H4, W4, 3)
        for i in range(B4):
            img4[i].uniform_(0.0, 1.0).pow_(2.2).mul_(0.9)
        out4 = out32.view(-1)[: B4 * H4 * W4 * 3].view(B4, H4, W4, 3)
        ccm4 = matrix_colour_correction(M_T, M_R, method="Finlayson 2015", degree=2)
        d4 = make_desc(H4, W4, B4, ccm=ccm4, gains=[1.0], sh=(0.5, -0.5, 7.0), final_clip=True, out="f32")
        res.append(entry(f"configs[3] x{B4}: 45 MP, root-polynomial CCM degree 2 + clip + Lab shadows/highlights, float32 out",
                         "chain_kernel<CCM_SMALL, S/H, F32_LINEAR>", 24.0, float(B4) * H4 * W4,
                         timed(lambda: ctx.chain_run(d4, img4.data_ptr(), out4.data_ptr()))))
        ccm4b = matrix_colour_correction(M_T, M_R, method="Finlayson 2015", degree=3)
        d4b = make_desc(H4, W4, B4, ccm=ccm4b, gains=[1.0], sh=(0.5, -0.5, 7.0), final_clip=True, out="f32")
        res.append(entry(f"configs[3] x{B4} with the root-polynomial of degree 3 (13 terms; its fit's coefficients are "
                         "beyond float32: the correction runs as a float64 pass into a staging image)",
                         "ccm_f64_kernel<ROOT, 13> + chain_kernel<CCM_NONE, S/H, F32_LINEAR>", 24.0, float(B4) * H4 * W4,
                         timed(lambda: ctx.chain_run(d4b, img4.data_ptr(), out4.data_ptr()))))
        # configs[2]: colour-negative look of fresh data, percentile search + inversion + CCM
        orange = torch.tensor([1.0, 0.75, 0.55], device=dev)
        for i in range(B):
            images[i].uniform_(0.0, 1.0).pow_(2.2).mul_(0.9).mul_(-1.0).add_(1.0).mul_(0.8).mul_(orange).add_(0.08)
        params = torch.empty((B, _lib.SS_NEG_PARAM_STRIDE), dtype=torch.float32, device=dev)
        d3 = make_desc(H, W, B, neg=True, neg_params_dev=params.data_ptr(), ccm=ccm, out="f32")
        torch.cuda.synchronize(dev)

        def run3(low=0.005):
            ctx.negative_params(images.data_ptr(), H, W, B, low, 0.995, True, params.data_ptr())
            ctx.chain_run(d3, images.data_ptr(), out32.data_ptr())

        res.append(entry(f"configs[2] x{B}: percentiles (0.5 / 99.5) + inversion + CCM 3 terms, float32 out",
                         "pct_* + chain_kernel<CCM_LIN3, F32_LINEAR>", 24.15, n_px, timed(run3)))
        res.append(entry(f"configs[2] x{B} with the reference's default percentiles (5 / 99.5)",
                         "pct_* + chain_kernel<CCM_LIN3, F32_LINEAR>", 24.15, n_px, timed(lambda: run3(0.05))))
        return res

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if os.environ.get("SS_BENCH_VERBOSE"):
        print("kernel ms per step:", [round(a.elapsed_time(b), 3) for a, b in ev], file=sys.stderr)
    px_per_launch = float(B) * H * W
    algo_bytes = 24.0 if config5 else ALGO_BYTES_PER_PX  # config5: 12 in + 12 out (float32)
    achieved_gbs = algo_bytes * px_per_launch / (kernel_ms * 1e-3) / 1e9
    mp_total = n_total * H * W / 1e6 * args.steps
    value = mp_total / elapsed

    # sanity: the output must not be all zeros / all saturated (catches a silently skipped launch)
    if config5:
        sample = out[0, ::400, ::600]
        assert float(sample.max()) > 0.0 and float(sample.min()) < 1.0, "implausible output"
    else:
        sample = out[0, ::400, ::600].to(torch.int32) & 0xFFFF
        assert int(sample.max()) > 0 and int(sample.min()) < 65535, "implausible output"
    g = gains.cpu().numpy()
    assert np.all(np.isfinite(g)) and abs(float(np.mean(g)) - 1.0) < 0.2, "implausible exposure multipliers"

    if rank == 0:
        line = {
            "metric": "megapixels/sec (whole node), full 24MP correction chain",
            "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not args.rehearse_on_one_gpu else "synthetic (REHEARSAL: all ranks on one GPU, gloo)",
            "config": {
                "workload": (f"{B}x24MP per GPU: exposure normalisation + Cheung-2004 3x3 CCM + gain + clip + sRGB encode, "
                             "float32 out (the dataset-normalisation scaling case)") if config5 else
                            (f"{B}x24MP per GPU: exposure normalisation + Cheung-2004 3x3 CCM + gain + clip + unsharp "
                             f"mask (radius 2, amount {args.amount:g}) + sRGB encode + 16-bit quantise "
                             "(BASELINE.json configs[1])"),
                "images_per_gpu": B, "height": H, "width": W, "parallelism": f"image-sharded x{world}",
                "collective": (f"all_gather of per-image brightness (RCCL, through {args.collective})") if world > 1 else "none",
            },
        }
        pmc, pmc_sha = (None, None) if config5 else pmc_summary()
        px_per_s = px_per_launch / (kernel_ms * 1e-3)
        roof = {
            # what limits the kernel: the pointwise chain streams (HBM); the tile kernel is bound by the vector ALU
            "bound": "hbm" if config5 else "valu",
            "kernel": "chain_kernel<CCM_LIN3, no S/H, F32_SRGB>" if config5 else "usm_kernel<6, CCM_LIN3, U16_SRGB, plain>",
            # the HBM roofline all the same: algorithmic bytes per launch / measured kernel time against 8 TB/s
            "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_px": algo_bytes, "px_per_launch": px_per_launch, "kernel_ms": round(kernel_ms, 4),
            "traffic": None,
        }
        if pmc is not None:
            clock_hz = pmc["clock_ghz"] * 1e9
            roof.update({
                # HBM bytes per launch from the counter passes (FETCH_SIZE doubled as gfx950 needs + WRITE_SIZE)
                "traffic": round((pmc["fetch_bytes_per_px"] + pmc["write_bytes_per_px"]) * px_per_launch),
                "valu_instructions_per_px": pmc["valu_per_px"], "valu_trans_per_px": pmc["trans_per_px"],
                "valu_busy_profiled": pmc["valu_busy"], "clock_ghz_profiled": pmc["clock_ghz"],
                "pmc_file": os.path.relpath(PMC_SUMMARY, ROOT), "pmc_sha256": pmc_sha,
            })
            priced = priced_valu_cycles(pmc)
            if priced is not None:
                # vector-ALU roofline: the cycles the instruction mix costs, class by class, against the cycles a SIMD
                # had per 64 pixels in this run (kernel time x the profiled clock / the wave-instructions' worth of
                # pixels per SIMD); the rest of the time the vector unit had nothing it could issue
                cycles_per_64px = kernel_ms * 1e-3 * clock_hz / (px_per_launch / 64.0 / N_SIMD)
                roof.update({"valu_mix_per_px": pmc["valu_mix"], "valu_cycle_prices": VALU_CYCLES,
                             "valu_priced_cycles_per_64px": round(priced, 1),
                             "valu_cycles_per_64px": round(cycles_per_64px, 1),
                             "valu_frac": round(priced / cycles_per_64px, 4),
                             "valu_stall_frac": round(1.0 - priced / cycles_per_64px, 4),
                             "wait_any_share_profiled": pmc.get("wait_any_share"),
                             "wait_issue_share_profiled": pmc.get("wait_issue_share")})
        line["roofline"] = roof
        if world == 1 and not args.no_others and not config5:
            line["others"] = others()
        if world == 1 and not args.no_cpu_baseline and not config5:
            def library_result(img, gain):
                """The same chain through the library for one host image (the restatement's own input)."""
                src = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).to(dev)
                dst = torch.empty((1, H, W, 3), dtype=torch.int16, device=dev)
                d = make_desc(H, W, 1, ccm=ccm, gains=[gain], usm=(args.amount, 2.0, 0.0), final_clip=True, out="u16")
                ctx.chain_run(d, src.data_ptr(), dst.data_ptr())
                ctx.sync()
                return dst[0].cpu().numpy().view(np.uint16)

            line["cpu_baseline"] = cpu_baseline(args.amount, check=library_result)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
