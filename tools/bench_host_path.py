#!/usr/bin/env python3
"""Wall time of the host-buffer entry points on one 24 MP image (numpy in, numpy out): what a caller of the
drop-in Python functions sees.  PCIe and host memory bound; never the benchmark's headline value."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanspaceimageprocessor_amd import default_context
from scanspaceimageprocessor_amd.chain import make_desc
from scanspaceimageprocessor_amd.ImageProcessor import editingTools as et

H, W = 4000, 6000
rng = np.random.default_rng(0)
img = (rng.random((H, W, 3), dtype=np.float32) ** 2.2 * 0.9).astype(np.float32)
ctx = default_context()


def timeit(name, fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    print(f"{name:52s} {t * 1e3:8.1f} ms  {H * W / 1e6 / t:8.0f} MP/s", flush=True)


ccm = (np.eye(3) * 0.9 + 0.03, 1, 3)
d16 = make_desc(H, W, 1, ccm=ccm, gains=[1.1], usm=(40.0, 1.0, 0.0), final_clip=True, out="u16")
d32 = make_desc(H, W, 1, ccm=ccm, gains=[1.1], final_clip=True, out="f32")
timeit("ss_chain_run_host ccm+gain+usm -> u16 (288 MB in, 144 out)", lambda: ctx.chain_run_host(d16, img))
pin_in = ctx.pinned_empty(img.shape, np.float32)
pin_in[...] = img
pin_out = ctx.pinned_empty(img.shape, np.uint16)
timeit("  the same between page-locked buffers (ctx.pinned_empty)", lambda: ctx.chain_run_host(d16, pin_in, out=pin_out))
timeit("ss_chain_run_host ccm+gain -> f32 (288 MB in, 288 out)", lambda: ctx.chain_run_host(d32, img))
timeit("editingTools.adjust_sharpen (f32 in, f32 out)", lambda: et.adjust_sharpen(img, 40.0, 1.0, 0.0))
timeit("numpy copy of the image, for scale", lambda: img.copy())

from scanspaceimageprocessor_amd.ImageProcessor import imageProcessorWorker as ipw

M_R = 0.03 + 0.87 * np.random.default_rng(7).random((24, 3))
M_T = (M_R @ (np.eye(3) * 0.95 + 0.02).T).astype(np.float32)
ipw.set_reference_swatches(M_R)
raw16 = (img * 65535.0).astype(np.uint16)
timeit("host: np.array(rgb, dtype=np.float32) / 65535.0 (the reference's conversion)", lambda: np.array(raw16, dtype=np.float32) / 65535.0)
for fmt, depth, masked in ((".tiff", 16, False), (".jpg", 8, False), (".tiff", 16, True)):
    w = ipw.ImageCorrectionWorker(["a.nef"], M_T, "/tmp", None, output_format=fmt, tiff_bitdepth=depth, shadow_adj=10.0,
                                  export_masked=masked)
    timeit(f"worker.process_array chart mode -> {fmt} {depth}-bit{' masked' if masked else ''} (+ 8-bit preview)",
           lambda: w.process_array(img, "a.nef"))
    if not masked:
        timeit(f"  the same from the decoder's uint16 image", lambda: w.process_array(raw16, "a.nef"))
        w.reuse_output_buffers = True
        timeit(f"  ... into reused page-locked output buffers", lambda: w.process_array(raw16, "a.nef"))

# The worker's loop itself (`run`): 8 x 24 MP, the decoder's uint16 image from a hook, 16-bit TIFF output + preview handed
# to hooks that do nothing, reused page-locked output buffers; one after the other and two deep.
def worker_run(depth, n=8):
    paths = [f"img{i}.nef" for i in range(n)]
    w = ipw.ImageCorrectionWorker(paths, M_T, "/tmp", None, output_format=".tiff", tiff_bitdepth=16, shadow_adj=10.0)
    w.raw_loader = lambda path: raw16
    w.image_writer = lambda out_path, arr, ext: None
    w.reuse_output_buffers = True
    w.pipeline_depth = depth
    w.run()  # warm-up: workspaces, page-locked arrays
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        w.run()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts)) / n
    print(f"worker.run, {n} x 24 MP -> 16-bit TIFF + preview, pipeline depth {depth}: {t * 1e3:8.1f} ms per image  "
          f"{H * W / 1e6 / t:8.0f} MP/s", flush=True)


worker_run(1)
worker_run(2)
raw16_pinned = ctx.pinned_empty(raw16.shape, np.uint16)
raw16_pinned[...] = raw16
raw16 = raw16_pinned
print("the same with the decoder's image in page-locked memory:")
worker_run(1)
worker_run(2)

# The reference runs several images at once, one worker thread each (4 by default, image_space.py:238): one context
# (own stream, own staging buffers) per thread, so one thread's upload overlaps another's download and kernels.
import threading

from scanspaceimageprocessor_amd import Context


def threads_run(n_threads, per_thread=6):
    ctxs = [Context(0) for _ in range(n_threads)]
    for c in ctxs:
        c.chain_run_host(d16, img)  # warm-up: staging buffers, workspaces
    barrier = threading.Barrier(n_threads + 1)

    def work(c):
        barrier.wait()
        for _ in range(per_thread):
            c.chain_run_host(d16, img)

    ts = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    for t in ts:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    n = n_threads * per_thread
    print(f"{n_threads} worker threads x {per_thread} images, ss_chain_run_host -> u16: {dt / n * 1e3:8.1f} ms per image  "
          f"{n * H * W / 1e6 / dt:8.0f} MP/s aggregate", flush=True)


for n_threads in (1, 4):
    threads_run(n_threads)
