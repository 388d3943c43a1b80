"""Shadow / highlight region masking (the worker's _apply_masking) through the C ABI, against
oracle/masking.py.  The zeroed pixel set and its size must match exactly."""

import numpy as np
import pytest

from gpu_util import DeviceBuffer, fuzz_seeds
from oracle import inputs
from oracle import masking as om
from scanspaceimageprocessor_amd import default_context

pytestmark = pytest.mark.gpu


def run_mask(imgs, shadow=0.05, highlight=0.98, min_area=200, float32_luminance=False):
    ctx = default_context()
    imgs = np.ascontiguousarray(imgs, dtype=np.float32)
    if imgs.ndim == 3:
        imgs = imgs[None]
    b, h, w, _ = imgs.shape
    buf = DeviceBuffer.from_array(ctx, imgs)
    cnt = DeviceBuffer(ctx, 8 * b)
    try:
        ctx.mask_regions(buf.ptr, h, w, b, shadow, highlight, min_area, cnt.ptr, float32_luminance=float32_luminance)
        ctx.sync()
        return buf.to_array(imgs.shape, np.float32), cnt.to_array((b,), np.uint64)
    finally:
        buf.free(); cnt.free()


def check(imgs, **kw):
    imgs = np.ascontiguousarray(imgs, dtype=np.float32)
    if imgs.ndim == 3:
        imgs = imgs[None]
    got, counts = run_mask(imgs, **kw)
    for i in range(imgs.shape[0]):
        want, n = om.apply_masking(imgs[i], kw.get("shadow", 0.05), kw.get("highlight", 0.98), kw.get("min_area", 200),
                                   kw.get("float32_luminance", False))
        np.testing.assert_array_equal(got[i], want.astype(np.float32))
        assert int(counts[i]) == n


def from_mask(mask, seed=0):
    """An image whose candidate mask is exactly `mask` (dark where set, mid-grey noise elsewhere)."""
    rng = np.random.default_rng(seed)
    img = (0.2 + 0.6 * rng.random(mask.shape + (3,))).astype(np.float32)
    img[mask] = 0.01
    return img


def test_blobs_and_specks():
    rng = np.random.default_rng(1)
    h, w = 300, 500
    mask = np.zeros((h, w), bool)
    for _ in range(40):
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(2, 25)
        yy, xx = np.ogrid[:h, :w]
        mask |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    mask |= rng.random((h, w)) < 0.02  # specks, most of them below the area threshold
    check(from_mask(mask))


@pytest.mark.parametrize("density", [0.3, 0.4, 0.5, 0.6])
def test_random_masks_around_the_percolation_threshold(density):
    rng = np.random.default_rng(int(density * 100))
    check(from_mask(rng.random((257, 389)) < density, seed=3), min_area=50)


def test_long_thin_structures():
    h, w = 200, 1300  # wider than one scan chunk of 1024 pixels
    mask = np.zeros((h, w), bool)
    mask[::4, :] = True            # long rows ...
    mask[:, 5] = True              # ... joined by one column into a comb
    mask[1::4, 700:] = False
    spiral = np.zeros((h, w), bool)
    for k in range(0, 90, 4):      # nested rectangles linked at a corner each: a long winding component
        spiral[k, k:w - k] = True
        spiral[h - 1 - k, k:w - k] = True
        spiral[k:h - k, k] = True
        spiral[k:h - k, w - 1 - k] = True
    check(from_mask(mask))
    check(from_mask(spiral))
    diag = np.zeros((300, 300), bool)
    idx = np.arange(300)
    diag[idx, idx] = True          # 8-connectivity only
    diag[idx[:-1], idx[:-1] + 1] |= (idx[:-1] % 7 == 0)
    check(from_mask(diag))
    anti = diag[:, ::-1].copy()    # contacts arrive from the upper right
    check(from_mask(anti))


@pytest.mark.parametrize("w", [1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4100])
def test_row_widths_around_the_scan_boundaries(w):
    # runs that start, end or continue across lane / wave / stripe / chunk boundaries of the row scan, and
    # runs touching either border
    rng = np.random.default_rng(w)
    h = 9
    mask = rng.random((h, w)) < 0.55
    mask[0, :] = True                      # one run as long as the row
    mask[1, :] = False
    mask[2, ::2] = True                    # the most runs a row can hold, starting at column 0 ...
    mask[2, 1::2] = False
    mask[3, ::2] = False                   # ... and at column 1: a checkerboard pair, 8-connected
    mask[3, 1::2] = True
    mask[4, :] = False
    mask[5, -1] = True
    mask[6, 0] = True
    check(from_mask(mask, seed=1), min_area=3)
    check(from_mask(mask, seed=1), min_area=1)


def test_checkerboard_is_one_component():
    h, w = 64, 301
    yy, xx = np.indices((h, w))
    check(from_mask((yy + xx) % 2 == 0), min_area=h * w // 2)
    check(from_mask((yy + xx) % 2 == 0), min_area=h * w // 2 + 2)   # one more than it has: nothing is masked


def test_single_row_and_single_column():
    rng = np.random.default_rng(11)
    check(from_mask(rng.random((1, 3000)) < 0.7), min_area=4)
    check(from_mask(rng.random((3000, 1)) < 0.7), min_area=4)
    check(from_mask(np.ones((1, 1), bool)), min_area=1)


def test_wide_rows_with_many_runs():
    rng = np.random.default_rng(12)
    check(from_mask(rng.random((40, 5000)) < 0.45, seed=2), min_area=30)
    # vertical stripes joined along the bottom row: every row holds w / 3 runs of one component
    mask = np.zeros((50, 4500), bool)
    mask[:, ::3] = True
    mask[-1, :] = True
    check(from_mask(mask), min_area=1000)


@pytest.mark.parametrize("seed", fuzz_seeds(60))
def test_random_masks_shapes_and_thresholds(seed):
    rng = np.random.default_rng(7000 + seed)
    h, w = int(rng.integers(1, 90)), int(rng.integers(1, 700))
    density = float(rng.choice([0.05, 0.2, 0.45, 0.55, 0.8, 0.97]))
    mask = rng.random((h, w)) < density
    if rng.random() < 0.5:   # coarser structure: blocks of 3 x 5 pixels
        mask = np.kron(rng.random((h // 3 + 1, w // 5 + 1)) < density, np.ones((3, 5), bool))[:h, :w]
    b = int(rng.integers(1, 4))
    imgs = np.stack([from_mask(np.roll(mask, i, axis=1), seed=seed + i) for i in range(b)])
    check(imgs, min_area=int(rng.choice([1, 2, 7, 50, 200, 5000])))


def test_area_threshold_and_limits_are_inclusive():
    mask = np.zeros((60, 60), bool)
    mask[2:12, 3:23] = True    # exactly 200
    mask[30:40, 3:22] = True   # 190
    check(from_mask(mask))
    img = np.full((20, 20, 3), 0.5, np.float32)
    img[:, :10] = np.float32(0.05)   # float32(0.05) > 0.05 in float64: not a candidate by the shadow limit
    img[:, 10:] = np.float32(0.98)   # float32(0.98) < 0.98: not a candidate either
    check(img, min_area=1)
    img[:, :10] = np.float32(0.049999997)
    check(img, min_area=1)
    check(inputs.linear_image(5, 120, 160), shadow=0.02, highlight=0.5, min_area=3)


def test_everything_and_nothing_masked_and_a_batch():
    check(np.zeros((50, 70, 3), np.float32))
    check(np.full((50, 70, 3), 0.5, np.float32))
    imgs = np.stack([inputs.linear_image(20 + i, 180, 240) for i in range(3)])
    check(imgs, shadow=0.03, highlight=0.6, min_area=20)


def test_24mp_dark_background_with_an_object():
    h, w = 4000, 6000
    rng = np.random.default_rng(7)
    img = (0.01 * rng.random((h, w, 3))).astype(np.float32)            # dark backdrop: one huge component
    yy, xx = np.ogrid[:h, :w]
    obj = ((yy - 2000) / 1500.0) ** 2 + ((xx - 3000) / 2200.0) ** 2 <= 1.0
    img[obj] = (0.08 + 0.8 * rng.random((int(obj.sum()), 3))).astype(np.float32)
    holes = obj & (rng.random((h, w)) < 0.001)                          # dark specks inside the object: too small
    img[holes] = 0.0
    img[1900:2100, 2900:3100] = 0.995                                   # a blown highlight patch
    check(img)


def _pixels_on_the_limit(limit, seed):
    """Colourful float32 pixels whose luminance is `limit` to within a float32 rounding, and their two tests."""
    rng = np.random.default_rng(seed)
    f = np.float32
    rg = rng.random((2000000, 2)).astype(np.float32) * f(0.2)
    b = (limit - 0.2126 * rg[:, 0].astype(np.float64) - 0.7152 * rg[:, 1]) / 0.0722
    ok = (b > 0) & (b < 1)
    px = np.stack([rg[ok, 0], rg[ok, 1], b[ok].astype(np.float32)], axis=-1)
    lum32 = (f(0.2126) * px[:, 0] + f(0.7152) * px[:, 1]) + f(0.0722) * px[:, 2]
    lum64 = 0.2126 * px[:, 0].astype(np.float64) + 0.7152 * px[:, 1] + 0.0722 * px[:, 2]
    return px, lum32 <= f(limit), lum64 <= limit


@pytest.mark.parametrize("limit", [0.05, 0.06])   # float32(0.05) > 0.05, float32(0.06) < 0.06
def test_luminance_in_float32_or_float64_at_the_limits(limit):
    """Pixels whose luminance lands on the other side of the shadow limit in float32 than in float64 (numpy < 2
    tests a float32 image in float32: no chart, exposure / white balance only; imageProcessorWorker.py:989-993):
    a block of them, large enough to be a region, is zeroed in one mode and kept in the other, as the oracle says."""
    px, in32, in64 = _pixels_on_the_limit(limit, 21)
    differ = px[in32 != in64]
    assert len(differ) >= 400, len(differ)
    seen32 = bool((in32 & ~in64).sum() > (in64 & ~in32).sum())  # which test sees most of them as shadow
    block = px[(in32 & ~in64) if seen32 else (in64 & ~in32)][:400].reshape(20, 20, 3)
    rng = np.random.default_rng(22)
    img = (0.3 + 0.4 * rng.random((64, 96, 3))).astype(np.float32)
    img[4:24, 4:24] = block
    for mode in (False, True):
        check(img, shadow=limit, float32_luminance=mode)
    got32, n32 = run_mask(img, shadow=limit, float32_luminance=True)
    got64, n64 = run_mask(img, shadow=limit, float32_luminance=False)
    assert (int(n32[0]), int(n64[0])) == ((400, 0) if seen32 else (0, 400))
    assert got32[0, 4:24, 4:24].any() != seen32 and got64[0, 4:24, 4:24].any() == seen32
