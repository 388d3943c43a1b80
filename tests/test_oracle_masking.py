"""Hand-made cases for the masking restatement (oracle/masking.py): connectivity, the area threshold, the
luminance limits."""

import numpy as np

from oracle import masking as om


def test_diagonal_neighbours_are_connected():
    m = np.zeros((40, 40), bool)
    for i in range(30):
        m[i, i] = True  # a pure diagonal: one component under 8-connectivity, 30 singletons under 4
    assert om.clean_mask(m, 30).sum() == 30
    assert om.clean_mask(m, 31).sum() == 0


def test_area_threshold_is_inclusive():
    m = np.zeros((50, 50), bool)
    m[2:12, 3:23] = True     # 200 pixels
    m[30:40, 3:22] = True    # 190 pixels
    m[30:35, 40:42] = True   # 10 pixels
    kept = om.clean_mask(m, 200)
    assert kept[2:12, 3:23].all() and kept.sum() == 200


def test_two_blobs_joined_by_a_corner():
    m = np.zeros((30, 30), bool)
    m[0:10, 0:10] = True
    m[10:20, 10:20] = True  # touches the first only at the corner (9,9)-(10,10)
    assert om.clean_mask(m, 200).sum() == 200
    assert om.clean_mask(m, 201).sum() == 0


def test_luminance_limits_are_inclusive_and_float64():
    img = np.zeros((1, 4, 3))
    img[0, 0] = 0.05                      # lum = 0.05 (to rounding) -> candidate by <=
    img[0, 1] = 0.0500001
    img[0, 2] = 0.98
    img[0, 3] = 0.9799999
    lum = 0.2126 * img[:, :, 0] + 0.7152 * img[:, :, 1] + 0.0722 * img[:, :, 2]
    want = (lum <= 0.05) | (lum >= 0.98)
    np.testing.assert_array_equal(om.candidate_mask(img), want)
    assert not want[0, 1] and not want[0, 3]


def test_apply_masking_zeroes_and_counts():
    rng = np.random.default_rng(0)
    img = 0.2 + 0.6 * rng.random((64, 64, 3))
    img[10:30, 10:30] = 0.01   # 400 dark pixels
    img[50:55, 50:55] = 0.99   # 25 bright pixels: too small
    out, n = om.apply_masking(img)
    assert n == 400 and (out[10:30, 10:30] == 0).all()
    np.testing.assert_array_equal(out[50:55, 50:55], img[50:55, 50:55])


def test_float32_luminance_mode_rounds_products_sum_and_limit():
    """numpy < 2 on a float32 image: 0.2126 * c0 etc. are float32 products, the limit is compared as float32.
    float32(0.05) = 0.0500000007 lies above 0.05: a grey pixel of exactly that value is a shadow candidate in
    float32 (lum == limit) and not in float64 (lum > 0.05)."""
    f = np.float32
    g = f(0.05)
    img = np.full((1, 1, 3), g, dtype=np.float32)
    lum32 = (f(0.2126) * g + f(0.7152) * g) + f(0.0722) * g
    lum64 = 0.2126 * float(g) + 0.7152 * float(g) + 0.0722 * float(g)
    assert om.candidate_mask(img, 0.05, 0.98, float32_luminance=True)[0, 0] == bool(lum32 <= f(0.05))
    assert om.candidate_mask(img, 0.05, 0.98, float32_luminance=False)[0, 0] == bool(lum64 <= 0.05)
    assert lum64 > 0.05  # (the float64 test never takes this pixel)
