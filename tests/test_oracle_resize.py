"""Known-answer tests of the area-resampling restatement (oracle/resize.py).  OpenCV is not available, so these
pin the restatement to properties and hand-computed cases of the published algorithm, not to library output."""

import numpy as np
import pytest

from oracle import resize as orz


def test_integer_scale_is_the_block_mean():
    rng = np.random.default_rng(0)
    img = rng.random((12, 20, 3))
    got = orz.resize_area(img, (5, 4))  # scale 4 in x, 3 in y
    want = img.reshape(4, 3, 5, 4, 3).mean(axis=(1, 3))
    assert np.max(np.abs(got - want)) < 1e-7


def test_same_size_is_identity_and_weights_sum_to_one():
    rng = np.random.default_rng(1)
    img = rng.random((7, 9, 3))
    assert np.max(np.abs(orz.resize_area(img, (9, 7)) - img)) < 1e-7
    for ssize, dsize in [(6000, 1224), (4000, 816), (4000, 341), (17, 5), (9, 9), (10, 1), (3, 2)]:
        for row in orz.area_table(ssize, dsize):
            assert abs(sum(float(a) for _, a in row) - 1.0) < 1e-6
            assert all(0 <= sx < ssize for sx, _ in row)


def test_hand_computed_fractional_case():
    # 5 -> 2: scale 2.5; cell 0 covers [0, 2.5): 1, 1, half of index 2; cell 1 covers [2.5, 5): half of 2, 3, 4
    t = orz.area_table(5, 2)
    assert [(sx, round(float(a), 6)) for sx, a in t[0]] == [(0, 0.4), (1, 0.4), (2, 0.2)]
    assert [(sx, round(float(a), 6)) for sx, a in t[1]] == [(2, 0.2), (3, 0.4), (4, 0.4)]
    x = np.arange(5, dtype=np.float64).reshape(1, 5)
    got = orz.resize_area(x, (2, 1))
    assert np.allclose(got, [[0.4 * 0 + 0.4 * 1 + 0.2 * 2, 0.2 * 2 + 0.4 * 3 + 0.4 * 4]], atol=1e-7)


def test_constant_images_stay_constant_and_the_mean_is_kept():
    img = np.full((33, 47, 3), 0.37)
    assert np.max(np.abs(orz.resize_area(img, (10, 7)) - 0.37)) < 1e-6
    rng = np.random.default_rng(2)
    img = rng.random((600, 900, 3))
    out = orz.resize_area(img, (183, 122))
    assert abs(out.mean() - img.mean()) < 2e-4


def test_upscaling_is_rejected():
    with pytest.raises(ValueError):
        orz.area_table(4, 5)


def test_reference_size_rules():
    # imageLoader.py:104-111: 24 MP -> 1 MP
    assert orz.preview_size(4000, 6000, 1_000_000) == (1224, 816)
    assert orz.preview_size(800, 1000, 1_000_000) is None
    # image_space.py:2977-2980: 45 MP -> 4 MP
    assert orz.preview_size(5504, 8256, 4_000_000) == (2449, 1632)
    # imageLoader.py:299-302
    assert orz.thumbnail_size(4000, 6000, (512, 512)) == (512, 341)
    assert orz.thumbnail_size(300, 400, (512, 512)) is None
