"""Analytic known-answer tests for the restated third-party leaves
(colour-science / OpenCV / numpy percentile).  These are the only guard the
leaves have: the reference ships no fixture for them (parity unpinned)."""

import numpy as np
import pytest
import scipy.ndimage

from oracle import leaves as lv
from oracle import inputs


def test_srgb_cctf_known_values_and_round_trip():
    assert lv.eotf_inverse_sRGB(0.0) == 0.0
    assert lv.eotf_inverse_sRGB(1.0) == pytest.approx(1.0, abs=1e-15)
    assert lv.eotf_inverse_sRGB(0.0031308) == pytest.approx(0.040449936, abs=1e-12)
    assert lv.eotf_inverse_sRGB(0.5) == pytest.approx(0.7353569830524495, abs=1e-12)
    assert lv.eotf_sRGB(0.5) == pytest.approx(0.21404114048223255, abs=1e-12)
    # negative values go through the linear segment, values > 1 through the power segment
    assert lv.eotf_inverse_sRGB(-0.25) == pytest.approx(-0.25 * 12.92)
    assert lv.eotf_sRGB(-0.25) == pytest.approx(-0.25 / 12.92)
    x = np.linspace(-0.2, 1.4, 2001)
    np.testing.assert_allclose(lv.eotf_sRGB(lv.eotf_inverse_sRGB(x)), x, atol=1e-7)


def test_spow_is_sign_preserving():
    np.testing.assert_allclose(lv.spow(np.array([-8.0, 0.0, 27.0]), 1 / 3), [-2.0, 0.0, 3.0], atol=1e-14)


def test_white_and_black_in_lab():
    lab = lv.XYZ_to_Lab(lv.sRGB_to_XYZ(np.array([[1.0, 1.0, 1.0], [0.0, 0.0, 0.0]])))
    assert lab[0, 0] == pytest.approx(100.0, abs=1e-9)
    assert abs(lab[0, 1]) < 0.02 and abs(lab[0, 2]) < 0.02  # rounded sRGB matrix vs D65 xy
    np.testing.assert_allclose(lab[1], [0.0, 0.0, 0.0], atol=1e-12)
    # XYZ of the D65 white itself is exactly (100, 0, 0)
    np.testing.assert_allclose(lv.XYZ_to_Lab(lv.D65_XYZ[None]), [[100.0, 0.0, 0.0]], atol=1e-12)


def test_lab_of_the_srgb_primaries_against_published_values():
    # CIELAB (D65) of the sRGB primaries and secondaries as commonly tabulated (e.g. Lindbloom's calculator, which
    # uses an sRGB matrix with more digits than the 4-digit IEC one colour-science applies: agreement to ~0.03)
    published = {
        (1, 0, 0): (53.2408, 80.0925, 67.2032),
        (0, 1, 0): (87.7347, -86.1827, 83.1793),
        (0, 0, 1): (32.2970, 79.1875, -107.8602),
        (1, 1, 0): (97.1393, -21.5537, 94.4780),
        (0, 1, 1): (91.1132, -48.0875, -14.1312),
        (1, 0, 1): (60.3242, 98.2343, -60.8249),
    }
    for rgb, lab in published.items():
        got = lv.XYZ_to_Lab(lv.sRGB_to_XYZ(np.array([rgb], dtype=np.float64)))[0]
        np.testing.assert_allclose(got, lab, atol=0.05)


def test_colour_science_docstring_examples():
    # the worked examples in the docstrings of colour.XYZ_to_Lab / Lab_to_XYZ / sRGB_to_XYZ / XYZ_to_sRGB /
    # eotf_inverse_sRGB (colour-science 0.4.x), eight digits as printed there
    XYZ = np.array([0.20654008, 0.12197225, 0.05136952])
    np.testing.assert_allclose(lv.XYZ_to_Lab(XYZ), [41.52787529, 52.63858304, 26.92317922], atol=5e-9)
    np.testing.assert_allclose(lv.Lab_to_XYZ(np.array([41.52787529, 52.63858304, 26.92317922])), XYZ, atol=5e-9)
    np.testing.assert_allclose(lv.XYZ_to_sRGB(XYZ), [0.70573936, 0.19248266, 0.22354169], atol=5e-9)
    np.testing.assert_allclose(lv.sRGB_to_XYZ(np.array([0.70573936, 0.19248266, 0.22354169])),
                               [0.2065429, 0.1219794, 0.0513714], atol=5e-8)
    assert lv.eotf_inverse_sRGB(0.18) == pytest.approx(0.46135612950044164, abs=1e-15)
    # the two IEC matrices are rounded independently: the round trip is the identity only to ~4e-5
    rt = lv.MATRIX_XYZ_TO_sRGB @ lv.MATRIX_sRGB_TO_XYZ
    assert 1e-5 < np.max(np.abs(rt - np.eye(3))) < 5e-5


def test_lab_and_xyz_round_trips():
    rng = np.random.default_rng(0)
    rgb = rng.random((500, 3))
    xyz = lv.sRGB_to_XYZ(rgb)
    np.testing.assert_allclose(lv.Lab_to_XYZ(lv.XYZ_to_Lab(xyz)), xyz, atol=1e-12)
    # sRGB -> XYZ -> sRGB is exact only up to the rounding of the IEC matrix pair (and the encode's slope)
    back = lv.XYZ_to_sRGB(xyz)
    assert 1e-6 < np.max(np.abs(back - rgb)) < 3e-4
    np.testing.assert_allclose(lv.eotf_sRGB(back), (lv.MATRIX_XYZ_TO_sRGB @ lv.MATRIX_sRGB_TO_XYZ @ lv.eotf_sRGB(rgb).T).T,
                               atol=1e-12)
    # mid grey: L* of Y=0.18 is 49.496
    assert lv.XYZ_to_Lab(np.array([[0.18 * lv.D65_XYZ[0], 0.18, 0.18 * lv.D65_XYZ[2]]]))[0, 0] == pytest.approx(49.4961, abs=1e-3)


@pytest.mark.parametrize("terms", lv.CHEUNG_TERMS)
def test_cheung_expansion_shapes(terms):
    rgb = np.random.default_rng(1).random((10, 3))
    e = lv.matrix_augmented_Cheung2004(rgb, terms)
    assert e.shape == (10, terms)
    np.testing.assert_array_equal(e[:, :3], rgb)
    if terms in (5, 7, 8, 10, 11, 14, 17, 20):
        np.testing.assert_array_equal(e[:, -1], 1.0)


def test_cheung_bad_terms():
    with pytest.raises(ValueError):
        lv.matrix_augmented_Cheung2004(np.zeros((1, 3)), 4)


@pytest.mark.parametrize("degree", [1, 2, 3, 4])
def test_root_polynomial_is_degree_one_homogeneous(degree):
    """Finlayson 2015's point: root-polynomial terms scale linearly with exposure."""
    rgb = np.random.default_rng(2).random((50, 3))
    e1 = lv.polynomial_expansion_Finlayson2015(rgb, degree, True)
    e2 = lv.polynomial_expansion_Finlayson2015(rgb * 3.5, degree, True)
    assert e1.shape == (50, lv.ROOTPOLY_TERMS[degree])
    np.testing.assert_allclose(e2, 3.5 * e1, rtol=1e-12)


def test_colour_correction_identity_and_recovery():
    M_T, M_R = inputs.synthetic_swatches()
    img = inputs.linear_image(3, 16, 20)
    # same chart on both sides -> identity
    np.testing.assert_allclose(lv.colour_correction(img, M_R, M_R), img.astype(np.float64), atol=1e-12)
    # an exact linear camera -> the fit recovers its inverse
    A = np.array([[1.1, -0.05, 0.02], [0.03, 0.9, 0.08], [-0.02, 0.1, 1.2]])
    ccm = lv.matrix_colour_correction(M_R @ A.T, M_R)
    np.testing.assert_allclose(ccm, np.linalg.inv(A), atol=1e-10)
    out = lv.colour_correction(img, M_T, M_R)
    assert out.shape == img.shape and out.dtype == np.float64


def test_usm_kernel_size_rule():
    assert [lv.usm_kernel_size(s) for s in (0.5, 1.0, 1.5, 2.0, 5.0)] == [5, 7, 11, 13, 31]


def test_gaussian_kernel_properties():
    for k, s in ((7, 1.0), (13, 2.0), (11, 1.5)):
        g = lv.gaussian_kernel_f32(k, s)
        assert g.dtype == np.float32 and g.shape == (k,)
        assert abs(float(g.astype(np.float64).sum()) - 1.0) < 1e-6
        np.testing.assert_array_equal(g, g[::-1])
        assert g[k // 2] == g.max()
    # sigma=1, k=7: centre tap = 1/sum(exp(-i^2/2)), i=-3..3
    i = np.arange(-3, 4)
    assert lv.gaussian_kernel_f32(7, 1.0)[3] == np.float32(1.0 / np.exp(-i * i / 2.0).sum())


def test_gaussian_taps_and_cheung_terms_against_published_values():
    # cv2.getGaussianKernel(7, 1.0) and (5, 1.0) as commonly printed (8 digits)
    np.testing.assert_allclose(lv.gaussian_kernel_f32(7, 1.0)[:4], [0.00443305, 0.05400558, 0.24203623, 0.39905028], atol=2e-8)
    np.testing.assert_allclose(lv.gaussian_kernel_f32(5, 1.0)[:3], [0.05448868, 0.24420134, 0.40261996], atol=2e-8)
    # colour.characterisation.matrix_augmented_Cheung2004's docstring example (terms=5)
    np.testing.assert_allclose(lv.expand(np.array([[0.1722481, 0.0917066, 0.06416938]]), "Cheung 2004", terms=5)[0],
                               [0.1722481, 0.0917066, 0.06416938, 0.00101364, 1.0], atol=5e-9)


def test_blur_constant_and_border_against_scipy():
    const = np.full((20, 30), 0.37, dtype=np.float32)
    np.testing.assert_allclose(lv.gaussian_blur_reflect(const, 13, 2.0), const, atol=2e-7)
    rng = np.random.default_rng(5)
    x = rng.random((37, 53)).astype(np.float32)
    for k, s in ((7, 1.0), (13, 2.0), (31, 5.0)):
        taps = lv.gaussian_kernel_f32(k, s).astype(np.float64)
        # scipy 'reflect' == OpenCV BORDER_REFLECT (edge pixel repeated)
        ref = scipy.ndimage.correlate1d(x.astype(np.float64), taps, axis=1, mode="reflect")
        ref = scipy.ndimage.correlate1d(ref.astype(np.float32).astype(np.float64), taps, axis=0, mode="reflect")
        got = lv.gaussian_blur_reflect(x, k, s)
        assert got.dtype == np.float32
        np.testing.assert_allclose(got, ref, atol=5e-7)


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 296815])
@pytest.mark.parametrize("q", [0.0, 0.5, 5.0, 50.0, 99.5, 100.0])
def test_percentile_matches_numpy(n, q):
    rng = np.random.default_rng(n)
    x = rng.random((n, 3)).astype(np.float32)
    val, x_lo, x_hi, lo, gamma = lv.percentile_linear(x, q, legacy=False)
    np.testing.assert_array_equal(val, np.percentile(x, q, axis=0))  # numpy>=2 semantics, bit-exact
    srt = np.sort(x, axis=0)
    np.testing.assert_array_equal(x_lo, srt[lo])
    np.testing.assert_array_equal(x_hi, srt[min(lo + 1, n - 1)] if gamma != 0 or lo + 1 < n else srt[lo])
    # numpy<2 semantics == float64 index arithmetic and interpolation
    val64, *_ = lv.percentile_linear(x, q, legacy=True)
    assert val64.dtype == np.float64
    np.testing.assert_allclose(val64, np.percentile(x.astype(np.float64), q, axis=0), rtol=0, atol=1e-7)
