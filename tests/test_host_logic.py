"""Host-side logic and the C-ABI surface, without a GPU: the library loads, exports every symbol
the header declares, its structs match the ctypes mirrors, and the pure-host entry points agree
with the oracle.  No kernel is launched here."""

import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import chain as oc
from oracle import inputs
from oracle import leaves as lv
from scanspaceimageprocessor_amd import _lib
from scanspaceimageprocessor_amd.chain import make_desc
from scanspaceimageprocessor_amd.colour_fit import matrix_colour_correction
from scanspaceimageprocessor_amd.dataset import shard_range
from scanspaceimageprocessor_amd.ImageProcessor import editingTools as et
from scanspaceimageprocessor_amd.ImageProcessor import imageProcessorWorker as ipw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "scanspace.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load_library()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/scanspace.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES out of sync with the header"
    assert b"gfx950" in lib.ss_version()


def test_header_is_plain_c_and_struct_layouts_match_ctypes(tmp_path):
    fields = [f[0] for f in _lib.ChainDesc._fields_]
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){',
             'printf("%zu\\n", sizeof(ss_chain_desc));']
    lines += [f'printf("%zu\\n", offsetof(ss_chain_desc, {f}));' for f in fields]
    lines += ['printf("%zu\\n", sizeof(ss_percentile_detail));', 'printf("%zu\\n", sizeof(ss_host_opts));',
              'return 0;}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(src)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == C.sizeof(_lib.ChainDesc)
    for f, off in zip(fields, out[1:1 + len(fields)]):
        assert getattr(_lib.ChainDesc, f).offset == off, f
    assert out[-2] == C.sizeof(_lib.PercentileDetail)
    assert out[-1] == C.sizeof(_lib.HostOpts)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_means_a_loud_error_not_a_fallback():
    with pytest.raises(_lib.ScanSpaceError):
        _lib.Context(0)
    img = inputs.linear_image(1, 8, 8)
    with pytest.raises(_lib.ScanSpaceError):
        et.adjust_exposure(img, 1.0)
    with pytest.raises(_lib.ScanSpaceError):
        et.ensure_positive_if_negative(img, True)  # a missing device is not swallowed


@pytest.mark.parametrize("legacy", [True, False])
def test_percentile_rank_matches_numpy_index_arithmetic(legacy):
    rng = np.random.default_rng(0)
    ns = [1, 2, 3, 10, 1000, 296815, 269664, 357500, 16777217] + list(rng.integers(1, 5_000_000, 40))
    for n in ns:
        for frac in (0.0, 0.005, 0.05, 0.5, 0.995, 1.0, float(rng.random())):
            lo, hi, g = _lib.percentile_rank(int(n), frac, legacy)
            elo, ehi, eg = lv.percentile_ranks(int(n), frac * 100.0, legacy)
            assert (lo, hi) == (elo, ehi), (n, frac)
            assert g == float(eg), (n, frac)


def test_percentile_rank_known_answers_of_the_numpy1_branch():
    """Hand-computed cases of np.percentile's index arithmetic as numpy < 2 does it for float32 data (float64
    throughout): virtual index (n - 1) q, floor / next neighbour, weight, the clamp at q = 1.  Independent of the
    oracle's restatement."""
    kat = [  # n, q -> lo, hi, gamma
        (5, 0.5, 2, 3, 0.0),       # 4 * 0.5 = 2 exactly
        (5, 0.0, 0, 1, 0.0),
        (5, 1.0, 4, 4, 0.0),       # the upper bound: both neighbours are the last element
        (1, 0.3, 0, 0, 0.0),       # a single value
        (11, 0.25, 2, 3, 0.5),     # 10 * 0.25 = 2.5: the weight at which numpy's lerp switches form
        (11, 0.75, 7, 8, 0.5),
        (9, 0.125, 1, 2, 0.0),     # 8 * 0.125 = 1
        (3, 0.75, 1, 2, 0.5),
    ]
    for n, q, lo, hi, g in kat:
        assert _lib.percentile_rank(n, q, True) == (lo, hi, g), (n, q)
    # not representable exactly: 296814 * 0.005 = 1484.07, 999 * 0.995 = 994.005
    lo, hi, g = _lib.percentile_rank(296815, 0.005, True)
    assert (lo, hi) == (1484, 1485) and abs(g - 0.07) < 1e-9
    lo, hi, g = _lib.percentile_rank(1000, 0.995, True)
    assert (lo, hi) == (994, 995) and abs(g - 0.005) < 1e-9
    # numpy >= 2 rounds q and the index to float32: n - 1 = 2^24, q = 0.3 -> float32(0.3) = 10066330 / 2^25, so the
    # index is 5033165 exactly; numpy < 2 keeps float64: 2^24 * 0.3 = 5033164.8
    assert _lib.percentile_rank(16777217, 0.3, False) == (5033165, 5033166, 0.0)
    lo, hi, g = _lib.percentile_rank(16777217, 0.3, True)
    assert (lo, hi) == (5033164, 5033165) and abs(g - 0.8) < 1e-9


def test_sample_step_matches_reference_rule():
    for h, w in [(1, 1), (96, 128), (1100, 1300), (4000, 6000), (5504, 8256), (499, 501), (500, 500), (1000, 1000)]:
        assert _lib.negative_sample_step(h, w) == oc.negative_sample_step(h, w)
    assert _lib.negative_sample_step(4000, 6000) == 9 and _lib.negative_sample_step(5504, 8256) == 13


@pytest.mark.parametrize("method,kw", [("Cheung 2004", dict(terms=t)) for t in lv.CHEUNG_TERMS]
                         + [("Finlayson 2015", dict(degree=d)) for d in (1, 2, 3, 4)])
def test_ccm_fit_matches_oracle(method, kw):
    M_T, M_R = inputs.synthetic_swatches()
    ccm, kind, terms = matrix_colour_correction(M_T, M_R, method=method, **kw)
    want = lv.matrix_colour_correction(M_T, M_R, method, **kw)
    assert ccm.shape == want.shape == (3, terms)
    np.testing.assert_allclose(ccm, want, rtol=1e-9, atol=1e-9)
    assert kind == (_lib.SS_CCM_CHEUNG if method == "Cheung 2004" else _lib.SS_CCM_ROOTPOLY)


def test_ccm_fit_rejects_bad_input():
    M_T, M_R = inputs.synthetic_swatches()
    with pytest.raises(ValueError):
        matrix_colour_correction(M_T, M_R, terms=4)
    with pytest.raises(ValueError):
        matrix_colour_correction(M_T[:, :2], M_R)
    with pytest.raises(ValueError):
        matrix_colour_correction(M_T, M_R, method="Vandermonde")


def test_make_desc_fills_the_descriptor():
    M_T, M_R = inputs.synthetic_swatches()
    d = make_desc(10, 20, 3, neg=([0.1, 0.2, 0.3], [2.0, 3.0, 4.0]), ccm=matrix_colour_correction(M_T, M_R),
                  gains=[1.5, (1.0, 1.1, 1.2)], sh=(0.5, -0.5), usm=(50, 2.0), final_clip=True, out="u16")
    assert (d.height, d.width, d.batch) == (10, 20, 3)
    assert d.neg_enabled == 1 and d.neg_p_low[1] == np.float32(0.2) and d.neg_inv_denom[2] == 4.0
    assert d.ccm_kind == _lib.SS_CCM_CHEUNG and d.ccm_terms == 3
    assert d.n_gain_stages == 2 and list(d.gain[0]) == [1.5, 1.5, 1.5] and d.gain[1][2] == np.float32(1.2)
    assert d.sh_enabled == 1 and d.sh_falloff_power == 7.0 and d.usm_enabled == 1 and d.usm_radius == 2.0
    assert d.final_clip == 1 and d.out_kind == _lib.SS_OUT_U16_SRGB
    with pytest.raises(ValueError):
        make_desc(1, 1, gains=[1, 1, 1, 1])


def test_white_balance_multipliers_known_answers():
    assert et.calculate_white_balance_multipliers(5500, 5500) == pytest.approx((1.0, 1.0, 1.0))
    warm = et.calculate_white_balance_multipliers(6500, 3000)  # towards a warmer white: red up, blue down
    assert min(warm) == pytest.approx(1.0) and warm[0] > warm[1] > warm[2]
    tint = et.calculate_white_balance_multipliers(5500, 5500, 50.0)
    assert tint[1] / tint[0] == pytest.approx(1.05)
    # colour.temperature.CCT_to_xy_CIE_D's docstring example (the default method of CCT_to_xy)
    x, y = et._cct_to_xy_cie_d(6504.38938305)
    assert (x, y) == pytest.approx((0.3127077, 0.3291128), abs=1e-7)   # digits printed there, truncated
    # D65 maps to (almost) equal sRGB channels: from 6504 K to 6504 K nothing changes, from there to 5000 K
    # (0.3457, 0.3587 on the daylight locus) red gains on blue
    x5, y5 = et._cct_to_xy_cie_d(5000.0)
    assert abs(x5 - 0.3457) < 2e-4 and abs(y5 - 0.3587) < 2e-4
    m = et.calculate_white_balance_multipliers(6504.38938305, 5000.0)
    assert m[2] == pytest.approx(1.0) and m[0] > m[1] > 1.0


def test_adjustment_stage_translation_keeps_reference_prescaling_and_order():
    kw = et.adjustment_stages(exposure=1.0, shadows=35.0, highlights=0.5, current_wb=5000, target_wb=6000,
                              sharpen_amount=20.0)
    assert len(kw["gains"]) == 2 and kw["gains"][1] == 2.0  # white balance first, then exposure
    assert kw["sh"] == (35.0 / 70, 0.5 * 1.3, 7.0)
    assert kw["usm"] == (20.0, 1.0, 0.0)
    assert et.adjustment_stages() == dict(gains=[])
    with pytest.raises(NotImplementedError):
        et.adjustment_stages(denoise_strength=10)


def test_worker_stage_selection_follows_the_reference():
    M_T, M_R = inputs.synthetic_swatches()
    ipw.set_reference_swatches(M_R)
    w = ipw.ImageCorrectionWorker(["a.nef"], M_T, "/tmp", None, output_format=".tiff", tiff_bitdepth=16,
                                  sharpen_amount=30.0, shadow_adj=14.0)
    w.image_metadata_map = {"a.nef": {"average_exposure": "1.25"}, "b.nef": {"average_exposure": None}}
    kw = w.chain_stages("a.nef")
    assert kw["gains"] == [1.25] and kw["sh"] == (14.0 / 70, 0.0, 7.0) and kw["final_clip"]
    assert "usm" not in kw  # chart mode never sharpens
    assert w.chain_stages("b.nef")["gains"] == [1.0] and w.chain_stages("missing.nef")["gains"] == [1.0]
    assert w._out_kind(".tiff") == _lib.SS_OUT_U16_SRGB and w._out_kind(".jpg") == _lib.SS_OUT_U8_SRGB
    assert w._out_kind(".exr") == _lib.SS_OUT_F32_LINEAR
    w2 = ipw.ImageCorrectionWorker(["a.nef"], M_T, "/tmp", None, use_chart=False, sharpen_amount=30.0,
                                   exposure_adj=0.5)
    kw2 = w2.chain_stages("a.nef")
    assert "ccm" not in kw2 and kw2["usm"] == (30.0, 1.0, 0.0) and kw2["gains"] == [2.0 ** 0.5]


def test_worker_reports_errors_and_continues(tmp_path):
    M_T, M_R = inputs.synthetic_swatches()
    ipw.set_reference_swatches(M_R)
    sig = ipw.WorkerSignals()
    events = []
    sig.status.connect(lambda *a: events.append(a))
    w = ipw.ImageCorrectionWorker(["bad1.nef", "bad2.nef"], M_T, str(tmp_path), sig)

    def loader(path):
        raise IOError("cannot decode " + path)

    w.raw_loader = loader
    w.run()
    assert [e[1] for e in events] == ["started", "error", "started", "error"]


def test_c_abi_shard_range_is_the_python_one():
    """ss_shard_range (what a host in another language calls) and dataset.shard_range agree everywhere."""
    from scanspaceimageprocessor_amd import _lib

    for n in (0, 1, 2, 7, 63, 64, 65, 1000):
        for world in (1, 2, 3, 4, 8):
            for r in range(world):
                assert _lib.shard_range(n, r, world) == shard_range(n, r, world)
    with pytest.raises(ValueError):
        _lib.shard_range(4, 2, 2)
    with pytest.raises(ValueError):
        _lib.shard_range(-1, 0, 1)


def test_shard_range_partitions_contiguously():
    for n in (0, 1, 7, 64, 1024, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_neutral_settings_need_no_gpu():
    img = inputs.linear_image(1, 8, 8)
    assert et.adjust_exposure(img, 0) is img
    assert et.adjust_shadows_highlights_combined(img, 0, 0) is img
    assert et.adjust_sharpen(img, 0) is img
    assert et.adjust_white_balance(img, 5500, 5500, 0) is img
    assert et.ensure_positive_if_negative(img, enabled=False) is img
    out = et.apply_all_adjustments(img)
    assert out is not img and np.array_equal(out, img)
    g = et.GlobalAdjustments()
    assert g.apply_to_image(img) is img and not g.has_adjustments()


def test_white_balance_picked_from_a_point_against_reference_vectors(golden):
    """sample_white_balance_from_point (reference editingTools.py:643-711): host arithmetic on the picked window."""
    from conftest import golden_input
    from scanspaceimageprocessor_amd.ImageProcessor import editingTools as et

    meta, data = golden
    for name in ("wb_sample_centre", "wb_sample_corner", "wb_sample_wide"):
        img = golden_input(meta[name])
        got = et.sample_white_balance_from_point(img, **meta[name]["kwargs"])
        assert isinstance(got, float)
        np.testing.assert_allclose(got, float(data[name]), rtol=1e-7)
    dark = np.full((9, 9, 3), 0.0005, np.float32)
    assert et.sample_white_balance_from_point(dark, 4, 4, 5100.0) == 5100.0   # too dark to judge: unchanged
    red = np.zeros((9, 9, 3), np.float32) + np.array([0.9, 0.3, 0.001], np.float32)
    assert et.sample_white_balance_from_point(red, 4, 4, 9000.0) == 12000.0   # clamped


def test_cache_hooks_of_the_preview_path_exist_and_hold_nothing():
    from scanspaceimageprocessor_amd.ImageProcessor import editingTools as et

    et.set_current_image_id("a.nef")
    et.clear_adjustment_cache()
    stats = et.get_cache_stats()
    assert stats["entries"] == 0 and set(stats) == {"entries", "memory_mb", "memory_percent", "hits", "misses",
                                                    "evictions", "hit_rate"}
    assert et.check_memory_usage() is False


def test_committed_bench_line_names_the_committed_counter_summary():
    """The bench line quotes counter-derived figures from a file it names with its SHA-256: the committed line and
    the committed file must belong together, and the figures in the line must be the file's."""
    import hashlib
    import json

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    line = json.load(open(os.path.join(root, "profiles", "r03_bench_line.json")))
    roof = line["roofline"]
    raw = open(os.path.join(root, roof["pmc_file"]), "rb").read()
    assert hashlib.sha256(raw).hexdigest() == roof["pmc_sha256"]
    pmc = json.loads(raw)
    assert roof["valu_instructions_per_px"] == pmc["valu_per_px"] and roof["valu_trans_per_px"] == pmc["trans_per_px"]
    assert roof["traffic"] == round((pmc["fetch_bytes_per_px"] + pmc["write_bytes_per_px"]) * roof["px_per_launch"])
    assert roof["bound"] == "valu" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    for name in pmc["sources"]:
        assert os.path.exists(os.path.join(root, "profiles", name))
    assert {"threads4", "threads_all"} <= set(line["cpu_baseline"]) and len(line["others"]) >= 3
    # the vector-ALU figures are priced per instruction class from the same file: a fraction, and its complement
    assert roof["valu_mix_per_px"] == pmc["valu_mix"] and 0.0 < roof["valu_frac"] <= 1.0
    assert abs(roof["valu_frac"] + roof["valu_stall_frac"] - 1.0) < 2e-4
    assert abs(roof["valu_frac"] - roof["valu_priced_cycles_per_64px"] / roof["valu_cycles_per_64px"]) < 1e-3


def test_design_quotes_the_committed_bench_line():
    """DESIGN.md's measurement section quotes value, kernel_ms and frac of the committed line: they must be its."""
    import json
    import re

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    line = json.load(open(os.path.join(root, "profiles", "r03_bench_line.json")))
    text = open(os.path.join(root, "DESIGN.md")).read()
    m = re.search(r"Round 3 line \(`profiles/r03_bench_line\.json`\)[^\n]*", text)
    assert m, "DESIGN.md must quote the round-3 bench line"
    quoted = m.group(0)

    def number(label):
        mm = re.search(label + r"\D{0,12}?([0-9][0-9,]*\.?[0-9]*)", quoted)
        assert mm, (label, quoted)
        return float(mm.group(1).replace(",", ""))

    assert abs(number("`value`") - line["value"]) <= 0.5
    assert abs(number("`kernel_ms`") - line["roofline"]["kernel_ms"]) <= 5e-4
    assert abs(number("`frac`") - line["roofline"]["frac"]) <= 5e-4
