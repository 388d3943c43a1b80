"""The job record of the reference's network mode (networkProcessor.py:507-526) and its translation into a worker
(networkProcessor.py:2752-2806); output naming rules (imageProcessorWorker.py:664-746).  Host logic only."""

import json
import sys
import types

import numpy as np

from scanspaceimageprocessor_amd.ImageProcessor import imageProcessorWorker as ipw
from scanspaceimageprocessor_amd.ImageProcessor.jobs import ProcessingJob, worker_from_job


def test_job_json_round_trip():
    sw = np.linspace(0.05, 0.9, 72).reshape(24, 3).tolist()
    job = ProcessingJob("j1", "/in/a.nef", "/out/a.tiff", sw, {"output_format": ".tiff", "tiff_bitdepth": 16},
                        created_time=12.5)
    back = ProcessingJob.from_json(job.to_json())
    assert back == job
    assert set(json.loads(job.to_json())) == {"job_id", "image_path", "output_path", "swatches", "settings",
                                               "created_time", "start_time", "timeout"}
    assert back.timeout == 300.0 and back.start_time is None


def test_worker_from_job_applies_the_reference_defaults():
    w = worker_from_job(ProcessingJob("j", "/in/a.nef", "/out/sub/a.jpg", None, {}))
    assert (w.images, w.output_folder, w.network_output_path) == (["/in/a.nef"], "/out/sub", "/out/sub/a.jpg")
    assert w.from_network and not w.use_chart and w.swatches is None
    assert (w.jpeg_quality, w.output_format, w.tiff_bitdepth, w.exr_colorspace) == (100, ".jpg", 8, "sRGB")
    assert (w.exposure_adj, w.shadow_adj, w.highlight_adj, w.white_balance_adj) == (0.0, 0.0, 0.0, 5500)
    assert (w.denoise_strength, w.sharpen_amount, w.use_export_schema) == (0.0, 0.0, False)
    assert w.construct_export_path("/in/a.nef") == "/out/sub/a.jpg"     # the server's path wins in network mode


def test_worker_from_job_settings_and_white_balance_switch():
    sw = np.full((24, 3), 0.5).tolist()
    st = {"output_format": ".tiff", "tiff_bitdepth": 16, "exposure_adj": 0.3, "shadow_adj": 12.0, "highlight_adj": -5.0,
          "white_balance_adj": 4200, "sharpen_amount": 30.0, "jpeg_quality": 90, "group_name": "g"}
    w = worker_from_job(ProcessingJob("j", "a.nef", "/o/a.tiff", sw, st))
    assert w.use_chart and w.white_balance_adj == 5500      # white balance is ignored unless enabled
    assert (w.exposure_adj, w.shadow_adj, w.highlight_adj, w.sharpen_amount) == (0.3, 12.0, -5.0, 30.0)
    assert (w.output_format, w.tiff_bitdepth, w.jpeg_quality, w.group_name) == (".tiff", 16, 90, "g")
    w = worker_from_job(ProcessingJob("j", "a.nef", "/o/a.tiff", sw, dict(st, enable_white_balance=True)))
    assert w.white_balance_adj == 4200


def test_export_path_rules(tmp_path, monkeypatch):
    out = str(tmp_path)
    mk = lambda **kw: ipw.ImageCorrectionWorker(["/in/IMG_0042.nef"], None, out, None, output_format=".png",
                                                use_chart=False, **kw)
    p = "/in/IMG_0042.nef"
    assert mk().construct_export_path(p) == f"{out}/IMG_0042.png"
    w = mk(rename_map={p: 7}, name_base="scan", padding=4)
    assert w.construct_export_path(p) == f"{out}/scan_0007.png"
    w.use_original_filenames = True
    assert w.construct_export_path(p) == f"{out}/IMG_0042.png"
    assert mk(rename_map={p: 7}).construct_export_path(p) == f"{out}/IMG_0042.png"          # no base name
    # network mode without a server path: sequence defaults to 1
    w = mk(from_network=True, name_base="scan", padding=3)
    assert w.construct_export_path(p) == f"{out}/scan_001.png"
    # a naming schema needs the host application's module; without it the default name is used and the error logged
    logs = []
    w = mk(export_schema="[r]/[n]", use_export_schema=True)
    w.signals.log.connect(logs.append)
    assert w.construct_export_path(p) == f"{out}/IMG_0042.png"
    assert logs and logs[0].startswith("[Schema Error]")
    # ... and with it, the schema's answer (directories are created)
    mod = types.ModuleType("fileNamingSchema")
    seen = {}

    def apply_naming_schema(**kw):
        seen.update(kw)
        return f"{out}/deep/er/x.png"

    mod.apply_naming_schema = apply_naming_schema
    monkeypatch.setitem(sys.modules, "fileNamingSchema", mod)
    w = mk(export_schema="[r]/[n]", use_export_schema=True, custom_name="c", group_name="g", rename_map={p: 3})
    assert w.construct_export_path(p) == f"{out}/deep/er/x.png"
    assert (tmp_path / "deep" / "er").is_dir()
    assert seen == dict(schema="[r]/[n]", input_path=p, output_base_dir=out, custom_name="c", image_number=3,
                        output_extension=".png", root_folder="", group_name="g")


class _Log:
    def __init__(self):
        self.lines = []
        self.log = types.SimpleNamespace(emit=self.lines.append)
        self.status = self.preview = types.SimpleNamespace(emit=lambda *a: None)


def test_as_shot_white_balance_follows_the_reference_estimate(monkeypatch):
    """imageProcessorWorker.py:397-478: camera multipliers -> red/blue ratio -> three linear pieces, clamped; each
    outcome logged; daylight (with its log line) when nothing is known."""
    est = ipw.ImageCorrectionWorker.wb_temperature_from_multipliers
    assert est([2.0, 1.0, 1.0, 1.0]) == 2000 + (2.0 / (1.0 + 1e-10) - 1.5) * 1000   # very warm
    assert est([1.2, 1.0, 1.0]) == 3000 + (1.2 / (1.0 + 1e-10) - 1.0) * 4000   # warm to neutral
    assert est([1.0, 1.0, 2.0]) == 5000 + (1.0 - 1.0 / (2.0 + 1e-10)) * 5000   # cool
    assert est([30.0, 1.0, 1.0]) == 12000.0 and est([1.0, 0.0, 1.0]) is None and est([1.0, 1.0]) is None
    assert est(None) is None

    sig = _Log()
    w = worker_from_job(ProcessingJob("j", "/in/a.nef", "/out/a.jpg", None, {}), sig)

    class Raw:
        camera_whitebalance = [2.1, 1.0, 1.4, 1.0]

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    monkeypatch.setitem(sys.modules, "rawpy", types.SimpleNamespace(imread=lambda path: Raw()))
    want = 3000 + (2.1 / (1.4 + 1e-10) - 1.0) * 4000
    assert w._extract_white_balance_from_exif("/in/a.nef") == want
    assert sig.lines[-1] == f"[WB] Found white balance from rawpy: {want}K"
    # a value the caller already knows wins and opens no file
    w.image_metadata_map = {"/in/a.nef": {"current_wb": 4300}}
    assert w._extract_white_balance_from_exif("/in/a.nef") == 4300.0
    # neither LibRaw nor EXIF: daylight, and the log says that it is a default
    w.image_metadata_map = {}
    monkeypatch.setitem(sys.modules, "rawpy", None)
    monkeypatch.setitem(sys.modules, "OpenImageIO", None)
    sig.lines.clear()
    assert w._extract_white_balance_from_exif("/in/b.nef") == 5500.0
    assert sig.lines[-1] == "[WB] Using default white balance: 5500.0K"
    assert any(l.startswith("[WB] Error reading rawpy white balance") for l in sig.lines)


def _project(tmp_out="/out"):
    sw = np.linspace(0.05, 0.9, 72).reshape(24, 3).round(6).tolist()
    return {
        "metadata": {"export_format_version": "1.0"},
        "processing_settings": {"output_directory": tmp_out, "export_format": ".tiff", "output_format": ".tiff",
                                "tiff_bitdepth": 16, "exposure_adj": 0.25, "root_folder": "scan_042",
                                "use_export_schema": False, "export_schema": ""},
        "images": [
            {"index": 0, "filename": "Group A", "full_path": "", "group": "A"},                       # a header row
            {"index": 1, "filename": "a1.nef", "full_path": "/in/A/a1.nef", "group": "A", "selected": True},
            {"index": 2, "filename": "b1.nef", "full_path": "/in/B/b1.nef", "group": "B", "selected": True},
            {"index": 3, "filename": "a2.nef", "full_path": "/in/A/a2.nef", "group": "A", "selected": True},
            {"index": 4, "filename": "hdr", "full_path": "/x", "group": "A", "metadata": {"is_group_header": True}},
            {"index": 5, "filename": "c1.nef", "full_path": "/in/C/c1.nef", "group": "C", "selected": True},
        ],
        "image_groups": {
            "A": {"has_calibration": True, "chart_swatches": sw, "chart_swatches_count": 24},
            "B": {"has_calibration": True, "chart_swatches": (np.array(sw) * 0.9).tolist(), "chart_swatches_count": 24},
            "C": {"has_calibration": False, "chart_swatches": [], "chart_swatches_count": 0},
        },
    }


def test_jobs_of_a_saved_project_follow_the_server():
    """image_space.py:1844-2100 (the exported structure) through networkProcessor.py:1166-1320: header rows and
    images without a path are skipped, a group without 24 swatches fails its images, settings travel whole."""
    from scanspaceimageprocessor_amd.ImageProcessor.jobs import jobs_from_project

    ticks = iter(range(100, 200))
    jobs, failed = jobs_from_project(_project(), clock=lambda: float(next(ticks)))
    assert [j.image_path for j in jobs] == ["/in/A/a1.nef", "/in/B/b1.nef", "/in/A/a2.nef"]
    assert [j.output_path for j in jobs] == ["/out/a1.tiff", "/out/b1.tiff", "/out/a2.tiff"]
    assert len({j.job_id for j in jobs}) == 3 and all(j.job_id.startswith("proj_") for j in jobs)
    assert jobs[0].swatches == _project()["image_groups"]["A"]["chart_swatches"]
    assert jobs[1].swatches == _project()["image_groups"]["B"]["chart_swatches"]
    assert jobs[0].settings["exposure_adj"] == 0.25
    assert failed == [{"image_path": "/in/C/c1.nef", "group": "C",
                       "error": "Group 'C' has invalid chart swatches (length: 0)"}]
    w = worker_from_job(jobs[0])
    assert w.use_chart and w.output_format == ".tiff" and w.tiff_bitdepth == 16 and w.exposure_adj == 0.25


def test_project_naming_schema_gets_a_sequence_number_per_group(monkeypatch):
    from scanspaceimageprocessor_amd.ImageProcessor.jobs import jobs_from_project

    seen = []

    def apply_naming_schema(**kw):
        seen.append((kw["group_name"], kw["image_number"], kw["root_folder"], kw["output_extension"]))
        return f"{kw['output_base_dir']}/{kw['group_name']}_{kw['image_number']:03d}{kw['output_extension']}"

    pkg = types.ModuleType("ImageProcessor")
    mod = types.ModuleType("ImageProcessor.fileNamingSchema")
    mod.apply_naming_schema = apply_naming_schema
    pkg.fileNamingSchema = mod
    monkeypatch.setitem(sys.modules, "ImageProcessor", pkg)
    monkeypatch.setitem(sys.modules, "ImageProcessor.fileNamingSchema", mod)
    p = _project()
    p["processing_settings"].update(use_export_schema=True, export_schema="[g]_[n3][e]")
    jobs, _ = jobs_from_project(p)
    assert [j.output_path for j in jobs] == ["/out/A_001.tiff", "/out/B_001.tiff", "/out/A_002.tiff"]
    assert seen == [("A", 1, "scan_042", ".tiff"), ("B", 1, "scan_042", ".tiff"), ("A", 2, "scan_042", ".tiff")]


def test_run_project_drives_one_worker_per_image(tmp_path, monkeypatch):
    from scanspaceimageprocessor_amd.ImageProcessor import jobs as jobs_mod

    ran = []
    monkeypatch.setattr(ipw.ImageCorrectionWorker, "run", lambda self: ran.append((self.images[0], self.raw_loader)))
    path = tmp_path / "project.json"
    path.write_text(json.dumps(_project(str(tmp_path))))
    loader = lambda p: None
    jobs, failed = jobs_mod.run_project(str(path), raw_loader=loader)
    assert [r[0] for r in ran] == [j.image_path for j in jobs] and all(r[1] is loader for r in ran)
    assert len(failed) == 1
