"""The job record of the reference's network mode and its mapping onto a worker, without any networking.

``ProcessingJob`` (networkProcessor.py:507-526) is what the server serialises as JSON for one image: paths, the
chart's 24 x 3 swatches (or none) and a settings dictionary.  ``worker_from_job`` is the client's translation of
that record into an ``ImageCorrectionWorker`` (networkProcessor.py:2737-2806), defaults included, so a headless
runner can feed saved jobs to the GPU path.
"""

from __future__ import annotations

import json
import os
import time
from dataclasses import asdict, dataclass, field
from typing import Any, Dict, List, Optional

from .imageProcessorWorker import ImageCorrectionWorker, WorkerSignals


@dataclass
class ProcessingJob:
    job_id: str
    image_path: str
    output_path: str
    swatches: Optional[List[List[float]]]   # 24 x 3 chart swatches, None for the no-chart mode
    settings: Dict[str, Any] = field(default_factory=dict)
    created_time: float = field(default_factory=time.time)
    start_time: Optional[float] = None
    timeout: float = 300.0

    def to_dict(self) -> Dict:
        return asdict(self)

    @classmethod
    def from_dict(cls, data: Dict) -> "ProcessingJob":
        return cls(**data)

    def to_json(self) -> str:
        return json.dumps(self.to_dict())

    @classmethod
    def from_json(cls, text: str) -> "ProcessingJob":
        return cls.from_dict(json.loads(text))


def worker_from_job(job: ProcessingJob, signals: Optional[WorkerSignals] = None) -> ImageCorrectionWorker:
    """networkProcessor.py:2752-2806: every setting with the reference's default; white balance only counts when
    ``enable_white_balance`` is set; chart mode iff the job carries swatches."""
    st = job.settings or {}
    white_balance_adj = st.get("white_balance_adj", 5500)
    if not st.get("enable_white_balance", False):
        white_balance_adj = 5500
    worker = ImageCorrectionWorker(
        images=[job.image_path],
        swatches=job.swatches,
        output_folder=os.path.dirname(job.output_path),
        signals=signals,
        jpeg_quality=st.get("jpeg_quality", 100),
        output_format=st.get("output_format", ".jpg"),
        tiff_bitdepth=st.get("tiff_bitdepth", 8),
        exr_colorspace=st.get("exr_colorspace", "sRGB"),
        from_network=True,
        network_output_path=job.output_path,
        group_name=st.get("group_name", ""),
        use_chart=job.swatches is not None,
        exposure_adj=st.get("exposure_adj", 0.0),
        shadow_adj=st.get("shadow_adj", 0.0),
        highlight_adj=st.get("highlight_adj", 0.0),
        white_balance_adj=white_balance_adj,
        denoise_strength=st.get("denoise_strength", 0.0),
        sharpen_amount=st.get("sharpen_amount", 0.0),
        export_schema=st.get("export_schema", ""),
        use_export_schema=st.get("use_export_schema", False),
        custom_name=st.get("custom_name", ""),
        root_folder=st.get("root_folder", ""),
    )
    worker.use_original_filenames = getattr(job, "use_original_filenames", False)
    worker.image_metadata_map = getattr(job, "image_metadata_map", {})
    return worker


def output_path_from_settings(image_data: dict, processing_settings: dict) -> str:
    """Where the server writes an image of a project (networkProcessor.py:1337-1383): the host application's naming
    schema when the project asks for one and its module is importable, else ``<output_directory>/<stem><format>``."""
    input_path = image_data["full_path"]
    output_dir = processing_settings.get("output_directory", os.path.dirname(input_path))
    output_format = processing_settings.get("export_format", ".jpg")
    if processing_settings.get("use_export_schema", False) and processing_settings.get("export_schema", ""):
        try:
            from ImageProcessor.fileNamingSchema import apply_naming_schema  # the host application's module

            return apply_naming_schema(
                schema=processing_settings["export_schema"], input_path=input_path, output_base_dir=output_dir,
                custom_name=processing_settings.get("custom_name", ""),
                image_number=(image_data.get("metadata") or {}).get("sequence_number", 1),
                output_extension=output_format, root_folder=processing_settings.get("root_folder", ""),
                group_name=image_data.get("group", ""))
        except Exception as e:  # the reference falls through to the default name as well
            print(f"[Server] Schema Error: Failed to apply naming schema for {input_path}: {e}")
    stem = os.path.splitext(os.path.basename(input_path))[0]
    return os.path.join(output_dir, f"{stem}{output_format}")


def jobs_from_project(project_data: dict, clock=time.time):
    """The jobs of a saved project (the JSON the main window exports, image_space.py:1844-2100), as the processing
    server makes them (networkProcessor.py:1166-1320): every image with a path, in list order; the 24 chart swatches
    of the image's group; one sequence number per group for the naming schema; the project's processing settings as
    the job's settings.  Returns ``(jobs, failed)``; ``failed`` lists the images whose group has no usable
    calibration (``{'image_path', 'group', 'error'}``), which the server reports and skips."""
    images = project_data.get("images", [])
    image_groups = project_data.get("image_groups", {})
    settings = project_data.get("processing_settings", {})
    jobs, failed, sequence = [], [], {}
    for image in images:
        if (image.get("metadata") or {}).get("is_group_header", False) or not image.get("full_path"):
            continue
        group = image.get("group", "All Images")
        swatches = (image_groups.get(group) or {}).get("chart_swatches", [])
        if not swatches or len(swatches) != 24:
            failed.append({"image_path": image.get("full_path", "unknown"), "group": group,
                           "error": f"Group '{group}' has invalid chart swatches (length: {len(swatches)})"})
            continue
        sequence[group] = sequence.get(group, 0) + 1
        numbered = dict(image)
        numbered["metadata"] = dict(image.get("metadata") or {}, sequence_number=sequence[group])
        now = clock()
        jobs.append(ProcessingJob(job_id=f"proj_{int(now * 1000)}_{len(jobs)}", image_path=image["full_path"],
                                  output_path=output_path_from_settings(numbered, settings), swatches=swatches,
                                  settings=settings, created_time=now))
    return jobs, failed


def run_project(project, signals: Optional[WorkerSignals] = None, **worker_hooks):
    """Process a saved project headless, image by image, on this process's GPU: ``project`` is the exported
    dictionary or the path of its JSON file; ``worker_hooks`` (``raw_loader``, ``image_writer``) are set on every
    worker.  Returns ``(jobs, failed)`` as ``jobs_from_project`` does; per-image errors are reported through
    ``signals.status`` like in every other mode."""
    if not isinstance(project, dict):
        with open(project, "r", encoding="utf-8") as f:
            project = json.load(f)
    jobs, failed = jobs_from_project(project)
    for job in jobs:
        worker = worker_from_job(job, signals)
        for name, hook in worker_hooks.items():
            setattr(worker, name, hook)
        worker.run()
    return jobs, failed
