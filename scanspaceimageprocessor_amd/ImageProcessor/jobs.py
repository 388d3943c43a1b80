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
