"""Counterparts of the reference's ``ImageProcessor`` package for the correction chain:
``editingTools`` (pixel operators), ``imageProcessorWorker`` (per-image worker) and the preview /
thumbnail producers of ``imageLoader``."""

from . import editingTools  # noqa: F401
from . import imageLoader  # noqa: F401
