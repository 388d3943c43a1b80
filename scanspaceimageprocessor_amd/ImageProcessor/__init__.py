"""Counterparts of the reference's ``ImageProcessor`` package for the correction chain:
``editingTools`` (pixel operators) and ``imageProcessorWorker`` (per-image worker)."""

from . import editingTools  # noqa: F401
