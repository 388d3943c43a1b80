"""MI355X-native correction chain of "Image Space" (hglmz/scanSpaceImageProcessor).

The Python entry points of the reference's ``ImageProcessor/editingTools.py`` and
``ImageProcessor/imageProcessorWorker.py`` keep their names and signatures
(``scanspaceimageprocessor_amd.ImageProcessor``) and call hand-written gfx950
HIP kernels through the C ABI of ``libscanspace_hip.so`` (``include/scanspace.h``).
"""

from . import _lib  # noqa: F401
from ._lib import Context, ScanSpaceError, default_context, load_library  # noqa: F401

__all__ = ["Context", "ScanSpaceError", "default_context", "load_library"]
