"""eioku_amd - MI355X (gfx950) implementation of the eioku ml-service hot path.

Scene scoring, YOLOv8 detection, all-MiniLM-L6-v2 segment embedding and flat-L2 kNN run as
hand-written HIP kernels in ``libeioku_hip.so`` (C ABI: ``include/eioku_hip.h``); this package is
the thin Python host that keeps the reference's ``ModelManager`` / ``process_ml_task`` interface.
"""

__version__ = "0.1.0"

from . import _lib  # noqa: F401  (dlopen is lazy: importing the package needs no GPU)
