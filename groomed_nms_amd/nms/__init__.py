"""Classical hard NMS -- mirror of the reference's lib/nms package."""
from .gpu_nms import gpu_nms  # noqa: F401
from .cpu_nms import cpu_nms  # noqa: F401
from .py_cpu_nms import py_cpu_nms  # noqa: F401
