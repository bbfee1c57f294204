"""groomed_nms_amd -- MI355X-native (gfx950) GrooMeD-NMS layer.

  groomed_nms_amd.groomed_nms   mirror of the reference's lib/groomed_nms.py (differentiable_nms, get_groups, ...)
  groomed_nms_amd.overlaps      mirror of the overlap helpers in lib/core.py / lib/math_3d.py
  groomed_nms_amd.nms           mirror of lib/nms (gpu_nms over the C symbol `_nms`, cpu_nms, py_cpu_nms)
  groomed_nms_amd.nms_others    mirror of lib/nms_others.py
  groomed_nms_amd.build         hipcc build of libgroomed_nms_hip.so (C ABI: include/groomed_nms_hip.h)
"""
from .groomed_nms import (differentiable_nms, differentiable_nms_batched, differentiable_nms_from_boxes_batched, differentiable_nms_with_iou2d_batched, differentiable_nms_with_iou3d_batched, soft_sort, pruning_function, sigmoid_numpy,  # noqa: F401
                          cast_to_cpu_cuda_tensor, get_groups, indices_copy, GroomedNMS)

__version__ = "0.1.0"
