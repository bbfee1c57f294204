"""ctypes binding of libgroomed_nms_hip.so (the C ABI declared in include/groomed_nms_hip.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, importing the ops raises.
torch is imported first so that the library's libamdhip64.so.7 dependency resolves to the HIP runtime
torch already loaded (same soname) and device pointers/streams are shared with torch.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNMS_LIB_PATH") or os.path.join(_HERE, "libgroomed_nms_hip.so")   # (GNMS_LIB_PATH: a developer build, e.g. tools/build_timing.sh)

c_f32p = ctypes.c_void_p
c_vp = ctypes.c_void_p


class GnmsParams(ctypes.Structure):
    """struct gnms_params (include/groomed_nms_hip.h) == keyword arguments of lib/groomed_nms.py:10."""
    _fields_ = [("nms_threshold", ctypes.c_float), ("temperature", ctypes.c_float),
                ("valid_box_prob_threshold", ctypes.c_float), ("pruning_method", ctypes.c_int32),
                ("return_sorted_prob", ctypes.c_int32), ("group_boxes", ctypes.c_int32),
                ("mask_group_boxes", ctypes.c_int32), ("group_size", ctypes.c_int32), ("presorted", ctypes.c_int32)]


class GnmsError(RuntimeError):
    pass


_SIGNATURES = {
    "gnms_abi_version": (ctypes.c_int, []),
    "gnms_last_error": (ctypes.c_char_p, []),
    "gnms_default_params": (None, [ctypes.POINTER(GnmsParams)]),
    "gnms_iou2d": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int64, c_vp]),
    "gnms_corners_of_cuboid": (ctypes.c_int, [c_vp, ctypes.c_int64, c_vp, c_vp]),
    "gnms_iou2d_f64": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int64, c_vp]),
    "gnms_corners_of_cuboid_f64": (ctypes.c_int, [c_vp, ctypes.c_int64, ctypes.c_int, c_vp, c_vp]),
    "gnms_iou3d_approximate": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp,
                                              ctypes.c_int64, c_vp]),
    "gnms_iou3d_from_params": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp,
                                              ctypes.c_int64, c_vp]),
    "gnms_nms_overlap3d_from_params": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, c_vp, ctypes.c_int64, c_vp]),
    "gnms_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(GnmsParams)]),
    "gnms_forward": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_vp, ctypes.POINTER(GnmsParams),
                                    c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "gnms_forward_with_iou2d": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_vp, ctypes.POINTER(GnmsParams),
                                               c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "gnms_forward_with_iou3d": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_vp, ctypes.POINTER(GnmsParams),
                                               c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "gnms_backward": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_vp,
                                     ctypes.POINTER(GnmsParams), c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "gnms_counts_to_host": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_vp, c_vp]),
    "gnms_host_counts_slot": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp)]),
    "gnms_host_counts_wait": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_vp]),
    "gnms_host_counts_release": (ctypes.c_int, [c_vp, c_vp]),
    "gnms_test_mailbox_slots": (ctypes.c_int, [ctypes.c_int]),
    "gnms_forward_from_boxes": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.POINTER(GnmsParams), c_vp, c_vp, c_vp,
                                               c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "gnms_backward_from_boxes": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.POINTER(GnmsParams), c_vp,
                                                c_vp, ctypes.c_size_t, c_vp]),
    "gnms_profile_bitmask": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_vp, ctypes.c_float, c_vp,
                                            ctypes.c_size_t, c_vp]),
    "gnms_profile_bitmask_boxes": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_float, c_vp, ctypes.c_size_t, c_vp]),
    "gnms_profile_events": (ctypes.c_int, [ctypes.c_int]),
    "gnms_profile_collect": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "gnms_profile_write_kernel_name": (ctypes.c_char_p, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "gnms_profile_fill": (ctypes.c_int, [c_vp, ctypes.c_size_t, c_vp]),
    "gnms_profile_fill_tiles": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, c_vp]),
    "gnms_profile_fill_sym": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp]),
    "gnms_profile_read": (ctypes.c_int, [c_vp, ctypes.c_size_t, c_vp, c_vp]),
    "gnms_get_groups": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_int, c_vp, c_vp, c_vp,
                                       c_vp, ctypes.c_size_t, c_vp]),
    "gnms_pruning_function": (ctypes.c_int, [c_vp, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_int, c_vp, c_vp]),
    "gnms_pruning_function_backward": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_int, c_vp, c_vp]),
    "gnms_soft_sort": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int64, ctypes.c_float, c_vp, c_vp, c_vp, c_vp,
                                      ctypes.c_size_t, c_vp]),
    "gnms_soft_sort_backward_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "gnms_soft_sort_backward": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_vp,
                                               c_vp, c_vp, ctypes.c_size_t, c_vp, ctypes.c_size_t, c_vp]),
    "gnms_sgemm": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                  ctypes.c_int64, c_vp]),
    "gnms_profile_sgemm": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_int, c_vp]),
    "_nms": (None, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int]),
    "gnms_nms_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "gnms_nms_sorted": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "gnms_nms_sorted_shift": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, c_vp, c_vp, c_vp,
                                             ctypes.c_size_t, c_vp]),
    "gnms_nms_sorted_shift_f64": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, c_vp, c_vp, c_vp,
                                                 ctypes.c_size_t, c_vp]),
    "gnms_soft_nms_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "gnms_soft_nms": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                     ctypes.c_double, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "gnms_bbox_transform_inv": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, c_vp]),
    "gnms_select_topk": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int, c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp,
                                         c_vp]),
    "gnms_project_boxes3d": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp]),
    "gnms_best_targets": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp, ctypes.c_float, c_vp, c_vp,
                                          c_vp, c_vp]),
    "gnms_aploss": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_float, ctypes.c_float, c_vp, c_vp, c_vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """Loads the shared library (once).  Raises GnmsError with a build hint when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GnmsError("%s not found: build it with `python -m groomed_nms_amd.build` (hipcc, gfx950). "
                        "There is no CPU fallback for the GrooMeD-NMS ops." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise GnmsError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.gnms_abi_version() != 1:
        raise GnmsError("ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gnms_last_error()
        msg = msg.decode() if msg else ""
        if rc == -2 and "not implemented" in msg:
            raise NotImplementedError("Pruning method not implemented!")   # lib/groomed_nms.py:178
        raise GnmsError("%s failed (%d): %s" % (what, rc, msg))


def ptr(t):
    """Device address of a tensor as a plain int (None -> NULL): every entry declares c_void_p argtypes, which take ints as they are;
    wrapping each in a c_void_p object cost ~0.3 us a piece, a dozen per call on a path that is host-bound below N = 4096."""
    return t.data_ptr() if t is not None else None


class _Noop:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOOP = _Noop()


def on_device(dev):
    """Context that makes `dev` the current HIP device for the library call -- a no-op object when it already is
    (torch.cuda.device() costs ~4 us per entry, twice per training step on the launch-bound path)."""
    idx = dev.index
    if idx is None or idx == torch.cuda.current_device():
        return _NOOP
    return torch.cuda.device(dev)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` (default: the current device) as an int (0 = the null stream)."""
    if _raw_stream is not None:           # ~0.3 us; torch.cuda.current_stream() builds a Stream object (~10 us)
        if device is None:
            idx = torch.cuda.current_device()
        elif isinstance(device, int):
            idx = device
        else:
            idx = torch.device(device).index
            if idx is None:
                idx = torch.cuda.current_device()
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream
