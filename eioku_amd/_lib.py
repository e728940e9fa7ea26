"""ctypes binding of ``libeioku_hip.so`` (the C ABI declared in ``include/eioku_hip.h``).

The product path has no CPU fallback: if the shared library is missing, or a GPU call is made
without a gfx950 device, this module raises - it never routes around the HIP kernels.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

_PKG_DIR = Path(__file__).resolve().parent
# EIOKU_HIP_LIB: another build of the same ABI (tests load libeioku_hip_bc.so, the bounds-check build, in a child process)
LIB_PATH = Path(os.environ["EIOKU_HIP_LIB"]) if os.environ.get("EIOKU_HIP_LIB") else _PKG_DIR / "libeioku_hip.so"

MEM_HOST = 0
MEM_DEVICE = 1


class EiokuHipError(RuntimeError):
    """A libeioku_hip call returned a non-zero code (message from ``eioku_last_error``)."""


_c_u8p = C.c_void_p  # all data pointers travel as integers (host ndarray / torch data_ptr)

# name -> (restype, argtypes).  Kept in one table so tests can compare it with the header.
SIGNATURES = {
    "eioku_abi_version": (C.c_int, []),
    "eioku_init": (C.c_int, [C.c_int]),
    "eioku_shutdown": (None, []),
    "eioku_last_error": (C.c_char_p, []),
    "eioku_device_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "eioku_prof_enable": (C.c_int, [C.c_int]),
    "eioku_prof_reset": (C.c_int, []),
    "eioku_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "eioku_synth_u64": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "eioku_synth_bytes": (C.c_int, [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "eioku_synth_normal_f32": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "eioku_synth_frames_bgr": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "eioku_scene_sad_luma": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "eioku_scene_hsv_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p]),
    "eioku_scene_sad_luma_bgr": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p]),
    "eioku_scene_scores_from_sad": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "eioku_scene_scores_luma": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_double,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "eioku_scene_content_scores": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]),
    "eioku_scene_content_cuts": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.POINTER(C.c_int)]),
    "eioku_scene_content": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p,
                                      C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_void_p]),
    "eioku_yuv420_to_bgr": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "eioku_bgr2hsv": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    "eioku_conv2d_f16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p]),
    "eioku_debug_bounds": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int]),
    "eioku_yolo_create": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "eioku_yolo_destroy": (None, [C.c_void_p]),
    "eioku_yolo_num_convs": (C.c_int, [C.c_void_p]),
    "eioku_yolo_conv_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "eioku_yolo_set_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "eioku_yolo_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_void_p), C.c_void_p]),
    "eioku_yolo_last_conv_flops": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "eioku_yolo_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "eioku_letterbox_f16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eioku_yolo_postprocess": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int),
                                         C.POINTER(C.c_int), C.c_int, C.c_float, C.c_float, C.c_int, C.c_float,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eioku_index_flat_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "eioku_index_destroy": (None, [C.c_void_p]),
    "eioku_index_ntotal": (C.c_longlong, [C.c_void_p]),
    "eioku_index_reset": (C.c_int, [C.c_void_p]),
    "eioku_index_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    "eioku_index_attach": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "eioku_index_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p]),
    "eioku_index_search_after": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_void_p]),
    "eioku_index_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_longlong]),
    "eioku_topk_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "eioku_comm_unique_id": (C.c_int, [C.c_void_p]),
    "eioku_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "eioku_comm_destroy": (None, [C.c_void_p]),
    "eioku_comm_rank": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "eioku_index_search_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "eioku_bert_create": (C.c_int, [C.c_int] * 7 + [C.c_float, C.POINTER(C.c_void_p)]),
    "eioku_bert_destroy": (None, [C.c_void_p]),
    "eioku_bert_num_tensors": (C.c_int, [C.c_void_p]),
    "eioku_bert_tensor_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int),
                                         C.POINTER(C.c_int)]),
    "eioku_bert_set_tensor": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "eioku_bert_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p]),
    "eioku_bert_last_flops": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "eioku_topk_merge_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "eioku_kmeans_update": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "eioku_kmeans_accumulate": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "eioku_kmeans_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "eioku_pq_assign": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "eioku_ivf_histogram": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    "eioku_ivf_scatter": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "eioku_ivfpq_scan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "eioku_ivfpq_tables": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                     C.c_void_p]),
    "eioku_ivfpq_scan_tables": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "eioku_resnet18_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "eioku_resnet18_destroy": (None, [C.c_void_p]),
    "eioku_resnet18_num_convs": (C.c_int, [C.c_void_p]),
    "eioku_resnet18_conv_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                           C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "eioku_resnet18_set_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "eioku_resnet18_set_fc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "eioku_places_preprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "eioku_resnet18_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "eioku_resnet18_classify": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p]),
    "eioku_resnet18_last_flops": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "eioku_ivfpq_lists_aux": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eioku_ivfpq_lists_workspace": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int]),
    "eioku_ivfpq_search_lists": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_longlong,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()
_initialised_device: int | None = None


def load() -> C.CDLL:
    """dlopen the library and attach prototypes (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise EiokuHipError(
                f"{LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C eioku_amd/csrc`; there is no CPU fallback for the hot path")
        # One HIP runtime per process: torch bundles its own libamdhip64 (SONAME libamdhip64.so.7,
        # the soname this library links against).  Loading torch FIRST makes the dynamic linker
        # resolve our dependency to that already-loaded copy; the other order leaves two runtimes
        # in the process and the second one sees no device.  torch owns device memory and streams
        # for the Python host anyway, so it is always present here.
        import torch  # noqa: F401

        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.eioku_abi_version() != 1:
            raise EiokuHipError(f"ABI version mismatch: library reports {lib.eioku_abi_version()}")
        _lib = lib
    return _lib


def last_error() -> str:
    return (load().eioku_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise EiokuHipError(f"{what} failed (code {rc}): {last_error()}")


def init(device_id: int | None = None) -> int:
    """Select the worker's HIP device (``EIOKU_HIP_DEVICE`` / ``LOCAL_RANK`` / 0)."""
    global _initialised_device
    lib = load()
    if device_id is None:
        device_id = int(os.environ.get("EIOKU_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if _initialised_device == device_id:
        return device_id
    check(lib.eioku_init(device_id), f"eioku_init({device_id})")
    _initialised_device = device_id
    return device_id


def device_info() -> dict:
    lib = load()
    init()
    name = C.create_string_buffer(256)
    cus = C.c_int(0)
    hbm = C.c_uint64(0)
    check(lib.eioku_device_info(name, 256, C.byref(cus), C.byref(hbm)), "eioku_device_info")
    return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}


PROF_SCENE_SAD, PROF_SCENE_HSV, PROF_CONV, PROF_KNN, PROF_GEMM, PROF_IVFPQ = 0, 1, 2, 3, 4, 5


def prof_enable(on: bool, tags=None) -> None:
    """hipEvent brackets on/off; ``tags`` (iterable of PROF_* ids) restricts them to those kernels."""
    flag = int(bool(on))
    if on and tags is not None:
        flag = 0
        for t in tags:
            flag |= 1 << (int(t) + 1)
    check(load().eioku_prof_enable(flag), "eioku_prof_enable")


def prof_reset() -> None:
    check(load().eioku_prof_reset(), "eioku_prof_reset")


def prof_read(tag: int) -> tuple[float, int]:
    """(summed kernel milliseconds, launches) for one tagged kernel since the last reset."""
    ms = C.c_double(0)
    cnt = C.c_uint64(0)
    check(load().eioku_prof_read(tag, C.byref(ms), C.byref(cnt)), "eioku_prof_read")
    return ms.value, cnt.value
