"""ctypes binding of libpose_amd.so (the C ABI declared in include/pose_amd.h).

There is no fallback: if the shared library is missing or a call fails, an exception is
raised.  Nothing here computes on the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POSE_AMD_LIB") or os.path.join(_HERE, "libpose_amd.so")   # override: experiments with other builds


class PoseAmdError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("select_mode", C.c_int), ("eps", C.c_double), ("early_exit", C.c_int),
                ("inlier_divisor", C.c_uint32), ("inlier_absolute", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("found", C.c_int), ("focal", C.c_double), ("C", C.c_double * 3), ("Rc", C.c_double * 9),
                ("inliers", C.c_uint32), ("area_fit", C.c_int64), ("area_all", C.c_int64), ("steps", C.c_uint32),
                ("best_hyp", C.c_uint32), ("best_sol", C.c_uint32), ("poses_scored", C.c_uint64)]


class QueryDesc(C.Structure):
    """pose_amd_query: one query of pose_amd_ransac_batch (host pointers)."""
    _fields_ = [("n", C.c_int), ("kx", C.POINTER(C.c_float)), ("ky", C.POINTER(C.c_float)), ("X", C.POINTER(C.c_float)),
                ("Y", C.POINTER(C.c_float)), ("Z", C.POINTER(C.c_float)), ("lindex", C.POINTER(C.c_uint32)),
                ("cam_off", C.POINTER(C.c_uint32)), ("cam_ids", C.POINTER(C.c_uint32)), ("width", C.c_int), ("height", C.c_int),
                ("focal", C.c_double), ("query_id", C.c_uint32)]


_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_uint32)
_u8 = C.POINTER(C.c_uint8)
_u64 = C.POINTER(C.c_uint64)
_i64 = C.POINTER(C.c_int64)
_ctx = C.c_void_p

# name -> (restype, argtypes); must list every function include/pose_amd.h declares
SIGNATURES = {
    "pose_amd_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "pose_amd_destroy": (None, [_ctx]),
    "pose_amd_last_error": (C.c_char_p, [_ctx]),
    "pose_amd_device_info": (C.c_int, [_ctx, C.c_char_p, C.c_int, _ip, _ip]),
    "pose_amd_synchronize": (C.c_int, [_ctx]),
    "pose_amd_p3p_batch": (C.c_int, [_ctx, C.c_int, _dp, _dp, _dp, _ip]),
    "pose_amd_quartic_batch": (C.c_int, [_ctx, C.c_int, _dp, _dp]),
    "pose_amd_p4pf_batch": (C.c_int, [_ctx, C.c_int, _dp, _dp, _ip, _dp, _dp, _dp]),
    "pose_amd_p4pf_batch_noalloc": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pose_amd_set_query": (C.c_int, [_ctx, C.c_int, _fp, _fp, _fp, _fp, _fp, _up, C.c_int, C.c_int, C.c_double]),
    "pose_amd_set_camsets": (C.c_int, [_ctx, _up, _up]),
    "pose_amd_compute_matrices_batch": (C.c_int, [_ctx, C.c_int, _dp, _dp, _dp, _dp]),
    "pose_amd_score_batch": (C.c_int, [_ctx, C.c_int, _dp, C.c_double, _ip, _u64]),
    "pose_amd_coverage_batch": (C.c_int, [_ctx, C.c_int, _dp, C.c_double, _i64, _i64]),
    "pose_amd_project_batch": (C.c_int, [_ctx, C.c_int, _dp, _dp, _dp]),
    "pose_amd_best_fit": (C.c_int, [_ctx, _u8, _dp, _dp, _ip]),
    "pose_amd_sample": (C.c_int, [_ctx, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _up]),
    "pose_amd_set_samples": (C.c_int, [_ctx, _up, C.c_uint32]),
    "pose_amd_ransac_run": (C.c_int, [_ctx, C.POINTER(Params), C.POINTER(Result), _ip, _ip]),
    "pose_amd_best_mask": (C.c_int, [_ctx, _u8]),
    "pose_amd_ransac_batch": (C.c_int, [_ctx, C.c_int, C.POINTER(QueryDesc), C.c_int, C.c_uint64, C.c_uint32, C.POINTER(Params),
                                        C.POINTER(Result)]),
    "pose_amd_batch_create": (C.c_int, [_ctx, C.c_int, C.POINTER(QueryDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "pose_amd_batch_size": (C.c_int, [C.c_void_p]),
    "pose_amd_batch_run": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.POINTER(Params), C.POINTER(Result)]),
    "pose_amd_batch_destroy": (None, [C.c_void_p]),
    "pose_amd_batch_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "pose_amd_batch_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "pose_amd_partition_queries": (C.c_int, [C.c_int, _ip, C.c_int, _ip]),
    "pose_amd_ransac_batch_multi": (C.c_int, [_ip, C.c_int, C.c_int, C.POINTER(QueryDesc), C.c_int, C.c_uint64, C.c_uint32, C.POINTER(Params),
                                              C.POINTER(Result)]),
    "pose_amd_multi_last_error": (C.c_char_p, []),
    "pose_amd_multi_release": (None, []),
    "pose_amd_ransac_range": (C.c_int, [_ctx, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.POINTER(Result), _u64]),
    "pose_amd_knn_set_model": (C.c_int, [_ctx, C.c_int, _fp]),
    "pose_amd_knn_set_model_views": (C.c_int, [_ctx, C.c_int, _up, _u8, _fp]),
    "pose_amd_knn_match": (C.c_int, [_ctx, C.c_int, _u8, C.c_float, _ip, _fp, _u8, _ip]),
    "pose_amd_knn_last_timing": (C.c_int, [_ctx, C.POINTER(C.c_float)]),
    "pose_amd_last_timing": (C.c_int, [_ctx, C.POINTER(C.c_float)]),
}

_lib = None


def load():
    """Loads libpose_amd.so and declares every prototype.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PoseAmdError(f"{LIB_PATH} not found: build it with `make` (hipcc --offload-arch=gfx950)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
