"""ctypes binding of ``libbasq_hip.so`` (the C ABI declared in ``include/basq_hip.h``).

There is deliberately NO fallback: if the shared library is missing or does not
export a declared symbol, importing the binding raises.  The product path never
computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BASQ_HIP_LIB selects a tuning-variant build of the SAME library (tools/bench_blocksum.py); never a fallback.
LIB_PATH = os.environ.get("BASQ_HIP_LIB") or os.path.join(_HERE, "csrc", "libbasq_hip.so")

# error codes / enums (mirror include/basq_hip.h)
BASQ_OK = 0
FAMILY = {"rbf": 0, "matern52": 1, "matern32": 2}
ROLE_A, ROLE_B = 0, 1
MAX_DIM = 38
ABI_VERSION = 15


class KernelSpecC(C.Structure):
    _fields_ = [("family", C.c_int32), ("d", C.c_int32), ("lengthscale", C.c_double), ("outputscale", C.c_double),
                ("flags", C.c_int32), ("reserved", C.c_int32)]


SPEC_ACCURATE_EXP = 1            # include/basq_hip.h: BASQ_SPEC_ACCURATE_EXP


_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_specp = C.POINTER(KernelSpecC)

# name -> (restype, argtypes); must list EVERY symbol of include/basq_hip.h (tests/test_abi.py checks).
SIGNATURES = {
    "basq_strerror": (C.c_char_p, [C.c_int]),
    "basq_abi_version": (C.c_int, []),
    "basq_kp": (C.c_int, [C.c_int]),
    "basq_shader_clock_mhz": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "basq_col_mean_f64": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp]),
    "basq_pack_points_f64": (C.c_int, [_specp, _vp, _i64, _vp, C.c_int, _vp, _vp]),
    "basq_gram_f64": (C.c_int, [_specp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "basq_kernel_matvec_f64": (C.c_int, [_specp, _vp, _i64, _vp, _i64, _vp, _f64, _vp, _vp]),
    "basq_blocksum_f64": (C.c_int, [_specp, _vp, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp,
                                    _vp]),
    "basq_regroup_classes_f64": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "basq_project_f64": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _f64, _i32, _vp, _vp, _vp]),
    "basq_project_chunks_f64": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _f64, _i32, _vp, _vp, _vp]),
    "basq_sum_parts_f64": (C.c_int, [_vp, _i32, _i64, _vp, _vp]),
    "basq_finalize_f64": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _f64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "basq_nullspace_f64": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "basq_reduction_ws_doubles": (C.c_int64, [_i32, _i32]),
    "basq_car_eliminate_f64": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "basq_reweight_compact_f64": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _i32,
                                            _i64, _vp, _vp, _vp, _vp, _vp]),
    "basq_finalize_geo_f64": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _f64, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "basq_tail_weights_geo_f64": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "basq_round_next_i64": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "basq_regroup_round_next_f64": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "basq_blocksum_geo_f64": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "basq_reweight_compact_geo_f64": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _i64,
                                                _i32, _vp, _vp, _vp, _vp, _vp]),
    "basq_epoch_turn_f64": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "basq_reweight_compact_rounds_f64": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i32,
                                                   _vp, _vp, _vp, _vp, _vp]),
    "basq_init_state_f64": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "basq_dense_blocksum_f64": (C.c_int, [_vp, _i32, _i64, _i64, _vp, _i64, _i64, _i32, _f64, _i32, _vp, _vp, _vp]),
    "basq_blocksum_sq_f64": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp,
                                       _i64, _i32, _f64, _vp, _vp]),
    "basq_cov_diag_f64": (C.c_int, [_vp, _vp, _i32, _vp, _i64, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _f64, _vp, _vp]),
    "basq_blocksum_sq_geo_f64": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i64,
                                           _i32, _f64, _vp, _vp]),
    "basq_cov_diag_geo_f64": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _f64, _vp, _vp]),
    "basq_sq_noise_part_ws_doubles": (C.c_int64, [_i32]),
    "basq_sq_noise_part_geo_f64": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "basq_box_muller_f64": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "basq_chol_inv_f64": (C.c_int, [_vp, _i32, _vp, _vp, _f64, _vp]),
    "basq_chol_factor_f64": (C.c_int, [_vp, _i32, _vp, _f64, _vp]),
    "basq_trsm_rows_f64": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp]),
    "basq_cholqr_f64": (C.c_int, [_vp, _i32, _vp, _f64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "basq_skinny_gemm_f64": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "basq_gemm_f64": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _f64, _vp]),
}

_lib = None


class BasqHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and attach prototypes.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BasqHipError(
            f"{LIB_PATH} not found: build the HIP extension first (python -m basq_amd._build, needs hipcc). "
            "basq_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise BasqHipError(f"{LIB_PATH} does not export {name}")
        fn.restype = res
        fn.argtypes = args
    if lib.basq_abi_version() != ABI_VERSION:
        raise BasqHipError(f"ABI version mismatch: library {lib.basq_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != BASQ_OK:
        msg = load().basq_strerror(rc).decode()
        raise BasqHipError(f"{what} failed: {msg} ({rc})")
