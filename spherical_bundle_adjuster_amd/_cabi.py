"""ctypes binding of the C-ABI in ``include/sba_hip.h`` (``libsba_hip.so``).

This is plumbing only: every call lands in the HIP library.  There is no Python or CPU
implementation of the hot path in this package; if the shared library is missing the import
fails loudly (``LibraryNotBuilt``), and without a HIP device every compute call raises ``SbaError``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libsba_hip.so"

# status codes / enums (mirror include/sba_hip.h)
SBA_OK = 0
SBA_ERR_INVALID_ARG, SBA_ERR_NO_DEVICE, SBA_ERR_HIP, SBA_ERR_NOT_UPLOADED = -1, -2, -3, -4
SBA_ERR_COMM, SBA_ERR_NUMERIC, SBA_ERR_UNSUPPORTED = -5, -6, -7
MODE_ROT, MODE_TRAN, MODE_RT = 0, 1, 2
DEPTH_UNIFORM, DEPTH_PER_MATCH = 0, 1
STORE_F64, STORE_F32 = 0, 1
TRAN_FREE, TRAN_SPHERE = 0, 1
KERNEL_FACTORED, KERNEL_EXPLICIT = 0, 1
PACK_SIZE = 24
ABI_VERSION = 2          # SBA_ABI_VERSION of include/sba_hip.h
COMM_ID_BYTES = 128
PEER_HANDLE_BYTES = 64
TERMINATION = {1: "CONVERGENCE_FUNCTION", 2: "CONVERGENCE_GRADIENT", 3: "CONVERGENCE_PARAMETER",
               4: "NO_CONVERGENCE", 5: "MIN_RADIUS", 6: "FAILURE"}


class LibraryNotBuilt(ImportError):
    pass


class SbaError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"sba error {code}: {message}")
        self.code = code
        self.message = message


class NormalEq(C.Structure):
    _fields_ = [("H", C.c_double * 36), ("g", C.c_double * 6), ("cost", C.c_double),
                ("sum_w", C.c_double), ("n_outlier", C.c_double)]


class LmOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int),
                ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double),
                ("jacobi_scaling", C.c_int),
                ("huber_delta", C.c_double),
                ("tran_param", C.c_int),
                ("verbose", C.c_int),
                ("max_num_line_search_step_size_iterations", C.c_int),
                ("line_search_sufficient_function_decrease", C.c_double),
                ("max_line_search_step_contraction", C.c_double),
                ("min_line_search_step_contraction", C.c_double),
                ("min_line_search_step_size", C.c_double)]


class LmSummary(C.Structure):
    _fields_ = [("termination", C.c_int), ("num_iterations", C.c_int),
                ("num_successful_steps", C.c_int), ("num_evaluations", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_gradient_max_norm", C.c_double), ("final_radius", C.c_double),
                ("seconds_total", C.c_double), ("seconds_eval", C.c_double),
                ("num_line_search_steps", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); must list every function include/sba_hip.h declares
SIGNATURES = {
    "sba_abi_version": (C.c_int, []),
    "sba_last_error": (C.c_char_p, []),
    "sba_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sba_lm_options_default": (None, [C.POINTER(LmOptions)]),
    "sba_problem_create": (C.c_int, [C.POINTER(_vp), C.c_int, _vp]),
    "sba_problem_destroy": (C.c_int, [_vp]),
    "sba_problem_upload": (C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t, C.c_int]),
    "sba_problem_upload_keypoints": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _vp, C.c_int]),
    "sba_problem_upload_device": (C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t, C.c_int]),
    "sba_problem_size": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "sba_problem_set_kernel": (C.c_int, [_vp, C.c_int]),
    "sba_problem_set_depths": (C.c_int, [_vp, _vp]),
    "sba_problem_eval": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double,
                                   C.c_double, C.POINTER(NormalEq)]),
    "sba_problem_eval_pack": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double,
                                        C.c_double, _dp]),
    "sba_problem_eval_timed": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double,
                                         C.c_double, C.c_int, _dp, _dp, _dp]),
    "sba_problem_eval_launch_times": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double,
                                                C.c_double, C.c_int, C.POINTER(C.c_float)]),
    "sba_problem_eval_steps": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double,
                                         C.c_double, C.c_int, _dp, _dp]),
    "sba_expand_pack": (C.c_int, [C.c_int, _dp, C.POINTER(NormalEq)]),
    "sba_problem_solve": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double,
                                    C.POINTER(LmOptions), C.POINTER(LmSummary)]),
    "sba_problem_solve_depths": (C.c_int, [_vp, _dp, _dp, C.c_double, C.c_double, C.POINTER(LmOptions), _vp,
                                           C.POINTER(LmSummary)]),
    "sba_problem_epipolar_moments": (C.c_int, [_vp, _dp]),
    "sba_initial_guess_from_moments": (C.c_int, [_dp, C.c_int, C.c_double, C.c_ulonglong, _dp, _dp,
                                                 C.POINTER(C.c_int)]),
    "sba_problem_initial_guess": (C.c_int, [_vp, C.c_int, C.c_double, C.c_ulonglong, _dp, _dp, C.POINTER(C.c_int)]),
    "sba_reference_rand_seed": (C.c_int, [C.c_uint]),
    "sba_reference_rand_next": (C.c_int, []),
    "sba_reference_trial_subsets": (C.c_int, [C.c_int, C.c_int, C.c_double, _vp, C.POINTER(C.c_int)]),
    "sba_problem_epipolar_subset_moments": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _dp]),
    "sba_problem_initial_guess_reference": (C.c_int, [_vp, C.c_int, C.c_double, _dp, _dp, C.POINTER(C.c_int)]),
    "sba_comm_unique_id": (C.c_int, [C.c_char_p]),
    "sba_problem_comm_init_rank": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p]),
    "sba_rccl_available": (C.c_int, []),
    "sba_problem_comm_destroy": (C.c_int, [_vp]),
    "sba_problem_peer_export": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p]),
    "sba_problem_peer_connect": (C.c_int, [_vp, C.c_char_p]),
    "sba_problem_peer_selftest": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "sba_problem_peer_disable": (C.c_int, [_vp]),
    "sba_set_host_threads": (C.c_int, [C.c_int]),
    "sba_problem_set_shard": (C.c_int, [_vp, C.c_int, C.c_int]),
    "sba_problem_set_allreduce": (C.c_int, [_vp, ALLREDUCE_FN, _vp]),
    "sba_problem_pack_device_ptr": (C.c_int, [_vp, C.POINTER(_vp)]),
    "sba_batch_create": (C.c_int, [C.POINTER(_vp), C.c_int, _vp]),
    "sba_batch_destroy": (C.c_int, [_vp]),
    "sba_batch_set_kernel": (C.c_int, [_vp, C.c_int]),
    "sba_batch_upload": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(C.c_size_t), C.c_int, C.c_int]),
    "sba_batch_size": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sba_batch_eval": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, _dp]),
    "sba_batch_eval_timed": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_int, _dp,
                                       _dp, _dp, _dp, _dp]),
    "sba_batch_sweep_launch_times": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_int,
                                               C.POINTER(C.c_float)]),
    "sba_batch_step_launch_times": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_int,
                                              C.POINTER(C.c_float)]),
    "sba_batch_step_is_fused": (C.c_int, [_vp]),
    "sba_batch_solve": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.POINTER(LmOptions),
                                  C.POINTER(LmSummary), C.POINTER(C.c_int)]),
    "sba_batch_solve_depths": (C.c_int, [_vp, _dp, _dp, C.c_double, C.c_double, C.POINTER(LmOptions), _vp,
                                         C.POINTER(LmSummary), C.POINTER(C.c_int)]),
    "sba_batch_solve_problem": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_ulonglong, _dp, _dp, C.POINTER(LmOptions), _vp, _dp,
                                          C.POINTER(C.c_int), C.POINTER(LmSummary), C.POINTER(LmSummary), C.POINTER(LmSummary),
                                          C.POINTER(C.c_int)]),
    "sba_batch_set_depths": (C.c_int, [_vp, _dp]),
    "sba_batch_epipolar_moments": (C.c_int, [_vp, _dp]),
    "sba_batch_initial_guess": (C.c_int, [_vp, C.c_int, C.c_double, C.c_ulonglong, _dp, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sba_keypoints_to_sphere": (C.c_int, [C.c_int, _vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _vp]),
    "sba_rotate_keypoints": (C.c_int, [C.c_int, _vp, C.c_size_t, C.c_size_t, C.c_float, C.c_int, C.c_int]),
    "sba_cube2equi_keypoints": (C.c_int, [C.c_int, _vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "sba_crop_rotated_image": (C.c_int, [C.c_int, _vp, C.c_int, C.c_int, C.c_float, _vp]),
    "sba_rotate_keypoints_device": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t, C.c_size_t, C.c_float, C.c_int, C.c_int]),
    "sba_cube2equi_keypoints_device": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "sba_crop_rotated_image_device": (C.c_int, [C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_int, _vp]),
    "sba_map_table_host_decided": (C.c_long, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sba_map_table_tiles": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]),
    "sba_equi2cube": (C.c_int, [C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "sba_equi2cube_device": (C.c_int, [C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
}

_lib = None


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """Load libsba_hip.so (once).  Raises LibraryNotBuilt when it has not been compiled."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # SBA_LIBRARY_PATH: a differently built libsba_hip.so (tools' A/B measurements of build variants only)
    p = Path(path) if path else Path(os.environ.get("SBA_LIBRARY_PATH", LIB_PATH))
    if not p.exists():
        raise LibraryNotBuilt(
            f"{p} not found: build it with `make -C {PKG_DIR / 'csrc'}` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no Python/CPU fallback for the HIP hot path.")
    # One HIP runtime per process: torch bundles libamdhip64.so.7; if torch is going to be used in
    # this process, let it load its copy first so this library binds to the same one.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    if lib.sba_abi_version() != ABI_VERSION:
        raise LibraryNotBuilt(f"{p}: ABI version {lib.sba_abi_version()} != {ABI_VERSION}, rebuild")
    if path is None:
        _lib = lib
    return lib


def last_error(lib: C.CDLL) -> str:
    msg = lib.sba_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(lib: C.CDLL, rc: int) -> None:
    if rc != SBA_OK:
        msg = lib.sba_last_error()
        raise SbaError(rc, msg.decode("utf-8", "replace") if msg else "")
