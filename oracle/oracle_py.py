"""ORACLE -- TEST INFRASTRUCTURE ONLY (parity unpinned, see sba_oracle.cpp header).

ctypes binding of oracle/libsba_oracle.so.  Importable only from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; the product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "libsba_oracle.so"

MODE_ROT, MODE_TRAN, MODE_RT = 0, 1, 2


class LmOptions(C.Structure):   # same field order as sba_lm_options
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
    _fields_ = [("termination", C.c_int), ("num_iterations", C.c_int), ("num_successful_steps", C.c_int),
                ("num_evaluations", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_gradient_max_norm", C.c_double), ("final_radius", C.c_double),
                ("num_line_search_steps", C.c_int)]


def default_options(**kw) -> LmOptions:
    o = LmOptions(50, 1e4, 1e16, 1e-32, 1e-3, 1e-6, 1e32, 1e-6, 1e-10, 1e-8, 1, 1.0, 0, 0, 20, 1e-4, 1e-3, 0.6, 1e-9)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


@dataclass
class Eval:
    H: np.ndarray
    g: np.ndarray
    cost: float
    sum_w: float
    n_outlier: float


_lib = None


def build() -> None:
    subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        _lib = C.CDLL(str(LIB))
        _lib.orc_num_procs.restype = C.c_int
        _lib.orc_equi2cube.restype = C.c_long
        _lib.orc_lm_solve.restype = C.c_int
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def num_procs() -> int:
    return lib().orc_num_procs()


def point(mode, cam1, cam2, rot, tran, d1, d2):
    """Residual e (3,) and Jacobian J (3,6) over [rot|tran] of ONE residual block (dual numbers)."""
    e, J = np.zeros(3), np.zeros((3, 6))
    lib().orc_point(C.c_int(mode), _p(_f64(cam1)), _p(_f64(cam2)), _p(_f64(rot)), _p(_f64(tran)),
                    C.c_double(d1), C.c_double(d2), _p(e), _p(J))
    return e, J


def huber(a: float, s: float) -> np.ndarray:
    rho = np.zeros(3)
    lib().orc_huber(C.c_double(a), C.c_double(s), _p(rho))
    return rho


def rotate(w, p) -> np.ndarray:
    out = np.zeros(3)
    lib().orc_rotate(_p(_f64(w)), _p(_f64(p)), _p(out))
    return out


def _eval(fn, mode, x1, x2, d12, rot, tran, d1, d2, delta, threads) -> Eval:
    x1, x2 = _f64(x1).reshape(-1, 3), _f64(x2).reshape(-1, 3)
    n = x1.shape[0]
    per_match = d12 is not None
    dd = _f64(d12).reshape(-1, 2) if per_match else np.zeros((1, 2))
    out = np.zeros(45)
    fn(C.c_int(mode), C.c_int(1 if per_match else 0), _p(x1), _p(x2), _p(dd), C.c_size_t(n), _p(_f64(rot)),
       _p(_f64(tran)), C.c_double(d1), C.c_double(d2), C.c_double(delta), C.c_int(threads), _p(out))
    return Eval(out[:36].reshape(6, 6).copy(), out[36:42].copy(), float(out[42]), float(out[43]), float(out[44]))


def evaluate(mode, x1, x2, rot, tran, d1=1.0, d2=1.0, delta=1.0, d12=None, threads=0) -> Eval:
    """Faithful loop: per-match functor through dual numbers + per-match trig, Huber corrector,
    long-double accumulation (reference .cpp:843-1002 as Ceres would evaluate it)."""
    return _eval(lib().orc_eval, mode, x1, x2, d12, rot, tran, d1, d2, delta, threads)


def evaluate_f64(mode, x1, x2, rot, tran, d1=1.0, d2=1.0, delta=1.0, d12=None, threads=0) -> Eval:
    """The faithful loop with plain-double accumulation (what is TIMED as the CPU baseline)."""
    return _eval(lib().orc_eval_f64, mode, x1, x2, d12, rot, tran, d1, d2, delta, threads)


def evaluate_hoisted(mode, x1, x2, rot, tran, d1=1.0, d2=1.0, delta=1.0, d12=None, threads=0) -> Eval:
    """Optimised-CPU loop: rotation matrices hoisted, analytic per-match arithmetic in double."""
    return _eval(lib().orc_eval_hoisted, mode, x1, x2, d12, rot, tran, d1, d2, delta, threads)


def lm_solve(mode, x1, x2, rot, tran, d1=1.0, d2=1.0, d12=None, options: LmOptions | None = None,
             threads=0, faithful=True):
    x1, x2 = _f64(x1).reshape(-1, 3), _f64(x2).reshape(-1, 3)
    n = x1.shape[0]
    per_match = d12 is not None
    dd = _f64(d12).reshape(-1, 2) if per_match else np.zeros((1, 2))
    rot, tran = _f64(rot).copy(), _f64(tran).copy()
    o = options if options is not None else default_options()
    s = LmSummary()
    rc = lib().orc_lm_solve(C.c_int(mode), C.c_int(1 if per_match else 0), _p(x1), _p(x2), _p(dd), C.c_size_t(n),
                            _p(rot), _p(tran), C.c_double(d1), C.c_double(d2), C.byref(o), C.c_int(threads),
                            C.c_int(1 if faithful else 0), C.byref(s))
    return rot, tran, s, rc


def depth_solve(x1, x2, rot, tran, d12, lam=1.0, c=1.0, options: LmOptions | None = None):
    """d-only stage (reference .cpp:1004-1063) as one bounded trust-region problem; returns (d12, summary, rc)."""
    x1, x2 = _f64(x1).reshape(-1, 3), _f64(x2).reshape(-1, 3)
    d = _f64(d12).reshape(-1, 2).copy()
    o = options if options is not None else default_options()
    s = LmSummary()
    lib().orc_depth_solve.restype = C.c_int
    rc = lib().orc_depth_solve(_p(x1), _p(x2), C.c_size_t(x1.shape[0]), _p(_f64(rot)), _p(_f64(tran)),
                               C.c_double(lam), C.c_double(c), _p(d), C.byref(o), C.byref(s))
    return d, s, rc


def keypoints_to_sphere(kp: np.ndarray, im_w: int, im_h: int) -> np.ndarray:
    kp = np.ascontiguousarray(kp)
    n = kp.shape[0]
    out = np.zeros((n, 3))
    lib().orc_keypoints_to_sphere(_p(kp), C.c_size_t(n), C.c_size_t(kp.strides[0] if n else 28), C.c_int(im_w),
                                  C.c_int(im_h), _p(out))
    return out


def equi2cube(im: np.ndarray, cube: int, clamp: bool = True):
    im = np.ascontiguousarray(im, dtype=np.uint8)
    out = np.zeros((cube, 6 * cube, 3), dtype=np.uint8)
    clamped = lib().orc_equi2cube(_p(im), C.c_int(im.shape[0]), C.c_int(im.shape[1]), C.c_int(cube),
                                  C.c_int(1 if clamp else 0), _p(out))
    return out, int(clamped)


def rotate_keypoints(kp: np.ndarray, pitch_deg: float, im_w: int, im_h: int) -> np.ndarray:
    kp = np.ascontiguousarray(kp).copy()
    lib().orc_rotate_keypoints(_p(kp), C.c_size_t(kp.shape[0]), C.c_size_t(kp.strides[0] if kp.shape[0] else 28),
                               C.c_float(pitch_deg), C.c_int(im_w), C.c_int(im_h))
    return kp


def cube2equi_keypoints(kp: np.ndarray, cube: int, im_w: int, im_h: int) -> np.ndarray:
    kp = np.ascontiguousarray(kp).copy()
    lib().orc_cube2equi_keypoints(_p(kp), C.c_size_t(kp.shape[0]), C.c_size_t(kp.strides[0] if kp.shape[0] else 28),
                                  C.c_int(cube), C.c_int(im_w), C.c_int(im_h))
    return kp


def crop_rotated_image(im: np.ndarray, pitch_deg: float) -> np.ndarray:
    im = np.ascontiguousarray(im, dtype=np.uint8)
    out = np.zeros((im.shape[0] // 4, im.shape[1], 3), dtype=np.uint8)
    lib().orc_crop_rotated_image(_p(im), C.c_int(im.shape[0]), C.c_int(im.shape[1]), C.c_float(pitch_deg), _p(out))
    return out


def interpolating_polynomial(samples):
    """samples: rows (x, value, gradient) -- every value and gradient valid.  Coefficients, highest degree first."""
    a = _f64(samples).reshape(-1, 3)
    out = np.zeros(2 * a.shape[0])
    lib().orc_ls_interpolating_polynomial(_p(a), C.c_int(a.shape[0]), _p(out))
    return out


def minimize_polynomial(poly, x_min, x_max):
    poly = _f64(poly)
    out = np.zeros(2)
    lib().orc_ls_minimize_polynomial(_p(poly), C.c_int(poly.size), C.c_double(x_min), C.c_double(x_max), _p(out))
    return float(out[0]), float(out[1])


PHI_CB = C.CFUNCTYPE(C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def armijo(phi, cost0, slope0, direction_max_norm=1.0, options: LmOptions | None = None):
    """Ceres' Armijo search (first trial 1.0) on phi(a) -> (value, slope).  Returns (success, step size, contractions)."""
    o = options if options is not None else default_options()

    def _cb(a, v, g, _u):
        try:
            v[0], g[0] = map(float, phi(a))
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return -1
    out = np.zeros(3)
    lib().orc_ls_armijo.restype = C.c_int
    rc = lib().orc_ls_armijo(C.byref(o), C.c_double(cost0), C.c_double(slope0), C.c_double(direction_max_norm),
                             PHI_CB(_cb), None, _p(out))
    if rc != 0:
        raise RuntimeError("phi callback failed")
    return bool(out[0]), float(out[1]), int(out[2])


def reference_trial_subsets(n: int, trials: int = 80, reseed: bool = True) -> np.ndarray:
    """(trials, int(n * 0.25)) match indices: the reference's random_array (std::iota + the real libstdc++
    std::random_shuffle on the process-wide rand()) per trial -- oracle/reference_shuffle.cpp.  reseed: srand(1) first, the
    state of a process that never seeded and has drawn nothing."""
    m = C.c_int(0)
    lib().orc_reference_trial_subsets(C.c_int(n), C.c_int(trials), C.c_int(0), None, C.byref(m))
    out = np.zeros((trials, m.value), dtype=np.int32)
    lib().orc_reference_trial_subsets(C.c_int(n), C.c_int(trials), C.c_int(1 if reseed else 0), _p(out), C.byref(m))
    return out


def c_srand(seed: int) -> None:
    """srand() of the C library this process shares with libsba_hip.so (rand() is process state)."""
    lib().orc_srand(C.c_uint(seed))


def c_rand() -> int:
    return int(lib().orc_rand())


def initial_guess_recipe(x1: np.ndarray, x2: np.ndarray, subsets: np.ndarray):
    """The reference's initial_guess (spherical_bundle_adjuster.cpp:118-181) restated in numpy on GIVEN subsets (rows of
    `subsets` = the match indices of one trial): explicit rows kron(left, right) (.cpp:53-68), LAPACK SVD, last row of vt (for
    fewer than 9 rows that is row m - 1 of the economy vt, as cv::SVDecomp returns it), rank-2 projection (.cpp:75-80),
    R1 / R2 / t as cv::decomposeEssentialMat builds them, float32 Euler angles (rot2euler returns cv::Vec3f), validity
    |angle| < 1.57 (.cpp:101-115), 20-80 % trimmed-mean consensus (.cpp:162-180).  Returns (R_vec_out, T_vec_out, number
    of candidates) or (None, None, 0).  Test infrastructure, like everything under oracle/."""
    def euler_of(R):
        sy = np.hypot(R[0, 0], R[1, 0])
        return np.array([np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], sy), np.arctan2(R[1, 0], R[0, 0])])
    A = (x1[:, :, None] * x2[:, None, :]).reshape(len(x1), 9)
    W = np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]])
    cands, tvecs = [], []
    for idx in subsets:
        vt = np.linalg.svd(A[idx], full_matrices=False)[2]
        E = vt[-1].reshape(3, 3)
        U, s, Vt = np.linalg.svd(E)
        U, s, Vt = np.linalg.svd(U @ np.diag([s[0], s[1], 0.0]) @ Vt)
        if np.linalg.det(U) < 0:
            U = -U
        if np.linalg.det(Vt) < 0:
            Vt = -Vt
        for R in (U @ W @ Vt, U @ W.T @ Vt):
            e = euler_of(R).astype(np.float32)
            if np.abs(e).max() < 1.57:
                cands.append(e)
                tvecs.append(U[:, 2])
    c = np.array(cands, dtype=np.float32)
    if len(c) == 0:
        return None, None, 0
    d = np.sort(np.linalg.norm((c[:, None, :] - c[None, :, :]).astype(np.float64), axis=2), axis=1)
    lo, hi = int(len(c) * 0.2), int(len(c) * 0.8)
    pick = int(np.argmin(d[:, lo:hi].mean(axis=1)))
    return c[pick], tvecs[pick], len(c)
