"""Host-side Python mirror of the C-ABI (thin; the C++ mirror of the reference class lives in
``csrc/spherical_bundle_adjuster.hpp``).  Names follow the reference's domain: a *problem* holds
one shard of *correspondences* (matched key-points as unit vectors), a *sweep* is one residual +
Jacobian evaluation over them, the *solve stage* is the LM loop of ``solve_problem``
(reference spherical_bundle_adjuster.cpp:183-217)."""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass

import numpy as np

from . import _cabi as cabi
from ._cabi import (DEPTH_PER_MATCH, DEPTH_UNIFORM, KERNEL_EXPLICIT, KERNEL_FACTORED, MODE_ROT, MODE_RT, MODE_TRAN, STORE_F32,  # noqa: F401
                    STORE_F64, TRAN_FREE, TRAN_SPHERE, SbaError)


@dataclass
class NormalEquations:
    H: np.ndarray          # (6, 6) sum rho' J^T J over [rot | tran]
    g: np.ndarray          # (6,)   sum rho' J^T e
    cost: float            # 1/2 sum rho
    sum_w: float
    n_outlier: float


@dataclass
class SolveSummary:
    termination: str
    num_iterations: int
    num_successful_steps: int
    num_evaluations: int
    initial_cost: float
    final_cost: float
    final_gradient_max_norm: float
    final_radius: float
    seconds_total: float
    seconds_eval: float
    num_line_search_steps: int = 0


def _f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def device_count() -> int:
    lib = cabi.load_library()
    n = C.c_int(0)
    cabi.check(lib, lib.sba_device_count(C.byref(n)))
    return n.value


def default_lm_options(**overrides) -> cabi.LmOptions:
    lib = cabi.load_library()
    o = cabi.LmOptions()
    lib.sba_lm_options_default(C.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def expand_pack(mode: int, pack) -> NormalEquations:
    lib = cabi.load_library()
    pack = _f64(pack, (cabi.PACK_SIZE,))
    ne = cabi.NormalEq()
    cabi.check(lib, lib.sba_expand_pack(mode, _dptr(pack), C.byref(ne)))
    return _to_ne(ne)


def _to_ne(ne: cabi.NormalEq) -> NormalEquations:
    return NormalEquations(np.array(ne.H, dtype=np.float64).reshape(6, 6), np.array(ne.g, dtype=np.float64),
                           float(ne.cost), float(ne.sum_w), float(ne.n_outlier))


def _summary(s: cabi.LmSummary) -> SolveSummary:
    return SolveSummary(cabi.TERMINATION.get(s.termination, str(s.termination)), s.num_iterations,
                        s.num_successful_steps, s.num_evaluations, s.initial_cost, s.final_cost,
                        s.final_gradient_max_norm, s.final_radius, s.seconds_total, s.seconds_eval,
                        s.num_line_search_steps)


class Problem:
    """One shard of correspondences resident on one GPU (``sba_problem``)."""

    def __init__(self, device: int = 0, stream: int | None = None, lib=None):
        self._lib = lib if lib is not None else cabi.load_library()
        self._h = C.c_void_p()
        cabi.check(self._lib, self._lib.sba_problem_create(C.byref(self._h), device, C.c_void_p(stream or 0)))
        self._hook_keepalive = None
        self.device = device

    # -- life cycle ---------------------------------------------------------------------------
    def close(self) -> None:
        """Destroy the handle.  `destroy_status` keeps the library's answer: non-zero means the handle was POISONED (a
        device wait timed out or the device faulted) and its device resources were leaked instead of freed -- the
        process should then exit non-zero; never raise from here, close() runs inside `with` blocks that are unwinding."""
        if getattr(self, "_h", None) is not None and self._h:
            self.destroy_status = int(self._lib.sba_problem_destroy(self._h))
            self._h = C.c_void_p()
            if self.destroy_status != 0:
                import warnings
                warnings.warn("sba_problem_destroy: " + cabi.last_error(self._lib), ResourceWarning, stacklevel=2)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data ------------------------------------------------------------------------------------
    def upload(self, left_xyz, right_xyz, d12=None, store: int = STORE_F64) -> None:
        """left_xyz/right_xyz: (n,3) f64 unit vectors (cv::Point3d layout); d12: (n,2) or None."""
        x1 = _f64(left_xyz).reshape(-1, 3)
        x2 = _f64(right_xyz).reshape(-1, 3)
        if x1.shape != x2.shape:
            raise ValueError("left/right shapes differ")
        n = x1.shape[0]
        dp = None
        if d12 is not None:
            d = _f64(d12).reshape(-1, 2)
            if d.shape[0] != n:
                raise ValueError("d12 length differs")
            dp = d.ctypes.data_as(C.c_void_p)
        cabi.check(self._lib, self._lib.sba_problem_upload(
            self._h, x1.ctypes.data_as(C.c_void_p), x2.ctypes.data_as(C.c_void_p), dp, n, store))

    def upload_keypoints(self, left_kp: np.ndarray, right_kp: np.ndarray, im_width: int, im_height: int, d12=None,
                         store: int = STORE_F64) -> None:
        """Matched cv::KeyPoint-like records (first two float32 = pt.x, pt.y) of both images -> device planes,
        pixel -> unit sphere computed on the device (reference .cpp:271-298)."""
        kl, kr = np.ascontiguousarray(left_kp), np.ascontiguousarray(right_kp)
        if kl.shape != kr.shape:
            raise ValueError("left/right key-point arrays differ")
        n = kl.shape[0]
        stride = kl.strides[0] if n > 0 else 28
        dp = None
        if d12 is not None:
            d = _f64(d12).reshape(-1, 2)
            if d.shape[0] != n:
                raise ValueError("d12 length differs")
            dp = d.ctypes.data_as(C.c_void_p)
        cabi.check(self._lib, self._lib.sba_problem_upload_keypoints(
            self._h, kl.ctypes.data_as(C.c_void_p), kr.ctypes.data_as(C.c_void_p), n, stride, im_width, im_height, dp,
            store))

    def upload_device(self, left_ptr: int, right_ptr: int, d12_ptr: int | None, n: int,
                      store: int = STORE_F64) -> None:
        cabi.check(self._lib, self._lib.sba_problem_upload_device(
            self._h, C.c_void_p(left_ptr), C.c_void_p(right_ptr), C.c_void_p(d12_ptr or 0), n, store))

    def set_depths(self, d12) -> None:
        """(Re)upload only the per-match depths (n, 2) of the resident correspondences."""
        d = _f64(d12).reshape(-1, 2)
        if d.shape[0] != self.size:
            raise ValueError("d12 length differs from the uploaded correspondences")
        cabi.check(self._lib, self._lib.sba_problem_set_depths(self._h, d.ctypes.data_as(C.c_void_p)))

    def set_kernel(self, kind: int) -> None:
        """KERNEL_FACTORED (default) or KERNEL_EXPLICIT."""
        cabi.check(self._lib, self._lib.sba_problem_set_kernel(self._h, kind))

    @property
    def size(self) -> int:
        n = C.c_size_t(0)
        cabi.check(self._lib, self._lib.sba_problem_size(self._h, C.byref(n)))
        return n.value

    # -- sweeps -----------------------------------------------------------------------------------
    def eval(self, mode, rot, tran, d1=1.0, d2=1.0, huber_delta=1.0, depth_mode=DEPTH_UNIFORM) -> NormalEquations:
        rot, tran = _f64(rot, (3,)), _f64(tran, (3,))
        ne = cabi.NormalEq()
        cabi.check(self._lib, self._lib.sba_problem_eval(self._h, mode, depth_mode, _dptr(rot), _dptr(tran),
                                                         d1, d2, huber_delta, C.byref(ne)))
        return _to_ne(ne)

    def eval_pack(self, mode, rot, tran, d1=1.0, d2=1.0, huber_delta=1.0, depth_mode=DEPTH_UNIFORM) -> np.ndarray:
        rot, tran = _f64(rot, (3,)), _f64(tran, (3,))
        pack = np.zeros(cabi.PACK_SIZE)
        cabi.check(self._lib, self._lib.sba_problem_eval_pack(self._h, mode, depth_mode, _dptr(rot), _dptr(tran),
                                                              d1, d2, huber_delta, _dptr(pack)))
        return pack

    def eval_timed(self, mode, rot, tran, d1=1.0, d2=1.0, huber_delta=1.0, depth_mode=DEPTH_UNIFORM,
                   repeat: int = 10):
        """`repeat` back-to-back sweeps timed with HIP events on the problem's stream.
        Returns (pack, mean ms per step [sweep+finalize(+all-reduce)], mean ms of the sweep kernel alone)."""
        rot, tran = _f64(rot, (3,)), _f64(tran, (3,))
        pack = np.zeros(cabi.PACK_SIZE)
        step_ms, sweep_ms = C.c_double(0), C.c_double(0)
        cabi.check(self._lib, self._lib.sba_problem_eval_timed(self._h, mode, depth_mode, _dptr(rot), _dptr(tran),
                                                               d1, d2, huber_delta, repeat, _dptr(pack),
                                                               C.byref(step_ms), C.byref(sweep_ms)))
        return pack, step_ms.value, sweep_ms.value

    def eval_launch_times(self, mode, rot, tran, d1=1.0, d2=1.0, huber_delta=1.0, depth_mode=DEPTH_UNIFORM,
                          repeat: int = 20) -> np.ndarray:
        """Device time (ms) of each of `repeat` back-to-back sweep-kernel launches (HIP event between every two)."""
        rot, tran = _f64(rot, (3,)), _f64(tran, (3,))
        ms = np.zeros(repeat, dtype=np.float32)
        cabi.check(self._lib, self._lib.sba_problem_eval_launch_times(
            self._h, mode, depth_mode, _dptr(rot), _dptr(tran), d1, d2, huber_delta, repeat,
            ms.ctypes.data_as(C.POINTER(C.c_float))))
        return ms.astype(np.float64)

    def eval_steps(self, mode, rot, tran, d1=1.0, d2=1.0, huber_delta=1.0, depth_mode=DEPTH_UNIFORM, steps: int = 1):
        """`steps` host-synchronous sweeps in a row inside the library; returns (last pack, seconds of the loop)."""
        rot, tran = _f64(rot, (3,)), _f64(tran, (3,))
        pack = np.zeros(cabi.PACK_SIZE)
        sec = C.c_double(0)
        cabi.check(self._lib, self._lib.sba_problem_eval_steps(self._h, mode, depth_mode, _dptr(rot), _dptr(tran),
                                                               d1, d2, huber_delta, steps, _dptr(pack), C.byref(sec)))
        return pack, sec.value

    # -- solve stage --------------------------------------------------------------------------------
    def solve(self, mode, rot, tran, d1=1.0, d2=1.0, depth_mode=DEPTH_UNIFORM, options: cabi.LmOptions | None = None):
        """LM solve; returns (rot, tran, SolveSummary).  Inputs are not modified."""
        rot = _f64(rot, (3,)).copy()
        tran = _f64(tran, (3,)).copy()
        opt = options if options is not None else default_lm_options()
        s = cabi.LmSummary()
        cabi.check(self._lib, self._lib.sba_problem_solve(self._h, mode, depth_mode, _dptr(rot), _dptr(tran),
                                                          d1, d2, C.byref(opt), C.byref(s)))
        return rot, tran, _summary(s)

    def solve_depths(self, rot, tran, lam=1.0, c=1.0, options: cabi.LmOptions | None = None):
        """d-only stage: refines the uploaded per-match depths in place; returns (d12 (n,2), SolveSummary)."""
        rot, tran = _f64(rot, (3,)), _f64(tran, (3,))
        out = np.zeros((self.size, 2))
        opt = options if options is not None else default_lm_options()
        s = cabi.LmSummary()
        cabi.check(self._lib, self._lib.sba_problem_solve_depths(self._h, _dptr(rot), _dptr(tran), lam, c,
                                                                 C.byref(opt), out.ctypes.data_as(C.c_void_p),
                                                                 C.byref(s)))
        return out, _summary(s)

    # -- 8-point initial guess ----------------------------------------------------------------------
    def epipolar_moments(self) -> np.ndarray:
        """(64, 45): upper triangle of A^T A of the kron(left, right) rows per interleaved group (i//2) % 64."""
        g = np.zeros((64, 45))
        cabi.check(self._lib, self._lib.sba_problem_epipolar_moments(self._h, _dptr(g)))
        return g

    def initial_guess(self, trials: int = 80, subset_fraction: float = 0.25, seed: int = 0):
        """Returns (euler (3,), tran (3,), number of candidates): R_vec_out / T_vec_out of the reference."""
        e, t, n = np.zeros(3), np.zeros(3), C.c_int(0)
        cabi.check(self._lib, self._lib.sba_problem_initial_guess(self._h, trials, subset_fraction, seed, _dptr(e),
                                                                  _dptr(t), C.byref(n)))
        return e, t, n.value

    def epipolar_subset_moments(self, indices) -> np.ndarray:
        """indices: (trials, m) int32 match indices -> (trials, 45): A^T A of the kron(left, right) rows of each list."""
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        if idx.ndim != 2:
            raise ValueError("indices must be (trials, m)")
        out = np.zeros((idx.shape[0], 45))
        cabi.check(self._lib, self._lib.sba_problem_epipolar_subset_moments(self._h, idx.ctypes.data_as(C.c_void_p), idx.shape[0],
                                                                            idx.shape[1], _dptr(out)))
        return out

    def initial_guess_reference(self, trials: int = 80, subset_fraction: float = 0.25):
        """The initial guess from the subsets the reference itself would draw (std::random_shuffle on this process's
        rand() stream, reference .hpp:182-211 / .cpp:130-141).  Returns (euler, tran, number of candidates)."""
        e, t, n = np.zeros(3), np.zeros(3), C.c_int(0)
        cabi.check(self._lib, self._lib.sba_problem_initial_guess_reference(self._h, trials, subset_fraction, _dptr(e), _dptr(t),
                                                                            C.byref(n)))
        return e, t, n.value

    # -- multi-GPU ------------------------------------------------------------------------------------
    def comm_init_rank(self, nranks: int, rank: int, unique_id: bytes) -> None:
        if len(unique_id) != cabi.COMM_ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        cabi.check(self._lib, self._lib.sba_problem_comm_init_rank(self._h, nranks, rank, unique_id))

    def comm_destroy(self) -> None:
        cabi.check(self._lib, self._lib.sba_problem_comm_destroy(self._h))

    def peer_export(self, nranks: int, rank: int) -> bytes:
        """Allocate this rank's inbox for the direct peer exchange; returns its 64-byte IPC handle."""
        buf = C.create_string_buffer(cabi.PEER_HANDLE_BYTES)
        cabi.check(self._lib, self._lib.sba_problem_peer_export(self._h, nranks, rank, buf))
        return buf.raw

    def peer_connect(self, handles: bytes) -> None:
        """handles: the nranks 64-byte IPC handles in rank order, concatenated."""
        cabi.check(self._lib, self._lib.sba_problem_peer_connect(self._h, handles))

    def peer_selftest(self, rounds: int = 8) -> bool:
        ok = C.c_int(0)
        cabi.check(self._lib, self._lib.sba_problem_peer_selftest(self._h, rounds, C.byref(ok)))
        return bool(ok.value)

    def peer_disable(self) -> None:
        cabi.check(self._lib, self._lib.sba_problem_peer_disable(self._h))

    def set_shard(self, rank: int, nranks: int) -> None:
        """Which shard of the correspondences this problem holds (needed by the d-only stage over the user hook)."""
        cabi.check(self._lib, self._lib.sba_problem_set_shard(self._h, rank, nranks))

    def set_allreduce(self, fn) -> None:
        """fn(device_ptr: int, count: int, stream: int) -> int (0 = ok), or None to clear."""
        if fn is None:
            cb = C.cast(None, cabi.ALLREDUCE_FN)
        else:
            def _tramp(buf, count, stream, _user):
                try:
                    return int(fn(buf or 0, count, stream or 0) or 0)
                except Exception:  # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return -1
            cb = cabi.ALLREDUCE_FN(_tramp)
        self._hook_keepalive = cb
        cabi.check(self._lib, self._lib.sba_problem_set_allreduce(self._h, cb, None))

    @property
    def pack_device_ptr(self) -> int:
        p = C.c_void_p()
        cabi.check(self._lib, self._lib.sba_problem_pack_device_ptr(self._h, C.byref(p)))
        return p.value or 0


class Batch:
    """Many independent two-view problems ("pairs") on one GPU (``sba_batch``; BASELINE config C5)."""

    def __init__(self, device: int = 0, stream: int | None = None, lib=None):
        self._lib = lib if lib is not None else cabi.load_library()
        self._h = C.c_void_p()
        cabi.check(self._lib, self._lib.sba_batch_create(C.byref(self._h), device, C.c_void_p(stream or 0)))
        self.num_pairs = 0
        self._total = 0

    def close(self) -> None:
        """As Problem.close(): `destroy_status` != 0 = the batch was poisoned and its device resources were leaked."""
        if getattr(self, "_h", None) is not None and self._h:
            self.destroy_status = int(self._lib.sba_batch_destroy(self._h))
            self._h = C.c_void_p()
            if self.destroy_status != 0:
                import warnings
                warnings.warn("sba_batch_destroy: " + cabi.last_error(self._lib), ResourceWarning, stacklevel=2)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_kernel(self, kind: int) -> None:
        cabi.check(self._lib, self._lib.sba_batch_set_kernel(self._h, kind))

    def upload(self, left_xyz, right_xyz, offsets, d12=None, store: int = STORE_F64) -> None:
        """left/right: (total, 3); offsets: (num_pairs+1,) row offsets of each pair; d12: (total, 2) or None."""
        x1 = _f64(left_xyz).reshape(-1, 3)
        x2 = _f64(right_xyz).reshape(-1, 3)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        if off.ndim != 1 or off.size < 1 or int(off[-1]) > x1.shape[0] or x1.shape != x2.shape:
            raise ValueError("bad offsets / array shapes")
        dp = None
        if d12 is not None:
            d = _f64(d12).reshape(-1, 2)
            if d.shape[0] != x1.shape[0]:
                raise ValueError("d12 length differs")
            dp = d.ctypes.data_as(C.c_void_p)
        self.num_pairs = off.size - 1
        self._total = int(off[-1])
        cabi.check(self._lib, self._lib.sba_batch_upload(
            self._h, x1.ctypes.data_as(C.c_void_p), x2.ctypes.data_as(C.c_void_p), dp,
            off.ctypes.data_as(C.POINTER(C.c_size_t)), self.num_pairs, store))

    def set_depths(self, d12) -> None:
        """Re-send only the per-match depths (total, 2), laid out like the d12 of upload(); the coordinates stay resident."""
        d = _f64(d12).reshape(-1, 2)
        if self.num_pairs and d.shape[0] != self._total:
            raise ValueError("d12 length differs from the uploaded pairs")
        cabi.check(self._lib, self._lib.sba_batch_set_depths(self._h, _dptr(d)))

    @property
    def blocks_per_pair(self) -> int:
        n, b = C.c_int(0), C.c_int(0)
        cabi.check(self._lib, self._lib.sba_batch_size(self._h, C.byref(n), C.byref(b)))
        return b.value

    def _pp(self, a, width):
        if a is None:
            return None, None
        arr = _f64(a).reshape(self.num_pairs, width) if width > 1 else _f64(a).reshape(self.num_pairs)
        return arr, _dptr(arr)

    def eval(self, mode, rot, tran, d1=None, d2=None, huber_delta=1.0, depth_mode=DEPTH_UNIFORM) -> np.ndarray:
        """rot, tran: (num_pairs, 3).  Returns packs (num_pairs, 24) in the SBA_PACK layout."""
        rot, rp = self._pp(rot, 3)
        tran, tp = self._pp(tran, 3)
        d1a, d1p = self._pp(d1, 1)
        d2a, d2p = self._pp(d2, 1)
        packs = np.zeros((self.num_pairs, cabi.PACK_SIZE))
        cabi.check(self._lib, self._lib.sba_batch_eval(self._h, mode, depth_mode, rp, tp, d1p, d2p, huber_delta,
                                                       _dptr(packs)))
        return packs

    def eval_timed(self, mode, rot, tran, steps, d1=None, d2=None, huber_delta=1.0, depth_mode=DEPTH_UNIFORM):
        """`steps` host-synchronous batched steps in a C loop.  Returns (packs, dict of mean ms: step / prepare /
        device / convert)."""
        rot, rp = self._pp(rot, 3)
        tran, tp = self._pp(tran, 3)
        d1a, d1p = self._pp(d1, 1)
        d2a, d2p = self._pp(d2, 1)
        packs = np.zeros((self.num_pairs, cabi.PACK_SIZE))
        ms = [C.c_double(0) for _ in range(4)]
        cabi.check(self._lib, self._lib.sba_batch_eval_timed(
            self._h, mode, depth_mode, rp, tp, d1p, d2p, huber_delta, steps, _dptr(packs),
            *[C.cast(C.byref(m), C.POINTER(C.c_double)) for m in ms]))
        return packs, dict(zip(("step_ms", "prepare_ms", "device_ms", "convert_ms"), (m.value for m in ms)))

    def _launch_times(self, fn, mode, rot, tran, d1, d2, huber_delta, depth_mode, repeat) -> np.ndarray:
        rot, rp = self._pp(rot, 3)
        tran, tp = self._pp(tran, 3)
        d1a, d1p = self._pp(d1, 1)
        d2a, d2p = self._pp(d2, 1)
        ms = np.zeros(repeat, dtype=np.float32)
        cabi.check(self._lib, fn(self._h, mode, depth_mode, rp, tp, d1p, d2p, huber_delta, repeat,
                                 ms.ctypes.data_as(C.POINTER(C.c_float))))
        return ms.astype(np.float64)

    def sweep_launch_times(self, mode, rot, tran, d1=None, d2=None, huber_delta=1.0, depth_mode=DEPTH_UNIFORM,
                           repeat: int = 20) -> np.ndarray:
        """Device time (ms) of each of `repeat` back-to-back launches of the batched sweep kernel alone."""
        return self._launch_times(self._lib.sba_batch_sweep_launch_times, mode, rot, tran, d1, d2, huber_delta, depth_mode,
                                  repeat)

    def step_launch_times(self, mode, rot, tran, d1=None, d2=None, huber_delta=1.0, depth_mode=DEPTH_UNIFORM,
                          repeat: int = 20) -> np.ndarray:
        """The same for the dominant kernel of the step this batch really runs (`step_is_fused`: the one-launch
        batch_step_kernel, else the batched sweep kernel)."""
        return self._launch_times(self._lib.sba_batch_step_launch_times, mode, rot, tran, d1, d2, huber_delta, depth_mode,
                                  repeat)

    @property
    def step_is_fused(self) -> bool:
        rc = self._lib.sba_batch_step_is_fused(self._h)
        if rc < 0:
            cabi.check(self._lib, rc)
        return rc == 1

    def solve(self, mode, rot, tran, d1=None, d2=None, depth_mode=DEPTH_UNIFORM, options: cabi.LmOptions | None = None):
        """Per-pair LM in lock-step.  Returns (rot (B,3), tran (B,3), [SolveSummary], status (B,))."""
        rot = _f64(rot).reshape(self.num_pairs, 3).copy()
        tran = _f64(tran).reshape(self.num_pairs, 3).copy()
        d1a, d1p = self._pp(d1, 1)
        d2a, d2p = self._pp(d2, 1)
        opt = options if options is not None else default_lm_options()
        sums = (cabi.LmSummary * max(self.num_pairs, 1))()
        status = np.zeros(max(self.num_pairs, 1), dtype=np.int32)
        cabi.check(self._lib, self._lib.sba_batch_solve(self._h, mode, depth_mode, _dptr(rot), _dptr(tran), d1p, d2p,
                                                        C.byref(opt), sums, status.ctypes.data_as(C.POINTER(C.c_int))))
        return rot, tran, [_summary(sums[i]) for i in range(self.num_pairs)], status[:self.num_pairs]

    def solve_depths(self, rot, tran, lam: float = 1.0, c: float = 1.0, options: cabi.LmOptions | None = None,
                     total_matches: int | None = None, want_depths: bool = True):
        """The d-only stage for every pair (its own trust region / line search / convergence each), in lock-step.
        Returns (d12 (offsets[-1], 2) indexed like the uploaded depths, [SolveSummary], status (B,))."""
        rot = _f64(rot).reshape(self.num_pairs, 3)
        tran = _f64(tran).reshape(self.num_pairs, 3)
        opt = options if options is not None else default_lm_options()
        total = int(self._total) if total_matches is None else int(total_matches)
        d12 = np.zeros((total, 2)) if want_depths else None           # None: the refined depths stay on the device only
        sums = (cabi.LmSummary * max(self.num_pairs, 1))()
        status = np.zeros(max(self.num_pairs, 1), dtype=np.int32)
        cabi.check(self._lib, self._lib.sba_batch_solve_depths(self._h, _dptr(rot), _dptr(tran), lam, c, C.byref(opt),
                                                               d12.ctypes.data_as(C.c_void_p) if want_depths else None, sums,
                                                               status.ctypes.data_as(C.POINTER(C.c_int))))
        return d12, [_summary(sums[i]) for i in range(self.num_pairs)], status[:self.num_pairs]


    def solve_problem(self, rot=None, tran=None, use_initial_guess: bool = True, trials: int = 80, subset_fraction: float = 0.25,
                      seed: int = 0, options: cabi.LmOptions | None = None, want_depths: bool = False, check: bool = True):
        """The reference's per-pair pipeline for every pair (initial guess -> d-only -> rot-only -> tran-only,
        reference .cpp:302-331, :183-217).  rot / tran: start values (B, 3) (needed when use_initial_guess is False; with the
        guess they only serve pairs that have no valid candidate).  Returns a dict: rot, tran (B, 3), d_uniform (B, 2),
        guess_candidates (B,), depth / rot / tran stage summaries, status (B,), d12 (if want_depths)."""
        B = self.num_pairs
        r = np.zeros((max(B, 1), 3)) if rot is None else _f64(rot).reshape(B, 3).copy()
        t = np.zeros((max(B, 1), 3)) if tran is None else _f64(tran).reshape(B, 3).copy()
        opt = options if options is not None else default_lm_options()
        d12 = np.zeros((int(self._total), 2)) if want_depths else None
        du = np.zeros((max(B, 1), 2))
        nc, status = np.zeros(max(B, 1), dtype=np.int32), np.zeros(max(B, 1), dtype=np.int32)
        sums = [(cabi.LmSummary * max(B, 1))() for _ in range(3)]
        t0 = time.perf_counter()
        rc = self._lib.sba_batch_solve_problem(self._h, 1 if use_initial_guess else 0, trials, subset_fraction, seed, _dptr(r), _dptr(t),
                                               C.byref(opt), d12.ctypes.data_as(C.c_void_p) if want_depths else None, _dptr(du),
                                               nc.ctypes.data_as(C.POINTER(C.c_int)), sums[0], sums[1], sums[2],
                                               status.ctypes.data_as(C.POINTER(C.c_int)))
        seconds = time.perf_counter() - t0       # the library call alone (the summaries below are 3 B Python objects)
        if check or rc != cabi.SBA_ERR_NUMERIC:
            cabi.check(self._lib, rc)
        return {"rot": r[:B], "tran": t[:B], "d_uniform": du[:B], "guess_candidates": nc[:B], "status": status[:B], "d12": d12,
                "seconds_inside_the_library": seconds,
                "depth_stage": [_summary(sums[0][i]) for i in range(B)], "rot_stage": [_summary(sums[1][i]) for i in range(B)],
                "tran_stage": [_summary(sums[2][i]) for i in range(B)]}

    def epipolar_moments(self):
        """Group moments of every pair: (B, 64, 45)."""
        g = np.zeros((max(self.num_pairs, 1), 64, 45))
        cabi.check(self._lib, self._lib.sba_batch_epipolar_moments(self._h, _dptr(g)))
        return g[:self.num_pairs]

    def initial_guess(self, trials: int = 80, subset_fraction: float = 0.25, seed: int = 0, check: bool = True):
        """The 8-point initial guess of every pair (reference initial_guess, .cpp:47-181, once per pair).
        Returns (rot_euler (B, 3), tran (B, 3), num_candidates (B,), status (B,)); check=False: a pair without a valid
        candidate is reported in status only."""
        B = self.num_pairs
        e, t = np.zeros((max(B, 1), 3)), np.zeros((max(B, 1), 3))
        nc, status = np.zeros(max(B, 1), dtype=np.int32), np.zeros(max(B, 1), dtype=np.int32)
        rc = self._lib.sba_batch_initial_guess(self._h, trials, subset_fraction, seed, _dptr(e), _dptr(t),
                                               nc.ctypes.data_as(C.POINTER(C.c_int)), status.ctypes.data_as(C.POINTER(C.c_int)))
        if check or rc != cabi.SBA_ERR_NUMERIC:
            cabi.check(self._lib, rc)
        return e[:B], t[:B], nc[:B], status[:B]


def set_host_threads(n: int) -> None:
    """Host threads for the host-side trial loop of the initial guess (reference: set_omp).  0 = auto."""
    lib = cabi.load_library()
    cabi.check(lib, lib.sba_set_host_threads(n))


def initial_guess_from_moments(groups, trials: int = 80, subset_fraction: float = 0.25, seed: int = 0):
    """Host-only part of the initial guess (no device needed)."""
    lib = cabi.load_library()
    g = _f64(groups, (64, 45))
    e, t, n = np.zeros(3), np.zeros(3), C.c_int(0)
    cabi.check(lib, lib.sba_initial_guess_from_moments(_dptr(g), trials, subset_fraction, seed, _dptr(e), _dptr(t),
                                                       C.byref(n)))
    return e, t, n.value


def reference_rand_seed(seed: int = 1) -> None:
    """Reset the library's copy of the reference's rand() stream (1 = the never-seeded state a fresh process starts in)."""
    lib = cabi.load_library()
    cabi.check(lib, lib.sba_reference_rand_seed(seed))


def reference_rand_next() -> int:
    return int(cabi.load_library().sba_reference_rand_next())


def reference_trial_subsets(n: int, trials: int = 80, subset_fraction: float = 0.25) -> np.ndarray:
    """Host-only: (trials, int(n * subset_fraction)) match indices the reference's trials draw from the library's copy of
    the reference's rand() stream at this point (sba_reference_trial_subsets)."""
    lib = cabi.load_library()
    m = C.c_int(0)
    cabi.check(lib, lib.sba_reference_trial_subsets(n, trials, subset_fraction, None, C.byref(m)))
    out = np.zeros((trials, m.value), dtype=np.int32)
    cabi.check(lib, lib.sba_reference_trial_subsets(n, trials, subset_fraction, out.ctypes.data_as(C.c_void_p), C.byref(m)))
    return out


def rccl_available() -> bool:
    """librccl loadable and bound in this process (checked on every rank before anyone enters ncclCommInitRank)."""
    return bool(cabi.load_library().sba_rccl_available())


def comm_unique_id() -> bytes:
    lib = cabi.load_library()
    buf = C.create_string_buffer(cabi.COMM_ID_BYTES)
    cabi.check(lib, lib.sba_comm_unique_id(buf))
    return buf.raw


def keypoints_to_sphere(keypoints: np.ndarray, im_width: int, im_height: int, device: int = 0) -> np.ndarray:
    """keypoints: structured/2-D array whose rows start with float32 pt.x, pt.y (cv::KeyPoint: 28 B)."""
    lib = cabi.load_library()
    kp = np.ascontiguousarray(keypoints)
    n = kp.shape[0]
    stride = kp.strides[0] if n > 0 else 28
    out = np.zeros((n, 3))
    cabi.check(lib, lib.sba_keypoints_to_sphere(device, kp.ctypes.data_as(C.c_void_p), n, stride, im_width,
                                                im_height, out.ctypes.data_as(C.c_void_p)))
    return out


def equi2cube(erp: np.ndarray, cube_size: int, device: int = 0) -> np.ndarray:
    """erp: (H, W, 3) uint8 -> (S, 6S, 3) uint8 strip, faces left,front,right,back,top,bottom."""
    lib = cabi.load_library()
    im = np.ascontiguousarray(erp, dtype=np.uint8)
    if im.ndim != 3 or im.shape[2] != 3:
        raise ValueError("erp must be (H, W, 3) uint8")
    out = np.zeros((cube_size, 6 * cube_size, 3), dtype=np.uint8)
    cabi.check(lib, lib.sba_equi2cube(device, im.ctypes.data_as(C.c_void_p), im.shape[0], im.shape[1], cube_size,
                                      out.ctypes.data_as(C.c_void_p)))
    return out


def _kp_inplace(fn, keypoints: np.ndarray, *args, device: int = 0) -> np.ndarray:
    lib = cabi.load_library()
    kp = np.ascontiguousarray(keypoints).copy()
    n = kp.shape[0]
    stride = kp.strides[0] if n > 0 else 28
    cabi.check(lib, getattr(lib, fn)(device, kp.ctypes.data_as(C.c_void_p), n, stride, *args))
    return kp


def rotate_keypoints(keypoints: np.ndarray, pitch_deg: float, im_width: int, im_height: int, device: int = 0):
    """spherical_surf::rotate_keypoint on cv::KeyPoint-like records (first two float32 = pt.x, pt.y); returns a copy."""
    return _kp_inplace("sba_rotate_keypoints", keypoints, C.c_float(pitch_deg), im_width, im_height, device=device)


def cube2equi_keypoints(keypoints: np.ndarray, cube_size: int, im_width: int, im_height: int, device: int = 0):
    """equi2cube_surf::cube2equi_pixel on key-point records; returns a copy."""
    return _kp_inplace("sba_cube2equi_keypoints", keypoints, cube_size, im_width, im_height, device=device)


def crop_rotated_image(erp: np.ndarray, pitch_deg: float, device: int = 0) -> np.ndarray:
    """spherical_surf::crop_rotated_image: (H, W, 3) uint8 -> (H//4, W, 3) rotated equatorial band."""
    lib = cabi.load_library()
    im = np.ascontiguousarray(erp, dtype=np.uint8)
    if im.ndim != 3 or im.shape[2] != 3:
        raise ValueError("erp must be (H, W, 3) uint8")
    out = np.zeros((im.shape[0] // 4, im.shape[1], 3), dtype=np.uint8)
    cabi.check(lib, lib.sba_crop_rotated_image(device, im.ctypes.data_as(C.c_void_p), im.shape[0], im.shape[1],
                                               C.c_float(pitch_deg), out.ctypes.data_as(C.c_void_p)))
    return out
