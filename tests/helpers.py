"""Shared test helpers (tests/ only)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
HEADER = ROOT / "include" / "sba_hip.h"

# tolerances (stated once, used everywhere)
REL_TOL_F64 = 1e-12      # normal-equation entries, f64 planes: relative to the largest |entry| of the block
REL_TOL_F32 = 5e-6       # f32 planes (inputs rounded to 24 bits)
RT_TOL_F64 = 1e-9        # recovered rot (rad) / tran, f64 planes
RT_TOL_F32 = 1e-5


def declared_functions():
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sba_[a-z0-9_]+)\s*\(", txt)))


def pack_from_eval(mode, ev):
    """oracle Eval (6x6 H, g, cost, ...) -> the 24-double pack layout of include/sba_hip.h."""
    p = np.zeros(24)
    if mode in (0, 2):
        k = 0
        for a in range(3):
            for b in range(a, 3):
                p[k] = ev.H[a, b]
                k += 1
        p[16:19] = ev.g[:3]
    if mode in (1, 2):
        p[15] = ev.sum_w
        p[19:22] = ev.g[3:]
    if mode == 2:
        p[6:15] = ev.H[:3, 3:].reshape(-1)
    p[22] = ev.cost
    p[23] = ev.n_outlier
    return p


def assert_normal_eq_close(got, ref, rel, what=""):
    """got: api.NormalEquations, ref: oracle Eval.  Block-relative comparison."""
    sH = max(np.abs(ref.H).max(), 1e-300)
    assert np.abs(got.H - ref.H).max() <= rel * sH, f"{what} H rel err {np.abs(got.H - ref.H).max() / sH:.3e}"
    sg = max(np.abs(ref.g).max(), rel * sH, 1e-300)
    assert np.abs(got.g - ref.g).max() <= rel * sg * 10, f"{what} g rel err {np.abs(got.g - ref.g).max() / sg:.3e}"
    assert abs(got.cost - ref.cost) <= rel * max(abs(ref.cost), 1e-300), f"{what} cost {got.cost} vs {ref.cost}"


_harness = None


def lm_harness():
    """Product host LM (csrc/sba_lm.hpp) compiled for CPU with a callback evaluator."""
    global _harness
    if _harness is None:
        so = ROOT / "tests" / "harness" / "liblm_harness.so"
        src = ROOT / "tests" / "harness" / "lm_harness.cpp"
        hdr = ROOT / "spherical_bundle_adjuster_amd" / "csrc" / "sba_lm.hpp"
        if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(so), str(src)], check=True)
        _harness = C.CDLL(str(so))
    return _harness


EVAL_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def harness_solve(mode, rot, tran, evaluator, **opt_overrides):
    """evaluator(rot(3,), tran(3,)) -> pack(24,).  Returns rot, tran, summary(ctypes), rc."""
    from spherical_bundle_adjuster_amd import _cabi as cabi
    h = lm_harness()
    o = cabi.LmOptions()
    h.harness_default_options(C.byref(o))
    for k, v in opt_overrides.items():
        setattr(o, k, v)
    rot = np.array(rot, dtype=np.float64)
    tran = np.array(tran, dtype=np.float64)

    def _cb(r, t, pack, _u):
        try:
            out = evaluator(np.array([r[0], r[1], r[2]]), np.array([t[0], t[1], t[2]]))
            for i in range(24):
                pack[i] = float(out[i])
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return -1
    cb = EVAL_CB(_cb)
    s = cabi.LmSummary()
    h.harness_lm_solve.restype = C.c_int
    rc = h.harness_lm_solve(C.c_int(mode), rot.ctypes.data_as(C.POINTER(C.c_double)),
                            tran.ctypes.data_as(C.POINTER(C.c_double)), C.byref(o), cb, None, C.byref(s))
    return rot, tran, s, rc
