#!/usr/bin/env python3
"""Where do the first solve calls after a batch's first upload spend their time?  Times the raw C call (ctypes) of
sba_batch_solve next to the library's own clock, call by call."""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from spherical_bundle_adjuster_amd import _cabi as cabi  # noqa: E402
from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

B, n = 256, 50000
cs = [synthetic.full_rt(n, seed=7000 + g) for g in range(B)]
off = (np.arange(B + 1) * n).astype(np.uint64)
x1, x2, d12 = (np.concatenate([getattr(c, k) for c in cs]) for k in ("x1", "x2", "d12"))
rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
opt = api.default_lm_options(tran_param=api.TRAN_SPHERE)
lib = cabi.load_library()


def raw(b, label, calls=4):
    out = []
    for _ in range(calls):
        rot, tran = rot0.copy(), tran0.copy()
        sums = (cabi.LmSummary * B)()
        status = np.zeros(B, dtype=np.int32)
        t0 = time.perf_counter()
        rc = lib.sba_batch_solve(b._h, api.MODE_RT, api.DEPTH_PER_MATCH, rot.ctypes.data_as(C.POINTER(C.c_double)),
                                 tran.ctypes.data_as(C.POINTER(C.c_double)), None, None, C.byref(opt), sums,
                                 status.ctypes.data_as(C.POINTER(C.c_int)))
        out.append((round((time.perf_counter() - t0) * 1e3, 2), round(sums[0].seconds_total * 1e3, 2), rc))
    print(label, out, flush=True)


with api.Batch(0) as b:
    b.upload(x1, x2, off, d12)
    raw(b, "after the first upload (C call ms, library's own ms, rc):")
    b.upload(x1, x2, off, d12)
    raw(b, "after a re-upload:")
with api.Batch(0) as b:
    b.upload(x1, x2, off, d12)
    t0 = time.perf_counter(); b.eval(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH); print("first eval ms", (time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); b.eval(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH); print("second eval ms", (time.perf_counter() - t0) * 1e3)
    raw(b, "fresh batch, after two evals:")
