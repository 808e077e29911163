#!/usr/bin/env python3
"""Config C5 (256 pairs x 50 000 matches) through sba_batch_solve_problem -- the reference's per-pair pipeline for every pair --
`reps` times, for rocprofv3 (--kernel-trace --stats): batch_epipolar_moments_kernel, batch_guess_kernel,
batch_depth_solve_kernel, batch_lm_kernel (rot-only, tran-only).  Prints one JSON line with the wall clock per stage.
Usage: python tools/batch_pipeline_workload.py [pairs] [matches] [reps]"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cs = [synthetic.full_rt(n, seed=7000 + g) for g in range(B)]
off = (np.arange(B + 1) * n).astype(np.uint64)
x1, x2 = (np.concatenate([getattr(c, k) for c in cs]) for k in ("x1", "x2"))
d0 = np.full((B * n, 2), 6.0)
with api.Batch(0) as b:
    b.upload(x1, x2, off, d0)
    b.solve_problem(seed=1, check=False)
    stage = {"guess": [], "d_only": [], "rot_only": [], "tran_only": [], "pipeline": []}
    for _ in range(reps):
        b.upload(x1, x2, off, d0)
        t0 = time.perf_counter()
        e, t, nc, st = b.initial_guess(80, 0.25, 1, check=False)
        t1 = time.perf_counter()
        rot, tran = -e, t
        d, sd, _ = b.solve_depths(rot, tran, want_depths=True)
        t2 = time.perf_counter()
        du1, du2 = d[off[:-1].astype(int), 0], d[off[:-1].astype(int) + 1, 0]
        t2b = time.perf_counter()
        r1, tr1, s1, _ = b.solve(api.MODE_ROT, rot, tran, du1, du2, depth_mode=api.DEPTH_UNIFORM)
        t3 = time.perf_counter()
        r2, tr2, s2, _ = b.solve(api.MODE_TRAN, r1, tr1, du1, du2, depth_mode=api.DEPTH_UNIFORM)
        t4 = time.perf_counter()
        b.upload(x1, x2, off, d0)
        t5 = time.perf_counter()
        res = b.solve_problem(seed=1, check=False)
        t6 = time.perf_counter()
        for k, v in zip(stage, (t1 - t0, t2 - t1, t3 - t2b, t4 - t3, t6 - t5)):
            stage[k].append(v)
        assert np.array_equal(res["rot"], r2) and np.array_equal(res["tran"], tr2)
    import os
    dyn = {}
    for name, flag in (("one_launch", "0"), ("dynamic_shares", "1"), ("hybrid_default", None)):
        if flag is None:
            os.environ.pop("SBA_BATCH_DYNAMIC", None)
        else:
            os.environ["SBA_BATCH_DYNAMIC"] = flag
        out = {}
        for stage, mode, r_in, t_in in (("rot_only", api.MODE_ROT, rot, tran), ("tran_only", api.MODE_TRAN, r1, tr1)):
            b.solve(mode, r_in, t_in, du1, du2, depth_mode=api.DEPTH_UNIFORM)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                rr, tt, ss, _ = b.solve(mode, r_in, t_in, du1, du2, depth_mode=api.DEPTH_UNIFORM)
                ts.append(time.perf_counter() - t0)
            out[stage] = (float(np.median(ts)) * 1e3, rr, tt, [q.num_iterations for q in ss])
        dyn[name] = out
    os.environ.pop("SBA_BATCH_DYNAMIC", None)
    for stage in ("rot_only", "tran_only"):
        assert dyn["one_launch"][stage][3] == dyn["dynamic_shares"][stage][3] == dyn["hybrid_default"][stage][3], "iteration counts differ"
    print(json.dumps({f"{stage}_{name}_ms": dyn[name][stage][0] for stage in ("rot_only", "tran_only") for name in dyn} |
                     {"max_abs_difference_to_one_launch": float(max(np.abs(dyn["one_launch"][st][k] - dyn[nm][st][k]).max()
                                                                   for st in ("rot_only", "tran_only") for nm in ("dynamic_shares", "hybrid_default") for k in (1, 2)))}))
    print(json.dumps({"pairs": B, "matches": n, "reps": reps, **{k + "_ms": float(np.median(v)) * 1e3 for k, v in stage.items()},
                      "d_only_iterations_min_max": [min(q.num_iterations for q in sd), max(q.num_iterations for q in sd)],
                      "d_only_pair_passes": int(sum(q.num_evaluations for q in sd)),
                      "rot_iterations_max": max(q.num_iterations for q in s1), "tran_iterations_max": max(q.num_iterations for q in s2),
                      "note": "d_only_ms includes the read-back of all refined depths (want_depths); the pipeline keeps them on the device"}))
