#!/usr/bin/env python3
"""Soak of the one-launch d-only stage of small single problems (the default up to 2 560 matches: the problem as a batch of one pair
through batch_depth_solve_kernel) against the resident evaluator (SBA_SMALL_ONE_LAUNCH=0): random sizes 1 ... 2 560, start depths,
noise levels, f64 / f32 planes; counts must agree, depths to 1e-9.  python tools/small_depth_soak.py [problems=400]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

os.environ.setdefault("SBA_WAIT_TIMEOUT_S", "20")
problems = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(3)
t0 = time.time()
mismatch = 0
worst = 0.0
with api.Problem(0) as p:
    for it in range(problems):
        n = int(rng.choice([rng.integers(1, 64), rng.integers(64, 600), rng.integers(600, 2561)]))
        c = synthetic.full_rt(n, seed=70_000 + it, sigma=float(rng.choice([1e-4, 1e-3, 5e-3])), outlier_fraction=float(rng.choice([0.0, 0.05])))
        start = np.full((n, 2), float(rng.choice([0.5, 1.5, 4.0, 9.0])))
        store = int(rng.integers(0, 2))
        res = []
        for flag in (None, "0"):
            if flag is None:
                os.environ.pop("SBA_SMALL_ONE_LAUNCH", None)
            else:
                os.environ["SBA_SMALL_ONE_LAUNCH"] = flag
            p.upload(c.x1, c.x2, start, store=store)
            d, s = p.solve_depths(c.rot_init, c.tran_init)
            res.append((d, (s.num_iterations, s.num_successful_steps, s.num_line_search_steps, s.num_evaluations, s.termination)))
        os.environ.pop("SBA_SMALL_ONE_LAUNCH", None)
        if res[0][1] != res[1][1]:
            mismatch += 1
            print("  count mismatch: problem", it, "n", n, res[0][1], res[1][1], flush=True)
        worst = max(worst, float(np.abs(res[0][0] - res[1][0]).max() / max(1.0, np.abs(res[1][0]).max())))
        if it % 100 == 99:
            print(f"[{time.time() - t0:6.1f} s] {it + 1} problems, {mismatch} count mismatches, worst relative depth difference {worst:.3e}", flush=True)
print("done:", problems, "problems,", mismatch, "count mismatches, worst relative depth difference", worst)
