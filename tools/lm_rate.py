#!/usr/bin/env python3
"""Per-evaluation cost inside a real LM solve against the tight host-synchronous step loop, for the two final-reduction
modes (SBA_FUSED = 0: sweep + finalize kernel, 1: last-arriving block folds inside the sweep), at mid-size problems.
Usage: python tools/lm_rate.py [n ...]"""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [20_000, 100_000, 1_000_000, 3_000_000]
for n in sizes:
    c = synthetic.rotation_only(n, seed=synthetic.BASE_SEED + 1)
    for fused in ("0", "1"):
        os.environ["SBA_FUSED"] = fused
        with api.Problem(0) as p:
            p.upload(c.x1, c.x2)
            p.eval_launch_times(api.MODE_ROT, c.rot_init, c.tran_init, repeat=300)
            _, sec = p.eval_steps(api.MODE_ROT, c.rot_init, c.tran_init, steps=300)
            _, sec = p.eval_steps(api.MODE_ROT, c.rot_init, c.tran_init, steps=300)
            per_eval = []
            for _ in range(30):
                r, t, s = p.solve(api.MODE_ROT, c.rot_init, c.tran_init)
                per_eval.append(s.seconds_total / s.num_evaluations * 1e6)
            print(json.dumps({"n": n, "fused": int(fused), "step_us": sec / 300 * 1e6, "lm_us_per_eval_median": float(np.median(per_eval)),
                              "lm_us_per_eval_min": float(np.min(per_eval)), "lm_us_per_eval_max": float(np.max(per_eval)),
                              "evals_per_solve": s.num_evaluations}), flush=True)
