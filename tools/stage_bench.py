#!/usr/bin/env python3
"""Timing of the non-sweep stages at full size: d-only stage (per LM iteration), 8-point group moments, upload."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
c = synthetic.full_rt(n)
with api.Problem(0) as p:
    t0 = time.perf_counter(); p.upload(c.x1, c.x2, np.full((n, 2), 5.0)); up = time.perf_counter() - t0
    p.epipolar_moments()
    t0 = time.perf_counter()
    for _ in range(5):
        g = p.epipolar_moments()
    mom = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter(); e, t, ncand = p.initial_guess(80, 0.25, 0); guess = time.perf_counter() - t0
    p.solve_depths(c.rot_true, c.tran_true)            # warm-up (first-use costs: occupancy query, allocator)
    p.set_depths(np.full((n, 2), 5.0))
    d, s = p.solve_depths(c.rot_true, c.tran_true)
    per_iter = s.seconds_total / max(s.num_evaluations, 1)
    print(f"n={n}: upload {up*1e3:.1f} ms; 8-point moments {mom*1e6:.0f} us ({n*48/mom/1e9:.0f} GB/s of 48 B/match, incl. D2H+alloc); "
          f"full initial guess {guess*1e3:.2f} ms ({ncand} candidates); d-only stage {s.num_iterations} iterations, "
          f"{per_iter*1e6:.0f} us per device pass ({n*112/per_iter/1e9:.0f} GB/s of 112 B/match), total {s.seconds_total*1e3:.1f} ms, {s.termination}")
