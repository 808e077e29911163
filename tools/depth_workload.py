#!/usr/bin/env python3
"""The d-only stage alone on 10^7 matches (start d = 5: 27 iterations, no contraction) -- workload of the PMC A/B of
depth_step_kernel's candidate stores (tools/profile_depth_stores.sh; SBA_DEPTH_NT_STORES=0/1 selects plain / nt)."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
c = synthetic.full_rt(n)
with api.Problem(0) as p:
    p.upload(c.x1, c.x2, np.full((n, 2), 5.0))
    p.solve_depths(c.rot_true, c.tran_true)
    p.set_depths(np.full((n, 2), 5.0))
    d, s = p.solve_depths(c.rot_true, c.tran_true)
print(json.dumps({"nt_stores": os.environ.get("SBA_DEPTH_NT_STORES", "default"), "n": n, "iterations": s.num_iterations,
                  "passes": s.num_evaluations, "ms_total": s.seconds_total * 1e3,
                  "us_per_pass": s.seconds_total / max(s.num_evaluations, 1) * 1e6, "final_cost": s.final_cost}))
