#!/usr/bin/env python3
"""Batched d-only stage at C5: the one-launch kernel to the end (SBA_BATCH_DEPTH_FIRST_PASSES=0) against the hybrid (first N
passes in one launch, then passes with dynamic shares), wall clock of sba_batch_solve_depths, depths staying on the device.
Usage: python tools/depth_hybrid_ab.py [pairs] [matches] [start depth]"""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
d_start = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
cs = [synthetic.full_rt(n, seed=7000 + g) for g in range(B)]
off = (np.arange(B + 1) * n).astype(np.uint64)
x1, x2 = (np.concatenate([getattr(c, k) for c in cs]) for k in ("x1", "x2"))
d0 = np.full((B * n, 2), d_start)
rot = np.stack([c.rot_init for c in cs]); tran = np.stack([c.tran_init for c in cs])
ref = None
with api.Batch(0) as b:
    b.upload(x1, x2, off, d0)
    for first in ("0", "4", "8", "12", "16", "20", "24"):
        os.environ["SBA_BATCH_DEPTH_FIRST_PASSES"] = first
        b.set_depths(d0); b.solve_depths(rot, tran, want_depths=False)
        ts = []
        for _ in range(3):
            b.set_depths(d0)
            t0 = time.perf_counter()
            _, sums, status = b.solve_depths(rot, tran, want_depths=False)
            ts.append(time.perf_counter() - t0)
        b.set_depths(d0)
        d, sums, status = b.solve_depths(rot, tran)
        counts = [(q.num_iterations, q.num_line_search_steps, q.num_evaluations) for q in sums]
        if ref is None:
            ref = (d, counts)
        print(json.dumps({"first_passes": int(first), "ms": float(np.median(ts)) * 1e3, "counts_equal_one_launch": counts == ref[1],
                          "max_abs_depth_difference": float(np.abs(d - ref[0]).max()), "all_ok": bool((status == 0).all()),
                          "passes_min_max": [min(c[2] for c in counts), max(c[2] for c in counts)]}), flush=True)
