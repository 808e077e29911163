#!/usr/bin/env python3
"""Thread 0's clock inside batch_guess_kernel (library built with -DSBA_GUESS_PROFILE, SBA_LIBRARY_PATH) next to the wall
clock of the call, for several trial counts.  Usage: python tools/guess_profile.py [pairs] [matches]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
cs = [synthetic.full_rt(n, seed=7000 + g) for g in range(B)]
off = (np.arange(B + 1) * n).astype(np.uint64)
x1, x2 = (np.concatenate([getattr(c, k) for c in cs]) for k in ("x1", "x2"))
with api.Batch(0) as b:
    b.upload(x1, x2, off, np.ones((B * n, 2)))
    print("us: staging+occupancy, trials, collect, consensus, pick | wall clock of the call | candidates")
    for trials in (1, 8, 40, 80, 128):
        b.initial_guess(trials, 0.25, 1, check=False)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            e, t, nc, st = b.initial_guess(trials, 0.25, 1, check=False)
            ts.append(time.perf_counter() - t0)
        print(trials, "trials:", np.round(np.median(np.c_[e, t[:, :2]], axis=0) / 100.0, 1), "|", round(float(np.median(ts)) * 1e6, 1), "|", int(nc.min()), int(nc.max()),
              flush=True)
