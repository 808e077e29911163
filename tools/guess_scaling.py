#!/usr/bin/env python3
"""Wall clock of sba_batch_initial_guess against the number of pairs (small pairs: the moments pass is negligible) -- does the
trial / consensus kernel run all its blocks at once?  Usage: python tools/guess_scaling.py"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

n = 2000
for B in (1, 16, 32, 64, 96, 128, 192, 256, 512):
    cs = [synthetic.full_rt(n, seed=7000 + g) for g in range(B)]
    off = (np.arange(B + 1) * n).astype(np.uint64)
    x1, x2 = (np.concatenate([getattr(c, k) for c in cs]) for k in ("x1", "x2"))
    with api.Batch(0) as b:
        b.upload(x1, x2, off, np.ones((B * n, 2)))
        b.initial_guess(80, 0.25, 1, check=False)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            e, t, nc, st = b.initial_guess(80, 0.25, 1, check=False)
            ts.append(time.perf_counter() - t0)
        print(B, "pairs:", round(float(np.median(ts)) * 1e6, 1), "us; candidates", int(nc.min()), int(nc.max()), flush=True)
