#!/usr/bin/env python3
"""Sweep-kernel time vs N (fit t = a + N*bytes/BW): separates launch ramp/tail from steady-state bandwidth."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

ns = [250_000, 500_000, 1_000_000, 2_000_000, 5_000_000, 10_000_000, 20_000_000, 40_000_000]
c = synthetic.full_rt(max(ns))
rows = []
for n in ns:
    with api.Problem(0) as p:
        p.upload(c.x1[:n], c.x2[:n], c.d12[:n])
        best = 1e9
        for _ in range(4):
            _, step, sweep = p.eval_timed(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH, repeat=30)
            best = min(best, sweep)
        rows.append((n, best * 1e3))
        print(f"n={n:>9d} sweep {best*1e3:8.2f} us  {n*64/best/1e6:7.0f} GB/s", flush=True)
x = np.array([r[0] * 64.0 for r in rows[3:]]); y = np.array([r[1] for r in rows[3:]])
b, a = np.polyfit(x, y, 1)
print(f"fit over n >= 2M: fixed {a:.2f} us, steady-state {1.0/b/1e6:.0f} GB/s")
