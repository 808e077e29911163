#!/usr/bin/env python3
"""Upload of 10^7 correspondences (640 MB of host arrays -> device planes): the pipelined pinned-staging path against the
single-buffer path (SBA_UPLOAD_PIPELINE=0), fresh host arrays every time (a reused array is pinned by the runtime after
its first copy).  Usage: python tools/upload_probe.py [n]"""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from spherical_bundle_adjuster_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(1)
base = rng.standard_normal((n, 3))
for mode, env in (("single-buffer", {"SBA_UPLOAD_PIPELINE": "0"}), ("pipelined 1 thread", {"SBA_UPLOAD_THREADS": "1"}),
                  ("pipelined 3 threads", {"SBA_UPLOAD_THREADS": "3"}), ("pipelined 6 threads", {}), ("pipelined 12 threads", {"SBA_UPLOAD_THREADS": "12"})):
    for k in ("SBA_UPLOAD_PIPELINE", "SBA_UPLOAD_THREADS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    times = []
    with api.Problem(0) as p:
        for rep in range(4):
            x1, x2, d12 = base + rep, base - rep, np.abs(base[:, :2]) + 1.0 + rep       # fresh pageable arrays
            t0 = time.perf_counter()
            p.upload(x1, x2, d12)
            times.append(time.perf_counter() - t0)
            del x1, x2, d12
    print(json.dumps({"mode": mode, "n": n, "upload_ms": [round(t * 1e3, 2) for t in times], "GBps_best": n * 64 / min(times) / 1e9}), flush=True)
