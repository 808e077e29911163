#!/usr/bin/env python3
"""Few long pairs (several blocks per pair): per-pair LM through the host lock-step loop (default for blocks_per_pair > 1) against
the device-resident launches with dynamic shares (SBA_BATCH_DYNAMIC=1).  Usage: python tools/few_pairs_lm_ab.py"""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

for B, n in ((4, 1_000_000), (16, 250_000), (64, 100_000), (128, 50_000)):
    cs = [synthetic.full_rt(n, seed=9000 + g) for g in range(B)]
    off = (np.arange(B + 1) * n).astype(np.uint64)
    x1, x2, d12 = (np.concatenate([getattr(c, k) for c in cs]) for k in ("x1", "x2", "d12"))
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    rot0[::2] += 0.2                                         # half of the pairs start poorly: more iterations
    opt = api.default_lm_options(tran_param=api.TRAN_SPHERE)
    out = {"pairs": B, "matches": n}
    res = {}
    with api.Batch(0) as b:
        b.upload(x1, x2, off, d12)
        out["blocks_per_pair"] = b.blocks_per_pair
        for name, flag in (("host_lock_step", None), ("dynamic_shares", "1")):
            if flag is None:
                os.environ.pop("SBA_BATCH_DYNAMIC", None)
            else:
                os.environ["SBA_BATCH_DYNAMIC"] = flag
            b.solve(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH, options=opt)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                r, t, s, st = b.solve(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH, options=opt)
                ts.append(time.perf_counter() - t0)
            out[name + "_ms"] = float(np.median(ts)) * 1e3
            res[name] = (r, [q.num_iterations for q in s])
    os.environ.pop("SBA_BATCH_DYNAMIC", None)
    out["iterations_min_max"] = [min(res["host_lock_step"][1]), max(res["host_lock_step"][1])]
    out["counts_equal"] = res["host_lock_step"][1] == res["dynamic_shares"][1]
    out["max_abs_rot_difference"] = float(np.abs(res["host_lock_step"][0] - res["dynamic_shares"][0]).max())
    print(json.dumps(out), flush=True)
