"""Soak of the one-launch batched paths: many batches of random shape (pairs, ragged sizes, modes), each stepped through
batch_step_kernel and solved by batch_lm_kernel; every solve must finish (bounded wait), converge or fail cleanly, and
agree with the host lock-step loop on the iteration counts of all well-posed pairs.  python tools/batch_soak.py [batches=150]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

os.environ.setdefault("SBA_WAIT_TIMEOUT_S", "20")
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(2)
t0 = time.time()
solves = pairs_total = mismatches = 0
for it in range(batches):
    B = int(rng.integers(256, 400))
    hi = int(rng.choice([64, 300, 512]))
    sizes = rng.integers(0, hi + 1, size=B)
    sizes[rng.integers(0, B, size=3)] = 0
    cs = [synthetic.full_rt(int(n), seed=50_000 + 1000 * it + g) for g, n in enumerate(sizes)]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    x1 = np.concatenate([c.x1 for c in cs]); x2 = np.concatenate([c.x2 for c in cs]); d12 = np.concatenate([c.d12 for c in cs])
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    with api.Batch(0) as b:
        b.upload(x1, x2, off, d12, store=int(rng.integers(0, 2)))
        assert b.blocks_per_pair == 1 and b.step_is_fused
        for mode, tp in ((api.MODE_RT, api.TRAN_SPHERE), (api.MODE_ROT, api.TRAN_FREE), (api.MODE_TRAN, api.TRAN_FREE)):
            opt = api.default_lm_options(tran_param=tp)
            p = b.eval(mode, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH)
            assert np.isfinite(p).all()
            dev = b.solve(mode, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH, options=opt)
            os.environ["SBA_BATCH_DEVICE_LM"] = "0"
            host = b.solve(mode, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH, options=opt)
            del os.environ["SBA_BATCH_DEVICE_LM"]
            solves += 1; pairs_total += B
            for g in range(B):
                if sizes[g] >= 50 and (dev[2][g].num_iterations != host[2][g].num_iterations or dev[3][g] != host[3][g]):
                    mismatches += 1
    if it % 10 == 9:
        print(f"[{time.time() - t0:6.1f} s] {it + 1} batches, {solves} device solves over {pairs_total} pairs, "
              f"{mismatches} pairs (n >= 50) whose iteration count differs from the host loop", flush=True)
print("done:", batches, "batches,", solves, "solves,", pairs_total, "pairs,", mismatches, "count mismatches")
