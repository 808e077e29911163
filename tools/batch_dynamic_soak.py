#!/usr/bin/env python3
"""Soak of the batched stages with hand-over and dynamic shares: many batches of random shape (few long pairs ... many short
ones, ragged, with empty pairs, f64 / f32 planes), random caps of the first launch, poor and good starts.  For every batch:
  * the d-only stage with a random pass cap against the one-launch kernel run to the end (cap 0),
  * the rot-only LM with a random sweep cap / dynamic from the first sweep against the one-launch kernel (or, with several blocks
    per pair, the host lock-step loop),
  * the whole pipeline against the chain of its stages,
  * the device trials of the initial guess against the host trials.
Every call must return (bounded waits), every well-posed pair must take the same numbers of iterations / passes and end at the same
point to rounding.  python tools/batch_dynamic_soak.py [batches=60]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

os.environ.setdefault("SBA_WAIT_TIMEOUT_S", "20")
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(7)
t0 = time.time()
stats = {"batches": 0, "pairs": 0, "depth_count_mismatch": 0, "lm_count_mismatch": 0, "pipeline_mismatch": 0, "guess_mismatch": 0,
         "worst_depth_diff": 0.0, "worst_rot_diff": 0.0}


def counts(sums):
    return [(q.num_iterations, q.num_successful_steps, q.num_line_search_steps, q.num_evaluations, q.termination) for q in sums]


def env(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


for it in range(batches):
    shape = int(rng.integers(0, 3))
    if shape == 0:
        B = int(rng.integers(2, 9)); sizes = rng.integers(20_000, 120_000, size=B)          # few long pairs: several blocks per pair
    elif shape == 1:
        B = int(rng.integers(256, 330)); sizes = rng.integers(60, 900, size=B)                # many short pairs: one block per pair
    else:
        B = int(rng.integers(40, 200)); sizes = rng.integers(300, 6000, size=B)
    if B > 4:
        sizes[rng.integers(0, B, size=2)] = 0
    else:
        sizes[int(rng.integers(0, B))] = 0 if B > 2 else sizes[0]
    cs = [synthetic.full_rt(int(n), seed=90_000 + 1000 * it + g, sigma=2e-4, outlier_fraction=0.02) for g, n in enumerate(sizes)]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    x1 = np.concatenate([c.x1 for c in cs]); x2 = np.concatenate([c.x2 for c in cs])
    d0 = np.full((int(off[-1]), 2), float(rng.choice([1.5, 3.0, 6.0])))
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    rot0[:: int(rng.integers(2, 5))] += 0.2                                                   # some poor starts: long LM tails
    store = int(rng.integers(0, 2))
    ok = sizes >= 60
    with api.Batch(0) as b:
        b.upload(x1, x2, off, d0, store=store)
        # d-only: one launch to the end against a random cap
        env(SBA_BATCH_DEPTH_FIRST_PASSES=0)
        dA, sA, stA = b.solve_depths(rot0, tran0)
        b.set_depths(d0)
        env(SBA_BATCH_DEPTH_FIRST_PASSES=int(rng.integers(1, 30)))
        dB, sB, stB = b.solve_depths(rot0, tran0)
        env(SBA_BATCH_DEPTH_FIRST_PASSES=None)
        cA, cB = counts(sA), counts(sB)
        stats["depth_count_mismatch"] += sum(1 for g in range(B) if ok[g] and (cA[g] != cB[g] or stA[g] != stB[g]))
        stats["worst_depth_diff"] = max(stats["worst_depth_diff"], float(np.abs(dA - dB).max() / max(1.0, np.abs(dA).max())) if dA.size else 0.0)
        # rot-only LM from the refined depths' first two
        first = off[:-1].astype(int)
        du1 = np.where(sizes > 0, dA[np.minimum(first, len(dA) - 1), 0], 1.0); du2 = np.where(sizes > 1, dA[np.minimum(first + 1, len(dA) - 1), 0], du1)
        env(SBA_BATCH_DYNAMIC=0)
        rA = b.solve(api.MODE_ROT, rot0, tran0, du1, du2)
        if rng.integers(0, 2):
            env(SBA_BATCH_DYNAMIC=1)
        else:
            env(SBA_BATCH_DYNAMIC=None, SBA_BATCH_LM_FIRST_SWEEPS=int(rng.integers(1, 12)))
        rB = b.solve(api.MODE_ROT, rot0, tran0, du1, du2)
        env(SBA_BATCH_DYNAMIC=None, SBA_BATCH_LM_FIRST_SWEEPS=None)
        cA, cB = counts(rA[2]), counts(rB[2])
        stats["lm_count_mismatch"] += sum(1 for g in range(B) if ok[g] and (cA[g][:2] != cB[g][:2] or rA[3][g] != rB[3][g]))
        good = ok & (rA[3] == 0)
        if good.any():
            conv = np.array([q.termination.startswith("CONVERGENCE") for q in rA[2]])      # a solve that stops at the iteration limit
            diff = np.where(good & conv, np.abs(rA[0] - rB[0]).max(axis=1), 0.0)               # amplifies last-bit differences 50 times over
            stats["lm_not_converged"] = stats.get("lm_not_converged", 0) + int((good & ~conv).sum())
            g = int(np.argmax(diff))
            if diff[g] > stats["worst_rot_diff"]:
                stats["worst_rot_diff"] = float(diff[g])
                if diff[g] > 1e-9:      # a pair worth a look: its size, counts, final costs and gradient norms both ways
                    print("  note: batch", it, "pair", g, "n", int(sizes[g]), "rot diff", float(diff[g]), cA[g], cB[g],
                          "final cost", rA[2][g].final_cost, rB[2][g].final_cost, "gradient", rA[2][g].final_gradient_max_norm,
                          rB[2][g].final_gradient_max_norm, flush=True)
        # the whole pipeline against the chain of its stages (same drivers, defaults)
        b.set_depths(d0)
        e, t, nc, stg = b.initial_guess(80, 0.25, it, check=False)
        env(SBA_BATCH_DEVICE_GUESS=0)
        eh, th, nch, sth = b.initial_guess(80, 0.25, it, check=False)
        env(SBA_BATCH_DEVICE_GUESS=None)
        stats["guess_mismatch"] += int((nc != nch).sum() + (stg != sth).sum() + (np.abs(e - eh).max(axis=1) > 2.5e-7).sum())
        res = b.solve_problem(seed=it, check=False)
        b.set_depths(d0)
        r_in = np.where((stg == 0)[:, None], -e, 0.0); t_in = np.where((stg == 0)[:, None], t, 0.0)
        d, sd, _ = b.solve_depths(r_in, t_in)
        u1 = np.where(sizes > 0, d[np.minimum(first, len(d) - 1), 0], 0.0); u2 = np.where(sizes > 1, d[np.minimum(first + 1, len(d) - 1), 0], u1)
        r1, t1, s1, _ = b.solve(api.MODE_ROT, r_in, t_in, u1, u2)
        r2, t2, s2, _ = b.solve(api.MODE_TRAN, r1, t1, u1, u2)
        good = ok & (res["status"] == 0)
        stats["pipeline_mismatch"] += int((np.abs(res["rot"][good] - r2[good]).max(axis=1) > 1e-9).sum()) if good.any() else 0
    stats["batches"] += 1; stats["pairs"] += B
    if it % 10 == 9:
        print(f"[{time.time() - t0:6.1f} s]", stats, flush=True)
print("done:", stats)
