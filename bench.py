#!/usr/bin/env python3
"""Benchmark of the spherical-BA hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): residual+Jacobian evaluations per second.  Workload at N = 1: BASELINE config
C3, "10M synthetic correspondences, full R|t (5-DoF) LM BA, 1xMI355X" -- 10^7 unit-sphere
correspondences with per-match depths (64 algorithmic bytes per evaluation: 2 x 3 f64 unit-vector
components + 2 f64 depths), f64 arithmetic.  At N > 1: one GPU's shard of BASELINE config C4 ("100M
correspondences sharded 8xMI355X") = 12.5M of the same correspondences per GPU, one RCCL all-reduce of the
24-double pack per step (the collective north_star names; `--transport peer` selects the direct xGMI exchange).
At N = 1 the line also carries `c1`: BASELINE config C1 (the reference's real workload: 2 048 matches, initial guess + d-only +
rot-only + tran-only end to end, GPU beside the oracle's pipeline on the host cores; `--no-c1-leg`, which also drops `c2`:
BASELINE config C2, 1M rotation-only), `c5`: BASELINE config C5 (256 pairs x 50k: batched step, per-pair LM, 512-frame remap) from a
child run of `--workload c5` after the timed region (`--no-c5-leg` skips it), and `stages`: the 8-point initial guess and the
bounded d-only stage of `solve_problem` on the resident problem (`--no-stage-leg`).
At N > 1, only on request (`--peer-trial` / SBA_BENCH_PEER_TRIAL=1), the line also carries `peer_trial`: after the
quoted measurement -- whose line is then first written to stderr, so it survives whatever the trial does -- the same K
steps once more over the direct peer exchange: a second figure, never the quoted `value`.  Off by default until the
peer path has one cross-device run behind it.
A *step* is one pass of the hot path over the resident correspondences exactly as one LM iteration needs it: sweep
kernel (residual + analytic Jacobian + Huber + reduction), finalize kernel, the all-reduce when N > 1, and the pack
published to and awaited by the host.  Weak scaling: per-GPU work is fixed as N grows, no data-path communication
other than that one exchange per step.

Clock ramp: on this pool a GPU coming out of idle runs the first ~15 launches (1.5 ms) at full speed, then drops by
~10 % for some milliseconds before it settles (profiles/r02_ramp.json) -- and the driver's command line (5 warm-up +
20 timed steps = 2.8 ms) falls exactly into that dip.  So an UN-TIMED, DISCLOSED pre-conditioning phase
(`--precondition-ms`, default 60 ms of the same sweeps) runs before the W warm-up steps; W and K are what the command
line says and nothing inside the timed region changes.  The spread is in the line: `roofline.kernel_ms_*` are per-launch
minima / medians / maxima next to the mean.

One JSON line on rank 0.  `roofline` prices the sweep kernel alone against HBM (device time from HIP
events recorded on the problem's stream); `cpu_baseline` times the oracle's
faithful dual-number loop on this box's host cores on a bounded sample (reported, not the target).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# the pool's host driver only supports dmabuf IPC (RCCL and HIP IPC between ranks need it); set before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--matches", "--n", dest="n", type=int, default=0, help="correspondences per GPU; default: 10M at "
                    "N = 1 (config C3), 12.5M at N > 1 (one GPU's shard of config C4) "
                    "(use --matches under torch.distributed.run: its own parser trips over --n)")
    ap.add_argument("--precondition-ms", type=float, default=60.0,
                    help="un-timed sweeps of the same kernel before the warm-up steps, to get past the clock dip that "
                         "follows the first ~1.5 ms of load (0 = off); disclosed in the JSON line")
    ap.add_argument("--workload", choices=["rt", "rot", "c5"], default="rt",
                    help="rt = config C3 / C4 shard (full R|t, per-match depths); rot = config C2 shape (rotation-only); "
                         "c5 = config C5 (256 ERP pairs x 50k matches per GPU, one batched launch per step, per-pair LM, "
                         "equi2cube of 512 frames)")
    ap.add_argument("--pairs", type=int, default=256, help="c5: ERP pairs per GPU")
    ap.add_argument("--pair-matches", type=int, default=50_000, help="c5: matches per pair")
    ap.add_argument("--frames", type=int, default=512, help="c5: 3840x1920 ERP frames remapped to S = 600 cube strips")
    ap.add_argument("--store", choices=["f64", "f32"], default="f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--transport", choices=["auto", "peer", "rccl", "hook"], default="auto",
                    help="N > 1: exchange of the 24-double pack per step.  auto (default) = rccl (ncclAllReduce issued by "
                         "the shim: what north_star names) and, only if RCCL cannot be set up on every rank, the "
                         "torch.distributed hook; rccl = RCCL or fail; peer = direct xGMI exchange (opt-in)")
    ap.add_argument("--kernel", choices=["factored", "explicit"], default="factored")
    ap.add_argument("--no-stage-leg", action="store_true",
                    help="N = 1: skip the initial-guess / d-only-stage figures (`stages` in the line)")
    ap.add_argument("--no-c1-leg", action="store_true",
                    help="N = 1: skip the config-C1 figures (`c1` in the line: the reference's real workload end to end)")
    ap.add_argument("--no-c5-leg", action="store_true",
                    help="N = 1: skip the config-C5 figures (`c5` in the line: a child run of --workload c5)")
    ap.add_argument("--peer-trial", action="store_true",
                    help="N > 1: after the quoted measurement take a second, un-quoted one over the direct xGMI peer "
                         "exchange (opt-in; also SBA_BENCH_PEER_TRIAL=1)")
    ap.add_argument("--no-peer-trial", action="store_true", help="(accepted for older command lines; the trial is off by default)")
    ap.add_argument("--no-cold-leg", action="store_true",
                    help="skip the W + K steps taken BEFORE the pre-conditioning (`cold` in the line)")
    ap.add_argument("--no-scaling-reference", action="store_true",
                    help="N = 1: skip the K steps at 12.5M correspondences (`scaling_reference`: the per-GPU size of N > 1)")
    return ap.parse_args()


def usable_cores() -> int:
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def stream_probe(n):
    """The no-arithmetic streaming ceiling of the sweep kernel's access pattern (tools/stream_probe.hip: the same
    eight f64 planes read once, 16 B per lane, grid-stride, register prefetch, nt loads), measured in this run on this
    box as a child process.  Context for roofline.frac; never part of `value`."""
    import subprocess
    exe = ROOT / "spherical_bundle_adjuster_amd" / "csrc" / "build" / "stream_probe"
    if not exe.exists():
        return None
    try:
        r = subprocess.run([str(exe), str(n), "30", "quick"], capture_output=True, text=True, timeout=120)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(line[-1]) if r.returncode == 0 and line else None
    except Exception:
        return None


def cpu_baseline(c, mode, per_match, sample, seconds):
    """Oracle (kind "port"): faithful per-match dual-number functor + per-match trig + Huber corrector, J^T J / J^T e
    accumulated in double (orc_eval_f64 -- the long-double accumulation of the checker is not what is timed), OpenMP
    over the usable host cores, on the first `sample` correspondences of the same workload."""
    from oracle import oracle_py as orc
    n = min(sample, c.x1.shape[0])
    x1, x2 = c.x1[:n], c.x2[:n]
    d12 = c.d12[:n] if per_match else None
    cores = min(orc.num_procs(), usable_cores())
    orc.evaluate_f64(mode, x1[:10000], x2[:10000], c.rot_init, c.tran_init, d12=None if d12 is None else d12[:10000], threads=cores)
    passes, t0 = 0, time.perf_counter()
    while True:
        orc.evaluate_f64(mode, x1, x2, c.rot_init, c.tran_init, 1.0, 1.0, 1.0, d12, threads=cores)
        passes += 1
        el = time.perf_counter() - t0
        if el >= seconds or passes >= 50:
            break
    faithful = n * passes / el
    t0 = time.perf_counter()
    orc.evaluate_hoisted(mode, x1, x2, c.rot_init, c.tran_init, 1.0, 1.0, 1.0, d12, threads=cores)
    hoisted = n / (time.perf_counter() - t0)
    return {"value": faithful, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"{passes} passes over the first {n} correspondences of the workload, faithful "
                      f"dual-number loop with f64 accumulation (oracle/sba_oracle.cpp: evaluate_all_f64), {cores} OpenMP threads",
            "optimised_cpu_value": hoisted}


def pmc_traffic(kernel_sig: str, n: int):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes (profiles/pmc_latest.json, made by
    tools/profile_round.sh + tools/summarize_profiles.py with the gfx950 FETCH_SIZE x2 correction).
    PMC counters cannot be read from inside the timed process, so this is the profiled run of the SAME
    command; returns None when no matching profile is committed."""
    try:
        d = json.loads((ROOT / "profiles" / "pmc_latest.json").read_text())
        k = d["kernels"][kernel_sig]
        if d["bench_line_under_trace"]["config"]["correspondences_per_gpu"] != n:
            return None
        return k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"]
    except Exception:
        return None


def run_c5(a):
    """BASELINE config C5: 256 independent ERP pairs x 50k matches on each GPU.  A step = ONE batched pass over all pairs:
    per-pair sweep state prepared on the device, batch_sweep_kernel (every pair at its own R|t), fold + moment conversion
    + publication, packs awaited by the host.  Pairs are independent: N > 1 needs no collective."""
    import numpy as np
    import torch

    from spherical_bundle_adjuster_amd import _cabi as cabi
    from spherical_bundle_adjuster_amd import api, synthetic
    import ctypes as C

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    rehearsal = os.environ.get("SBA_BENCH_ONE_GPU") == "1"
    device_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)    # control only: the data path has no exchange

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    B, n = a.pairs, a.pair_matches
    cs = [synthetic.full_rt(n, seed=7000 + rank * B + g) for g in range(B)]
    off = (np.arange(B + 1) * n).astype(np.uint64)
    x1, x2, d12 = (np.concatenate([getattr(c, k) for c in cs]) for k in ("x1", "x2", "d12"))
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    kw = dict(depth_mode=api.DEPTH_PER_MATCH)
    with api.Batch(device_index) as b:
        t_up = time.perf_counter()
        b.upload(x1, x2, off, d12)
        upload_s = time.perf_counter() - t_up
        t_up = time.perf_counter()
        b.upload(x1, x2, off, d12)                  # same pairs again: every allocation is reused, only the data moves
        reupload_s = time.perf_counter() - t_up
        t_up = time.perf_counter()
        b.set_depths(d12)
        set_depths_s = time.perf_counter() - t_up
        barrier()
        precond = {"sweeps": 0, "ms": 0.0}
        if a.precondition_ms > 0:
            t_pc = time.perf_counter()
            while (time.perf_counter() - t_pc) * 1e3 < a.precondition_ms:
                b.sweep_launch_times(api.MODE_RT, rot0, tran0, repeat=50, **kw)
                precond["sweeps"] += 50
            precond["ms"] = (time.perf_counter() - t_pc) * 1e3
            barrier()
        if a.warmup > 0:
            b.eval_timed(api.MODE_RT, rot0, tran0, a.warmup, **kw)
        barrier()
        t0 = time.perf_counter()
        packs, split = b.eval_timed(api.MODE_RT, rot0, tran0, a.steps, **kw)
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # the dominant kernel of the step just timed: batch_step_kernel (the whole step in one launch) with one block per
        # pair, batch_sweep_kernel otherwise
        fused = b.step_is_fused
        per_launch = b.step_launch_times(api.MODE_RT, rot0, tran0, repeat=max(a.steps, 20), **kw)
        sweep_alone = b.sweep_launch_times(api.MODE_RT, rot0, tran0, repeat=max(a.steps, 20), **kw) if fused else per_launch
        # per-pair LM: all pairs solved in lock-step off the batched launches
        opt = api.default_lm_options(tran_param=api.TRAN_SPHERE)
        b.solve(api.MODE_RT, rot0, tran0, options=opt, **kw)
        lm_times = []
        for _ in range(5):          # median: the Python wrapper builds 256 summary objects per call, and one call in a few pays a GC pass
            t_lm = time.perf_counter()
            rot, tran, sums, status = b.solve(api.MODE_RT, rot0, tran0, options=opt, **kw)
            lm_times.append(time.perf_counter() - t_lm)
        lm_s = float(np.median(lm_times))
        bpp = b.blocks_per_pair
        # the d-only stage of every pair (the first stage of solve_problem, reference .cpp:196-197): one launch, one solver per
        # pair on the device (SBA_BATCH_DEVICE_DEPTH=0: host solvers in lock-step, one launch per pass)
        depth = None
        try:
            d0 = np.full((B * n, 2), 5.0)
            b.upload(x1, x2, off, d0)
            b.solve_depths(rot0, tran0, want_depths=False)               # first call: scratch allocations
            b.upload(x1, x2, off, d0)
            t_d = time.perf_counter()
            _, dsums, dstatus = b.solve_depths(rot0, tran0, want_depths=False)     # the refined depths stay on the device
            d_s = time.perf_counter() - t_d
            passes = max(q.num_evaluations for q in dsums)
            pair_passes = sum(q.num_evaluations for q in dsums)         # a pair leaves the lock-step when it has converged
            one_launch = os.environ.get("SBA_BATCH_DEVICE_DEPTH", "1") != "0"
            cap = os.environ.get("SBA_BATCH_DEPTH_FIRST_PASSES", "16")
            depth = {"seconds": d_s,
                     "driver": ((f"per-pair solvers on the device: first {cap} passes in one launch, the rest in launches with dynamic shares" if cap != "0"
                                 else "per-pair solvers on the device, one launch to the end") if one_launch else "host lock-step, one launch per pass"),
                     "passes_longest_pair": passes, "pair_passes": pair_passes,
                     "iterations_min_max": [min(q.num_iterations for q in dsums), max(q.num_iterations for q in dsums)],
                     "us_per_pass": d_s / max(passes, 1) * 1e6, "all_converged": bool((dstatus == 0).all() and all(q.termination.startswith("CONV") for q in dsums)),
                     "algorithmic_bytes": pair_passes * n * 96,
                     "frac": pair_passes * n * 96 / d_s / 1e9 / HBM_PEAK_GBPS,
                     "ideal_balanced_s": pair_passes * n * 96 / (0.72 * HBM_PEAK_GBPS * 1e9),
                     "what": "sba_batch_solve_depths: every pair's bounded d-only problem (own trust region, line search, convergence), "
                             "start d = 5 (pairs need 9-36 iterations); ideal_balanced_s = the same pair-passes spread evenly over the "
                             "device at the d-only kernel's own 0.72 of peak (one launch to the end is paced by the longest pair's passes "
                             "on its one CU; the default hands the late passes to launches whose blocks are dealt to the pairs still iterating)"}
        except Exception as e:      # noqa: BLE001 -- a secondary figure must never take the line down
            depth = {"ok": False, "error": f"{type(e).__name__}: {e}"}
        # the 8-point initial guess of every pair (reference .cpp:47-181): group moments of all pairs in one launch + trials
        guess = None
        try:
            b.initial_guess(80, 0.25, 1, check=False)
            t_g = time.perf_counter()
            ge, gt, gnc, gst = b.initial_guess(80, 0.25, 1, check=False)       # moments, 80 trials, consensus: all on the device
            g_dev = time.perf_counter() - t_g
            t_g = time.perf_counter()
            b.epipolar_moments()
            g_mom = time.perf_counter() - t_g
            os.environ["SBA_BATCH_DEVICE_GUESS"] = "0"
            try:
                t_g = time.perf_counter()
                he, ht, hnc, hst = b.initial_guess(80, 0.25, 1, check=False)
                g_1 = time.perf_counter() - t_g
                api.set_host_threads(0)
                t_g = time.perf_counter()
                b.initial_guess(80, 0.25, 1, check=False)
                g_auto = time.perf_counter() - t_g
            finally:
                api.set_host_threads(1)
                del os.environ["SBA_BATCH_DEVICE_GUESS"]
            guess = {"seconds": g_dev, "pairs_per_s": B / g_dev, "seconds_moments_and_copy_alone": g_mom,
                     "seconds_host_trials_1_thread": g_1, "seconds_host_trials_auto_threads": g_auto,
                     "pairs_with_a_candidate": int((gst == 0).sum()), "moments_algorithmic_bytes": B * n * 48,
                     "max_abs_difference_to_host_trials": float(max(np.abs(ge - he).max(), np.abs(gt - ht).max())),
                     "what": "sba_batch_initial_guess: 64 x 45 group moments of every pair in one launch, then 80 trials + consensus "
                             "pick per pair in a second (one block per pair); host-trial figures: SBA_BATCH_DEVICE_GUESS=0"}
        except Exception as e:      # noqa: BLE001
            guess = {"ok": False, "error": f"{type(e).__name__}: {e}"}
        # the reference's whole per-pair pipeline for every pair (initial guess -> d-only -> rot-only -> tran-only), and the
        # oracle's pipeline for a few of the same pairs on the host cores
        pipeline = None
        try:
            d0 = np.full((B * n, 2), 6.0)
            b.upload(x1, x2, off, d0)
            b.solve_problem(seed=1, check=False)
            times, inside = [], []
            for _ in range(5):
                b.set_depths(d0)
                t_p = time.perf_counter()
                res = b.solve_problem(seed=1, check=False)
                times.append(time.perf_counter() - t_p)
                inside.append(res["seconds_inside_the_library"])
            p_s = float(np.median(inside))
            counts = lambda g: (res["depth_stage"][g].num_iterations, res["depth_stage"][g].num_line_search_steps,   # noqa: E731
                                res["rot_stage"][g].num_iterations, res["tran_stage"][g].num_iterations)
            pipeline = {"seconds": p_s, "seconds_with_the_python_wrapper": float(np.median(times)), "pairs_per_s": B / p_s,
                        "pairs_ok": int((res["status"] == 0).sum()),
                        "iterations_min_median_max": {k: [int(np.min(v)), float(np.median(v)), int(np.max(v))] for k, v in
                                                      ((st, [q.num_iterations for q in res[st]]) for st in ("depth_stage", "rot_stage", "tran_stage"))},
                        "evaluations_sum_and_max": {k: [int(np.sum(v)), int(np.max(v))] for k, v in
                                                    ((st, [q.num_evaluations for q in res[st]]) for st in ("depth_stage", "rot_stage", "tran_stage"))},
                        "what": "sba_batch_solve_problem: 8-point consensus guess, d-only, rot-only, tran-only for all pairs "
                                "(start d = 6); depths already resident; the C call alone, median of 5 (the Python wrapper then builds "
                                "3 x 256 summary objects)"}
            if rank == 0:       # three pairs, well under a second: also in the child run of the default line
                from oracle import oracle_py as orc
                sample = [0, B // 2, B - 1][:max(1, min(3, B))]
                cores = max(1, min(orc.num_procs(), usable_cores()))
                e_g, t_g2, _, _ = b.initial_guess(80, 0.25, 1, check=False)

                def cpu_pairs(threads):
                    ts, ok, diff = [], True, 0.0
                    for g in sample:
                        c = cs[g]
                        t_c = time.perf_counter()
                        rot_s, tran_s = -e_g[g], t_g2[g]            # the oracle has no group-sampled guess of its own: stages only
                        dd, sdd, _ = orc.depth_solve(c.x1, c.x2, rot_s, tran_s, d0[:n])
                        r1, t1, s1, _ = orc.lm_solve(0, c.x1, c.x2, rot_s, tran_s, dd[0, 0], dd[1, 0], threads=threads)
                        r2, t2, s2, _ = orc.lm_solve(1, c.x1, c.x2, r1, t1, dd[0, 0], dd[1, 0], threads=threads)
                        ts.append(time.perf_counter() - t_c)
                        ok = ok and (sdd.num_iterations, sdd.num_line_search_steps, s1.num_iterations, s2.num_iterations) == counts(g)
                        diff = max(diff, float(np.abs(res["rot"][g] - r2).max()))
                    return ts, ok, diff
                runs = {th: cpu_pairs(th) for th in sorted({1, cores})}     # more threads than granted cores only add wake-ups
                cores = min(runs, key=lambda k: np.mean(runs[k][0]))
                cpu_t, agree, rot_diff = runs[cores]
                per_pair = float(np.mean(cpu_t))
                pipeline["cpu_oracle"] = {"seconds_per_pair": per_pair, "pairs_per_s": 1.0 / per_pair, "pairs_sampled": sample,
                                          "cores": cores, "iteration_counts_equal": bool(agree), "max_abs_rot_diff": rot_diff,
                                          "what": "oracle/sba_oracle.cpp: the three solve stages of the same pairs from the same "
                                                  "start values (the 8-point guess is not part of this figure)"}
        except Exception as e:      # noqa: BLE001
            pipeline = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    remap = None
    if rank == 0 and a.frames > 0:
        # "equi2cube remap on GPU": device-resident frames, 3840x1920 -> S = 600 strip, 6 B per output pixel
        lib = cabi.load_library()
        F, H, W, S = a.frames, 1920, 3840, 600
        src = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device="cuda")
        dst = torch.zeros((F, S, 6 * S, 3), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: cabi.check(lib, lib.sba_equi2cube_device(device_index, C.c_void_p(st), C.c_void_p(src.data_ptr()), H, W,
                                                                 S, F, C.c_void_p(dst.data_ptr())))
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            call()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        remap = {"frames": F, "ms_per_batch": ms, "frames_per_s": F / (ms * 1e-3),
                 "algorithmic_GBps": F * S * 6 * S * 6 / (ms * 1e-3) / 1e9,
                 "what": "sba_equi2cube_device: 3840x1920 8UC3 frames -> 600 x 3600 strips, 3 B gathered + 3 B stored per pixel",
                 **remap_rooflines(F, S, ms)}
        del src, dst
    if rank == 0:
        total = B * n * world
        sweep_ms = float(per_launch.mean())
        kernel_name = ("batch_step_kernel" if fused else "batch_sweep_kernel") + "<2, 1, double, 0, true>"
        achieved = B * n * 64 / (sweep_ms * 1e-3) / 1e9
        out = {
            "metric": "residual+Jacobian evals/sec", "value": total * a.steps / elapsed, "unit": "evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" + (" (ONE-GPU REHEARSAL of the multi-rank path: not a measurement)" if rehearsal else ""),
            "config": {"workload": f"{B} ERP pairs x {n} matches per GPU, batched full R|t sweep with per-match depths, one "
                                   "launch for all pairs, per-pair LM (BASELINE config C5)",
                       "pairs_per_gpu": B, "matches_per_pair": n, "bytes_per_eval": 64, "blocks_per_pair": bpp,
                       "allreduce": "none (pairs are independent)",
                       "step": ("ONE launch: batch_step_kernel (per-pair sweep state, sweep, fold, conversion, publication), "
                                "packs awaited by the host") if fused else
                               ("prepare kernel (per-pair sweep state) + batch_sweep_kernel + fold/convert/publish kernel, "
                                "packs awaited by the host")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic_named(kernel_name.split("<")[0], B * n),
                         "kernel": kernel_name, "kernel_ms": sweep_ms,
                         "kernel_ms_what": f"mean of {len(per_launch)} launches, HIP event between every two",
                         "kernel_ms_min": float(per_launch.min()), "kernel_ms_median": float(np.median(per_launch)),
                         "kernel_ms_max": float(per_launch.max()),
                         "algorithmic_bytes_per_launch": B * n * 64,
                         # the streaming part on its own (the chain's sweep kernel: no state build, fold to rows only)
                         "sweep_kernel_alone": {"kernel": "batch_sweep_kernel<2, 1, double, 0, true>",
                                                "kernel_ms": float(sweep_alone.mean()),
                                                "frac": B * n * 64 / (float(sweep_alone.mean()) * 1e-3) / 1e9 / HBM_PEAK_GBPS}},
            "preconditioning": {**precond, "what": "un-timed launches of the same kernel before the warm-up steps"},
            "step_split_ms": split,
            "lm": {"pairs_per_s": B / lm_s, "seconds": lm_s, "seconds_inside_the_library": max(s_.seconds_total for s_ in sums),
                   "max_iterations": max(s_.num_iterations for s_ in sums),
                   "all_converged": bool((status == 0).all() and all(s_.termination.startswith("CONV") for s_ in sums)),
                   "max_rot_err_rad": float(max(np.abs(rot[g] - cs[g].rot_true).max() for g in range(B)))},
            "depth_stage": depth, "initial_guess": guess, "pipeline": pipeline,
            "equi2cube": remap, "upload_s": upload_s, "reupload_s": reupload_s, "set_depths_s": set_depths_s,
            "upload_bytes": int(B * n * 64),
        }
        if not a.no_cpu_baseline and world == 1:
            from oracle import oracle_py as orc
            cores = min(orc.num_procs(), usable_cores())
            take = max(1, min(B, a.cpu_sample // n))
            passes, t0 = 0, time.perf_counter()
            while True:
                for g in range(take):
                    orc.evaluate_f64(2, cs[g].x1, cs[g].x2, cs[g].rot_init, cs[g].tran_init, 1.0, 1.0, 1.0, cs[g].d12, threads=cores)
                passes += 1
                el = time.perf_counter() - t0
                if el >= a.cpu_seconds or passes >= 50:
                    break
            out["cpu_baseline"] = {"value": take * n * passes / el, "unit": "evals/s", "cores": cores, "kind": "port",
                                   "sample": f"{passes} passes over the first {take} pairs ({take * n} correspondences), faithful "
                                             f"dual-number loop with f64 accumulation, {cores} OpenMP threads"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def remap_rooflines(frames: int, S: int, ms: float):
    """The remap is a byte gather, not a stream: priced against HBM by its algorithmic bytes (6 B per output pixel) AND
    against the L1's tag pipeline, which is what a scattered gather exercises -- tag look-ups per frame from the committed
    rocprofv3 pass of gather_tiled_kernel (profiles/pmc_gather_latest.json: TCP_TOTAL_CACHE_ACCESSES), one look-up per
    cycle and CU at the 2.4 GHz peak clock as the ceiling.  Neither is saturated (profiles/r03_gather_subtiles_ab.md)."""
    t = ms * 1e-3
    out = {"roofline": {"bound": "hbm", "achieved": frames * S * 6 * S * 6 / t / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": frames * S * 6 * S * 6 / t / 1e9 / HBM_PEAK_GBPS, "traffic": None}}
    try:
        d = json.loads((ROOT / "profiles" / "pmc_gather_latest.json").read_text())
        if d["geometry"] == {"H": 1920, "W": 3840, "S": S}:
            look = d["tag_lookups_per_frame"] * frames / t / 1e9
            peak = 256 * 2.4
            out["roofline"]["traffic"] = (d["hbm_read_bytes_per_frame"] + d["hbm_write_bytes_per_frame"]) * frames
            out["roofline_l1"] = {"bound": "l1", "achieved": look, "peak": peak, "unit": "G tag look-ups/s", "frac": look / peak,
                                  "tag_lookups_per_frame": d["tag_lookups_per_frame"],
                                  "peak_what": "256 CUs x 2.4 GHz x one tag look-up per cycle", "source": "profiles/pmc_gather_latest.json"}
    except Exception:
        pass
    return out


def pmc_traffic_named(kernel_prefix: str, units: int):
    """HBM bytes per launch of a secondary kernel from the committed stage profile (profiles/pmc_stages_latest.json)."""
    try:
        d = json.loads((ROOT / "profiles" / "pmc_stages_latest.json").read_text())
        for k, v in d["kernels"].items():
            if k.startswith(kernel_prefix) and v.get("units_per_launch") == units:
                return v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]
    except Exception:
        pass
    return None


def stage_leg(p, c, n: int):
    """N = 1, after the timed region, on the resident problem: the stages either side of the sweep in `solve_problem` --
    8-point moments + initial guess and the bounded d-only stage (f-2), wall clock, host-synchronous."""
    import numpy as np
    try:
        out = {"ok": True, "matches": n}
        p.epipolar_moments()
        t0 = time.perf_counter()
        for _ in range(10):
            p.epipolar_moments()
        out["epipolar_moments_ms_incl_fold_and_d2h"] = (time.perf_counter() - t0) / 10 * 1e3
        out["epipolar_algorithmic_bytes"] = n * 48
        t0 = time.perf_counter(); p.initial_guess(80, 0.25, 0); out["initial_guess_ms_80_trials"] = (time.perf_counter() - t0) * 1e3
        d0 = np.full((n, 2), 5.0)
        p.set_depths(d0); p.solve_depths(c.rot_true, c.tran_true)          # first call: scratch allocations
        p.set_depths(d0)
        _, s = p.solve_depths(c.rot_true, c.tran_true)
        out["depth_stage"] = {"iterations": s.num_iterations, "passes": s.num_evaluations, "line_search_steps": s.num_line_search_steps,
                              "ms_total": s.seconds_total * 1e3, "us_per_pass": s.seconds_total / max(s.num_evaluations, 1) * 1e6,
                              "algorithmic_bytes_per_pass": n * 96, "termination": s.termination}
        return out
    except Exception as e:      # noqa: BLE001 -- a secondary figure must never take the headline line down
        return {"ok": False, "error": f"{type(e).__name__}: {e}"}


def c2_leg(device_index: int, steps: int, warmup: int):
    """N = 1, after the timed region: BASELINE config C2 -- 1M synthetic correspondences, rotation-only BA (48 B per evaluation):
    K host-synchronous steps, the sweep kernel between HIP events, and the rot-only LM."""
    try:
        from spherical_bundle_adjuster_amd import api, synthetic
        n = 1_000_000
        c = synthetic.rotation_only(n, seed=synthetic.BASE_SEED + 1)
        with api.Problem(device_index) as p:
            p.upload(c.x1, c.x2)
            p.eval_launch_times(api.MODE_ROT, c.rot_init, c.tran_init, repeat=200)
            p.eval_steps(api.MODE_ROT, c.rot_init, c.tran_init, steps=max(warmup, 1))
            _, seconds = p.eval_steps(api.MODE_ROT, c.rot_init, c.tran_init, steps=steps)
            _, _, sweep_ms = p.eval_timed(api.MODE_ROT, c.rot_init, c.tran_init, repeat=steps)
            p.solve(api.MODE_ROT, c.rot_init, c.tran_init)
            runs = [p.solve(api.MODE_ROT, c.rot_init, c.tran_init) for _ in range(15)]
            r, t, s = sorted(runs, key=lambda q: q[2].seconds_total)[len(runs) // 2]          # median solve
        return {"ok": True, "workload": "1M synthetic correspondences, rotation-only (BASELINE config C2), 48 B per evaluation",
                "evals_per_s": n * steps / seconds, "ms_per_step": seconds / steps * 1e3, "kernel_ms": sweep_ms,
                "roofline": {"bound": "hbm", "achieved": n * 48 / (sweep_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": n * 48 / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                "lm": {"iterations": s.num_iterations, "iters_per_s": s.num_iterations / s.seconds_total if s.seconds_total > 0 else None,
                       "termination": s.termination, "rot_err_rad": float(abs(r - c.rot_true).max())}}
    except Exception as e:      # noqa: BLE001 -- a secondary figure must never take the headline line down
        return {"ok": False, "error": f"{type(e).__name__}: {e}"}


def c1_leg(device_index: int, n: int = 2048, reps: int = 30):
    """N = 1, after the timed region: BASELINE config C1 -- the reference's real workload, ~2 k matches, main/main.cpp's path
    (reference main/main.cpp:29-32 -> do_bundle_adjustment -> initial_guess -> solve_problem: d-only -> rot-only ->
    tran-only, spherical_bundle_adjuster.cpp:183-217).  End to end on the GPU through the C-ABI (upload, the initial guess
    from the reference's own subsets, the three stages; at this size the d-only stage is ONE launch with its solver on the device and the LM stages command a
    resident single-block kernel -- `gpu_default`; `gpu_one_launch` / `gpu_resident`: all stages one way or the other), wall clock, median of `reps`; beside it the same pipeline with a launch per sweep
    (SBA_RESIDENT_MAX_N=0), and the oracle's pipeline on the host cores.  Iteration counts of all three must agree."""
    import numpy as np
    try:
        from oracle import oracle_py as orc
        from spherical_bundle_adjuster_amd import api, synthetic
        c = synthetic.full_rt(n, seed=synthetic.BASE_SEED + 9, sigma=2e-4, outlier_fraction=0.02)
        exp_d = 6.0
        d0 = np.full((n, 2), exp_d)

        def gpu_pipeline(p):
            t = [time.perf_counter()]
            p.upload(c.x1, c.x2, d0); t.append(time.perf_counter())
            api.reference_rand_seed(1)   # a fresh process's (never-seeded) rand() stream: the subsets the reference's main() would draw
            e, tv, ncand = p.initial_guess_reference(80, 0.25); t.append(time.perf_counter())
            rot0, tran0 = -e, tv                                                   # .cpp:330-331
            d, sd = p.solve_depths(rot0, tran0); t.append(time.perf_counter())
            r1, t1, s1 = p.solve(api.MODE_ROT, rot0, tran0, d[0, 0], d[1, 0]); t.append(time.perf_counter())
            r2, t2, s2 = p.solve(api.MODE_TRAN, r1, t1, d[0, 0], d[1, 0]); t.append(time.perf_counter())
            return np.diff(t) * 1e6, (sd.num_iterations, sd.num_line_search_steps, s1.num_iterations, s2.num_iterations), (r2, t2)

        def measure(max_n, one_launch=None):
            saved = {k: os.environ.get(k) for k in ("SBA_RESIDENT_MAX_N", "SBA_SMALL_ONE_LAUNCH")}
            for k, v in (("SBA_RESIDENT_MAX_N", max_n), ("SBA_SMALL_ONE_LAUNCH", one_launch)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            try:
                with api.Problem(device_index) as p:
                    gpu_pipeline(p); gpu_pipeline(p)
                    runs = [gpu_pipeline(p) for _ in range(reps)]
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            us = np.median(np.array([r[0] for r in runs]), axis=0)
            return {"total_us": float(us.sum()), "upload_us": float(us[0]), "initial_guess_us": float(us[1]), "d_only_us": float(us[2]),
                    "rot_only_us": float(us[3]), "tran_only_us": float(us[4])}, runs[-1][1], runs[-1][2]
        default, counts_d, (rot_d, tran_d) = measure(None)                 # default: d-only stage in one launch, LM stages on resident kernels
        one_launch, counts_o, (rot_o, tran_o) = measure(None, "2")         # every stage one launch, its solver on the device
        resident, counts_r, (rot_r, tran_r) = measure(None, "0")           # host state machines commanding a resident kernel per stage
        launch, counts_l, (rot_l, tran_l) = measure("0")
        # the oracle's pipeline on the host (numpy 8-point recipe on the same subsets + the C++ restatement of the three stages)
        # ONE OpenMP thread: a 2 048-match sweep is ~0.1 ms of work, and more threads than the cgroup grants cores (the oracle's
        # default is omp_get_num_procs(), like the reference's set_omp) only add wake-ups; the faster of 1 and `cores` threads
        # is what gets reported
        cores = min(orc.num_procs(), usable_cores())

        def cpu_pipeline(threads):
            t = [time.perf_counter()]
            subsets = orc.reference_trial_subsets(n, 80, reseed=True)
            e, tv, _ = orc.initial_guess_recipe(c.x1, c.x2, subsets); t.append(time.perf_counter())
            rot0, tran0 = -e.astype(np.float64), np.asarray(tv, dtype=np.float64)
            dd, sdd, _ = orc.depth_solve(c.x1, c.x2, rot0, tran0, d0); t.append(time.perf_counter())
            r1, t1, s1, _ = orc.lm_solve(0, c.x1, c.x2, rot0, tran0, dd[0, 0], dd[1, 0], threads=threads); t.append(time.perf_counter())
            r2, t2, s2, _ = orc.lm_solve(1, c.x1, c.x2, r1, t1, dd[0, 0], dd[1, 0], threads=threads); t.append(time.perf_counter())
            return np.diff(t) * 1e6, (sdd.num_iterations, sdd.num_line_search_steps, s1.num_iterations, s2.num_iterations), (r2, t2)
        by_threads = {}
        for threads in sorted({1, cores}):
            cpu_pipeline(threads)
            by_threads[threads] = [cpu_pipeline(threads) for _ in range(3)]
        cores = min(by_threads, key=lambda k: np.median([r[0].sum() for r in by_threads[k]]))
        cpu_runs = by_threads[cores]
        cus = np.median(np.array([r[0] for r in cpu_runs]), axis=0)
        counts_c, (rot_c, tran_c) = cpu_runs[-1][1], cpu_runs[-1][2]
        # the product's T_vec sign may differ from numpy's (the reference never resolves it either): compare the rotation, and
        # the translation only when the 8-point T agreed in sign
        return {"ok": True, "matches": n, "reps": reps,
                "gpu_default": default, "gpu_one_launch": one_launch, "gpu_resident": resident, "gpu_launch_per_sweep": launch,
                "cpu_oracle": {"total_us": float(cus.sum()), "initial_guess_us": float(cus[0]), "d_only_us": float(cus[1]),
                               "rot_only_us": float(cus[2]), "tran_only_us": float(cus[3]), "cores": cores,
                               "what": "numpy restatement of initial_guess on the same subsets + oracle/sba_oracle.cpp stages"},
                "iterations": {"order": "d-only, d-only line-search contractions, rot-only, tran-only",
                               "gpu_default": counts_d, "gpu_one_launch": counts_o, "gpu_resident": counts_r, "gpu_launch_per_sweep": counts_l, "cpu_oracle": counts_c,
                               "equal": bool(counts_d == counts_o == counts_r == counts_l == counts_c)},
                "max_abs_rot_diff_gpu_vs_cpu": float(np.abs(rot_d - rot_c).max()),
                "max_abs_rot_diff_one_launch_vs_launch": float(np.abs(rot_o - rot_l).max()),
                "max_abs_rot_diff_resident_vs_launch": float(np.abs(rot_r - rot_l).max()),
                "what": "BASELINE config C1 (2 048 synthetic matches): upload -> initial guess (reference's own subsets) -> "
                        "d-only -> rot-only -> tran-only through the C-ABI, wall clock in microseconds (Python call overhead included)"}
    except Exception as e:      # noqa: BLE001 -- a secondary figure must never take the headline line down
        return {"ok": False, "error": f"{type(e).__name__}: {e}"}


def c5_leg(steps: int, warmup: int):
    """N = 1, after the timed region: BASELINE config C5 (256 ERP pairs x 50k matches: batched step, per-pair LM, 512-frame
    remap) measured by a child run of `bench.py --workload c5` with the same K / W, condensed.  A secondary figure in the
    same driver-run line; it cannot disturb the headline (own process, bounded time, errors reported as text)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "c5", "--steps", str(steps), "--warmup", str(warmup),
                            "--no-cpu-baseline"], capture_output=True, text=True, timeout=240, cwd=str(ROOT))
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"ok": False, "error": (r.stderr or r.stdout)[-300:]}
        d = json.loads(lines[-1])
        rf = d["roofline"]
        return {"ok": True, "workload": d["config"]["workload"], "evals_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                "steps": d["steps"], "warmup": d["warmup"], "step": d["config"]["step"],
                "roofline": {k: rf.get(k) for k in ("kernel", "kernel_ms", "achieved", "frac", "traffic", "algorithmic_bytes_per_launch",
                                                    "sweep_kernel_alone")},
                "lm": d["lm"], "depth_stage": d.get("depth_stage"), "initial_guess": d.get("initial_guess"), "pipeline": d.get("pipeline"), "equi2cube": d["equi2cube"],
                "what": "child run of `python bench.py --workload c5` (same K / W) after the timed region"}
    except Exception as e:      # noqa: BLE001 -- a secondary figure must never take the headline line down
        return {"ok": False, "error": f"{type(e).__name__}: {e}"}


def scaling_reference(api, synthetic, c, seed, device_index, a):
    """N = 1 only, after the timed region: the same K host-synchronous steps at 12.5M correspondences -- the per-GPU size
    every rank of an N > 1 run holds (one GPU's shard of config C4).  The quoted N = 1 value is config C3's 10M; a weak-
    scaling curve whose N = 1 point has less work per GPU than its N > 1 points flatters the efficiency (larger sweeps
    amortise the launch better), so this is the N = 1 point to divide by."""
    import numpy as np
    try:
        n_ref = 12_500_000
        extra = synthetic.full_rt(n_ref - a.n, seed=seed, shard=1)       # same two-view geometry, further points
        x1, x2, d12 = (np.concatenate([getattr(c, k), getattr(extra, k)]) for k in ("x1", "x2", "d12"))
        with api.Problem(device_index) as q:
            q.upload(x1, x2, d12)
            del x1, x2, d12
            q.eval_launch_times(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH, repeat=200)
            q.eval_steps(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH, steps=max(a.warmup, 1))
            _, seconds = q.eval_steps(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH, steps=a.steps)
            _, _, sweep_ms = q.eval_timed(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH, repeat=a.steps)
        return {"ok": True, "correspondences_per_gpu": n_ref, "value": n_ref * a.steps / seconds, "ms_per_step": seconds / a.steps * 1e3,
                "kernel_ms": sweep_ms, "frac": n_ref * 64 / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "what": "N = 1 at the per-GPU size of the N > 1 runs (12.5M = one GPU's shard of config C4): the reference "
                        "point of the weak-scaling curve"}
    except Exception as e:      # noqa: BLE001 -- a secondary figure must never take the headline line down
        return {"ok": False, "error": f"{type(e).__name__}: {e}"}


def peer_trial(p, transport, dist, torch, run_steps, barrier, ref_pack, world, n):
    """N > 1, after the quoted measurement: the same K steps over the direct xGMI peer exchange (tools: DESIGN.md 5).
    Never the quoted `value`; it exists so that the first multi-GPU run also says what one-launch peer stores cost
    next to the RCCL all-reduce.  Every rank goes through the same control collectives whatever fails locally
    (distributed._all_agree after every phase), and every device-side wait of the peer kernel is bounded in time."""
    import numpy as np
    from spherical_bundle_adjuster_amd import distributed
    res = {"ok": False, "what": "same shard, same K host-synchronous steps, all-reduce by direct peer stores over xGMI "
                                "(one launch per step: fold + exchange + publication); not the quoted value"}

    def phase(name, fn):
        ok = True
        try:
            fn()
        except Exception as e:          # noqa: BLE001 -- recorded, and agreed on below
            ok = False
            res.setdefault("errors", []).append(f"{name}: {type(e).__name__}: {e}")
        agreed = distributed._all_agree(torch, dist, ok)
        if not agreed:
            res.setdefault("failed_phase", name)
        return agreed

    def detach():
        if transport == "rccl-native":
            p.comm_destroy()
        elif transport == "torch-hook":
            p.set_allreduce(None)
        elif transport == "xgmi-peer":
            p.peer_disable()

    if not phase("detach", detach):
        return res
    attached = [False]
    if not phase("attach", lambda: attached.__setitem__(0, distributed._try_peer(p, torch, dist))) or not attached[0]:
        res.setdefault("failed_phase", "attach")
        return res
    out = {}
    if not phase("warmup", lambda: run_steps(5)):
        return res
    barrier()

    def timed():
        t0 = time.perf_counter()
        out["pack"], out["steps"] = run_steps(None)
        out["elapsed"] = time.perf_counter() - t0
    if not phase("steps", timed):
        return res
    barrier()
    t = torch.tensor([out["elapsed"]], dtype=torch.float64, device=distributed._ctrl_device(torch, dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    scale = max(float(np.abs(ref_pack).max()), 1e-300)
    res.update(ok=True, ms_per_step=elapsed / out["steps"] * 1e3, value=n * world * out["steps"] / elapsed,
               max_rel_diff_to_quoted_pack=float(np.abs(out["pack"] - ref_pack).max() / scale))
    return res


def main():
    a = parse()
    if a.workload == "c5":
        return run_c5(a)
    if a.n <= 0:
        a.n = 10_000_000 if a.gpus == 1 else 12_500_000
    import numpy as np
    import torch

    from spherical_bundle_adjuster_amd import api, distributed, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # SBA_BENCH_ONE_GPU=1: rehearsal of the N > 1 code path on a box with ONE GPU (all ranks on device 0, gloo for
    # the control messages; an NCCL group cannot place several ranks on one device).  Not a measurement.
    rehearsal = os.environ.get("SBA_BENCH_ONE_GPU") == "1"
    device_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    rt = a.workload == "rt"
    mode = api.MODE_RT if rt else api.MODE_ROT
    depth_mode = api.DEPTH_PER_MATCH if rt else api.DEPTH_UNIFORM
    store = api.STORE_F64 if a.store == "f64" else api.STORE_F32
    bytes_per_eval = {("rt", "f64"): 64, ("rt", "f32"): 40, ("rot", "f64"): 48, ("rot", "f32"): 24}[(a.workload, a.store)]
    gen = synthetic.full_rt if rt else synthetic.rotation_only
    seed = synthetic.BASE_SEED + (2 if rt else 1)
    c = gen(a.n, seed=seed, shard=rank)

    # an explicitly requested hook transport runs on torch's current stream (no stream switch per collective);
    # everything else -- including the hook that `auto` falls back to -- on the shim's own stream
    use_hook = world > 1 and a.transport == "hook"
    stream = torch.cuda.current_stream().cuda_stream if use_hook else None
    p = api.Problem(device_index, stream=stream)
    p.set_kernel(api.KERNEL_EXPLICIT if a.kernel == "explicit" else api.KERNEL_FACTORED)
    t_up = time.perf_counter()
    p.upload(c.x1, c.x2, c.d12 if rt else None, store=store)      # host cv::Point3d-layout arrays -> device planes
    upload_s = time.perf_counter() - t_up
    transport = "none"
    if world > 1:
        transport = distributed.attach(p, transport=a.transport)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    rot, tran = c.rot_init.copy(), c.tran_init.copy()
    # W warm-up steps, then EXACTLY K steps, each host-synchronous (launch -> reduction -> all-reduce -> result on
    # the host) and driven from C++ like the LM loop drives them (sba_problem_eval_steps), bracketed by barriers.
    barrier()   # ranks generate and upload their shards at different speeds: enter the first exchange together
    # `cold`: the command line taken literally -- W warm-up + K timed steps straight after the upload, BEFORE any
    # pre-conditioning (what round 1's driver run measured: the K steps fall into the clock dip after the first ~1.5 ms
    # of load).  Reported beside the quoted value, never instead of it.
    cold = None
    if not a.no_cold_leg:
        if a.warmup > 0:
            p.eval_steps(mode, rot, tran, depth_mode=depth_mode, steps=a.warmup)
        barrier()
        t0 = time.perf_counter()
        p.eval_steps(mode, rot, tran, depth_mode=depth_mode, steps=a.steps)
        barrier()
        cold_elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([cold_elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cold_elapsed = float(t.item())
        cold = {"value": a.n * world * a.steps / cold_elapsed, "ms_per_step": cold_elapsed / a.steps * 1e3,
                "what": f"{a.warmup} warm-up + {a.steps} timed steps straight after the upload, before the pre-conditioning"}
    # Un-timed, disclosed pre-conditioning (see the module docstring): the same sweep kernel, no exchange, not a step.
    precond = {"sweeps": 0, "ms": 0.0}
    if a.precondition_ms > 0:
        t_pc = time.perf_counter()
        while (time.perf_counter() - t_pc) * 1e3 < a.precondition_ms:
            p.eval_launch_times(mode, rot, tran, depth_mode=depth_mode, repeat=50)
            precond["sweeps"] += 50
        precond["ms"] = (time.perf_counter() - t_pc) * 1e3
        barrier()
    if a.warmup > 0:
        p.eval_steps(mode, rot, tran, depth_mode=depth_mode, steps=a.warmup)
    barrier()
    t0 = time.perf_counter()
    pack, _ = p.eval_steps(mode, rot, tran, depth_mode=depth_mode, steps=a.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel-only timing (HIP events on the problem's stream, same K sweeps) for the roofline figure
    _, step_ms, sweep_ms = p.eval_timed(mode, rot, tran, depth_mode=depth_mode, repeat=a.steps)
    # ... and launch by launch (an event between every two launches: each figure includes one event boundary, so the
    # mean of these sits 1-2 us above sweep_ms): the spread a mean hides
    per_launch = p.eval_launch_times(mode, rot, tran, depth_mode=depth_mode, repeat=max(a.steps, 20))
    # LM iterations per second of a real solve of this workload (secondary metric)
    opt = api.default_lm_options(tran_param=api.TRAN_SPHERE if rt else api.TRAN_FREE)
    p.solve(mode, c.rot_init, c.tran_init, depth_mode=depth_mode, options=opt)         # first call: one-time costs
    # median of several solves: a single 3-iteration solve is ~0.4 ms and one scheduling hiccup doubles it
    # (profiles/r03_lm_rate.log); every rank runs the same number of solves (each is a sequence of collectives at N > 1)
    lm_runs = [p.solve(mode, c.rot_init, c.tran_init, depth_mode=depth_mode, options=opt) for _ in range(7)]
    r_s, t_s, summ = sorted(lm_runs, key=lambda q: q[2].seconds_total)[len(lm_runs) // 2]
    barrier()
    want_trial = world > 1 and not a.no_peer_trial and (a.peer_trial or os.environ.get("SBA_BENCH_PEER_TRIAL", "0") == "1")
    trial = None

    def headline():
        total = a.n * world
        return {"metric": "residual+Jacobian evals/sec", "value": total * a.steps / elapsed, "unit": "evals/s", "n_gpus": world,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "allreduce": transport}
    if want_trial:
        if rank == 0:      # the quoted figure is on record before the un-quoted leg touches the transport
            print("bench.py: quoted measurement before the peer trial: " + json.dumps(headline()), file=sys.stderr, flush=True)

        def run_steps(k):
            k = a.steps if k is None else k
            return p.eval_steps(mode, rot, tran, depth_mode=depth_mode, steps=k)[0], k
        trial = peer_trial(p, transport, dist, torch, run_steps, barrier, pack, world, a.n)

    if rank == 0:
        total = a.n * world
        value = total * a.steps / elapsed
        achieved = a.n * bytes_per_eval / (sweep_ms * 1e-3) / 1e9
        kernel_sig = (f"sweep_kernel<{mode}, {depth_mode}, {'double' if a.store == 'f64' else 'float'}, "
                      f"{1 if a.kernel == 'explicit' else 0}, true>")
        out = {
            "metric": "residual+Jacobian evals/sec", "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # the arithmetic type of the path: always f64; with --store f32 the resident unit vectors are f32
            "dtype": "f64" if a.store == "f64" else "f64 arithmetic on f32-stored unit vectors",
            "data": "synthetic" + (" (ONE-GPU REHEARSAL of the multi-rank path: not a measurement)" if rehearsal else ""),
            "config": {"workload": ("10M synthetic unit-sphere correspondences per GPU, full R|t sweep, per-match "
                                    "depths (BASELINE config C3)" if rt and a.n == 10_000_000 and world == 1 else
                                    "12.5M synthetic unit-sphere correspondences per GPU = one GPU's shard of BASELINE "
                                    "config C4 (100M over 8 GPUs), full R|t sweep, per-match depths, one all-reduce of "
                                    "the 24-double pack per step" if rt and a.n == 12_500_000 and world > 1 else
                                    f"{a.n} synthetic correspondences per GPU, {a.workload} sweep"),
                       "correspondences_per_gpu": a.n, "mode": a.workload, "storage": a.store,
                       "bytes_per_eval": bytes_per_eval, "allreduce": transport, "kernel": a.kernel,
                       "step": "sweep kernel + finalize kernel + all-reduce(24 f64, N>1) + pack published to the host and awaited"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(kernel_sig, a.n),
                         "traffic_source": "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                           "of this command; FETCH_SIZE x2 per MI355X_MICROARCH.md)",
                         "kernel": kernel_sig, "kernel_ms": sweep_ms,
                         "kernel_ms_what": f"mean of {a.steps} back-to-back launches under one HIP event pair",
                         "kernel_ms_min": float(per_launch.min()), "kernel_ms_median": float(np.median(per_launch)),
                         "kernel_ms_mean_per_launch_events": float(per_launch.mean()),
                         "kernel_ms_max": float(per_launch.max()),
                         "kernel_ms_per_launch": [round(float(x), 4) for x in per_launch[:32]],
                         "algorithmic_bytes_per_launch": a.n * bytes_per_eval},
            "preconditioning": {**precond, "what": "un-timed sweeps of the same kernel before the warm-up steps "
                                                   "(clock dip after the first ~1.5 ms of load; --precondition-ms)",
                                # everything the device ran between the upload and the first timed step
                                "launches_before_the_timed_region": precond["sweeps"] + a.warmup
                                                                    + (0 if cold is None else a.warmup + a.steps)},
            "cold": cold,
            "kernel_only_evals_per_s": a.n / (sweep_ms * 1e-3),
            "device_step_ms": step_ms,
            "lm": {"iters_per_s": summ.num_iterations / summ.seconds_total if summ.seconds_total > 0 else None,
                   "what": "median of 7 solves", "iterations": summ.num_iterations, "termination": summ.termination,
                   "rot_err_rad": float(np.abs(r_s - c.rot_true).max()),
                   "tran_err": float(np.abs(t_s - c.tran_true).max())},
            "peer_trial": trial,
            "upload_s": upload_s,   # once per problem (H2D over PCIe + re-layout); never part of `value`
            "pcie_inclusive_evals_per_s_if_reuploaded_every_sweep": a.n / (upload_s + elapsed / a.steps),
            "cost": float(pack[22]),
        }
        if world == 1 and not rehearsal and bytes_per_eval == 64 and a.store == "f64":
            probe = stream_probe(a.n)
            if probe:
                best = max(probe["GBps_1_block_per_cu"], probe["GBps_2_blocks_per_cu"])
                out["roofline"]["stream_probe"] = {
                    "GBps_1_block_per_cu": probe["GBps_1_block_per_cu"],
                    "GBps_2_blocks_per_cu": probe["GBps_2_blocks_per_cu"], "GBps_best": best,
                    "sweep_over_probe_best": achieved / best,
                    "what": "tools/stream_probe.hip, same box, same run: the same 8 planes streamed once with the sweep "
                            "kernel's access pattern and no residual/Jacobian arithmetic"}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(c, mode, rt, a.cpu_sample, a.cpu_seconds)
    if rank == 0:
        if world == 1 and not rehearsal and rt and a.n == 10_000_000 and a.store == "f64" and not a.no_scaling_reference:
            out["scaling_reference"] = scaling_reference(api, synthetic, c, seed, device_index, a)
        if world == 1 and not rehearsal and not a.no_stage_leg and rt:
            out["stages"] = stage_leg(p, c, a.n)
        if world == 1 and not rehearsal and not a.no_c1_leg:
            out["c1"] = c1_leg(device_index)
            out["c2"] = c2_leg(device_index, a.steps, a.warmup)
        if world == 1 and not rehearsal and not a.no_c5_leg and rt and a.store == "f64":
            p.close()                       # the child gets the GPU to itself
            out["c5"] = c5_leg(a.steps, a.warmup)
        print(json.dumps(out), flush=True)
    p.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
