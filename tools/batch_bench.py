#!/usr/bin/env python3
"""BASELINE config C5: 256 ERP pairs x 50k matches, per-pair LM, equi2cube remap on the GPU.
   python tools/batch_bench.py [--pairs 256] [--n 50000]                                  one GPU
   python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 tools/batch_bench.py
                                                                                           G GPUs: the pairs are split
       between the ranks (pairs are independent: NO collective on the data path); times are the max
       over ranks, throughputs the totals.  SBA_BENCH_ONE_GPU=1 rehearses that on one GPU (not a measurement).
Prints JSON: batched sweep time (all pairs, one launch per rank), evals/s, GB/s, LM pairs/s, remap GB/s."""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from spherical_bundle_adjuster_amd import _cabi as cabi  # noqa: E402
from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--n", type=int, default=50_000)
    ap.add_argument("--frames", type=int, default=32, help="ERP frames (3840x1920) remapped per batch")
    a = ap.parse_args()
    import os
    import torch
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    rehearsal = os.environ.get("SBA_BENCH_ONE_GPU") == "1"
    device = 0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")                      # control messages only: the data path has no exchange

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    n = a.n
    first, last = a.pairs * rank // world, a.pairs * (rank + 1) // world     # this rank's pairs
    B = last - first
    cs = [synthetic.full_rt(n, seed=7000 + g) for g in range(first, last)]
    off = (np.arange(B + 1) * n).astype(np.uint64)
    x1, x2, d12 = (np.concatenate([getattr(c, k) for c in cs]) for k in ("x1", "x2", "d12"))
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    out = {"config": f"{a.pairs} pairs x {n} matches, full R|t, per-match depths, f64 planes", "n_gpus": world,
           "pairs_per_gpu": B}
    if rehearsal and world > 1:
        out["note"] = "ONE-GPU REHEARSAL of the multi-rank path: not a measurement"
    with api.Batch(device) as b:
        b.upload(x1, x2, off, d12)
        out["blocks_per_pair"] = b.blocks_per_pair
        for _ in range(3):
            b.eval(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            b.eval(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH)
        dt = (time.perf_counter() - t0) / reps
        _, split = b.eval_timed(api.MODE_RT, rot0, tran0, 50, depth_mode=api.DEPTH_PER_MATCH)
        out["c_loop_ms"] = split
        dt = max_over_ranks(min(dt, split["step_ms"] * 1e-3))
        out["batched_sweep_ms_host_synchronous"] = dt * 1e3
        out["evals_per_s"] = a.pairs * n / dt
        out["algorithmic_GBps"] = a.pairs * n * 64 / dt / 1e9
        t0 = time.perf_counter()
        rot, tran, sums, status = b.solve(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH,
                                          options=api.default_lm_options(tran_param=api.TRAN_SPHERE))
        dt = max_over_ranks(time.perf_counter() - t0)
        out["lm_pairs_per_s"] = a.pairs / dt
        out["lm_seconds"] = dt
        out["lm_max_iterations"] = max(s.num_iterations for s in sums)
        out["lm_all_converged"] = bool((status == 0).all() and all(s.termination.startswith("CONV") for s in sums))
        out["max_rot_err_rad"] = float(max(np.abs(rot[g] - cs[g].rot_true).max() for g in range(B)))
        for threads in (4,):        # host threads for the per-pair LM steps (sba_set_host_threads)
            api.set_host_threads(threads)
            b.solve(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH,
                    options=api.default_lm_options(tran_param=api.TRAN_SPHERE))
            t0 = time.perf_counter()
            rot_t, _, _, _ = b.solve(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH,
                                     options=api.default_lm_options(tran_param=api.TRAN_SPHERE))
            out[f"lm_seconds_{threads}_host_threads"] = time.perf_counter() - t0
            assert np.array_equal(rot_t, rot)
        api.set_host_threads(1)
    if rank != 0:                 # the remap figure below is a one-GPU kernel measurement
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return
    # equi2cube on device-resident frames: 3840x1920 -> S=600 strip, 6 B per output pixel
    lib = cabi.load_library()
    F, H, W, S = a.frames, 1920, 3840, 600
    src = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device="cuda")
    dst = torch.zeros((F, S, 6 * S, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: cabi.check(lib, lib.sba_equi2cube_device(0, C.c_void_p(st), C.c_void_p(src.data_ptr()), H, W, S, F,
                                                             C.c_void_p(dst.data_ptr())))
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out["equi2cube_ms_per_batch"] = ms
    out["equi2cube_frames_per_s"] = F / (ms * 1e-3)
    out["equi2cube_algorithmic_GBps"] = F * S * 6 * S * 6 / (ms * 1e-3) / 1e9
    print(json.dumps(out))
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
