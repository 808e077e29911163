#!/usr/bin/env python3
"""The secondary kernels at full size, for rocprofv3 (tools/profile_stages.sh): the d-only stage and the 8-point
moments on 10^7 matches, config C5's batched step (256 pairs x 50k) and per-pair LM, equi2cube of 512 frames.
Prints one JSON line: algorithmic bytes per launch of every kernel it exercises (what summarize_profiles.py prices
the trace against) plus the wall-clock figures of the stages.  Usage: python3 tools/stage_workload.py [--frames 512]"""
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

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--pairs", type=int, default=256)
ap.add_argument("--pair-matches", type=int, default=50_000)
ap.add_argument("--frames", type=int, default=512)
a = ap.parse_args()
import torch  # noqa: E402

out = {"algorithmic_bytes_per_launch": {}, "units_per_launch": {}}
n = a.n
c = synthetic.full_rt(n)
with api.Problem(0) as p:
    p.upload(c.x1, c.x2, np.full((n, 2), 5.0))
    p.epipolar_moments()
    t0 = time.perf_counter()
    for _ in range(10):
        p.epipolar_moments()
    out["epipolar_moments_ms_incl_d2h"] = (time.perf_counter() - t0) / 10 * 1e3
    t0 = time.perf_counter(); p.initial_guess(80, 0.25, 0); out["initial_guess_ms"] = (time.perf_counter() - t0) * 1e3
    p.solve_depths(c.rot_true, c.tran_true)
    p.set_depths(np.full((n, 2), 5.0))
    d, s = p.solve_depths(c.rot_true, c.tran_true)
    out["depth_stage"] = {"iterations": s.num_iterations, "passes": s.num_evaluations, "line_search_steps": s.num_line_search_steps,
                          "ms_total": s.seconds_total * 1e3, "us_per_pass": s.seconds_total / max(s.num_evaluations, 1) * 1e6,
                          "termination": s.termination}
out["algorithmic_bytes_per_launch"]["epipolar_moments_kernel<double>"] = n * 48
out["units_per_launch"]["epipolar_moments_kernel<double>"] = n
# steady-state pass of the d-only stage: 48 B coordinates + 16 B depths + 16 B scaling + 16 B candidates
out["algorithmic_bytes_per_launch"]["depth_step_kernel<double, 1>"] = n * 96   # <ST, AHEAD>
out["units_per_launch"]["depth_step_kernel<double, 1>"] = n
del c

B, m = a.pairs, a.pair_matches
cs = [synthetic.full_rt(m, seed=7000 + g) for g in range(B)]
off = (np.arange(B + 1) * m).astype(np.uint64)
x1, x2, d12 = (np.concatenate([getattr(q, k) for q in cs]) for k in ("x1", "x2", "d12"))
rot0 = np.stack([q.rot_init for q in cs]); tran0 = np.stack([q.tran_init for q in cs])
with api.Batch(0) as b:
    b.upload(x1, x2, off, d12)
    kw = dict(depth_mode=api.DEPTH_PER_MATCH)
    b.eval_timed(api.MODE_RT, rot0, tran0, 50, **kw)
    _, split = b.eval_timed(api.MODE_RT, rot0, tran0, 50, **kw)
    out["c5_step_ms"] = split
    out["c5_step_is_fused"] = b.step_is_fused
    opt = api.default_lm_options(tran_param=api.TRAN_SPHERE)
    b.solve(api.MODE_RT, rot0, tran0, options=opt, **kw)
    t0 = time.perf_counter()
    rot, tran, sums, status = b.solve(api.MODE_RT, rot0, tran0, options=opt, **kw)
    out["c5_lm"] = {"seconds": time.perf_counter() - t0, "max_iterations": max(q.num_iterations for q in sums),
                    "all_converged": bool((status == 0).all())}
for kname in ("batch_step_kernel<2, 1, double, 0, true>", "batch_sweep_kernel<2, 1, double, 0, true>"):
    out["algorithmic_bytes_per_launch"][kname] = B * m * 64      # whichever of the two the step runs (fused or chain)
    out["units_per_launch"][kname] = B * m

if a.frames > 0:
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
    for _ in range(5):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    out["equi2cube"] = {"frames": F, "ms_per_batch": ms, "frames_per_s": F / (ms * 1e-3),
                        "algorithmic_GBps": F * S * 6 * S * 6 / (ms * 1e-3) / 1e9,
                        "host_decided_table_entries": int(lib.sba_map_table_host_decided(0, 0, S, H, W))}
    for kname in ("gather_tiled_kernel<true>", "gather_kernel<true>"):      # whichever of the two the remap runs
        out["algorithmic_bytes_per_launch"][kname] = F * S * 6 * S * 6
        out["units_per_launch"][kname] = F * S * 6 * S
    tiles, staged, lds = C.c_int(0), C.c_int(0), C.c_int(0)
    lib.sba_map_table_tiles(0, 0, S, H, W, C.byref(tiles), C.byref(staged), C.byref(lds))
    out["equi2cube"].update(tiles=tiles.value, staged_tiles=staged.value, lds_bytes_per_frame=lds.value)
print(json.dumps(out))
