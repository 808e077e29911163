#!/usr/bin/env python3
"""bench.py at every one-GPU BASELINE config shape -> a markdown table (profiles/r02_configs.md).
Usage (GPU box): python tools/configs_bench.py > gpurun_out/r02_configs.md"""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
RUNS = [
    ("c1 (2 048 matches, rot-only)", ["--matches", "2048", "--workload", "rot", "--steps", "200", "--cpu-sample", "2048", "--cpu-seconds", "2"]),
    ("c2 (10^6, rot-only)", ["--matches", "1000000", "--workload", "rot", "--steps", "100", "--cpu-sample", "1000000", "--cpu-seconds", "4"]),
    ("c3 headline (10^7, R|t, f64, factored)", ["--steps", "50", "--no-cpu-baseline"]),
    ("c3 f32 planes", ["--steps", "50", "--store", "f32", "--no-cpu-baseline"]),
    ("c3 explicit-Jacobian kernel", ["--steps", "50", "--kernel", "explicit", "--no-cpu-baseline"]),
    ("c4 share, R|t (12.5 M)", ["--matches", "12500000", "--steps", "50", "--no-cpu-baseline"]),
    ("c4 share, rot-only (12.5 M)", ["--matches", "12500000", "--workload", "rot", "--steps", "50", "--no-cpu-baseline"]),
    ("c4 total on one GPU (10^8, R|t)", ["--matches", "100000000", "--steps", "20", "--no-cpu-baseline"]),
]
print("# One-GPU measurements at the BASELINE config shapes (round 2)\n")
print("`python bench.py <args>` on one MI355X, default pre-conditioning (60 ms).  Config C5: profiles/r02_bench_c5.json.\n")
print("| run | matches | mode/storage/kernel | evals/s (host-synchronous steps) | step us | sweep kernel us (min..max per launch) | algorithmic GB/s | % of 8 TB/s | LM iters/s | CPU faithful evals/s |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, args in RUNS:
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args + ["--no-c5-leg", "--no-stage-leg", "--no-c1-leg", "--no-scaling-reference"], capture_output=True, text=True, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        print(f"| {name} | FAILED rc={r.returncode} {r.stderr[-200:]!r} |")
        continue
    d = json.loads(line[-1])
    rf, c = d["roofline"], d["config"]
    cpu = d.get("cpu_baseline", {}).get("value")
    print(f"| {name} | {c['correspondences_per_gpu']} | {c['mode']}/{c['storage']}/{c['kernel']} | {d['value']:.3e} | {d['ms_per_step'] * 1e3:.1f} | "
          f"{rf['kernel_ms'] * 1e3:.1f} ({rf['kernel_ms_min'] * 1e3:.1f}..{rf['kernel_ms_max'] * 1e3:.1f}) | {rf['achieved']:.0f} | {rf['frac'] * 100:.1f} | "
          f"{d['lm']['iters_per_s']:.0f} | {cpu if cpu is None else format(cpu, '.3e')} |", flush=True)
