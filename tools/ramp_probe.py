"""Diagnostic (GPU box): per-launch device time of the headline sweep kernel from a cold start -- does the clock ramp
explain a slow first few milliseconds?  Prints launch times of the first launches after process start, after idle gaps,
and the stream probe for the same box.  Usage: python tools/ramp_probe.py [n]"""
import json
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
c = synthetic.full_rt(n, seed=synthetic.BASE_SEED + 2)
out = {"n": n}
with api.Problem(0) as p:
    p.upload(c.x1, c.x2, c.d12)
    kw = dict(depth_mode=api.DEPTH_PER_MATCH)
    t = p.eval_launch_times(api.MODE_RT, c.rot_init, c.tran_init, repeat=400, **kw) * 1e3
    out["cold_first_400_us"] = [round(float(x), 1) for x in t]
    out["cold_mean_first20_us"] = float(t[:20].mean())
    out["cold_mean_last100_us"] = float(t[-100:].mean())
    out["cold_min_us"] = float(t.min())
    for gap in (0.0, 0.01, 0.1, 1.0, 3.0):
        time.sleep(gap)
        t = p.eval_launch_times(api.MODE_RT, c.rot_init, c.tran_init, repeat=25, **kw) * 1e3
        out[f"after_{gap}s_idle_first25_us"] = [round(float(x), 1) for x in t]
    # what the bench does: 5 warm-up steps, 20 timed steps, then 20 timed launches
    time.sleep(2.0)
    p.eval_steps(api.MODE_RT, c.rot_init, c.tran_init, steps=5, **kw)
    _, sec = p.eval_steps(api.MODE_RT, c.rot_init, c.tran_init, steps=20, **kw)
    out["bench_like_step_us"] = sec / 20 * 1e6
    _, step_ms, sweep_ms = p.eval_timed(api.MODE_RT, c.rot_init, c.tran_init, repeat=20, **kw)
    out["bench_like_sweep_us"] = sweep_ms * 1e3
    # the same after 300 pre-conditioning sweeps
    time.sleep(2.0)
    p.eval_timed(api.MODE_RT, c.rot_init, c.tran_init, repeat=300, **kw)
    p.eval_steps(api.MODE_RT, c.rot_init, c.tran_init, steps=5, **kw)
    _, sec = p.eval_steps(api.MODE_RT, c.rot_init, c.tran_init, steps=20, **kw)
    out["preconditioned_step_us"] = sec / 20 * 1e6
    _, step_ms, sweep_ms = p.eval_timed(api.MODE_RT, c.rot_init, c.tran_init, repeat=20, **kw)
    out["preconditioned_sweep_us"] = sweep_ms * 1e3
    t = p.eval_launch_times(api.MODE_RT, c.rot_init, c.tran_init, repeat=200, **kw) * 1e3
    out["warm_launch_us"] = {"min": float(t.min()), "median": float(np.median(t)), "mean": float(t.mean()), "max": float(t.max())}
exe = ROOT / "spherical_bundle_adjuster_amd" / "csrc" / "build" / "stream_probe"
r = subprocess.run([str(exe), str(n), "30", "quick"], capture_output=True, text=True, timeout=120)
line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
out["stream_probe"] = json.loads(line[-1]) if line else r.stdout[-500:]
print(json.dumps(out, indent=1))
