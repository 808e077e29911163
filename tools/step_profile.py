"""Where one fused C5 step spends its time (needs libsba_hip.so built with EXTRA=-DSBA_STEP_PROFILE: pack slots 0..4 then
hold thread 0's ticks of the 100 MHz wall clock instead of results)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spherical_bundle_adjuster_amd import api, synthetic
B, n = 256, 50_000
cs = [synthetic.full_rt(n, seed=7000 + i) for i in range(B)]
off = (np.arange(B + 1) * n).astype(np.uint64)
x1, x2, d12 = (np.concatenate([getattr(c, k) for c in cs]) for k in ("x1", "x2", "d12"))
rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
with api.Batch(0) as b:
    b.upload(x1, x2, off, d12)
    for _ in range(300):
        p = b.eval(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH)
    acc = np.zeros((20, B, 5))
    for i in range(20):
        acc[i] = b.eval(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH)[:, :5]
    us = acc[..., :4] / 100.0
    print("per block, mean over 20 steps and 256 blocks [us]: state read %.2f, state build %.2f, sweep+fold %.2f, conversion %.2f"
          % tuple(us.mean(axis=(0, 1))))
    print("max over blocks (mean over steps): ", us.mean(axis=0).max(axis=0))
    start = acc[..., 4]
    print("block start spread within a step [us]: mean %.2f max %.2f" % (((start.max(axis=1) - start.min(axis=1)) / 100).mean(),
                                                                         ((start.max(axis=1) - start.min(axis=1)) / 100).max()))
    # where the kernel's time between HIP events goes beyond a block's own 121 us: per step, every block's end stamp
    # (start + its four phases) against the first block's start -- the in-kernel span -- and against the mean block
    end = start + acc[..., :4].sum(axis=2)
    span = (end.max(axis=1) - start.min(axis=1)) / 100
    print("in-kernel span (first block start -> last block end, thread 0's clock) [us]: mean %.2f min %.2f max %.2f" % (span.mean(), span.min(), span.max()))
    print("last block end - mean block end [us]: mean %.2f; slowest block's sweep+fold - mean sweep+fold [us]: mean %.2f"
          % (((end.max(axis=1) - end.mean(axis=1)) / 100).mean(), (us[..., 2].max(axis=1) - us[..., 2].mean(axis=1)).mean()))
    late = (end - end.mean(axis=1, keepdims=True)).mean(axis=0) / 100          # per block, mean over steps
    by_xcd = [late[x::8].mean() for x in range(8)]
    print("mean lateness of a block's end by blockIdx % 8 [us]:", " ".join("%.2f" % v for v in by_xcd))
    print("the 8 latest blocks (index: us after the mean end):", ", ".join("%d: %.1f" % (i, late[i]) for i in np.argsort(late)[-8:]))
    launch_ms = b.step_launch_times(api.MODE_RT, rot0, tran0, repeat=50, depth_mode=api.DEPTH_PER_MATCH)
    print("kernel between HIP events (profile build) [us]: mean %.2f min %.2f" % (launch_ms.mean() * 1e3, launch_ms.min() * 1e3))
    opt = api.default_lm_options(tran_param=api.TRAN_SPHERE)
    ts = []
    for _ in range(12):
        rot, tran, sums, status = b.solve(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH, options=opt)
        ts.append(max(s.seconds_total for s in sums))
    print("device LM, seconds inside the library: min %.1f us median %.1f us" % (min(ts) * 1e6, np.median(ts) * 1e6))
