#!/usr/bin/env python3
"""Device-side timing of the sweep kernel variants (run on the GPU box):
   python tools/sweep_tune.py [--n 10000000] [--repeat 30]
Prints one line per (kernel kind, mode, depth, store, blocks/CU cap): sweep ms, algorithmic GB/s, evals/s."""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--repeat", type=int, default=30)
    ap.add_argument("--caps", default="8")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--libs", default="", help="comma-separated name=path of alternative libsba_hip builds to A/B")
    a = ap.parse_args()
    c = synthetic.full_rt(a.n)
    rows = []
    combos = [("rt", api.MODE_RT, api.DEPTH_PER_MATCH, 64, 40), ("rt-u", api.MODE_RT, api.DEPTH_UNIFORM, 48, 24),
              ("rot", api.MODE_ROT, api.DEPTH_UNIFORM, 48, 24), ("tran", api.MODE_TRAN, api.DEPTH_UNIFORM, 48, 24)]
    if a.quick:
        combos = combos[:1]
    from spherical_bundle_adjuster_amd import _cabi
    libs = [("default", None)]
    for item in filter(None, a.libs.split(",")):
        nm, path = item.split("=")
        libs.append((nm, _cabi.load_library(path)))
    for cap, (lname, lib) in [(int(x), l) for x in a.caps.split(",") for l in libs]:
        os.environ["SBA_BLOCKS_PER_CU"] = str(cap)
        for store, sname in ((api.STORE_F64, "f64"), (api.STORE_F32, "f32")):
            with api.Problem(0, lib=lib) as p:
                p.upload(c.x1, c.x2, c.d12, store=store)
                for kind, kname in ((api.KERNEL_FACTORED, "factored"), (api.KERNEL_EXPLICIT, "explicit")):
                    p.set_kernel(kind)
                    for name, mode, dm, b64, b32 in combos:
                        if mode == api.MODE_TRAN and kind == api.KERNEL_EXPLICIT:
                            continue
                        p.eval_timed(mode, c.rot_init, c.tran_init, depth_mode=dm, repeat=3)
                        best = 1e9
                        for _ in range(3):
                            _, step, sweep = p.eval_timed(mode, c.rot_init, c.tran_init, depth_mode=dm, repeat=a.repeat)
                            best = min(best, sweep)
                        bpe = b64 if store == api.STORE_F64 else b32
                        print(f"{lname:8s} cap={cap} {sname} {kname:9s} {name:5s} sweep {best*1e3:8.1f} us  step {step*1e3:8.1f} us  "
                              f"{a.n*bpe/best/1e6:8.0f} GB/s  {a.n/best/1e6:8.1f} Gevals/s", flush=True)


if __name__ == "__main__":
    main()
