"""A/B of the tiled gather's tuning knobs on the C5 remap (F frames 3840x1920 -> S = 600): LDS budget per tile and
frame, frames per block, XCD-aware order, against the pixel-per-lane kernel.  One child process per configuration (the
table cache is keyed by geometry only).  Usage: python tools/gather_sweep.py [frames=256]"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def child(F):
    import torch
    from spherical_bundle_adjuster_amd import _cabi as cabi
    H, W, S = 1920, 3840, 600
    lib = cabi.load_library()
    src = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device="cuda")
    dst = torch.zeros((F, S, 6 * S, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: cabi.check(lib, lib.sba_equi2cube_device(0, C.c_void_p(st), C.c_void_p(src.data_ptr()), H, W, S, F, C.c_void_p(dst.data_ptr())))
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tiles, staged, lds = C.c_int(0), C.c_int(0), C.c_int(0)
    lib.sba_map_table_tiles(0, 0, S, H, W, C.byref(tiles), C.byref(staged), C.byref(lds))
    print(json.dumps({"us_per_frame": ms * 1e3 / F, "GBps": F * S * 6 * S * 6 / (ms * 1e-3) / 1e9, "tiles": tiles.value,
                      "staged": staged.value, "lds_per_frame": lds.value}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
        sys.exit(0)
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    configs = [{"SBA_GATHER_TILED": "0"}]
    for budget in (6144, 8192, 12288, 16384, 24576):
        for fpb in (1, 2, 4):
            configs.append({"SBA_GATHER_LDS_BUDGET": str(budget), "SBA_GATHER_FPB": str(fpb)})
    configs.append({"SBA_GATHER_XCD": "0"})
    for cfg in configs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(F)], env=dict(os.environ, **cfg),
                           capture_output=True, text=True, timeout=120)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print(json.dumps(cfg), line[-1] if line else ("FAILED " + r.stderr[-300:]), flush=True)
