"""Diagnostic (GPU box): how many integer results of the key-point / image maps differ from the oracle, per pitch.
Usage: python tests/diag_maps_mismatch.py [H W]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import oracle_py as orc  # noqa: E402  (diagnostic tool: the oracle is the checker)
from spherical_bundle_adjuster_amd import api  # noqa: E402

sizes = [(1920, 3840), (480, 960), (1000, 2000), (1080, 2160), (960, 1920)]
if len(sys.argv) == 3:
    sizes = [(int(sys.argv[1]), int(sys.argv[2]))]
for H, W in sizes:
    rows = np.arange(H // 4)
    allkp = np.zeros((len(rows) * W, 7), dtype=np.float32)
    rr, cc = np.meshgrid(rows, np.arange(W), indexing="ij")
    allkp[:, 0], allkp[:, 1] = cc.ravel(), rr.ravel()
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    for pitch in (45.0, -45.0, -90.0, 0.0, 30.0, 90.0):
        got = api.rotate_keypoints(allkp, pitch, W, H)
        ref = orc.rotate_keypoints(allkp, pitch, W, H)
        bad = (got[:, :2] != ref[:, :2]).any(axis=1)
        g2 = api.crop_rotated_image(im, pitch)
        r2 = orc.crop_rotated_image(im, pitch)
        print(f"H={H} W={W} pitch={pitch:6.1f}: rotate_keypoints {int(bad.sum())} / {len(allkp)} differ; "
              f"crop {int((g2 != r2).any(axis=2).sum())} / {g2.shape[0] * g2.shape[1]} pixels differ", flush=True)
