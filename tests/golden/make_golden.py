"""Generates tests/golden/*.npz from the oracle (oracle/sba_oracle.cpp).

The reference ships no golden vectors for the BA path ("parity unpinned"), so
these fixtures are produced by this repo's own CPU restatement and pin it against regressions; they
are data only (inputs + expected outputs).  Run:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle_py as orc  # noqa: E402
from spherical_bundle_adjuster_amd import synthetic  # noqa: E402

OUT = Path(__file__).resolve().parent


def pointwise():
    rng = np.random.default_rng(11)
    rows = []
    cases = {
        "theta_zero": np.zeros(3),
        "theta_tiny": np.array([3e-9, -4e-9, 1e-9]),          # theta^2 < DBL_EPSILON -> small-angle branch
        "theta_just_above": np.array([1.2e-8, 0.9e-8, 0.0]),   # theta^2 slightly > DBL_EPSILON
        "theta_mid": np.array([0.21, -0.35, 0.11]),
        "theta_near_pi": np.array([1.8, -1.9, 1.7]) * (3.1 / np.linalg.norm([1.8, -1.9, 1.7])),
    }
    for name, rot in cases.items():
        for k in range(4):
            x1 = rng.standard_normal(3); x1 /= np.linalg.norm(x1)
            x2 = rng.standard_normal(3); x2 /= np.linalg.norm(x2)
            tran = rng.standard_normal(3) * (0.0 if k == 0 else 0.7)
            d1, d2 = (1.0, 1.0) if k < 2 else (float(rng.uniform(0.5, 9)), float(rng.uniform(0.5, 9)))
            for mode in (0, 1, 2):
                e, J = orc.point(mode, x1, x2, rot, tran, d1, d2)
                rho = orc.huber(1.0, float(e @ e))
                rows.append(dict(case=name, mode=mode, x1=x1, x2=x2, rot=rot, tran=tran, d1=d1, d2=d2, e=e, J=J, rho=rho))
    np.savez(OUT / "pointwise.npz",
             case=np.array([r["case"] for r in rows]), mode=np.array([r["mode"] for r in rows]),
             x1=np.stack([r["x1"] for r in rows]), x2=np.stack([r["x2"] for r in rows]),
             rot=np.stack([r["rot"] for r in rows]), tran=np.stack([r["tran"] for r in rows]),
             d1=np.array([r["d1"] for r in rows]), d2=np.array([r["d2"] for r in rows]),
             e=np.stack([r["e"] for r in rows]), J=np.stack([r["J"] for r in rows]),
             rho=np.stack([r["rho"] for r in rows]))


def reductions():
    out = {}
    for n in (1, 63, 64, 65, 2048):
        c = synthetic.full_rt(n, seed=1000 + n, outlier_fraction=0.1)
        out[f"n{n}_x1"], out[f"n{n}_x2"], out[f"n{n}_d12"] = c.x1, c.x2, c.d12
        out[f"n{n}_rot"], out[f"n{n}_tran"] = c.rot_init, c.tran_init
        for mode in (0, 1, 2):
            for dm, d12 in (("u", None), ("p", c.d12)):
                ev = orc.evaluate(mode, c.x1, c.x2, c.rot_init, c.tran_init, d1=1.3, d2=0.9, delta=1.0, d12=d12, threads=1)
                out[f"n{n}_m{mode}_{dm}_H"], out[f"n{n}_m{mode}_{dm}_g"] = ev.H, ev.g
                out[f"n{n}_m{mode}_{dm}_scalars"] = np.array([ev.cost, ev.sum_w, ev.n_outlier])
    np.savez(OUT / "reductions.npz", **out)


def solves():
    out = {}
    # (name, generator kwargs, mode, per-match depth?)
    c = synthetic.rotation_only(2048, seed=synthetic.BASE_SEED + 1)      # C1 twin: rot-only, noisy + outliers
    r, t, s, rc = orc.lm_solve(0, c.x1, c.x2, c.rot_init, c.tran_init, threads=1)
    assert rc == 0
    out.update(c1_x1=c.x1, c1_x2=c.x2, c1_rot0=c.rot_init, c1_tran0=c.tran_init, c1_rot=r, c1_tran=t,
               c1_meta=np.array([s.termination, s.num_iterations, s.num_successful_steps, s.final_cost]))
    c = synthetic.full_rt(2048, seed=synthetic.BASE_SEED + 2)            # C3 twin at small N
    for name, tp in (("rt6", 0), ("rt5", 1)):
        r, t, s, rc = orc.lm_solve(2, c.x1, c.x2, c.rot_init, c.tran_init, d12=c.d12, threads=1,
                                   options=orc.default_options(tran_param=tp))
        assert rc == 0
        out[f"{name}_rot"], out[f"{name}_tran"] = r, t
        out[f"{name}_meta"] = np.array([s.termination, s.num_iterations, s.num_successful_steps, s.final_cost])
    out.update(rt_x1=c.x1, rt_x2=c.x2, rt_d12=c.d12, rt_rot0=c.rot_init, rt_tran0=c.tran_init)
    np.savez(OUT / "solves.npz", **out)


def depth_stage():
    """d-only stage (reference .cpp:1004-1063) with Ceres' projected line search: starts whose full step fails Armijo,
    a start where it never does, and the same without the line search (what round 1 computed)."""
    out = {}
    cases = [("ls_d1", 500, 6, 1.0, 1.0, 1.0), ("ls_d005", 400, 9, 0.05, 1.0, 1.0), ("ls_reg", 64, 23, 0.01, 20.0, 4.0),
             ("plain_d2", 300, 12, 2.0, 1.0, 1.0)]
    for name, n, seed, d0, lam, c_ in cases:
        c = synthetic.full_rt(n, seed=seed)
        out[f"{name}_x1"], out[f"{name}_x2"], out[f"{name}_rot"], out[f"{name}_tran"] = c.x1, c.x2, c.rot_init, c.tran_init
        out[f"{name}_cfg"] = np.array([d0, lam, c_])
        for tag, ls in (("ceres", 20), ("nols", 0)):
            d, s, rc = orc.depth_solve(c.x1, c.x2, c.rot_init, c.tran_init, np.full((n, 2), d0), lam=lam, c=c_,
                                       options=orc.default_options(max_num_line_search_step_size_iterations=ls))
            assert rc == 0
            out[f"{name}_{tag}_d"] = d
            out[f"{name}_{tag}_meta"] = np.array([s.termination, s.num_iterations, s.num_successful_steps,
                                                   s.num_line_search_steps, s.final_cost])
    np.savez(OUT / "depth_stage.npz", **out)


def maps():
    """Integer / float32 coordinate maps of the matchers (spherical_surf.cpp:48-123, equi2cube_surf.cpp:19-76,
    equi2cube.cpp:12-302) on a small geometry, every pitch the reference uses plus the all-ties pitch 0."""
    rng = np.random.default_rng(77)
    H, W, S = 96, 192, 24
    im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    rr, cc = np.meshgrid(np.arange(H // 4), np.arange(W), indexing="ij")
    kp = np.zeros((rr.size, 7), dtype=np.float32)
    kp[:, 0], kp[:, 1] = cc.ravel(), rr.ravel()
    out = dict(im=im, kp=kp, geometry=np.array([H, W, S]))
    for pitch in (45.0, -45.0, -90.0, 0.0):
        tag = f"p{int(pitch)}".replace("-", "m")
        out[f"rotate_{tag}"] = orc.rotate_keypoints(kp, pitch, W, H)[:, :2]
        out[f"crop_{tag}"] = orc.crop_rotated_image(im, pitch)
    cube = np.zeros((400, 7), dtype=np.float32)
    cube[:, 0] = rng.uniform(0, 6 * S, 400); cube[:, 1] = rng.uniform(0, S, 400)
    cube[:6, 0] = [S / 2 + k * S for k in range(6)]; cube[:6, 1] = S / 2
    out["cube_kp"] = cube
    out["cube2equi"] = orc.cube2equi_keypoints(cube, S, W, H)[:, :2]
    out["equi2cube"], _ = orc.equi2cube(im, S)
    np.savez_compressed(OUT / "maps.npz", **out)


if __name__ == "__main__":
    pointwise()
    reductions()
    solves()
    depth_stage()
    maps()
    print("golden fixtures written to", OUT)
