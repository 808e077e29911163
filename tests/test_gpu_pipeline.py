"""GPU (-m gpu): the C++ mirror of the reference class (csrc/spherical_bundle_adjuster.cpp) driven through the CLI
with the reference's argument contract (main/main.cpp:8-27), against the oracle's three-stage pipeline
(pixel -> sphere, d-only, rot-only, tran-only with the init_d[0][0]/init_d[1][0] quirk)."""
import subprocess

import numpy as np
import pytest

from helpers import ROOT
from spherical_bundle_adjuster_amd import api, synthetic

pytestmark = pytest.mark.gpu

SBA_MAIN = ROOT / "spherical_bundle_adjuster_amd" / "csrc" / "build" / "sba_main"


def _sphere_to_pixels(x, W, H):
    colat = np.arccos(np.clip(x[:, 2], -1, 1))
    lon = np.mod(np.arctan2(x[:, 1], x[:, 0]), 2 * np.pi)
    return lon / (2 * np.pi) * W, colat / np.pi * H


def _write_keypoints(path, px, py, W, H):
    kp = np.zeros((len(px), 7), dtype=np.float32)      # cv::KeyPoint records (28 B)
    kp[:, 0], kp[:, 1] = px, py
    with open(path, "wb") as f:
        np.array([len(px), W, H, 0], dtype=np.int32).tofile(f)
        kp.tofile(f)
    return kp


def test_cli_three_stage_pipeline(oracle, tmp_path):
    assert SBA_MAIN.exists(), "build with make -C spherical_bundle_adjuster_amd/csrc"
    W, H, n = 3840, 1920, 2048                          # config C1: ~2k matches between two ERP frames
    c = synthetic.full_rt(n, seed=synthetic.BASE_SEED, sigma=2e-4, outlier_fraction=0.02)
    lx, ly = _sphere_to_pixels(c.x1, W, H)
    rx, ry = _sphere_to_pixels(c.x2, W, H)
    kl = _write_keypoints(tmp_path / "left.kp", lx, ly, W, H)
    kr = _write_keypoints(tmp_path / "right.kp", rx, ry, W, H)
    deg = np.rad2deg(c.rot_init)
    exp_d = 6.0
    args = [str(SBA_MAIN), str(tmp_path / "left.kp"), str(tmp_path / "right.kp"), *(f"{v:.17g}" for v in deg),
            *(f"{v:.17g}" for v in c.tran_init), f"{exp_d}"]
    import os
    r = subprocess.run(args, cwd=tmp_path, capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, SBA_INITIAL_GUESS="0"))       # start from the expected values (.cpp:328-329)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "rotation vector in degree" in r.stdout and "translation vector" in r.stdout and "Done." in r.stdout
    row = (tmp_path / "log.txt").read_text().strip().split(",")
    assert len(row) == 10 and int(row[9]) == n           # same 10 columns as the reference's log.txt (.cpp:348-354)
    got_rot = np.deg2rad([float(v) for v in row[3:6]])
    got_tran = np.array([float(v) for v in row[6:9]])

    # oracle pipeline on the same key-points
    x1 = oracle.keypoints_to_sphere(kl, W, H)
    x2 = oracle.keypoints_to_sphere(kr, W, H)
    rot0 = np.deg2rad(deg)                               # CLI passes degrees (main/main.cpp:20-22, .cpp:328)
    d, _, rc = oracle.depth_solve(x1, x2, rot0, c.tran_init, np.full((n, 2), exp_d))
    assert rc == 0
    d1, d2 = d[0, 0], d[1, 0]                            # init_d[0][0], init_d[1][0] for every match
    r1, t1, _, _ = oracle.lm_solve(0, x1, x2, rot0, c.tran_init, d1, d2)
    r2, t2, _, _ = oracle.lm_solve(1, x1, x2, r1, t1, d1, d2)
    # log.txt prints ~6 significant digits
    assert np.abs(got_rot - r2).max() < 2e-6 * max(1, np.abs(r2).max()) + 1e-7
    assert np.abs(got_tran - t2).max() < 2e-5 * max(1, np.abs(t2).max())
    # log_d.txt: one "d1,d2" line per match with the d-only stage's result (write_log_d, .cpp:219-225, :357)
    logged = np.loadtxt(tmp_path / "log_d.txt", delimiter=",").reshape(-1, 2)
    assert logged.shape == (n, 2) and np.abs(logged - d).max() <= 2e-5 * np.abs(d).max()


def test_cli_usage_and_errors(tmp_path):
    r = subprocess.run([str(SBA_MAIN)], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and "usage" in r.stdout     # the reference returns 0 on a usage error (main/main.cpp:11)
    r = subprocess.run([str(SBA_MAIN), "nope", "nope", *["0"] * 7], capture_output=True, text=True, timeout=30)
    assert r.returncode != 0


def test_cli_with_eight_point_initial_guess(oracle, tmp_path):
    """Default path like the reference: 8-point consensus -> init_rot = -Euler, init_tran = T (.cpp:304, :330-331),
    then d-only -> rot-only -> tran-only.  Checked against the same stages driven through the Python binding and
    the oracle from the SAME initial values."""
    from spherical_bundle_adjuster_amd import api
    W, H, n = 3840, 1920, 4096
    c = synthetic.full_rt(n, seed=synthetic.BASE_SEED + 9, sigma=2e-4, outlier_fraction=0.02)
    lx, ly = _sphere_to_pixels(c.x1, W, H)
    rx, ry = _sphere_to_pixels(c.x2, W, H)
    kl = _write_keypoints(tmp_path / "left.kp", lx, ly, W, H)
    kr = _write_keypoints(tmp_path / "right.kp", rx, ry, W, H)
    exp_d = 6.0
    args = [str(SBA_MAIN), str(tmp_path / "left.kp"), str(tmp_path / "right.kp"), "0", "0", "0", "0", "0", "0", f"{exp_d}"]
    r = subprocess.run(args, cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "E matrix estimation with SVD" in r.stdout
    row = (tmp_path / "log.txt").read_text().strip().split(",")
    got_rot = np.deg2rad([float(v) for v in row[3:6]]); got_tran = np.array([float(v) for v in row[6:9]])
    x1 = oracle.keypoints_to_sphere(kl, W, H); x2 = oracle.keypoints_to_sphere(kr, W, H)
    with api.Problem(0) as p:
        p.upload(x1, x2)
        # the CLI is a fresh process: the library's copy of the reference's never-seeded rand() stream, nothing drawn before
        # initial_guess -- the reference's own subsets (the mirror class's default up to 65 536 matches).  Same state here.
        api.reference_rand_seed(1)
        e, t, ncand = p.initial_guess_reference(80, 0.25)
    assert ncand > 40
    rot0, tran0 = -e, t
    d, _, rc = oracle.depth_solve(x1, x2, rot0, tran0, np.full((n, 2), exp_d))
    r1, t1, _, _ = oracle.lm_solve(0, x1, x2, rot0, tran0, d[0, 0], d[1, 0])
    r2, t2, _, _ = oracle.lm_solve(1, x1, x2, r1, t1, d[0, 0], d[1, 0])
    assert np.abs(got_rot - r2).max() < 2e-6 * max(1, np.abs(r2).max()) + 1e-7
    assert np.abs(got_tran - t2).max() < 2e-5 * max(1, np.abs(t2).max())
    # and the guess itself is in the neighbourhood of the true rotation
    assert np.abs(rot0 - c.rot_true).max() < 0.2


@pytest.mark.parametrize("n,store", [(7, api.STORE_F64), (63, api.STORE_F64), (2048, api.STORE_F64), (2049, api.STORE_F32),
                                     (8192, api.STORE_F64)])
def test_resident_evaluator_equals_the_launch_path(oracle, n, store, monkeypatch):
    """Small problems (up to 4 096 matches, 2 560 for the d-only stage: BASELINE config C1, the reference's real workload) are solved
    with ONE launch per stage -- the stage's solver runs on the device, the problem being a batch of one pair for
    batch_depth_solve_kernel (the default for the d-only stage) / batch_lm_kernel (SBA_SMALL_ONE_LAUNCH=2) -- or with ONE
    resident single-block kernel per stage (the default for the LM stages; SBA_SMALL_ONE_LAUNCH=0: for all three) that the host LM / d-only state machine commands through mapped
    memory (csrc/sba_resident.hpp) instead of two launches per sweep.  Same state machines, same per-match arithmetic;
    only the fold order of the sums differs (one block instead of a grid + finalize kernel): all three stages must
    agree with the launch-per-sweep path to rounding, with identical iteration / step / contraction counts -- and with
    the oracle."""
    c = synthetic.full_rt(n, seed=4200 + n, sigma=5e-4)
    start = np.full((n, 2), 3.0)
    res = {}
    # force each path at every size (defaults: up to 4 096 / 2 560 matches, one launch per stage when nobody watches the iterations)
    for name, max_n, one_launch in (("one_launch", "100000", "2"), ("resident", "100000", "0"), ("launch", "0", "0")):
        monkeypatch.setenv("SBA_RESIDENT_MAX_N", max_n)
        monkeypatch.setenv("SBA_SMALL_ONE_LAUNCH", one_launch)
        with api.Problem(0) as p:
            p.upload(c.x1, c.x2, start, store=store)
            d, sd = p.solve_depths(c.rot_init, c.tran_init)
            r1, t1, s1 = p.solve(api.MODE_ROT, c.rot_init, c.tran_init, d[0, 0], d[1 % n, 0])
            r2, t2, s2 = p.solve(api.MODE_TRAN, r1, t1, d[0, 0], d[1 % n, 0])
            r3, t3, s3 = p.solve(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH,
                                 options=api.default_lm_options(tran_param=api.TRAN_SPHERE))
            res[name] = (d, sd, r1, s1, t2, s2, r3, t3, s3)
    dL, sdL, r1L, s1L, t2L, s2L, r3L, t3L, s3L = res["launch"]
    for name in ("one_launch", "resident"):     # one_launch: the stage's solver itself runs on the device (batch kernels on a batch of one)
        d, sd, r1, s1, t2, s2, r3, t3, s3 = res[name]
        assert (sd.num_iterations, sd.num_successful_steps, sd.num_line_search_steps, sd.num_evaluations, sd.termination) == \
            (sdL.num_iterations, sdL.num_successful_steps, sdL.num_line_search_steps, sdL.num_evaluations, sdL.termination), name
        assert np.abs(d - dL).max() <= 1e-10 * max(1.0, np.abs(dL).max()), name
        for a, b in ((s1, s1L), (s2, s2L), (s3, s3L)):
            assert (a.num_iterations, a.num_successful_steps, a.num_evaluations, a.termination, a.final_cost == pytest.approx(b.final_cost, rel=1e-12)) == \
                (b.num_iterations, b.num_successful_steps, b.num_evaluations, b.termination, True), name
        assert np.abs(r1 - r1L).max() <= 1e-11 and np.abs(t2 - t2L).max() <= 1e-11, name
        assert np.abs(r3 - r3L).max() <= 1e-10 and np.abs(t3 - t3L).max() <= 1e-10, name
    d, sd, r1, s1, t2, s2, r3, t3, s3 = res["one_launch"]
    if store == api.STORE_F64:
        dref, sref, rc = oracle.depth_solve(c.x1, c.x2, c.rot_init, c.tran_init, start)
        assert rc == 0 and sd.num_iterations == sref.num_iterations and np.abs(d - dref).max() <= 1e-9 * max(1.0, np.abs(dref).max())
        ro, to, so, rc = oracle.lm_solve(0, c.x1, c.x2, c.rot_init, c.tran_init, dref[0, 0], dref[1 % n, 0])
        assert rc == 0 and s1.num_iterations == so.num_iterations and np.abs(r1 - ro).max() <= 1e-8


def test_resident_kernel_survives_an_idle_host(oracle, monkeypatch):
    """A resident kernel ends itself when no command arrives for SBA_RESIDENT_IDLE_S (a descheduled host thread must not
    leave a spinning kernel behind); the next command restarts it.  Forced here with an idle limit of 2 ms and a verbose
    LM whose per-iteration printing is slower than that now and then: the solve must still be the launch path's."""
    c = synthetic.full_rt(3000, seed=4301)
    with api.Problem(0) as p:
        p.upload(c.x1, c.x2, c.d12)
        r0, t0, s0 = p.solve(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
    monkeypatch.setenv("SBA_SMALL_ONE_LAUNCH", "0")               # the resident evaluator, not the one-launch stages
    monkeypatch.setenv("SBA_RESIDENT_IDLE_S", "0.000001")         # every gap between two commands is "idle"
    with api.Problem(0) as p:
        p.upload(c.x1, c.x2, c.d12)
        r1, t1, s1 = p.solve(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
        d, sd = p.solve_depths(c.rot_true, c.tran_true)
    assert s1.num_iterations == s0.num_iterations and np.array_equal(r1, r0) and np.array_equal(t1, t0)
    assert sd.termination.startswith("CONVERGENCE")


def test_bench_default_line_carries_roofline_cpu_baseline_and_c5(tmp_path):
    """`python bench.py` as the driver runs it at N = 1 (smaller problem, same code path): ONE JSON line with the contract's
    keys, `roofline` and `cpu_baseline`, and the config-C5 figures from the child run (`c5`)."""
    import json
    import subprocess
    import sys
    from helpers import ROOT
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--matches", "300000", "--steps", "5", "--warmup", "2",
                        "--cpu-seconds", "1", "--cpu-sample", "100000", "--precondition-ms", "5"],
                       capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    assert b["n_gpus"] == 1 and b["steps"] == 5 and b["dtype"] == "f64" and b["vs_baseline"] is None
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(b["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(b["cpu_baseline"]) and b["cpu_baseline"]["kind"] == "port"
    c5 = b["c5"]
    assert c5["ok"], c5
    assert c5["roofline"]["kernel"].startswith("batch_step_kernel") and 0.3 < c5["roofline"]["frac"] < 1.0
    assert c5["lm"]["all_converged"] and c5["equi2cube"]["frames"] == 512
    # the remap is priced against HBM by its algorithmic bytes and against the L1 tag pipeline, its real neighbourhood
    assert c5["equi2cube"]["roofline"]["bound"] == "hbm" and c5["equi2cube"]["roofline_l1"]["bound"] == "l1"
    assert 0.05 < c5["equi2cube"]["roofline_l1"]["frac"] < 1.0
    st = b["stages"]
    assert st["ok"] and st["depth_stage"]["termination"].startswith("CONVERGENCE") and st["initial_guess_ms_80_trials"] > 0
    # config C1, the reference's real workload end to end: resident kernels, launch per sweep and the oracle on the host
    # take the same numbers of iterations and land on the same rotation
    c1 = b["c1"]
    assert c1["ok"], c1
    assert c1["iterations"]["equal"], c1["iterations"]
    assert c1["max_abs_rot_diff_gpu_vs_cpu"] < 1e-6 and c1["max_abs_rot_diff_resident_vs_launch"] < 1e-10
    assert 0 < c1["gpu_default"]["total_us"] and 0 < c1["gpu_one_launch"]["total_us"] and 0 < c1["gpu_resident"]["total_us"]
    assert 0 < c1["cpu_oracle"]["total_us"]
    assert c1["max_abs_rot_diff_one_launch_vs_launch"] < 1e-10
    assert b["cold"]["value"] > 0
    c2 = b["c2"]
    assert c2["ok"] and c2["lm"]["termination"].startswith("CONVERGENCE") and 0.1 < c2["roofline"]["frac"] < 1.0
