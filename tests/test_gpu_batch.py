"""GPU (-m gpu): batched per-pair sweeps and LM (BASELINE config C5 shape, scaled down) against per-pair oracle /
single-problem results; ragged and empty pairs; both kernels; equi2cube on a batch of frames."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import REL_TOL_F64, RT_TOL_F64, pack_from_eval
from spherical_bundle_adjuster_amd import _cabi as cabi
from spherical_bundle_adjuster_amd import api, synthetic

pytestmark = pytest.mark.gpu


def _make_pairs(sizes, seed0=900, rt=True):
    cs = [(synthetic.full_rt if rt else synthetic.rotation_only)(n, seed=seed0 + i) for i, n in enumerate(sizes)]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    x1 = np.concatenate([c.x1 for c in cs]) if sum(sizes) else np.zeros((0, 3))
    x2 = np.concatenate([c.x2 for c in cs]) if sum(sizes) else np.zeros((0, 3))
    d12 = np.concatenate([c.d12 for c in cs]) if rt and sum(sizes) else (np.zeros((0, 2)) if rt else None)
    return cs, off, x1, x2, d12


@pytest.mark.parametrize("kind", [api.KERNEL_FACTORED, api.KERNEL_EXPLICIT], ids=["factored", "explicit"])
@pytest.mark.parametrize("store", [api.STORE_F64, api.STORE_F32], ids=["f64", "f32"])
def test_batch_eval_matches_oracle(oracle, kind, store):
    sizes = [1, 0, 63, 64, 65, 1000, 2, 513, 3001, 7]
    cs, off, x1, x2, d12 = _make_pairs(sizes)
    rot = np.stack([c.rot_init for c in cs]); tran = np.stack([c.tran_init for c in cs])
    with api.Batch(0) as b:
        b.set_kernel(kind)
        b.upload(x1, x2, off, d12, store=store)
        for mode in (api.MODE_ROT, api.MODE_TRAN, api.MODE_RT):
            for dm in (api.DEPTH_PER_MATCH, api.DEPTH_UNIFORM):
                d1 = np.linspace(0.8, 1.7, len(sizes)); d2 = np.linspace(1.3, 0.6, len(sizes))
                packs = b.eval(mode, rot, tran, d1, d2, 1.0, dm)
                for g, c in enumerate(cs):
                    a1, a2 = ((c.x1, c.x2) if store == api.STORE_F64 else
                              (c.x1.astype(np.float32).astype(np.float64), c.x2.astype(np.float32).astype(np.float64)))
                    ref = pack_from_eval(mode, oracle.evaluate(mode, a1, a2, c.rot_init, c.tran_init, d1[g], d2[g], 1.0,
                                                               c.d12 if dm == api.DEPTH_PER_MATCH else None))
                    scale = max(np.abs(ref).max(), 1e-300)
                    assert np.abs(packs[g] - ref).max() <= REL_TOL_F64 * scale, (mode, dm, g, sizes[g])


@pytest.mark.parametrize("sizes", [[1, 0, 63, 64, 65, 1000, 2, 513, 3001, 7, 255, 256, 257, 511, 512, 1025],
                                   [100_003, 0, 99_999, 100_000]], ids=["ragged_one_block_per_pair", "several_blocks_per_pair"])
@pytest.mark.parametrize("store", [api.STORE_F64, api.STORE_F32], ids=["f64", "f32"])
def test_interleaved_pair_layout_is_bit_identical_to_the_contiguous_one(sizes, store, monkeypatch):
    """Batches whose pairs are of similar length are laid out INTERLEAVED (tile t of every pair side by side in 4 KiB tiles,
    so that blocks sweeping their own pairs in step read one contiguous window per plane); ragged batches stay contiguous
    (SBA_BATCH_INTERLEAVE forces either).  Only addresses change -- every lane consumes the same vectors in the same
    order -- so packs and solves must agree to the last bit, with one block per pair (one-launch step, device LM) and with
    several (three-kernel chain, host lock-step LM), ragged tails, empty pairs, f64 and f32 planes."""
    cs, off, x1, x2, d12 = _make_pairs(sizes, seed0=3100)
    B = len(sizes)
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    d1 = np.linspace(0.8, 1.7, B); d2 = np.linspace(1.3, 0.6, B)
    got = {}
    for layout in ("0", "1"):
        monkeypatch.setenv("SBA_BATCH_INTERLEAVE", layout)
        with api.Batch(0) as b:
            b.upload(x1, x2, off, d12, store=store)
            packs = [b.eval(mode, rot0, tran0, d1, d2, 1.0, dm) for mode in (api.MODE_ROT, api.MODE_TRAN, api.MODE_RT)
                     for dm in (api.DEPTH_PER_MATCH, api.DEPTH_UNIFORM)]
            b.set_kernel(api.KERNEL_EXPLICIT)
            packs.append(b.eval(api.MODE_RT, rot0, tran0, d1, d2, 1.0, api.DEPTH_PER_MATCH))
            b.set_kernel(api.KERNEL_FACTORED)
            sol = b.solve(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH,
                          options=api.default_lm_options(tran_param=api.TRAN_SPHERE))
            got[layout] = (packs, sol, b.blocks_per_pair)
    assert got["0"][2] == got["1"][2]
    for a, c in zip(got["0"][0], got["1"][0]):
        assert np.array_equal(a, c)
    assert np.array_equal(got["0"][1][0], got["1"][1][0]) and np.array_equal(got["0"][1][1], got["1"][1][1])
    assert [q.num_iterations for q in got["0"][1][2]] == [q.num_iterations for q in got["1"][1][2]]


@pytest.mark.parametrize("store", [api.STORE_F64, api.STORE_F32], ids=["f64", "f32"])
@pytest.mark.parametrize("layout", ["0", "1"], ids=["contiguous", "interleaved"])
@pytest.mark.parametrize("driver", ["1", "0"], ids=["one-launch", "lock-step"])
def test_batch_depth_stage_matches_single_problem_stages(oracle, store, layout, driver, monkeypatch):
    """sba_batch_solve_depths: the d-only stage (reference .cpp:196-197, :1004-1063) for every pair of a batch -- B
    independent bounded problems, each with its own trust region, projected line search and convergence; driven by one
    DepthStageSolver per pair ON THE DEVICE in one launch (batch_depth_solve_kernel), or -- SBA_BATCH_DEVICE_DEPTH=0 -- by B
    host solvers in lock-step, one launch per pass.  Per pair it must do what sba_problem_solve_depths does for that pair alone: same
    iteration / accepted-step / contraction counts and termination, depths to 1e-9 -- ragged pairs, an empty one, a
    1-match one, pairs whose full step fails Armijo (start d = 1) next to pairs that converge at once -- and the oracle's
    numbers; the refined depths must be the ones a following per-match sweep sees."""
    monkeypatch.setenv("SBA_BATCH_INTERLEAVE", layout)
    monkeypatch.setenv("SBA_BATCH_DEVICE_DEPTH", driver)
    sizes = [500, 0, 257, 1, 300, 64, 1023, 2] + [150 + 7 * g for g in range(40)]
    cs, off, x1, x2, d12 = _make_pairs(sizes, seed0=6)                 # seed 6, n = 500: one contraction to a = 0.49
    B = len(sizes)
    start = np.ones_like(d12)
    start[off[4]:off[5]] = 3.0                                         # one pair starts elsewhere
    rot = np.stack([c.rot_init for c in cs]); tran = np.stack([c.tran_init for c in cs])
    with api.Batch(0) as b:
        b.upload(x1, x2, off, start, store=store)
        d, sums, status = b.solve_depths(rot, tran)
        assert (status == 0).all() and d.shape == start.shape
        packs = b.eval(api.MODE_RT, rot, tran, depth_mode=api.DEPTH_PER_MATCH)     # the batch's planes hold the refined depths
        d_again, sums2, _ = b.solve_depths(rot, tran)                              # ... and a second stage starts from them
    contractions = 0
    for g, c in enumerate(cs):
        n = sizes[g]
        lo, hi = int(off[g]), int(off[g + 1])
        if n == 0:
            assert sums[g].termination == "CONVERGENCE_GRADIENT" and sums[g].num_iterations == 0
            continue
        with api.Problem(0) as p:
            p.upload(c.x1, c.x2, start[lo:hi], store=store)
            d1, s1 = p.solve_depths(c.rot_init, c.tran_init)
            ref_pack = p.eval_pack(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
        assert (sums[g].num_iterations, sums[g].num_successful_steps, sums[g].num_line_search_steps, sums[g].termination) == \
            (s1.num_iterations, s1.num_successful_steps, s1.num_line_search_steps, s1.termination), (g, n)
        assert np.abs(d[lo:hi] - d1).max() <= 1e-9 * max(1.0, np.abs(d1).max()), (g, n)
        assert np.abs(packs[g] - ref_pack).max() <= 1e-9 * max(np.abs(ref_pack).max(), 1e-300), (g, n)
        assert sums2[g].num_iterations <= 1 or sums2[g].final_cost <= sums[g].final_cost * (1 + 1e-12)
        contractions += sums[g].num_line_search_steps
        if store == api.STORE_F64 and g in (0, 2, 4, 6):
            dref, sref, rc = oracle.depth_solve(c.x1, c.x2, c.rot_init, c.tran_init, start[lo:hi])
            assert rc == 0 and sums[g].num_iterations == sref.num_iterations and sums[g].num_line_search_steps == sref.num_line_search_steps
            assert np.abs(d[lo:hi] - dref).max() <= 1e-7 * max(1.0, np.abs(dref).max())
    assert contractions >= 1                                            # the batch really exercised the line search
    assert len({s_.num_evaluations for s_, n in zip(sums, sizes) if n > 0}) > 1      # ... and pairs finished at different passes


def test_batch_depth_stage_one_launch_equals_lock_step_bitwise(monkeypatch):
    """The device-resident solvers and the host lock-step solvers are the same source fed the same nine reductions: every
    pair's depths, counts and final cost must agree to the bit (options off the defaults too: no line search, few iterations)."""
    sizes = [3000, 1, 0, 777, 4096, 50] + [900 + 31 * g for g in range(30)]
    cs, off, x1, x2, d12 = _make_pairs(sizes, seed0=11)
    rot = np.stack([c.rot_init for c in cs]); tran = np.stack([c.tran_init for c in cs])
    start = np.full_like(d12, 1.5)
    monkeypatch.setenv("SBA_BATCH_DEPTH_FIRST_PASSES", "0")        # the one-launch kernel to the end: same reduction order as the host path
    for opts in (None, api.default_lm_options(max_num_line_search_step_size_iterations=0), api.default_lm_options(max_num_iterations=3)):
        got = {}
        for driver in ("1", "0"):
            monkeypatch.setenv("SBA_BATCH_DEVICE_DEPTH", driver)
            with api.Batch(0) as b:
                b.upload(x1, x2, off, start)
                got[driver] = b.solve_depths(rot, tran, options=opts)
        d_dev, s_dev, st_dev = got["1"]
        d_host, s_host, st_host = got["0"]
        assert np.array_equal(st_dev, st_host) and (st_dev == 0).all()
        assert np.array_equal(d_dev, d_host)
        for a, h in zip(s_dev, s_host):
            assert (a.num_iterations, a.num_successful_steps, a.num_line_search_steps, a.num_evaluations, a.termination, a.final_cost,
                    a.final_radius) == (h.num_iterations, h.num_successful_steps, h.num_line_search_steps, h.num_evaluations,
                                        h.termination, h.final_cost, h.final_radius)


@pytest.mark.parametrize("store", [api.STORE_F64, api.STORE_F32], ids=["f64", "f32"])
@pytest.mark.parametrize("first", ["1", "3", "7"])
def test_batch_depth_stage_hand_over_to_dynamic_shares(store, first, monkeypatch):
    """Default driver of sba_batch_solve_depths: the one-launch kernel runs the first passes of every pair, pairs that need more
    are handed over -- solver state, current / candidate planes, Jacobi scaling -- to per-pass launches whose blocks are dealt
    out to the pairs still iterating.  Whatever the hand-over point (1, 3, 7 passes: before, inside and after line searches),
    every pair must take exactly the passes, iterations and contractions of the one-launch kernel and end at its depths
    (1e-8: a pair swept by several blocks folds its nine reductions in another order); pairs that finish inside the first
    launch, an empty and a 1-match pair, a following stage starting from the refined depths."""
    sizes = [3000, 1, 0, 777, 4096, 50, 20000] + [900 + 31 * g for g in range(30)]
    cs, off, x1, x2, d12 = _make_pairs(sizes, seed0=11)
    rot = np.stack([c.rot_init for c in cs]); tran = np.stack([c.tran_init for c in cs])
    start = np.full_like(d12, 1.5)
    start[off[3]:off[4]] = d12[off[3]:off[4]]                        # one pair starts at the answer: done within the first launch
    got = {}
    for cap in ("0", first):
        monkeypatch.setenv("SBA_BATCH_DEPTH_FIRST_PASSES", cap)
        with api.Batch(0) as b:
            b.upload(x1, x2, off, start, store=store)
            d, sums, status = b.solve_depths(rot, tran)
            packs = b.eval(api.MODE_RT, rot, tran, depth_mode=api.DEPTH_PER_MATCH)     # the batch's planes hold the refined depths
            got[cap] = (d, sums, status, packs)
    d0, s0, st0, p0 = got["0"]
    d1, s1, st1, p1 = got[first]
    assert (st0 == 0).all() and (st1 == 0).all()
    assert [(q.num_iterations, q.num_successful_steps, q.num_line_search_steps, q.num_evaluations, q.termination) for q in s0] == \
           [(q.num_iterations, q.num_successful_steps, q.num_line_search_steps, q.num_evaluations, q.termination) for q in s1]
    assert np.abs(d0 - d1).max() <= 1e-8 * max(1.0, np.abs(d0).max())
    assert np.abs(p0 - p1).max() <= 1e-8 * np.abs(p0).max()
    assert max(q.num_evaluations for q in s1) > int(first) + 2          # pairs really went on after the hand-over


def test_batch_depth_stage_few_long_pairs_spread_over_the_device():
    """Fewer pairs than CUs: the d-only stage hands every pair over after its first pass, so that each pass of a pair is swept
    by G / pairs blocks instead of one.  Per pair it must still be the single-problem stage: same counts, depths to 1e-9."""
    sizes = [120_000, 90_001, 0, 150_000]
    cs, off, x1, x2, d12 = _make_pairs(sizes, seed0=7300)
    rot = np.stack([c.rot_init for c in cs]); tran = np.stack([c.tran_init for c in cs])
    start = np.full_like(d12, 2.0)
    with api.Batch(0) as b:
        b.upload(x1, x2, off, start)
        d, sums, status = b.solve_depths(rot, tran)
    assert (status == 0).all()
    for g, c in enumerate(cs):
        if sizes[g] == 0:
            continue
        lo, hi = int(off[g]), int(off[g + 1])
        with api.Problem(0) as p:
            p.upload(c.x1, c.x2, start[lo:hi])
            d1, s1 = p.solve_depths(c.rot_init, c.tran_init)
        assert (sums[g].num_iterations, sums[g].num_line_search_steps, sums[g].num_evaluations, sums[g].termination) == \
            (s1.num_iterations, s1.num_line_search_steps, s1.num_evaluations, s1.termination), g
        assert np.abs(d[lo:hi] - d1).max() <= 1e-9 * max(1.0, np.abs(d1).max()), g


@pytest.mark.parametrize("store", [api.STORE_F64, api.STORE_F32], ids=["f64", "f32"])
@pytest.mark.parametrize("layout", ["0", "1"], ids=["contiguous", "interleaved"])
def test_batch_initial_guess_matches_single_problem_guesses(store, layout, monkeypatch):
    """sba_batch_initial_guess: the 8-point initial guess (reference .cpp:47-181) once per pair.  A pair's 64 x 45 group
    moments equal numpy's on that pair's (stored) coordinates -- the group of a match is counted WITHIN its pair -- and
    the single-problem pass on the pair alone up to summation order; repeated launches are bit-identical; the guess of
    pair g is exactly what the host part makes of pair g's moments, and agrees with the single-problem guess; ragged
    pairs, an empty pair (no candidate: status says so, the others are served), pairs below 256 matches (empty groups)."""
    from test_initial_guess_cpu import group_moments
    monkeypatch.setenv("SBA_BATCH_INTERLEAVE", layout)
    sizes = [5000, 0, 257, 40, 3001, 64, 1023, 9] + [700 + 13 * g for g in range(24)]
    cs, off, x1, x2, d12 = _make_pairs(sizes, seed0=41)
    with api.Batch(0) as b:
        b.upload(x1, x2, off, d12, store=store)
        gm = b.epipolar_moments()
        assert np.array_equal(gm, b.epipolar_moments())
        e_dev, t_dev, nc_dev, status_dev = b.initial_guess(80, 0.25, 7, check=False)     # trials + consensus on the device
        assert all(np.array_equal(x, y) for x, y in zip((e_dev, t_dev, nc_dev, status_dev), b.initial_guess(80, 0.25, 7, check=False)))
        monkeypatch.setenv("SBA_BATCH_DEVICE_GUESS", "0")
        e, t, nc, status = b.initial_guess(80, 0.25, 7, check=False)                     # ... and on the host
        api.set_host_threads(4)
        try:
            e4, t4, nc4, status4 = b.initial_guess(80, 0.25, 7, check=False)     # pairs spread over host threads: same result
        finally:
            api.set_host_threads(1)
        e200 = b.initial_guess(200, 0.25, 7, check=False)                        # more trials than the kernel holds: host path
        monkeypatch.delenv("SBA_BATCH_DEVICE_GUESS")
        assert all(np.array_equal(x, y) for x, y in zip(e200, b.initial_guess(200, 0.25, 7, check=False)))
        with pytest.raises(api.SbaError):
            b.initial_guess(80, 0.25, 7)                                        # the empty pair makes the checked call fail
    assert np.array_equal(e, e4) and np.array_equal(t, t4) and np.array_equal(nc, nc4) and np.array_equal(status, status4)
    # device trials = the host's source, compiled for the device: same candidates, same pick.  The one libm call of a trial
    # (double atan2, rounded to float) may differ between the two libraries in the last float bit.
    assert np.array_equal(status_dev, status) and np.array_equal(nc_dev, nc)
    assert np.abs(e_dev - e).max() <= 2.5e-7 and np.abs(t_dev - t).max() <= 1.2e-7
    assert (np.abs(e_dev - e).max(axis=1) == 0).mean() >= 0.9
    for g, c in enumerate(cs):
        n = sizes[g]
        if n == 0:
            assert status[g] == cabi.SBA_ERR_NUMERIC and nc[g] == 0 and not gm[g].any()
            continue
        a1, a2 = ((c.x1, c.x2) if store == api.STORE_F64 else
                  (c.x1.astype(np.float32).astype(np.float64), c.x2.astype(np.float32).astype(np.float64)))
        ref, _, _ = group_moments(a1, a2)
        assert np.abs(gm[g] - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1.0), (g, n)
        eh, th, nh = api.initial_guess_from_moments(gm[g], 80, 0.25, 7)
        assert status[g] == 0 and nc[g] == nh and np.array_equal(e[g], eh) and np.array_equal(t[g], th), (g, n)
        with api.Problem(0) as p:
            p.upload(c.x1, c.x2, c.d12, store=store)
            gs = p.epipolar_moments()
            es, ts, ns = p.initial_guess(80, 0.25, 7)
        assert np.abs(gm[g] - gs).max() <= 1e-12 * max(np.abs(gs).max(), 1.0), (g, n)
        if n >= 256:      # well-conditioned pairs: a 1e-13 difference in the moments does not move the consensus pick
            assert ns == nc[g] and np.abs(es - e[g]).max() <= 1e-5 and np.abs(ts - t[g]).max() <= 1e-5, (g, n)


@pytest.mark.parametrize("driver", ["device", "host"])
def test_batch_pipeline_matches_single_problem_pipelines(oracle, driver, monkeypatch):
    """sba_batch_solve_problem: the reference's per-pair pipeline (do_bundle_adjustment's initial values + solve_problem,
    .cpp:302-331, :183-217) for every pair of a batch.  Per pair it must be the chain of the single-problem entry points
    the mirror class runs -- guess, init_rot = -R_vec_out, d-only, rot-only and tran-only with the first two refined
    depths -- started from the pair's own guess: same iteration counts per stage, results to 1e-9; the oracle's stages
    give the same numbers on sampled pairs.  `host`: the predecessors of the one-launch stages (host trials, lock-step
    d-only solvers) give the same pipeline."""
    if driver == "host":
        monkeypatch.setenv("SBA_BATCH_DEVICE_GUESS", "0")
        monkeypatch.setenv("SBA_BATCH_DEVICE_DEPTH", "0")
    sizes = [3000, 2048, 999, 4097, 1500] + [1200 + 50 * g for g in range(27)]
    cs = [synthetic.full_rt(n, seed=5200 + i, sigma=2e-4, outlier_fraction=0.02) for i, n in enumerate(sizes)]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    x1 = np.concatenate([c.x1 for c in cs]); x2 = np.concatenate([c.x2 for c in cs])
    B = len(sizes)
    d0 = np.full((int(off[-1]), 2), 6.0)                                  # expected_d of the reference's command line
    with api.Batch(0) as b:
        b.upload(x1, x2, off, d0)
        e, t, nc, st = b.initial_guess(80, 0.25, 5)
        res = b.solve_problem(seed=5, want_depths=True)
        b.upload(x1, x2, off, d0)
        res2 = b.solve_problem(rot=-e, tran=t, use_initial_guess=False)    # the same pipeline from explicit start values
    assert (res["status"] == 0).all() and np.array_equal(res["guess_candidates"], nc)
    assert np.array_equal(res["rot"], res2["rot"]) and np.array_equal(res["tran"], res2["tran"])
    assert np.array_equal(res["d_uniform"][:, 0], res["d12"][off[:-1].astype(int), 0])
    assert np.array_equal(res["d_uniform"][:, 1], res["d12"][off[:-1].astype(int) + 1, 0])
    for g, c in enumerate(cs):
        lo, hi = int(off[g]), int(off[g + 1])
        rot0, tran0 = -e[g], t[g]
        with api.Problem(0) as p:
            p.upload(c.x1, c.x2, d0[lo:hi])
            d, sd = p.solve_depths(rot0, tran0)
            r1, t1, s1 = p.solve(api.MODE_ROT, rot0, tran0, d[0, 0], d[1, 0])
            r2, t2, s2 = p.solve(api.MODE_TRAN, r1, t1, d[0, 0], d[1, 0])
        got = (res["depth_stage"][g].num_iterations, res["depth_stage"][g].num_line_search_steps, res["rot_stage"][g].num_iterations,
               res["tran_stage"][g].num_iterations)
        assert got == (sd.num_iterations, sd.num_line_search_steps, s1.num_iterations, s2.num_iterations), (g, got)
        assert np.abs(res["d12"][lo:hi] - d).max() <= 1e-9 * max(1.0, np.abs(d).max())
        assert np.abs(res["rot"][g] - r2).max() <= 1e-9 and np.abs(res["tran"][g] - t2).max() <= 1e-9, g
        # sanity only: the reference's last two stages give every match the depths of the pair's first two (.cpp:941-942), so
        # the pipeline's accuracy is the reference's (a few 0.01 rad on this data), not the per-match sweep's
        assert np.abs(res["rot"][g] - c.rot_true).max() < 0.15
        if g in (0, 3):
            dd, sdd, _ = oracle.depth_solve(c.x1, c.x2, rot0, tran0, d0[lo:hi])
            ro, to, so, _ = oracle.lm_solve(api.MODE_ROT, c.x1, c.x2, rot0, tran0, dd[0, 0], dd[1, 0])
            ro2, to2, so2, _ = oracle.lm_solve(api.MODE_TRAN, c.x1, c.x2, ro, to, dd[0, 0], dd[1, 0])
            assert (sdd.num_iterations, so.num_iterations, so2.num_iterations) == (got[0], got[2], got[3])
            assert np.abs(res["rot"][g] - ro2).max() <= 1e-7 and np.abs(res["tran"][g] - to2).max() <= 1e-7


@pytest.mark.parametrize("store", [api.STORE_F64, api.STORE_F32], ids=["f64", "f32"])
@pytest.mark.parametrize("layout", ["0", "1"], ids=["contiguous", "interleaved"])
def test_batch_upload_pipelined_relayout_and_reuse(store, layout, monkeypatch):
    """sba_batch_upload moves the concatenated arrays in 16 MiB chunks through pinned staging and ONE re-layout launch per
    chunk (every row finds its pair by bisection of the offsets): chunk boundaries fall inside pairs, empty pairs repeat an
    offset, the batch need not start at row 0.  The planes must hold exactly what a per-pair upload of the same data holds:
    every pair's pack equals the single-problem pack to the bit-level tolerance of the sweep, the refined depths read back
    land at the caller's rows.  A second upload with the same offsets reuses every allocation (new data must show), and
    sba_batch_set_depths replaces the depths alone."""
    monkeypatch.setenv("SBA_BATCH_INTERLEAVE", layout)
    sizes = [400_000, 0, 350_001, 0, 0, 299_999, 7, 450_000, 1, 380_000]          # 1.88 M rows: three chunks per array
    cs, off, x1, x2, d12 = _make_pairs(sizes, seed0=8800)
    lead = 1234                                                                   # the batch starts at row 1234 of the caller's arrays
    pad3, pad2 = np.full((lead, 3), 7.0), np.full((lead, 2), 9.0)
    X1, X2, D = np.concatenate([pad3, x1]), np.concatenate([pad3, x2]), np.concatenate([pad2, d12])
    off2 = off + np.uint64(lead)
    rot = np.stack([c.rot_init for c in cs]); tran = np.stack([c.tran_init for c in cs])
    with api.Batch(0) as b:
        b.upload(X1, X2, off2, D, store=store)
        packs = b.eval(api.MODE_RT, rot, tran, depth_mode=api.DEPTH_PER_MATCH)
        for g, c in enumerate(cs):
            if sizes[g] == 0:
                assert not packs[g].any()
                continue
            with api.Problem(0) as p:
                p.upload(c.x1, c.x2, c.d12, store=store)
                ref = p.eval_pack(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
            assert np.abs(packs[g] - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1e-300), (g, sizes[g])
        # same offsets, other data: allocations are reused, the data must be the new one
        D2 = D * 1.25
        X1b = X1.copy(); X1b[lead:] = x2; X2b = X2.copy(); X2b[lead:] = x1                  # left and right swapped
        b.upload(X1b, X2b, off2, D2, store=store)
        packs_b = b.eval(api.MODE_RT, rot, tran, depth_mode=api.DEPTH_PER_MATCH)
        g = 2
        with api.Problem(0) as p:
            p.upload(cs[g].x2, cs[g].x1, cs[g].d12 * 1.25, store=store)
            ref = p.eval_pack(api.MODE_RT, cs[g].rot_init, cs[g].tran_init, depth_mode=api.DEPTH_PER_MATCH)
        assert np.abs(packs_b[g] - ref).max() <= 1e-10 * np.abs(ref).max()
        # depths alone
        b.set_depths(D)
        packs_c = b.eval(api.MODE_RT, rot, tran, depth_mode=api.DEPTH_PER_MATCH)
        with api.Problem(0) as p:
            p.upload(cs[g].x2, cs[g].x1, cs[g].d12, store=store)
            ref = p.eval_pack(api.MODE_RT, cs[g].rot_init, cs[g].tran_init, depth_mode=api.DEPTH_PER_MATCH)
        assert np.abs(packs_c[g] - ref).max() <= 1e-10 * np.abs(ref).max()
        # and back out: the refined depths of a d-only stage land at the caller's rows
        d_out, sums, status = b.solve_depths(rot, tran, options=api.default_lm_options(max_num_iterations=2))
        assert d_out.shape == D.shape and not d_out[:lead].any()
        lo, hi = int(off2[7]), int(off2[8])
        with api.Problem(0) as p:
            p.upload(cs[7].x2, cs[7].x1, cs[7].d12, store=store)
            d1, _ = p.solve_depths(cs[7].rot_init, cs[7].tran_init, options=api.default_lm_options(max_num_iterations=2))
        assert np.abs(d_out[lo:hi] - d1).max() <= 1e-9 * max(1.0, np.abs(d1).max())
    with api.Batch(0) as b2:
        with pytest.raises(api.SbaError):
            b2.set_depths(D)                                                             # nothing uploaded yet


@pytest.mark.parametrize("store", [api.STORE_F64, api.STORE_F32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", [api.KERNEL_FACTORED, api.KERNEL_EXPLICIT], ids=["factored", "explicit"])
def test_batch_lm_dynamic_shares_equals_the_one_launch_solve(oracle, store, kind, monkeypatch):
    """SBA_BATCH_DYNAMIC=1: per-pair LM as launches per iteration whose blocks are dealt out to the pairs that are still
    iterating (converged pairs hand their CUs over).  Same LmSolver source, same sweeps: iteration counts and terminations
    equal the default path's for every pair (here, with fewer pairs than CUs, the host lock-step loop; the C5-sized comparison
    with the one-launch kernel is tools/batch_pipeline_workload.py), results to 1e-11 (a pair swept by several blocks folds its sums in another
    order), for every mode / depth mode, pairs that need very different iteration counts (good and poor starts), an empty
    pair, a 1-match pair, ragged tails; and few long pairs (several blocks per pair from the start)."""
    sizes = [3000, 0, 1, 257, 5000, 64, 1023, 2] + [400 + 37 * g for g in range(40)]
    cs, off, x1, x2, d12 = _make_pairs(sizes, seed0=7100)
    B = len(sizes)
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    rot0[::3] = np.stack([c.rot_true for c in cs])[::3] + 0.25          # poor starts: many more iterations for a third of the pairs
    d1 = np.linspace(0.9, 1.4, B); d2 = np.linspace(1.2, 0.8, B)
    got = {}
    for dyn in ("0", "1"):
        monkeypatch.setenv("SBA_BATCH_DYNAMIC", dyn)
        with api.Batch(0) as b:
            b.set_kernel(kind)
            b.upload(x1, x2, off, d12, store=store)
            got[dyn] = [b.solve(mode, rot0, tran0, d1, d2, depth_mode=dm, options=api.default_lm_options(tran_param=tp))
                        for mode, dm, tp in ((api.MODE_ROT, api.DEPTH_UNIFORM, api.TRAN_FREE), (api.MODE_TRAN, api.DEPTH_PER_MATCH, api.TRAN_FREE),
                                             (api.MODE_RT, api.DEPTH_PER_MATCH, api.TRAN_SPHERE))]
    spread = 0
    for (r0, t0, s0, st0), (r1, t1, s1, st1) in zip(got["0"], got["1"]):
        assert np.array_equal(st0, st1) and (st1 == 0).all()
        assert [(q.num_iterations, q.num_successful_steps, q.termination) for q in s0] == \
               [(q.num_iterations, q.num_successful_steps, q.termination) for q in s1]
        assert np.abs(r0 - r1).max() <= 1e-11 and np.abs(t0 - t1).max() <= 1e-11
        its = [q.num_iterations for q in s1]
        spread = max(spread, max(its) - min(its))
    assert spread >= 5                                                    # the shares really changed while the solve ran
    # few long pairs: several blocks per pair from the first iteration
    sizes = [120_000, 90_001, 0, 150_000]
    cs, off, x1, x2, d12 = _make_pairs(sizes, seed0=7300)
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    opt = api.default_lm_options(tran_param=api.TRAN_SPHERE)
    monkeypatch.setenv("SBA_BATCH_DYNAMIC", "1")
    with api.Batch(0) as b:
        b.set_kernel(kind)
        b.upload(x1, x2, off, d12, store=store)
        rot, tran, sums, status = b.solve(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH, options=opt)
    assert (status == 0).all()
    for g, c in enumerate(cs):
        if sizes[g] == 0:
            continue
        with api.Problem(0) as p:
            p.set_kernel(kind)
            p.upload(c.x1, c.x2, c.d12, store=store)
            r1, t1, s1 = p.solve(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH, options=opt)
        assert sums[g].num_iterations == s1.num_iterations and np.abs(rot[g] - r1).max() <= 1e-10 and np.abs(tran[g] - t1).max() <= 1e-10


def test_batch_solve_matches_single_problem_solves(oracle):
    sizes = [4000, 0, 2500, 3333, 1, 5000]
    cs, off, x1, x2, d12 = _make_pairs(sizes)
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    with api.Batch(0) as b:
        b.upload(x1, x2, off, d12)
        for mode, tp in ((api.MODE_RT, api.TRAN_SPHERE), (api.MODE_ROT, api.TRAN_FREE)):
            opt = api.default_lm_options(tran_param=tp)
            rot, tran, sums, status = b.solve(mode, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH, options=opt)
            assert (status == 0).all()
            for g, c in enumerate(cs):
                if sizes[g] == 0:
                    assert np.array_equal(rot[g], rot0[g]) and sums[g].num_evaluations == 1
                    continue
                ro, to, so, rc = oracle.lm_solve(mode, c.x1, c.x2, c.rot_init, c.tran_init, d12=c.d12,
                                                 options=oracle.default_options(tran_param=tp))
                assert rc == 0
                assert (sums[g].num_iterations, sums[g].num_successful_steps) == (so.num_iterations, so.num_successful_steps), g
                assert np.abs(rot[g] - ro).max() <= RT_TOL_F64 and np.abs(tran[g] - to).max() <= RT_TOL_F64
                with api.Problem(0) as p:       # and the single-problem API gives the same answer
                    p.upload(c.x1, c.x2, c.d12)
                    r1, t1, s1 = p.solve(mode, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH, options=opt)
                assert np.abs(rot[g] - r1).max() <= 1e-12 and np.abs(tran[g] - t1).max() <= 1e-12


def test_batch_config_c5_full_size_properties(oracle):
    """BASELINE config C5 at FULL size -- 256 pairs x 50k matches in one launch -- through size-independent properties:
    repeated launches are bit-identical; a pair's pack does not depend on where it sits in the batch (reversed order:
    bit-identical per pair); sampled pairs equal the single-problem sweep and the per-pair oracle; all 256 LMs converge
    to their own geometry."""
    B, n = 256, 50_000
    cs, off, x1, x2, d12 = _make_pairs([n] * B, seed0=5000)
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    with api.Batch(0) as b:
        b.upload(x1, x2, off, d12)
        p1 = b.eval(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH)
        assert np.array_equal(p1, b.eval(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH))
        ms = b.sweep_launch_times(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH, repeat=5)
        assert ms.shape == (5,) and (ms > 0).all()
        # one block per pair: the step is ONE launch (batch_step_kernel); the three-kernel chain gives the same bits
        assert b.blocks_per_pair == 1 and b.step_is_fused
        ms = b.step_launch_times(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH, repeat=5)
        assert ms.shape == (5,) and (ms > 0).all()
        assert np.array_equal(p1, b.eval(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH))
        os.environ["SBA_BATCH_FUSED_STEP"] = "0"
        try:
            assert not b.step_is_fused
            assert np.array_equal(p1, b.eval(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH))
        finally:
            del os.environ["SBA_BATCH_FUSED_STEP"]
        rot, tran, sums, status = b.solve(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH,
                                          options=api.default_lm_options(tran_param=api.TRAN_SPHERE))
        assert (status == 0).all() and all(s.termination.startswith("CONVERGENCE") for s in sums)
        err = np.array([np.abs(rot[g] - cs[g].rot_true).max() for g in range(B)])
        assert err.max() < 5e-3
        assert np.allclose(np.linalg.norm(tran, axis=1), 1.0, atol=1e-12)
    for g in (0, 17, 255):                                   # sampled pairs against the single-problem path and the oracle
        with api.Problem(0) as p:
            p.upload(cs[g].x1, cs[g].x2, cs[g].d12)
            single = p.eval_pack(api.MODE_RT, cs[g].rot_init, cs[g].tran_init, depth_mode=api.DEPTH_PER_MATCH)
        ref = pack_from_eval(2, oracle.evaluate(2, cs[g].x1, cs[g].x2, cs[g].rot_init, cs[g].tran_init, d12=cs[g].d12))
        scale = np.abs(ref).max()
        assert np.abs(p1[g] - single).max() <= REL_TOL_F64 * scale and np.abs(p1[g] - ref).max() <= REL_TOL_F64 * scale
    # the same pairs uploaded in reverse order
    order = np.arange(B)[::-1]
    with api.Batch(0) as b:
        b.upload(np.concatenate([cs[g].x1 for g in order]), np.concatenate([cs[g].x2 for g in order]), off,
                 np.concatenate([cs[g].d12 for g in order]))
        p2 = b.eval(api.MODE_RT, rot0[order], tran0[order], depth_mode=api.DEPTH_PER_MATCH)
    assert np.array_equal(p2[::-1], p1)


def test_batch_pipeline_config_c5_full_size_properties(oracle):
    """The reference's per-pair pipeline (guess -> d-only -> rot-only -> tran-only) at BASELINE config C5's FULL size, 256
    pairs x 50k matches, through size-independent properties: every pair is served (a candidate, three converged stages or
    the iteration limit, finite results); a repeated run from re-sent depths is bit-identical; a pair's result does not depend
    on where it sits in the batch (reversed order: bit-identical per pair, guess included); sampled pairs equal the
    single-problem chain and the oracle's stages from the same start (iteration counts, R|t)."""
    B, n = 256, 50_000
    cs = [synthetic.full_rt(n, seed=6100 + g, sigma=2e-4, outlier_fraction=0.02) for g in range(B)]
    off = (np.arange(B + 1) * n).astype(np.uint64)
    x1 = np.concatenate([c.x1 for c in cs]); x2 = np.concatenate([c.x2 for c in cs])
    d0 = np.full((B * n, 2), 6.0)
    with api.Batch(0) as b:
        b.upload(x1, x2, off, d0)
        e, t, nc, st = b.initial_guess(80, 0.25, 3)
        res = b.solve_problem(seed=3)
        b.set_depths(d0)
        res_again = b.solve_problem(seed=3)
    assert (res["status"] == 0).all() and (res["guess_candidates"] >= 1).all()
    assert np.isfinite(res["rot"]).all() and np.isfinite(res["tran"]).all() and (res["d_uniform"] > 0).all()
    for key in ("rot", "tran", "d_uniform"):
        assert np.array_equal(res[key], res_again[key]), key
    for stage in ("depth_stage", "rot_stage", "tran_stage"):
        assert all(q.termination.startswith("CONVERGENCE") or q.termination == "NO_CONVERGENCE" for q in res[stage])
        assert [q.num_iterations for q in res[stage]] == [q.num_iterations for q in res_again[stage]]
    assert np.median([np.abs(res["rot"][g] - cs[g].rot_true).max() for g in range(B)]) < 0.05
    for g in (0, 131, 255):
        c = cs[g]
        rot0, tran0 = -e[g], t[g]
        with api.Problem(0) as p:
            p.upload(c.x1, c.x2, d0[:n])
            d, sd = p.solve_depths(rot0, tran0)
            r1, t1, s1 = p.solve(api.MODE_ROT, rot0, tran0, d[0, 0], d[1, 0])
            r2, t2, s2 = p.solve(api.MODE_TRAN, r1, t1, d[0, 0], d[1, 0])
        got = (res["depth_stage"][g].num_iterations, res["depth_stage"][g].num_line_search_steps, res["rot_stage"][g].num_iterations,
               res["tran_stage"][g].num_iterations)
        assert got == (sd.num_iterations, sd.num_line_search_steps, s1.num_iterations, s2.num_iterations), (g, got)
        assert np.abs(res["rot"][g] - r2).max() <= 1e-9 and np.abs(res["tran"][g] - t2).max() <= 1e-9
        assert res["d_uniform"][g, 0] == pytest.approx(d[0, 0], rel=1e-9) and res["d_uniform"][g, 1] == pytest.approx(d[1, 0], rel=1e-9)
        if g == 131:
            dd, sdd, _ = oracle.depth_solve(c.x1, c.x2, rot0, tran0, d0[:n])
            ro, to, so, _ = oracle.lm_solve(api.MODE_ROT, c.x1, c.x2, rot0, tran0, dd[0, 0], dd[1, 0])
            ro2, to2, so2, _ = oracle.lm_solve(api.MODE_TRAN, c.x1, c.x2, ro, to, dd[0, 0], dd[1, 0])
            assert (sdd.num_iterations, so.num_iterations, so2.num_iterations) == (got[0], got[2], got[3])
            assert np.abs(res["rot"][g] - ro2).max() <= 1e-7 and np.abs(res["tran"][g] - to2).max() <= 1e-7
    order = np.arange(B)[::-1]
    with api.Batch(0) as b:
        b.upload(np.concatenate([cs[g].x1 for g in order]), np.concatenate([cs[g].x2 for g in order]), off, d0)
        rev = b.solve_problem(seed=3)
    assert np.array_equal(rev["rot"][::-1], res["rot"]) and np.array_equal(rev["tran"][::-1], res["tran"])
    assert np.array_equal(rev["guess_candidates"][::-1], res["guess_candidates"])


@pytest.mark.parametrize("kind", [api.KERNEL_FACTORED, api.KERNEL_EXPLICIT], ids=["factored", "explicit"])
@pytest.mark.parametrize("store", [api.STORE_F64, api.STORE_F32], ids=["f64", "f32"])
def test_one_block_per_pair_paths_equal_the_chain_and_the_oracle(oracle, kind, store):
    """Many small ragged pairs get one block each, which switches on the two one-launch paths: batch_step_kernel (a whole
    evaluation step) and batch_lm_kernel (a whole per-pair solve).  Both must give the numbers of the general paths -- the
    three-kernel chain and the host lock-step LM (to the last bit or two) -- for every mode, depth mode, kernel and plane type, and the oracle's
    numbers per pair (empty, 1-match and ragged-tail pairs included)."""
    sizes = [0, 1, 2, 3, 63, 64, 65, 127, 255, 256, 257, 500, 511, 512] + [17 * g % 509 for g in range(30)]
    cs, off, x1, x2, d12 = _make_pairs(sizes, seed0=12000)
    B = len(sizes)
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    d1 = np.linspace(0.8, 1.7, B); d2 = np.linspace(1.3, 0.6, B)
    with api.Batch(0) as b:
        b.set_kernel(kind)
        b.upload(x1, x2, off, d12, store=store)
        assert b.blocks_per_pair == 1 and b.step_is_fused
        for mode in (api.MODE_ROT, api.MODE_TRAN, api.MODE_RT):
            for dm in (api.DEPTH_PER_MATCH, api.DEPTH_UNIFORM):
                for delta in (1.0, 0.0):
                    fused = b.eval(mode, rot0, tran0, d1, d2, delta, dm)
                    os.environ["SBA_BATCH_FUSED_STEP"] = "0"
                    try:
                        chain = b.eval(mode, rot0, tran0, d1, d2, delta, dm)
                    finally:
                        del os.environ["SBA_BATCH_FUSED_STEP"]
                    # same sums in the same fold order; the two kernels are separate instantiations of the sweep core,
                    # and the compiler's FMA contraction of a ragged tail may differ between them in the last bit
                    assert np.abs(fused - chain).max() <= 1e-15 * max(np.abs(chain).max(), 1e-300), (mode, dm, delta)
                for g in (0, 1, 4, 9, 13, 20):
                    c = cs[g]
                    a1, a2 = ((c.x1, c.x2) if store == api.STORE_F64 else
                              (c.x1.astype(np.float32).astype(np.float64), c.x2.astype(np.float32).astype(np.float64)))
                    ref = pack_from_eval(mode, oracle.evaluate(mode, a1, a2, c.rot_init, c.tran_init, d1[g], d2[g], 0.0,
                                                               c.d12 if dm == api.DEPTH_PER_MATCH else None))
                    scale = max(np.abs(ref).max(), 1e-300)
                    assert np.abs(fused[g] - ref).max() <= REL_TOL_F64 * scale, (mode, dm, g, sizes[g])
        # the whole solve in one launch against the host lock-step loop (same LmSolver source, same sums: same bits)
        for mode, tp in ((api.MODE_RT, api.TRAN_SPHERE), (api.MODE_ROT, api.TRAN_FREE), (api.MODE_TRAN, api.TRAN_FREE)):
            opt = api.default_lm_options(tran_param=tp)
            dev = b.solve(mode, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH, options=opt)
            os.environ["SBA_BATCH_DEVICE_LM"] = "0"
            try:
                host = b.solve(mode, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH, options=opt)
            finally:
                del os.environ["SBA_BATCH_DEVICE_LM"]
            assert np.array_equal(dev[3], host[3])
            for g in range(B):
                sd, sh = dev[2][g], host[2][g]
                if sizes[g] == 0:
                    assert np.array_equal(dev[0][g], rot0[g]) and sd.num_evaluations == 1
                if sizes[g] < 50:        # under-determined toys: sin / cos differ in the last bit between host and device,
                    continue             # which such problems may amplify into another iteration count
                assert (sd.num_iterations, sd.num_successful_steps, sd.num_evaluations, sd.termination) == \
                       (sh.num_iterations, sh.num_successful_steps, sh.num_evaluations, sh.termination), (mode, g, sizes[g])
                assert np.abs(dev[0][g] - host[0][g]).max() <= 1e-12 and np.abs(dev[1][g] - host[1][g]).max() <= 1e-12, (mode, g)
            if store == api.STORE_F64:
                for g in (11, 12, 13):
                    c = cs[g]
                    ro, to, so, rc = oracle.lm_solve(mode, c.x1, c.x2, c.rot_init, c.tran_init, d12=c.d12,
                                                     options=oracle.default_options(tran_param=tp))
                    assert rc == 0 and dev[2][g].num_iterations == so.num_iterations
                    assert np.abs(dev[0][g] - ro).max() <= RT_TOL_F64 and np.abs(dev[1][g] - to).max() <= RT_TOL_F64


def test_equi2cube_512_frames_device_resident():
    """The remap leg of config C5 at its full size: 512 ERP frames 3840x1920 -> S = 600 strips in one batched call.
    Frame 0 carries its own pixel index in its three bytes, so its output IS the source-index table; every other frame
    (random bytes) must be exactly that gather of its own pixels (checked on the device with torch)."""
    torch = pytest.importorskip("torch")
    lib = cabi.load_library()
    F, H, W, S = 512, 1920, 3840, 600
    idx = torch.arange(H * W, dtype=torch.int64, device="cuda")
    frame0 = torch.stack([idx & 255, (idx >> 8) & 255, (idx >> 16) & 255], dim=1).to(torch.uint8).reshape(H, W, 3)
    src = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device="cuda")
    src[0] = frame0
    dst = torch.zeros((F, S, 6 * S, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    cabi.check(lib, lib.sba_equi2cube_device(0, C.c_void_p(stream), C.c_void_p(src.data_ptr()), H, W, S, F,
                                             C.c_void_p(dst.data_ptr())))
    torch.cuda.synchronize()
    o = dst[0].reshape(-1, 3).to(torch.int64)
    table = o[:, 0] | (o[:, 1] << 8) | (o[:, 2] << 16)
    assert int(table.min()) >= 0 and int(table.max()) < H * W
    for k in (1, 2, 255, 256, 511):
        want = src[k].reshape(-1, 3)[table].reshape(S, 6 * S, 3)
        assert torch.equal(dst[k], want), k
    # and the table itself is the oracle's: checked against one real frame in tests/test_gpu_side.py::test_equi2cube_bit_exact
    assert lib.sba_map_table_host_decided(0, 0, S, H, W) >= 0


def test_batch_solve_host_threads_change_nothing():
    """Host threads for the per-pair LM steps (sba_set_host_threads, the reference's set_omp; engaged from 64 pairs per
    thread): every pair is stepped by exactly one thread, so nothing may change -- not even the last bit."""
    B = 200
    cs, off, x1, x2, d12 = _make_pairs([1500 + 7 * g for g in range(B)], seed0=9000)
    rot0 = np.stack([c.rot_init for c in cs]); tran0 = np.stack([c.tran_init for c in cs])
    with api.Batch(0) as b:
        b.upload(x1, x2, off, d12)
        rot, tran, sums, status = b.solve(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH)
        try:
            for threads in (2, 3, 8):
                api.set_host_threads(threads)
                rot_t, tran_t, sums_t, status_t = b.solve(api.MODE_RT, rot0, tran0, depth_mode=api.DEPTH_PER_MATCH)
                assert np.array_equal(rot_t, rot) and np.array_equal(tran_t, tran) and np.array_equal(status_t, status)
                assert [s.num_iterations for s in sums_t] == [s.num_iterations for s in sums]
        finally:
            api.set_host_threads(1)


def test_batch_errors_and_empty():
    with api.Batch(0) as b:
        with pytest.raises(api.SbaError):
            b.eval(api.MODE_ROT, np.zeros((0, 3)), np.zeros((0, 3)))             # nothing uploaded
        b.upload(np.zeros((0, 3)), np.zeros((0, 3)), np.array([0], dtype=np.uint64))
        assert b.eval(api.MODE_ROT, np.zeros((0, 3)), np.zeros((0, 3))).shape == (0, 24)
        c = synthetic.rotation_only(100, seed=1)
        b.upload(c.x1, c.x2, np.array([0, 40, 100], dtype=np.uint64))
        with pytest.raises(api.SbaError):
            b.eval(api.MODE_ROT, np.zeros((2, 3)), np.zeros((2, 3)), depth_mode=api.DEPTH_PER_MATCH)
        with pytest.raises(ValueError):
            b.upload(c.x1, c.x2, np.array([0, 400], dtype=np.uint64))


def test_equi2cube_batched_device(oracle):
    """Batched, device-resident remap (config C5: "equi2cube remap on GPU") through torch-owned memory."""
    torch = pytest.importorskip("torch")
    lib = cabi.load_library()
    B, H, W, S = 3, 240, 480, 60
    rng = np.random.default_rng(9)
    ims = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    src = torch.from_numpy(ims).cuda()
    dst = torch.zeros((B, S, 6 * S, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    cabi.check(lib, lib.sba_equi2cube_device(0, C.c_void_p(stream), C.c_void_p(src.data_ptr()), H, W, S, B,
                                             C.c_void_p(dst.data_ptr())))
    torch.cuda.synchronize()
    out = dst.cpu().numpy()
    for k in range(B):
        ref, _ = oracle.equi2cube(ims[k], S)
        assert np.array_equal(out[k], ref)
