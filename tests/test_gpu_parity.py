"""GPU (-m gpu): the HIP path, called through the C-ABI, against the oracle on the same seeded inputs,
against the committed golden fixtures, and through size-independent properties at full size.

Tolerances (helpers.py): normal-equation entries rel 1e-12 (f64 planes) / 5e-6 (f32 planes) of the
block's largest entry; recovered R|t 1e-9 (f64) / 1e-5 (f32)."""
import numpy as np
import pytest

from helpers import GOLDEN, REL_TOL_F32, REL_TOL_F64, RT_TOL_F32, RT_TOL_F64, assert_normal_eq_close, pack_from_eval
from spherical_bundle_adjuster_amd import api, synthetic

pytestmark = pytest.mark.gpu

MODES = [api.MODE_ROT, api.MODE_TRAN, api.MODE_RT]


@pytest.fixture(scope="module", params=[api.KERNEL_FACTORED, api.KERNEL_EXPLICIT], ids=["factored", "explicit"])
def problem(request):
    """Every parity test runs against both sweep kernels: the factored (moment) form and the explicit
    per-match Jacobian form must each match the oracle."""
    p = api.Problem(0)
    p.set_kernel(request.param)
    yield p
    p.close()


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 2048, 100003])
def test_sweep_matches_oracle(problem, oracle, n):
    c = synthetic.full_rt(n, seed=2000 + n, outlier_fraction=0.1)
    problem.upload(c.x1, c.x2, c.d12)
    assert problem.size == n
    for mode in MODES:
        for dm, d12 in ((api.DEPTH_UNIFORM, None), (api.DEPTH_PER_MATCH, c.d12)):
            got = problem.eval(mode, c.rot_init, c.tran_init, 1.3, 0.9, 1.0, dm)
            ref = oracle.evaluate(mode, c.x1, c.x2, c.rot_init, c.tran_init, 1.3, 0.9, 1.0, d12)
            assert_normal_eq_close(got, ref, REL_TOL_F64, f"n={n} mode={mode} depth={dm}")
            assert got.n_outlier == ref.n_outlier
            assert abs(got.sum_w - ref.sum_w) <= REL_TOL_F64 * max(ref.sum_w, 1) or mode == api.MODE_ROT


def test_reduction_golden_fixtures(problem):
    z = np.load(GOLDEN / "reductions.npz", allow_pickle=False)
    for n in (1, 63, 64, 65, 2048):
        problem.upload(z[f"n{n}_x1"], z[f"n{n}_x2"], z[f"n{n}_d12"])
        for mode in MODES:
            for dm, key in ((api.DEPTH_UNIFORM, "u"), (api.DEPTH_PER_MATCH, "p")):
                got = problem.eval(mode, z[f"n{n}_rot"], z[f"n{n}_tran"], 1.3, 0.9, 1.0, dm)
                H, g, sc = z[f"n{n}_m{mode}_{key}_H"], z[f"n{n}_m{mode}_{key}_g"], z[f"n{n}_m{mode}_{key}_scalars"]
                assert np.abs(got.H - H).max() <= REL_TOL_F64 * max(np.abs(H).max(), 1e-300)
                assert np.abs(got.g - g).max() <= 10 * REL_TOL_F64 * max(np.abs(g).max(), 1e-300)
                assert abs(got.cost - sc[0]) <= REL_TOL_F64 * sc[0] and got.n_outlier == sc[2]


def test_pointwise_golden_fixtures(problem):
    """n = 1 sweeps reproduce e and J of single residual blocks (incl. theta = 0, theta^2 < eps, theta ~ pi)."""
    z = np.load(GOLDEN / "pointwise.npz", allow_pickle=False)
    for i in range(len(z["mode"])):
        mode = int(z["mode"][i])
        problem.upload(z["x1"][i][None], z["x2"][i][None])
        got = problem.eval(mode, z["rot"][i], z["tran"][i], float(z["d1"][i]), float(z["d2"][i]), 0.0)   # no loss
        e, J = z["e"][i], z["J"][i]
        scale = max(np.abs(J.T @ J).max(), 1e-300)
        # The fixtures carry the Ceres formula's own derivative noise (~eps/theta) just above its small-angle
        # threshold (tests/test_host_rotation_cpu.py); the device kernels are compared up to that noise.
        th = float(np.linalg.norm(z["rot"][i]))
        tol = 1e-13 + (0.0 if th * th <= np.finfo(float).eps else 8 * np.finfo(float).eps / th)
        assert np.abs(got.H - J.T @ J).max() <= tol * scale, (z["case"][i], mode)
        assert np.abs(got.g - J.T @ e).max() <= tol * max(np.abs(J.T @ e).max(), scale), (z["case"][i], mode)
        assert abs(got.cost - 0.5 * e @ e) <= 1e-14 * max(e @ e, 1e-300)
        # with Huber(1): cost = rho/2, weight = rho'
        goth = problem.eval(mode, z["rot"][i], z["tran"][i], float(z["d1"][i]), float(z["d2"][i]), 1.0)
        rho = z["rho"][i]
        assert abs(goth.cost - 0.5 * rho[0]) <= 4e-16 * max(rho[0], 1e-300) * 8
        assert np.abs(goth.H - rho[1] * (J.T @ J)).max() <= tol * scale


@pytest.mark.parametrize("delta", [0.0, 0.05, 1.0, 100.0])
def test_huber_regions(problem, oracle, delta):
    """delta = 0: no loss; 0.05: nearly everything is an outlier; 100: nothing is."""
    c = synthetic.rotation_only(5000, seed=91, outlier_fraction=0.2)
    problem.upload(c.x1, c.x2)
    for mode in MODES:
        got = problem.eval(mode, c.rot_init, [0.01, -0.02, 0.03], 1.0, 1.0, delta)
        ref = oracle.evaluate(mode, c.x1, c.x2, c.rot_init, [0.01, -0.02, 0.03], 1.0, 1.0, delta)
        assert_normal_eq_close(got, ref, REL_TOL_F64, f"delta={delta} mode={mode}")
        assert got.n_outlier == ref.n_outlier
    if delta == 100.0:
        assert got.n_outlier == 0
    if delta == 0.05:
        assert got.n_outlier > 4000


def test_empty_and_reupload(problem, oracle):
    problem.upload(np.zeros((0, 3)), np.zeros((0, 3)))
    got = problem.eval(api.MODE_RT, [0.1, 0.2, 0.3], [0, 0, 1.0])
    assert not got.H.any() and not got.g.any() and got.cost == 0 and problem.size == 0
    r, t, s = problem.solve(api.MODE_RT, [0.1, 0.2, 0.3], [0, 0, 1.0])
    assert s.termination == "CONVERGENCE_GRADIENT" and s.num_evaluations == 1
    # a smaller upload after a larger one must not see stale data
    c = synthetic.rotation_only(777, seed=5)
    problem.upload(c.x1, c.x2)
    c2 = synthetic.rotation_only(33, seed=6)
    problem.upload(c2.x1, c2.x2)
    got = problem.eval(api.MODE_ROT, c2.rot_init, c2.tran_init)
    assert_normal_eq_close(got, oracle.evaluate(0, c2.x1, c2.x2, c2.rot_init, c2.tran_init), REL_TOL_F64)


def test_reuploads_reuse_or_replace_the_resident_planes(oracle):
    """A handle keeps its plane allocations from upload to upload while they fit (one handle fed image pair after image
    pair): every sequence of sizes, storage types and with / without per-match depths must give the oracle's numbers --
    no stale tail from a larger predecessor (f32 vectors hold 4 matches, f64 vectors 2: the padding differs), depths gone
    when an upload brings none, the d-only stage's scratch following along."""
    steps = [(5000, api.STORE_F64, True), (4097, api.STORE_F32, True), (4999, api.STORE_F64, False), (63, api.STORE_F32, False),
             (40000, api.STORE_F64, True), (6, api.STORE_F64, True), (39999, api.STORE_F32, True)]
    with api.Problem(0) as p:
        for k, (n, store, with_d) in enumerate(steps):
            c = synthetic.full_rt(n, seed=7700 + k)
            x1, x2 = (c.x1, c.x2) if store == api.STORE_F64 else (c.x1.astype(np.float32).astype(np.float64),
                                                                  c.x2.astype(np.float32).astype(np.float64))
            p.upload(c.x1, c.x2, c.d12 if with_d else None, store=store)
            tol = REL_TOL_F64
            got = p.eval(api.MODE_ROT, c.rot_init, c.tran_init, 1.3, 0.7)
            assert_normal_eq_close(got, oracle.evaluate(0, x1, x2, c.rot_init, c.tran_init, 1.3, 0.7), tol, f"step {k} uniform")
            if with_d:
                got = p.eval(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
                assert_normal_eq_close(got, oracle.evaluate(2, x1, x2, c.rot_init, c.tran_init, d12=c.d12), tol, f"step {k} per match")
                if n >= 1000 and store == api.STORE_F64:
                    d, sd = p.solve_depths(c.rot_init, c.tran_init)
                    dref, sref, rc = oracle.depth_solve(c.x1, c.x2, c.rot_init, c.tran_init, c.d12)
                    assert rc == 0 and sd.num_iterations == sref.num_iterations
                    assert np.abs(d - dref).max() <= 1e-9 * max(1.0, np.abs(dref).max())
            else:
                with pytest.raises(api.SbaError):
                    p.eval(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)


def test_error_behaviour(problem):
    c = synthetic.rotation_only(10, seed=1)
    problem.upload(c.x1, c.x2)                                    # no per-match depths uploaded
    with pytest.raises(api.SbaError) as ei:
        problem.eval(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
    assert ei.value.code == -1
    with pytest.raises(api.SbaError):
        problem.eval(5, c.rot_init, c.tran_init)
    with pytest.raises(api.SbaError):
        problem.eval(api.MODE_ROT, [np.nan, 0, 0], c.tran_init)
    with pytest.raises(api.SbaError):
        api.Problem(99)
    fresh = api.Problem(0)
    with pytest.raises(api.SbaError) as ei:
        fresh.eval(api.MODE_ROT, c.rot_init, c.tran_init)
    assert ei.value.code == -4                                     # SBA_ERR_NOT_UPLOADED
    fresh.close()


def test_f32_planes(problem, oracle):
    c = synthetic.full_rt(50001, seed=17)
    problem.upload(c.x1, c.x2, c.d12, store=api.STORE_F32)
    x1f, x2f = c.x1.astype(np.float32).astype(np.float64), c.x2.astype(np.float32).astype(np.float64)
    for mode in MODES:
        got = problem.eval(mode, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
        # exact statement: f64 arithmetic on the f32-rounded inputs
        assert_normal_eq_close(got, oracle.evaluate(mode, x1f, x2f, c.rot_init, c.tran_init, d12=c.d12), REL_TOL_F64,
                               f"f32 planes mode={mode}")
        # and close to the f64-input answer
        assert_normal_eq_close(got, oracle.evaluate(mode, c.x1, c.x2, c.rot_init, c.tran_init, d12=c.d12), REL_TOL_F32)


# ---- LM solves ---------------------------------------------------------------------------------------
def test_solve_golden_and_oracle_lm(problem, oracle):
    z = np.load(GOLDEN / "solves.npz", allow_pickle=False)
    problem.upload(z["c1_x1"], z["c1_x2"])                        # config C1 twin: 2048 matches, rot-only
    r, t, s = problem.solve(api.MODE_ROT, z["c1_rot0"], z["c1_tran0"])
    assert np.abs(r - z["c1_rot"]).max() <= RT_TOL_F64 and np.array_equal(t, z["c1_tran0"])
    assert [s.num_iterations, s.num_successful_steps] == list(z["c1_meta"][1:3])
    problem.upload(z["rt_x1"], z["rt_x2"], z["rt_d12"])
    for name, tp in (("rt6", api.TRAN_FREE), ("rt5", api.TRAN_SPHERE)):
        r, t, s = problem.solve(api.MODE_RT, z["rt_rot0"], z["rt_tran0"], depth_mode=api.DEPTH_PER_MATCH,
                                options=api.default_lm_options(tran_param=tp))
        assert np.abs(r - z[f"{name}_rot"]).max() <= RT_TOL_F64 and np.abs(t - z[f"{name}_tran"]).max() <= RT_TOL_F64
        assert [s.num_iterations, s.num_successful_steps] == list(z[f"{name}_meta"][1:3])
        assert s.num_evaluations == s.num_iterations + 1          # one sweep per LM iteration


def test_three_stage_order_like_solve_problem(problem, oracle):
    """rot-only then tran-only with the uniform init_d quirk (.cpp:202-209, :941-942, :998-999)."""
    c = synthetic.full_rt(4000, seed=23, outlier_fraction=0.02)
    problem.upload(c.x1, c.x2, c.d12)
    d1, d2 = float(c.d12[0, 0]), float(c.d12[1, 0])               # init_d[0][0], init_d[1][0]
    r, t, _ = problem.solve(api.MODE_ROT, c.rot_init, c.tran_init, d1, d2)
    r2, t2, _ = problem.solve(api.MODE_TRAN, r, t, d1, d2)
    ro, to, _, _ = oracle.lm_solve(0, c.x1, c.x2, c.rot_init, c.tran_init, d1, d2)
    ro2, to2, _, _ = oracle.lm_solve(1, c.x1, c.x2, ro, to, d1, d2)
    assert np.abs(r - ro).max() <= RT_TOL_F64 and np.array_equal(r2, r)
    assert np.abs(t2 - to2).max() <= RT_TOL_F64 and np.array_equal(t, c.tran_init)


def test_known_answer_noise_free(problem):
    tight = dict(function_tolerance=1e-30, parameter_tolerance=1e-14, gradient_tolerance=1e-16)
    c = synthetic.rotation_only(20000, seed=61, sigma=0.0, outlier_fraction=0.0)
    problem.upload(c.x1, c.x2)
    r, _, s = problem.solve(api.MODE_ROT, c.rot_init, c.tran_init, options=api.default_lm_options(**tight))
    assert np.abs(r - c.rot_true).max() <= RT_TOL_F64
    c = synthetic.full_rt(20000, seed=62, sigma=0.0, outlier_fraction=0.0)
    problem.upload(c.x1, c.x2, c.d12)
    for tp in (api.TRAN_FREE, api.TRAN_SPHERE):
        r, t, s = problem.solve(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH,
                                options=api.default_lm_options(tran_param=tp, **tight))
        assert np.abs(r - c.rot_true).max() <= RT_TOL_F64 and np.abs(t - c.tran_true).max() <= RT_TOL_F64
    problem.upload(c.x1, c.x2, c.d12, store=api.STORE_F32)
    r, t, s = problem.solve(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH,
                            options=api.default_lm_options(**tight))
    assert np.abs(r - c.rot_true).max() <= RT_TOL_F32 and np.abs(t - c.tran_true).max() <= RT_TOL_F32


# ---- size-independent properties at BASELINE.json's full single-GPU sizes ------------------------------
@pytest.mark.parametrize("n,gen,mode,dm", [(1_000_000, "rot", api.MODE_ROT, api.DEPTH_UNIFORM),
                                           (10_000_000, "rt", api.MODE_RT, api.DEPTH_PER_MATCH)])
def test_full_size_properties(oracle, n, gen, mode, dm):
    c = synthetic.rotation_only(n, seed=synthetic.BASE_SEED + 1) if gen == "rot" else synthetic.full_rt(n)
    d12 = c.d12 if dm == api.DEPTH_PER_MATCH else None
    with api.Problem(0) as p:
        p.upload(c.x1, c.x2, d12)
        p.set_kernel(api.KERNEL_FACTORED)                  # whatever SBA_KERNEL says: the comparisons below are bitwise
        full = p.eval_pack(mode, c.rot_init, c.tran_init, depth_mode=dm)
        # the two kernels (factored moments vs explicit per-match Jacobian) agree at full size
        p.set_kernel(api.KERNEL_EXPLICIT)
        full_explicit = p.eval_pack(mode, c.rot_init, c.tran_init, depth_mode=dm)
        p.set_kernel(api.KERNEL_FACTORED)
        assert np.abs(full - full_explicit).max() <= REL_TOL_F64 * np.abs(full).max()
        assert full[23] == full_explicit[23]
        # determinism: the fixed-order reduction makes repeated sweeps bit-identical
        assert np.array_equal(full, p.eval_pack(mode, c.rot_init, c.tran_init, depth_mode=dm))
        # oracle on a bounded prefix + additivity over shards (the multi-GPU sharding identity)
        k = 200_000
        p.upload(c.x1[:k], c.x2[:k], None if d12 is None else d12[:k])
        head = p.eval_pack(mode, c.rot_init, c.tran_init, depth_mode=dm)
        ref = pack_from_eval(mode, oracle.evaluate(mode, c.x1[:k], c.x2[:k], c.rot_init, c.tran_init,
                                                   d12=None if d12 is None else d12[:k]))
        assert np.abs(head - ref).max() <= REL_TOL_F64 * np.abs(ref).max()
        p.upload(c.x1[k:], c.x2[k:], None if d12 is None else d12[k:])
        tail = p.eval_pack(mode, c.rot_init, c.tran_init, depth_mode=dm)
        assert np.abs(head + tail - full).max() <= REL_TOL_F64 * np.abs(full).max()
        assert head[23] + tail[23] == full[23]                      # outlier counts are exact integers
        # the generating R|t is (nearly) stationary: gradient tiny relative to the start point's
        p.upload(c.x1, c.x2, d12)
        at_truth = p.eval_pack(mode, c.rot_true, c.tran_true, depth_mode=dm)
        assert at_truth[22] < full[22]
        # and the solve stage converges to it within the noise level
        r, t, s = p.solve(mode, c.rot_init, c.tran_init, depth_mode=dm)
        assert s.termination.startswith("CONVERGENCE") and np.abs(r - c.rot_true).max() < 5e-3


def test_config_c4_full_size_on_one_gpu_and_the_eight_shard_identity(oracle):
    """BASELINE config C4: 100 M rotation-only correspondences, sharded 8 x 12.5 M with ONE all-reduce of the pack per
    sweep.  All of it fits one MI355X (6 planes x 800 MB = 4.8 GB of 288 GB), so the identity the sharding rests on --
    one residual block per match, every block sharing init_rot (reference spherical_bundle_adjuster.cpp:921-945), hence
    H, g, cost are plain sums over matches -- is checked at the real size: the packs of the 8 contiguous shards
    (synthetic.shard_range, what `bench.py --gpus 8` gives its ranks: 12.5 M each) sum to the full problem's pack to
    1e-12, outlier counts exactly.  Plus determinism, factored == explicit, the oracle on a bounded prefix, and the LM
    on a 12.5 M shard converging to the generating rotation.  The 4.8 GB of synthetic input are drawn ON the device
    (torch, seeded: numpy needs ~2 minutes of host time for them) with synthetic.rotation_only's recipe and handed
    over with sba_problem_upload_device -- the C-ABI's device-resident upload, cv::Point3d layout."""
    import torch
    world, per = 8, 12_500_000
    n = world * per
    c0 = synthetic.rotation_only(8, seed=synthetic.BASE_SEED + 3)        # ground truth + start point of the geometry
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(synthetic.BASE_SEED + 3)
    R = torch.tensor(synthetic.rodrigues(c0.rot_true), dtype=torch.float64, device=dev)
    x1 = torch.randn((n, 3), dtype=torch.float64, device=dev, generator=g)
    x1 /= x1.norm(dim=1, keepdim=True)
    x2 = x1 @ R.T
    x2 += 1e-3 * torch.randn((n, 3), dtype=torch.float64, device=dev, generator=g)
    x2 /= x2.norm(dim=1, keepdim=True)
    out = torch.rand(n, device=dev, generator=g) < 0.05                   # 5 % outliers: uniform points on the sphere
    rnd = torch.randn((int(out.sum().item()), 3), dtype=torch.float64, device=dev, generator=g)
    x2[out] = rnd / rnd.norm(dim=1, keepdim=True)
    del rnd, out
    torch.cuda.synchronize()
    assert [synthetic.shard_range(n, r, world) for r in range(world)] == [(r * per, (r + 1) * per) for r in range(world)]
    mode, dm = api.MODE_ROT, api.DEPTH_UNIFORM

    def upload(p, lo, hi):
        p.upload_device(x1.data_ptr() + 24 * lo, x2.data_ptr() + 24 * lo, None, hi - lo)

    with api.Problem(0) as p:
        upload(p, 0, n)
        assert p.size == n
        p.set_kernel(api.KERNEL_FACTORED)
        full = p.eval_pack(mode, c0.rot_init, c0.tran_init, depth_mode=dm)
        assert np.array_equal(full, p.eval_pack(mode, c0.rot_init, c0.tran_init, depth_mode=dm))      # deterministic
        p.set_kernel(api.KERNEL_EXPLICIT)
        full_explicit = p.eval_pack(mode, c0.rot_init, c0.tran_init, depth_mode=dm)
        p.set_kernel(api.KERNEL_FACTORED)
        assert np.abs(full - full_explicit).max() <= REL_TOL_F64 * np.abs(full).max() and full[23] == full_explicit[23]
        # uniform outliers further than 60 degrees from R x1 have |e|^2 = 2 - 2 cos > 1: 3/4 of the 5 % sit in Huber's linear region
        assert abs(full[23] / n - 0.0375) < 0.001
        at_truth = p.eval_pack(mode, c0.rot_true, c0.tran_true, depth_mode=dm)
        assert at_truth[22] < full[22]
        total = np.zeros(24)
        for r in range(world):
            lo, hi = synthetic.shard_range(n, r, world)
            upload(p, lo, hi)
            part = p.eval_pack(mode, c0.rot_init, c0.tran_init, depth_mode=dm)
            assert np.array_equal(part, p.eval_pack(mode, c0.rot_init, c0.tran_init, depth_mode=dm))
            total += part
            if r == world - 1:       # the shard rank 7 of the 8-GPU run holds: the LM converges on it
                rr, tt, s = p.solve(mode, c0.rot_init, c0.tran_init, depth_mode=dm)
                assert s.termination.startswith("CONVERGENCE") and np.abs(rr - c0.rot_true).max() < 5e-3
        assert np.abs(total - full).max() <= REL_TOL_F64 * np.abs(full).max()
        assert total[23] == full[23]                                   # outlier counts are exact integers
        k = 200_000
        upload(p, 0, k)
        head = p.eval_pack(mode, c0.rot_init, c0.tran_init, depth_mode=dm)
    h1, h2 = x1[:k].cpu().numpy(), x2[:k].cpu().numpy()
    ref = pack_from_eval(mode, oracle.evaluate(mode, h1, h2, c0.rot_init, c0.tran_init))
    assert np.abs(head - ref).max() <= REL_TOL_F64 * np.abs(ref).max()


# ---- d-only stage (first stage of solve_problem) ----------------------------------------------------------
@pytest.mark.parametrize("n,store", [(1, api.STORE_F64), (65, api.STORE_F64), (5000, api.STORE_F64),
                                     (5000, api.STORE_F32), (200001, api.STORE_F64)])
def test_depth_stage_matches_oracle(oracle, n, store):
    c = synthetic.full_rt(n, seed=300 + n)
    d0 = np.full((n, 2), 3.0)
    x1, x2 = (c.x1, c.x2) if store == api.STORE_F64 else (c.x1.astype(np.float32).astype(np.float64),
                                                          c.x2.astype(np.float32).astype(np.float64))
    dref, sref, rc = oracle.depth_solve(x1, x2, c.rot_init, c.tran_init, d0)
    assert rc == 0
    with api.Problem(0) as p:
        p.upload(c.x1, c.x2, d0, store=store)
        d, s = p.solve_depths(c.rot_init, c.tran_init)
        assert (s.num_iterations, s.num_successful_steps) == (sref.num_iterations, sref.num_successful_steps)
        assert s.termination == {1: "CONVERGENCE_FUNCTION", 2: "CONVERGENCE_GRADIENT", 3: "CONVERGENCE_PARAMETER",
                                 4: "NO_CONVERGENCE"}[sref.termination]
        assert np.abs(d - dref).max() <= 1e-9 * max(1.0, np.abs(dref).max()) and (d >= 0).all()
        assert abs(s.final_cost - sref.final_cost) <= 1e-10 * sref.final_cost
        # the refined depths stay on the device for the following stages
        got = p.eval(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
        ref = oracle.evaluate(2, x1, x2, c.rot_init, c.tran_init, d12=dref)
        assert_normal_eq_close(got, ref, 1e-8, "after depth stage")


# Ceres line-searches every step of this bounds-constrained stage (max_num_line_search_step_size_iterations = 20 by
# default, reference .cpp:334-338 + :1060-1061).  Starts / regularisers whose full step fails the Armijo test, so that
# contraction passes really run (tests/test_depth_line_search_cpu.py pins the same cases oracle-vs-numpy on the CPU).
LS_CASES = [(500, 6, 1.0, 1.0, 1.0), (400, 9, 0.05, 1.0, 1.0), (300, 11, 0.01, 1.0, 1.0), (64, 21, 0.05, 5.0, 3.0),
            (64, 23, 0.01, 20.0, 4.0), (64, 24, 0.2, 3.0, 8.0), (100003, 6, 1.0, 1.0, 1.0)]


@pytest.mark.parametrize("n,seed,d0,lam,c", LS_CASES)
@pytest.mark.parametrize("ls", [20, 0], ids=["ceres_default", "no_line_search"])
def test_depth_stage_line_search_matches_oracle(oracle, n, seed, d0, lam, c, ls):
    cs = synthetic.full_rt(n, seed=seed)
    start = np.full((n, 2), d0)
    dref, sref, rc = oracle.depth_solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, start, lam=lam, c=c,
                                        options=oracle.default_options(max_num_line_search_step_size_iterations=ls))
    assert rc == 0
    with api.Problem(0) as p:
        p.upload(cs.x1, cs.x2, start)
        d, s = p.solve_depths(cs.rot_init, cs.tran_init, lam=lam, c=c,
                              options=api.default_lm_options(max_num_line_search_step_size_iterations=ls))
    assert (s.num_iterations, s.num_successful_steps, s.num_line_search_steps) == \
        (sref.num_iterations, sref.num_successful_steps, sref.num_line_search_steps)
    if ls and n <= 500:
        assert s.num_line_search_steps >= 1            # the contraction path really ran on the device
    assert s.termination == {1: "CONVERGENCE_FUNCTION", 2: "CONVERGENCE_GRADIENT", 3: "CONVERGENCE_PARAMETER",
                             4: "NO_CONVERGENCE"}[sref.termination]
    assert np.abs(d - dref).max() <= 1e-7 * max(1.0, np.abs(dref).max()) and (d >= 0).all()
    assert abs(s.final_cost - sref.final_cost) <= 1e-10 * sref.final_cost


def test_depth_stage_golden_fixtures():
    """The HIP d-only stage against the COMMITTED fixtures (tests/golden/depth_stage.npz: Ceres' default and the
    no-line-search variant): counts equal, depths to 1e-7 of the largest depth."""
    z = np.load(GOLDEN / "depth_stage.npz", allow_pickle=False)
    for name in ("ls_d1", "ls_d005", "ls_reg", "plain_d2"):
        d0, lam, c_ = (float(v) for v in z[f"{name}_cfg"])
        n = len(z[f"{name}_x1"])
        for tag, ls in (("ceres", 20), ("nols", 0)):
            with api.Problem(0) as p:
                p.upload(z[f"{name}_x1"], z[f"{name}_x2"], np.full((n, 2), d0))
                d, s = p.solve_depths(z[f"{name}_rot"], z[f"{name}_tran"], lam=lam, c=c_,
                                      options=api.default_lm_options(max_num_line_search_step_size_iterations=ls))
            meta = z[f"{name}_{tag}_meta"]
            assert (s.num_iterations, s.num_successful_steps, s.num_line_search_steps) == tuple(int(v) for v in meta[1:4]), (name, tag)
            ref = z[f"{name}_{tag}_d"]
            assert np.abs(d - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max()), (name, tag)
            assert abs(s.final_cost - meta[4]) <= 1e-10 * meta[4]


def test_depth_stage_line_search_failure_keeps_full_step(oracle):
    """A search that cannot succeed (sufficient decrease 1 - 1e-12 can only be met by an exactly linear cost) leaves the
    step unscaled after the allowed contractions, like Ceres (`if (line_search_summary.success) delta *= ...`): the
    candidate planes are restored by one more pass at step size 1."""
    cs = synthetic.full_rt(257, seed=6)
    start = np.ones((257, 2))
    kw = dict(max_num_line_search_step_size_iterations=3, line_search_sufficient_function_decrease=1.0 - 1e-12,
              max_num_iterations=4)
    dref, sref, rc = oracle.depth_solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, start, options=oracle.default_options(**kw))
    with api.Problem(0) as p:
        p.upload(cs.x1, cs.x2, start)
        d, s = p.solve_depths(cs.rot_init, cs.tran_init, options=api.default_lm_options(**kw))
    assert sref.num_line_search_steps >= 3
    assert (s.num_iterations, s.num_successful_steps, s.num_line_search_steps) == \
        (sref.num_iterations, sref.num_successful_steps, sref.num_line_search_steps)
    assert np.abs(d - dref).max() <= 1e-9 * max(1.0, np.abs(dref).max())


@pytest.mark.parametrize("n", [5, 6, 4097, 4098])
def test_depth_stage_f32_planes_ragged_tail_stays_clean(oracle, n):
    """f32 coordinate planes hold 4 matches per 16-byte vector, the depth planes 2: with n % 4 in {1, 2} the last sweep
    vector reaches past the last depth pair.  After an ODD number of accepted steps the candidate planes are the
    problem's depth planes -- their padding must be zero like an uploaded plane's, or a per-match sweep turns NaN."""
    cs = synthetic.full_rt(n, seed=77)
    x1, x2 = cs.x1.astype(np.float32).astype(np.float64), cs.x2.astype(np.float32).astype(np.float64)
    for max_it in (1, 2, 3):                       # odd and even numbers of accepted steps
        start = np.full((n, 2), 3.0)
        with api.Problem(0) as p:
            p.upload(cs.x1, cs.x2, start, store=api.STORE_F32)
            d, s = p.solve_depths(cs.rot_init, cs.tran_init, options=api.default_lm_options(max_num_iterations=max_it))
            assert s.num_successful_steps == max_it
            got = p.eval(api.MODE_RT, cs.rot_init, cs.tran_init, depth_mode=api.DEPTH_PER_MATCH)
        ref = oracle.evaluate(2, x1, x2, cs.rot_init, cs.tran_init, d12=d)
        assert np.isfinite(got.H).all() and np.isfinite(got.g).all() and np.isfinite(got.cost)
        assert_normal_eq_close(got, ref, 1e-10, f"n={n} after {max_it} accepted steps")


def test_depth_stage_errors():
    c = synthetic.rotation_only(10, seed=1)
    with api.Problem(0) as p:
        p.upload(c.x1, c.x2)                       # no depths
        with pytest.raises(api.SbaError):
            p.solve_depths(c.rot_init, c.tran_init)
        p.upload(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 2)))
        d, s = p.solve_depths(c.rot_init, c.tran_init)
        assert d.shape == (0, 2) and s.termination == "CONVERGENCE_GRADIENT"


def test_fused_and_two_kernel_reduction_agree(oracle, monkeypatch):
    """The in-kernel fused last-block reduction (SBA_FUSED=1), the sweep + publishing finalize kernel (SBA_FUSED=0)
    and the legacy D2H copy (SBA_PUBLISH=0) give the same pack (different but fixed fold orders: equal to
    rounding), each deterministic."""
    c = synthetic.full_rt(300007, seed=404)
    packs = {}
    for fused, publish in (("1", "1"), ("0", "1"), ("0", "0"), ("2", "1")):
        monkeypatch.setenv("SBA_FUSED", fused)
        monkeypatch.setenv("SBA_PUBLISH", publish)
        with api.Problem(0) as p:
            p.upload(c.x1, c.x2, c.d12)
            a = p.eval_pack(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
            for _ in range(20):      # repeated launches reuse the arrival counter
                assert np.array_equal(a, p.eval_pack(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH))
            b, _, _ = p.eval_timed(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH, repeat=7)
            assert np.array_equal(a, b)
            packs[fused + publish] = a
    assert np.array_equal(packs["01"], packs["00"]) and np.array_equal(packs["01"], packs["21"])   # same kernels
    assert np.abs(packs["11"] - packs["01"]).max() <= 1e-13 * np.abs(packs["01"]).max()
    ref = pack_from_eval(2, oracle.evaluate(2, c.x1, c.x2, c.rot_init, c.tran_init, d12=c.d12))
    assert np.abs(packs["11"] - ref).max() <= REL_TOL_F64 * np.abs(ref).max()


# ---- degenerate inputs -----------------------------------------------------------------------------------
def test_degenerate_inputs_fail_loudly_or_terminate_sanely(oracle):
    with api.Problem(0) as p:
        # NaN in the data: the sweep returns a non-finite cost and the solve refuses it (no silent garbage)
        c = synthetic.rotation_only(1000, seed=3)
        bad = c.x1.copy(); bad[17, 1] = np.nan
        p.upload(bad, c.x2)
        assert not np.isfinite(p.eval(api.MODE_ROT, c.rot_init, c.tran_init).cost)
        with pytest.raises(api.SbaError) as ei:
            p.solve(api.MODE_ROT, c.rot_init, c.tran_init)
        assert ei.value.code == -6
        # rank-deficient geometry: every match is the same point -> rotation about that axis is unobservable.
        # The damped system stays positive definite; the solve must end in a finite state, like the oracle's.
        x = np.tile(np.array([[0.0, 0.6, 0.8]]), (500, 1))
        p.upload(x, x)
        r, t, s = p.solve(api.MODE_ROT, [0.01, 0.02, -0.01], [0, 0, 0])
        ro, to, so, rc = oracle.lm_solve(0, x, x, [0.01, 0.02, -0.01], [0, 0, 0])
        assert rc == 0 and np.isfinite(r).all() and s.termination != "FAILURE"
        assert np.abs(r - ro).max() < 1e-9 and s.num_iterations == so.num_iterations
        # a rotation of nearly pi: the start point sits where sin(theta)/theta is small but well conditioned
        c = synthetic.rotation_only(5000, seed=8, sigma=0.0, outlier_fraction=0.0)
        axis = c.rot_true / np.linalg.norm(c.rot_true)
        big = axis * 3.0
        x2 = c.x1 @ synthetic.rodrigues(big).T
        p.upload(c.x1, x2)
        r, t, s = p.solve(api.MODE_ROT, big + 0.02, [0, 0, 0], options=api.default_lm_options(huber_delta=0.0,
                          function_tolerance=1e-30, parameter_tolerance=1e-14, gradient_tolerance=1e-16))
        assert np.abs(r - big).max() < 1e-9
