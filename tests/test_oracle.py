"""CPU: the oracle against its golden vectors and three independent cross-checks.

Parity is UNPINNED at the Ceres boundary (the reference holds no golden vector for the BA path and
cannot be built here); what these tests pin is the oracle's restatement against (a) an independent
closed-form numpy Jacobian, (b) torch f64 autograd of the residual formula, (c) central finite
differences, and (d) the committed fixtures."""
import numpy as np
import pytest

import ref_numpy as rn
from helpers import GOLDEN
from spherical_bundle_adjuster_amd import synthetic


def test_pointwise_golden(oracle):
    z = np.load(GOLDEN / "pointwise.npz", allow_pickle=False)
    for i in range(len(z["mode"])):
        e, J = oracle.point(int(z["mode"][i]), z["x1"][i], z["x2"][i], z["rot"][i], z["tran"][i],
                            float(z["d1"][i]), float(z["d2"][i]))
        assert np.array_equal(e, z["e"][i]) and np.array_equal(J, z["J"][i]), z["case"][i]
        assert np.array_equal(oracle.huber(1.0, float(e @ e)), z["rho"][i])


def test_pointwise_vs_closed_form(oracle):
    z = np.load(GOLDEN / "pointwise.npz", allow_pickle=False)
    for i in range(len(z["mode"])):
        e, J = rn.residual_jacobian(int(z["mode"][i]), z["x1"][i], z["x2"][i], z["rot"][i], z["tran"][i],
                                    float(z["d1"][i]), float(z["d2"][i]))
        scale = max(1.0, float(z["d1"][i]))
        assert np.abs(e - z["e"][i]).max() <= 4e-15 * scale * 3, z["case"][i]
        # near theta = pi the closed form divides by sin-free terms only; 1e-13 covers cancellation
        assert np.abs(J - z["J"][i]).max() <= 1e-13 * scale, (z["case"][i], np.abs(J - z["J"][i]).max())


def test_small_angle_branch_is_first_order(oracle):
    # theta^2 <= DBL_EPSILON: R p = p + w x p exactly, dRp/dw = -[p]x exactly
    w = np.array([3e-9, -4e-9, 1e-9])
    p = np.array([0.3, -0.5, 0.81])
    assert np.array_equal(oracle.rotate(w, p), p + np.cross(w, p))
    e, J = oracle.point(0, p, p, w, np.zeros(3), 1.0, 1.0)
    assert np.array_equal(J[:, :3], rn.skew(p))     # e = x2 - R x1  ->  de/dw = -(-[p]x) = [p]x


def test_jacobian_vs_finite_differences(oracle):
    rng = np.random.default_rng(5)
    for _ in range(20):
        x1, x2 = synthetic._sphere(rng, 1)[0], synthetic._sphere(rng, 1)[0]
        rot, tran = rng.standard_normal(3) * 0.5, rng.standard_normal(3)
        d1, d2 = rng.uniform(0.5, 5, 2)
        _, J = oracle.point(2, x1, x2, rot, tran, d1, d2)
        h = 1e-6
        for k in range(6):
            dp = np.zeros(6); dp[k] = h
            ep, _ = oracle.point(2, x1, x2, rot + dp[:3], tran + dp[3:], d1, d2)
            em, _ = oracle.point(2, x1, x2, rot - dp[:3], tran - dp[3:], d1, d2)
            assert np.abs((ep - em) / (2 * h) - J[:, k]).max() < 5e-9


def test_jacobian_vs_torch_autograd(oracle):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(6)

    def residual(p, x1, x2, d1, d2):
        w, t = p[:3], p[3:]
        th = torch.sqrt((w * w).sum())
        k = w / th
        X = torch.tensor(x1, dtype=torch.float64) * d1
        Xr = X * torch.cos(th) + torch.linalg.cross(k, X) * torch.sin(th) + k * (k @ X) * (1 - torch.cos(th))
        return torch.tensor(x2, dtype=torch.float64) * d2 - (Xr - t)

    for _ in range(10):
        x1, x2 = synthetic._sphere(rng, 1)[0], synthetic._sphere(rng, 1)[0]
        p0 = np.concatenate([rng.standard_normal(3) * 0.4, rng.standard_normal(3)])
        d1, d2 = rng.uniform(0.5, 5, 2)
        Jt = torch.autograd.functional.jacobian(lambda p: residual(p, x1, x2, d1, d2),
                                                torch.tensor(p0, dtype=torch.float64)).numpy()
        e, J = oracle.point(2, x1, x2, p0[:3], p0[3:], d1, d2)
        assert np.abs(J - Jt).max() < 1e-13 * max(1, d1)
        assert np.abs(e - residual(torch.tensor(p0), x1, x2, d1, d2).numpy()).max() < 1e-14 * max(1, d1, d2)


def test_huber_is_blockwise_ceres_semantics(oracle):
    # inlier region: identity; outlier: rho = 2 a sqrt(s) - a^2, rho' = a / sqrt(s), rho'' < 0
    assert np.array_equal(oracle.huber(1.0, 0.25), [0.25, 1.0, 0.0])
    assert np.array_equal(oracle.huber(1.0, 1.0), [1.0, 1.0, 0.0])       # s == b is still inlier
    r = oracle.huber(1.0, 4.0)
    assert r[0] == 3.0 and r[1] == 0.5 and r[2] == -0.5 / 8.0
    r = oracle.huber(0.5, 4.0)
    assert r[0] == 2 * 0.5 * 2 - 0.25 and r[1] == 0.25


@pytest.mark.parametrize("n", [1, 63, 64, 65, 2048])
def test_reduction_golden(oracle, n):
    z = np.load(GOLDEN / "reductions.npz", allow_pickle=False)
    x1, x2, d12 = z[f"n{n}_x1"], z[f"n{n}_x2"], z[f"n{n}_d12"]
    rot, tran = z[f"n{n}_rot"], z[f"n{n}_tran"]
    for mode in (0, 1, 2):
        for dm, dd in (("u", None), ("p", d12)):
            ev = oracle.evaluate(mode, x1, x2, rot, tran, d1=1.3, d2=0.9, delta=1.0, d12=dd, threads=1)
            assert np.array_equal(ev.H, z[f"n{n}_m{mode}_{dm}_H"])
            assert np.array_equal(ev.g, z[f"n{n}_m{mode}_{dm}_g"])
            assert np.array_equal([ev.cost, ev.sum_w, ev.n_outlier], z[f"n{n}_m{mode}_{dm}_scalars"])
            # threads only change the (long double) summation order
            ev8 = oracle.evaluate(mode, x1, x2, rot, tran, d1=1.3, d2=0.9, delta=1.0, d12=dd, threads=8)
            assert np.abs(ev8.H - ev.H).max() <= 1e-15 * max(np.abs(ev.H).max(), 1e-300)


@pytest.mark.parametrize("n", [1, 65, 400])
def test_reduction_vs_closed_form(oracle, n):
    z = np.load(GOLDEN / "reductions.npz", allow_pickle=False)
    nn = 2048 if n == 400 else n
    x1, x2, d12 = z[f"n{nn}_x1"][:n], z[f"n{nn}_x2"][:n], z[f"n{nn}_d12"][:n]
    rot, tran = z[f"n{nn}_rot"], z[f"n{nn}_tran"]
    for mode in (0, 1, 2):
        for dd in (None, d12):
            H, g, cost, sw, nout = rn.normal_equations(mode, x1, x2, rot, tran, 1.3, 0.9, 1.0, dd)
            for ev in (oracle.evaluate(mode, x1, x2, rot, tran, 1.3, 0.9, 1.0, dd),
                       oracle.evaluate_hoisted(mode, x1, x2, rot, tran, 1.3, 0.9, 1.0, dd)):
                assert np.abs(ev.H - H).max() <= 1e-13 * max(np.abs(H).max(), 1e-300)
                assert np.abs(ev.g - g).max() <= 1e-12 * max(np.abs(g).max(), 1e-300)
                assert abs(ev.cost - cost) <= 1e-13 * cost and ev.n_outlier == nout and abs(ev.sum_w - sw) <= 1e-12 * sw


def test_empty_problem(oracle):
    ev = oracle.evaluate(2, np.zeros((0, 3)), np.zeros((0, 3)), [0.1, 0.2, 0.3], [0, 0, 1])
    assert not ev.H.any() and not ev.g.any() and ev.cost == 0


def test_lm_golden_and_known_answer(oracle):
    z = np.load(GOLDEN / "solves.npz", allow_pickle=False)
    r, t, s, rc = oracle.lm_solve(0, z["c1_x1"], z["c1_x2"], z["c1_rot0"], z["c1_tran0"], threads=1)
    assert rc == 0 and np.array_equal(r, z["c1_rot"]) and np.array_equal(t, z["c1_tran"])
    assert [s.termination, s.num_iterations, s.num_successful_steps] == list(z["c1_meta"][:3])
    # noise-free, outlier-free problems return the generating R|t
    c = synthetic.rotation_only(512, seed=77, sigma=0.0, outlier_fraction=0.0)
    o = oracle.default_options(function_tolerance=1e-30, parameter_tolerance=1e-14, gradient_tolerance=1e-16)
    r, _, s, rc = oracle.lm_solve(0, c.x1, c.x2, c.rot_init, c.tran_init, options=o)
    assert rc == 0 and np.abs(r - c.rot_true).max() < 1e-12
    c = synthetic.full_rt(512, seed=78, sigma=0.0, outlier_fraction=0.0)
    for tp in (0, 1):
        o = oracle.default_options(function_tolerance=1e-30, parameter_tolerance=1e-14, gradient_tolerance=1e-16,
                                   tran_param=tp)
        r, t, s, rc = oracle.lm_solve(2, c.x1, c.x2, c.rot_init, c.tran_init, d12=c.d12, options=o)
        assert rc == 0 and np.abs(r - c.rot_true).max() < 1e-10 and np.abs(t - c.tran_true).max() < 1e-10
        if tp == 1:
            assert abs(np.linalg.norm(t) - 1.0) < 1e-14


def test_lm_minimiser_vs_independent_numpy_cost_and_scipy(oracle):
    """The point the LM restatement converges to is a minimiser of the block-wise Huber objective as an INDEPENDENT numpy
    restatement computes it (cost written from scratch here: Rodrigues via scipy, s = |e|^2 per 3-vector block,
    rho(s) = s / 2 delta sqrt(s) - delta^2): the central-difference gradient of that cost vanishes there and scipy's
    BFGS, started from it, cannot lower the cost.  (scipy's own loss='huber' is per component, not per block -- not usable
    as an oracle)"""
    from scipy.optimize import minimize
    from scipy.spatial.transform import Rotation

    c = synthetic.full_rt(3000, seed=91)                       # noise + 5 % outliers: the Huber region is populated
    delta = 1.0

    def cost(p):
        R = Rotation.from_rotvec(p[:3]).as_matrix()
        e = c.d12[:, 1:2] * c.x2 - c.d12[:, 0:1] * (c.x1 @ R.T) + p[3:]
        s = np.einsum("ij,ij->i", e, e)
        rho = np.where(s <= delta * delta, s, 2.0 * delta * np.sqrt(np.maximum(s, 1e-300)) - delta * delta)
        return 0.5 * rho.sum()

    o = oracle.default_options(function_tolerance=1e-16, parameter_tolerance=1e-14, gradient_tolerance=1e-14,
                               max_num_iterations=200)
    r, t, s, rc = oracle.lm_solve(2, c.x1, c.x2, c.rot_init, c.tran_init, d12=c.d12, options=o)
    assert rc == 0
    p_lm = np.concatenate([r, t])
    assert abs(cost(p_lm) - s.final_cost) <= 1e-12 * s.final_cost       # the two cost definitions agree
    h = 1e-6
    grad = np.array([(cost(p_lm + h * np.eye(6)[k]) - cost(p_lm - h * np.eye(6)[k])) / (2 * h) for k in range(6)])
    p0 = np.concatenate([c.rot_init, c.tran_init])
    grad0 = np.array([(cost(p0 + h * np.eye(6)[k]) - cost(p0 - h * np.eye(6)[k])) / (2 * h) for k in range(6)])
    assert np.abs(grad).max() <= 1e-6 * np.abs(grad0).max()
    res = minimize(cost, p_lm, method="BFGS", options={"gtol": 1e-10, "maxiter": 200})
    assert res.fun >= cost(p_lm) * (1 - 1e-12)
    assert np.abs(res.x - p_lm).max() < 1e-6
    # and from the start point scipy reaches the same minimiser (same basin)
    res0 = minimize(cost, p0, method="BFGS", options={"gtol": 1e-9, "maxiter": 500})
    assert np.abs(res0.x - p_lm).max() < 1e-5 and abs(res0.fun - cost(p_lm)) <= 1e-9 * cost(p_lm)


def test_side_paths_basic(oracle):
    # pixel -> sphere: unit vectors, known pixels
    kp = np.zeros((3, 7), dtype=np.float32)
    kp[:, 0] = [0.0, 960.0, 1920.0]
    kp[:, 1] = [480.0, 480.0, 0.0]
    v = oracle.keypoints_to_sphere(kp, 3840, 960)
    assert np.allclose(np.linalg.norm(v, axis=1), 1.0, atol=1e-15)
    assert np.allclose(v[0], [1, 0, 0], atol=1e-15) and np.allclose(v[1], [0, 1, 0], atol=1e-15)
    assert np.allclose(v[2], [0, 0, 1], atol=1e-15)
    # equi2cube on an index-coded image: every output pixel must be a copy of some input pixel
    H, W, S = 64, 128, 16
    im = np.zeros((H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    im[..., 0], im[..., 1], im[..., 2] = yy, xx, (yy * 7 + xx * 3) % 251
    out, clamped = oracle.equi2cube(im, S)
    # Reference quirk (equi2cube.cpp:47-50, no clamp): for every EVEN cube size the bottom-face centre
    # (i = j = S/2) looks exactly at the south pole, theta = pi -> row = H, one row past the image.
    # The oracle (and the HIP kernel) clamp that single pixel to row H-1.
    assert clamped == 1 and int(H * np.arccos(-1.0) / np.pi) == H
    _, clamped_odd = oracle.equi2cube(im, S + 1)
    assert clamped_odd == 0
    assert np.array_equal(out[..., 2], (out[..., 0].astype(int) * 7 + out[..., 1].astype(int) * 3) % 251)
    # top face centre looks at the north pole (row 0), bottom face at the south pole
    assert out[S // 2, 4 * S + S // 2, 0] <= 1 and out[S // 2, 5 * S + S // 2, 0] >= H - 2


def test_depth_stage_oracle(oracle):
    """d-only stage restatement (.cpp:1004-1063): with the true R|t and clean data the depths are recovered up to
    the bias of the lambda*exp(-c*d) regularisers; bounds are respected; a start at the optimum stops at once."""
    c = synthetic.full_rt(300, seed=71, sigma=0.0, outlier_fraction=0.0)
    d, s, rc = oracle.depth_solve(c.x1, c.x2, c.rot_true, c.tran_true, np.full((300, 2), 4.0))
    assert rc == 0 and s.termination in (1, 2, 3) and s.final_cost < s.initial_cost
    assert np.median(np.abs(d - c.d12)) < 0.05 and (d >= 0).all()
    # regulariser only (x1 = x2 = 0 direction impossible on the sphere, so use tiny lambda): pure reprojection fit
    d2, s2, rc = oracle.depth_solve(c.x1, c.x2, c.rot_true, c.tran_true, np.full((300, 2), 4.0), lam=1e-9)
    assert rc == 0 and np.abs(d2 - c.d12).max() < 1e-3 * c.d12.max()
    # outliers push some depths onto the bound d = 0
    c = synthetic.full_rt(500, seed=6)
    d3, s3, rc = oracle.depth_solve(c.x1, c.x2, c.rot_init, c.tran_init, np.ones((500, 2)))
    assert rc == 0 and d3.min() == 0.0 and (d3 >= 0).all()
    # restart from the result: converges without moving
    d4, s4, rc = oracle.depth_solve(c.x1, c.x2, c.rot_init, c.tran_init, d3)
    assert rc == 0 and s4.num_successful_steps <= 1 and np.abs(d4 - d3).max() < 1e-3


def test_depth_stage_vs_scipy_per_match(oracle):
    """Run to tight tolerances, the one global bounded problem (.cpp:1004-1063) must land on the per-match optima:
    each match's two depths minimise its own five residuals subject to d >= 0 -- checked with scipy's bounded
    least_squares on independently written residuals (Rodrigues via scipy)."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation

    c = synthetic.full_rt(60, seed=8)                              # noise + outliers: some depths end on the bound
    o = oracle.default_options(function_tolerance=1e-16, parameter_tolerance=1e-13, gradient_tolerance=1e-13,
                               max_num_iterations=500)
    d, s, rc = oracle.depth_solve(c.x1, c.x2, c.rot_init, c.tran_init, np.full((60, 2), 3.0), options=o)
    assert rc == 0
    R = Rotation.from_rotvec(c.rot_init).as_matrix()
    on_bound = 0
    for i in range(60):
        q = R @ c.x1[i]

        def res(dd):
            return np.concatenate([dd[1] * c.x2[i] - dd[0] * q + c.tran_init, [np.exp(-dd[0]), np.exp(-dd[1])]])
        best = min((least_squares(res, start, bounds=(0.0, np.inf), xtol=1e-15, ftol=1e-15, gtol=1e-15)
                    for start in ([3.0, 3.0], np.maximum(d[i], 1e-3))), key=lambda r_: r_.cost)
        mine = 0.5 * np.sum(res(d[i]) ** 2)
        assert mine <= best.cost * (1 + 1e-9) + 1e-15, (i, mine, best.cost)   # never worse than scipy's optimum
        assert np.abs(d[i] - best.x).max() <= 1e-5 * max(1.0, np.abs(best.x).max()), (i, d[i], best.x)
        on_bound += int((d[i] == 0.0).any())
    assert (d >= 0).all()


def test_matcher_coordinate_maps_oracle(oracle):
    """rotate_keypoint / crop_rotated_image / cube2equi_pixel restatements: consistency properties."""
    H, W, S = 480, 960, 120
    # pitch 0: rotate_keypoint only adds the band offset H*3/8 and truncates like Vec2i
    kp = np.zeros((5, 7), dtype=np.float32)
    kp[:, 0] = [0.0, 10.7, 500.2, 959.0, 333.3]
    kp[:, 1] = [0.0, 5.5, 60.9, 119.0, 77.7]
    out = oracle.rotate_keypoints(kp, 0.0, W, H)
    assert np.abs(out[:, 0] - np.floor(kp[:, 0])).max() <= 1 and np.abs(out[:, 1] - np.floor(kp[:, 1] + H * 3 // 8)).max() <= 1
    # crop at pitch 0 is (up to the truncation round trip) the plain equatorial band im(roi) of do_all (.cpp:131-139)
    rng = np.random.default_rng(2)
    im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    band = oracle.crop_rotated_image(im, 0.0)
    assert band.shape == (H // 4, W, 3)
    same = (band == im[H * 3 // 8: H * 3 // 8 + H // 4]).all(axis=2).mean()
    assert same > 0.5            # the int truncation of acos/atan2 round trips shifts some pixels by one
    # rotating a band key-point by +45 then looking the pixel up in the rotated crop is consistent with the warp
    band45 = oracle.crop_rotated_image(im, 45.0)
    kq = np.zeros((1, 7), dtype=np.float32); kq[0, 0], kq[0, 1] = 200.0, 30.0
    back = oracle.rotate_keypoints(kq, 45.0, W, H)
    assert (band45[30, 200] == im[int(back[0, 1]), int(back[0, 0])]).all()
    # cube2equi_pixel: the centre of each cube face maps to the ERP pixel of that face's axis
    centres = np.zeros((6, 7), dtype=np.float32)
    centres[:, 0] = [S / 2 + k * S for k in range(6)]; centres[:, 1] = S / 2
    e = oracle.cube2equi_keypoints(centres, S, W, H)
    #            left(+y)   front(-x)  right(-y)   back(+x)  top(+z)   bottom(-z)
    assert np.allclose(e[:4, 1], H / 2, atol=1e-3)
    assert np.allclose(e[:4, 0], [W / 4, W / 2, 3 * W / 4, 0.0], atol=1e-3)
    assert e[4, 1] < 1e-3 and abs(e[5, 1] - H) < 1e-3


def test_timed_cpu_baseline_variant_matches_checker(oracle):
    """bench.py times evaluate_f64 (double accumulation); it must compute the same thing as the long-double checker."""
    c = synthetic.full_rt(20000, seed=99)
    for mode in (0, 1, 2):
        a = oracle.evaluate(mode, c.x1, c.x2, c.rot_init, c.tran_init, d12=c.d12)
        b = oracle.evaluate_f64(mode, c.x1, c.x2, c.rot_init, c.tran_init, d12=c.d12)
        assert np.abs(a.H - b.H).max() <= 1e-12 * np.abs(a.H).max() and abs(a.cost - b.cost) <= 1e-12 * a.cost
        assert np.abs(a.g - b.g).max() <= 1e-11 * max(np.abs(a.g).max(), 1e-300) and a.n_outlier == b.n_outlier


@pytest.mark.parametrize("mode,n,seed", [(0, 400, 31), (1, 400, 32), (2, 400, 33), (2, 150, 34), (0, 60, 35)])
def test_lm_trajectory_vs_independent_numpy_restatement(oracle, mode, n, seed):
    """The oracle's LM loop against a numpy restatement written from Ceres' documented schedule with an independent
    Jacobian (closed form) and numpy.linalg.solve: same termination, iteration and accepted-step counts, R|t to 1e-9.
    (The product's host LM is held against the oracle iterate for iterate in tests/test_host_lm_cpu.py; this test is what
    keeps that pair from being twins that agree with each other and with nothing else.)"""
    import ref_lm_numpy as rl
    gen = synthetic.full_rt if mode else synthetic.rotation_only
    c = gen(n, seed=seed)
    d12 = c.d12 if mode else None
    r_o, t_o, s_o, rc = oracle.lm_solve(mode, c.x1, c.x2, c.rot_init, c.tran_init, d12=d12)
    assert rc == 0
    r_n, t_n, info = rl.solve(mode, c.x1, c.x2, c.rot_init, c.tran_init, d12=d12)
    term = {1: "function", 2: "gradient", 3: "parameter", 4: "no_convergence", 5: "min_radius"}[s_o.termination]
    assert (term, s_o.num_iterations, s_o.num_successful_steps) == (info["termination"], info["iterations"], info["successful"])
    assert np.abs(r_o - r_n).max() <= 1e-9 and np.abs(t_o - t_n).max() <= 1e-9
    assert abs(s_o.final_cost - info["cost"]) <= 1e-10 * info["cost"] and abs(s_o.initial_cost - info["initial_cost"]) <= 1e-10 * info["initial_cost"]


def test_depth_stage_and_maps_golden(oracle):
    """The committed fixtures of the d-only stage (with and without Ceres' line search) and of the coordinate maps:
    the oracle must reproduce them exactly (same build flags, same libm)."""
    z = np.load(GOLDEN / "depth_stage.npz", allow_pickle=False)
    for name in ("ls_d1", "ls_d005", "ls_reg", "plain_d2"):
        d0, lam, c_ = z[f"{name}_cfg"]
        n = len(z[f"{name}_x1"])
        for tag, ls in (("ceres", 20), ("nols", 0)):
            d, s, rc = oracle.depth_solve(z[f"{name}_x1"], z[f"{name}_x2"], z[f"{name}_rot"], z[f"{name}_tran"],
                                          np.full((n, 2), d0), lam=lam, c=c_,
                                          options=oracle.default_options(max_num_line_search_step_size_iterations=ls))
            meta = z[f"{name}_{tag}_meta"]
            assert rc == 0 and np.array_equal(d, z[f"{name}_{tag}_d"]), (name, tag)
            assert (s.termination, s.num_iterations, s.num_successful_steps, s.num_line_search_steps) == tuple(int(v) for v in meta[:4])
    assert z["ls_d1_ceres_meta"][3] >= 1 and z["ls_d1_nols_meta"][1] > z["ls_d1_ceres_meta"][1]     # the search matters here
    assert np.array_equal(z["plain_d2_ceres_d"], z["plain_d2_nols_d"])                                # ... and not here
    m = np.load(GOLDEN / "maps.npz", allow_pickle=False)
    H, W, S = (int(v) for v in m["geometry"])
    for pitch in (45.0, -45.0, -90.0, 0.0):
        tag = f"p{int(pitch)}".replace("-", "m")
        assert np.array_equal(oracle.rotate_keypoints(m["kp"], pitch, W, H)[:, :2], m[f"rotate_{tag}"])
        assert np.array_equal(oracle.crop_rotated_image(m["im"], pitch), m[f"crop_{tag}"])
    assert np.array_equal(oracle.cube2equi_keypoints(m["cube_kp"], S, W, H)[:, :2], m["cube2equi"])
    assert np.array_equal(oracle.equi2cube(m["im"], S)[0], m["equi2cube"])
