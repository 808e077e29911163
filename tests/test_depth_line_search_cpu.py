"""CPU: Ceres' projected Armijo line search in the bounded d-only stage.

The reference's d-only problem sets lower bounds (spherical_bundle_adjuster.cpp:1060-1061) and leaves
Solver::Options::max_num_line_search_step_size_iterations at its default of 20 (.cpp:334-338), so Ceres' trust-region
minimizer line-searches every step of that stage.  Parity is UNPINNED (no Ceres here, no fixture in the reference), so
three independently written restatements are held against each other:
  * oracle/ceres_line_search.hpp  -- Ceres' own procedure: dense full-pivot LU for the monomial coefficients, all complex
    roots of the derivative, real parts tested;
  * csrc/sba_line_search.hpp (the product's host code, through tests/harness) -- Newton-form Hermite interpolant, real
    critical points by recursive isolation + bisection;
  * tests/ref_depth_numpy.py -- numpy.linalg.solve + numpy.roots.
"""
import ctypes as C

import numpy as np
import pytest

import ref_depth_numpy as rd
from helpers import lm_harness
from spherical_bundle_adjuster_amd import _cabi as cabi
from spherical_bundle_adjuster_amd import synthetic

TERM = {1: "function", 2: "gradient", 3: "parameter", 4: "no_convergence"}


def product_armijo(phi, cost0, slope0, dmax=1.0, **opt):
    h = lm_harness()
    o = cabi.LmOptions()
    h.harness_default_options(C.byref(o))
    for k, v in opt.items():
        setattr(o, k, v)
    cb_t = C.CFUNCTYPE(C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)

    def _cb(a, v, g, _u):
        v[0], g[0] = map(float, phi(a))
        return 0
    out = (C.c_double * 3)()
    h.harness_armijo.restype = C.c_int
    rc = h.harness_armijo(C.byref(o), C.c_double(cost0), C.c_double(slope0), C.c_double(dmax), cb_t(_cb), None, out)
    assert rc == 0
    return bool(out[0]), float(out[1]), int(out[2])


def product_hermite_argmin(samples, lo, hi):
    h = lm_harness()
    a = np.ascontiguousarray(samples, dtype=np.float64)
    h.harness_hermite_argmin.restype = C.c_double
    return h.harness_hermite_argmin(a.ctypes.data_as(C.c_void_p), C.c_int(a.shape[0]), C.c_double(lo), C.c_double(hi))


def test_interpolating_polynomial_and_its_minimum(oracle):
    """Cubic (two samples) and quintic (three samples) Hermite interpolants: oracle's LU solve == numpy's solve; the
    minimiser on [1e-3 a, 0.6 a] agrees between oracle (complex roots), product (real critical points) and numpy."""
    rng = np.random.default_rng(3)
    for trial in range(300):
        k = 2 + trial % 2
        cur_x = rng.uniform(0.02, 0.6)                      # as in a real search: current <= 0.6 * previous <= 0.6
        xs = np.array([0.0, cur_x] + ([min(1.0, cur_x / rng.uniform(0.05, 0.6))] if k == 3 else []))
        f0, g0 = rng.uniform(1, 100), -rng.uniform(0.1, 50)
        samples = [(0.0, f0, g0)] + [(x, f0 + rng.uniform(0, 30) * x, rng.uniform(-40, 80)) for x in xs[1:]]
        want = rd.hermite_polynomial(samples)
        got = oracle.interpolating_polynomial(samples)
        assert np.abs(got - want).max() <= 1e-8 * np.abs(want).max(), (trial, got, want)   # Vandermonde conditioning
        for (x, f, fp) in samples:                                  # it really interpolates values and slopes
            assert abs(np.polyval(got, x) - f) <= 1e-9 * max(1.0, abs(f))
            assert abs(np.polyval(np.polyder(got), x) - fp) <= 1e-8 * max(1.0, abs(fp), np.abs(got).max())
        cur = samples[1][0]
        lo, hi = 1e-3 * cur, 0.6 * cur
        x_np = rd.minimize_on_interval(want, lo, hi)
        x_orc, v_orc = oracle.minimize_polynomial(got, lo, hi)
        x_prod = product_hermite_argmin(samples, lo, hi)
        v = lambda x: np.polyval(want, x)
        scale = max(abs(v(lo)), abs(v(hi)), 1.0)
        # the three pick the same minimum value; the minimiser itself agrees unless the polynomial is flat there
        assert abs(v(x_orc) - v(x_np)) <= 1e-9 * scale and abs(v(x_prod) - v(x_np)) <= 1e-9 * scale, (trial, x_np, x_orc, x_prod)
        grid = np.linspace(lo, hi, 2001)
        assert v(x_prod) <= np.min(v(grid)) + 1e-9 * scale          # and it IS the minimum over the interval
        if abs(x_orc - x_np) > 1e-7 or abs(x_prod - x_np) > 1e-7:
            assert abs(v(x_orc) - v(x_prod)) <= 1e-12 * scale


PHIS = {
    # (phi(a) -> (value, slope), cost0, slope0): every one needs at least one contraction from a = 1
    "steep_power": (lambda a: (10.0 - 4.0 * a + 900.0 * a ** 6, -4.0 + 5400.0 * a ** 5), 10.0, -4.0),
    "exp_wall": (lambda a: (5.0 - 2.0 * a + np.expm1(12.0 * a) * 1e-2, -2.0 + 0.12 * np.exp(12.0 * a)), 5.0, -2.0),
    "kink_projection": (lambda a: (3.0 - a + 40.0 * max(a - 0.07, 0.0) ** 2, -1.0 + 80.0 * max(a - 0.07, 0.0)), 3.0, -1.0),
    "oscillating": (lambda a: (2.0 - 0.5 * a + 0.6 * np.sin(25 * a) ** 2 + 3 * a * a, -0.5 + 15 * np.sin(50 * a) + 6 * a), 2.0, -0.5),
    "tiny_basin": (lambda a: (1.0 - 1e-3 * a + 1e4 * a ** 4, -1e-3 + 4e4 * a ** 3), 1.0, -1e-3),
}


@pytest.mark.parametrize("name", sorted(PHIS))
def test_armijo_three_way(oracle, name):
    """Multi-contraction searches (the quintic, three-sample interpolation path): same accepted step size and the same
    number of contractions from the oracle, the product's host code and the numpy restatement."""
    phi, f0, g0 = PHIS[name]
    ok_o, a_o, it_o = oracle.armijo(phi, f0, g0)
    ok_p, a_p, it_p = product_armijo(phi, f0, g0)
    ok_n, a_n, it_n = rd.armijo(phi, f0, g0, 1.0)
    assert ok_o and ok_p and ok_n
    assert it_o == it_p == it_n and it_o >= 1, (it_o, it_p, it_n)
    assert abs(a_o - a_p) <= 1e-9 * a_p, (a_o, a_p)
    # numpy.linalg.solve on the monomial (Vandermonde) system is the least accurate of the three when the trial sizes
    # are tiny (a^5 ~ 1e-13 next to 1): 1e-6
    assert abs(a_o - a_n) <= 1e-6 * a_n and abs(a_p - a_n) <= 1e-6 * a_n, (a_o, a_p, a_n)
    assert phi(a_p)[0] <= f0 + 1e-4 * g0 * a_p                         # Armijo holds at the accepted size
    if name in ("steep_power", "tiny_basin"):
        assert it_o >= 2                                                # really went through the three-sample path


def test_armijo_failure_modes(oracle):
    """No sufficient decrease anywhere: 20 contractions then failure (delta left unscaled); step below
    min_line_search_step_size / |delta|_inf: failure at once.  Product and oracle agree."""
    up = lambda a: (2.0 + a, 1.0)              # a jump up although the initial slope claims descent
    for fn in (oracle.armijo, product_armijo):
        ok, a, it = fn(up, 1.0, -1.0, 1e300)    # |delta|_inf huge: the minimum-step test never fires
        assert not ok and a == 1.0 and it == 20
        ok, a, it = fn(up, 1.0, -1.0, 1e-12)    # |delta|_inf = 1e-12: the first contraction is already too small
        assert not ok and a == 1.0 and it == 1
    ok, a, it = product_armijo(up, 1.0, -1.0, 1e300, max_num_line_search_step_size_iterations=3)
    assert not ok and it == 3
    ok, a, it = oracle.armijo(up, 1.0, -1.0, 1e300, options=oracle.default_options(max_num_line_search_step_size_iterations=3))
    assert not ok and it == 3
    # non-finite value: halve (clamped into [1e-3 a, 0.6 a])
    wall = lambda a: (np.inf, np.nan) if a > 0.3 else (1.0 - a, -1.0)
    for fn in (oracle.armijo, product_armijo):
        ok, a, it = fn(wall, 1.0, -1.0)
        assert ok and a == 0.25 and it == 2


CASES = [   # (n, seed, initial depth, lambda, c): starts / regularisers that make full steps fail Armijo
    (500, 6, 1.0, 1.0, 1.0), (400, 9, 0.05, 1.0, 1.0), (300, 11, 0.01, 1.0, 1.0), (64, 21, 0.05, 5.0, 3.0),
    (64, 23, 0.01, 20.0, 4.0), (64, 24, 0.2, 3.0, 8.0), (300, 12, 2.0, 1.0, 1.0),
]


@pytest.mark.parametrize("n,seed,d0,lam,c", CASES)
def test_depth_stage_oracle_vs_numpy_trajectory(oracle, n, seed, d0, lam, c):
    """The oracle's bounded trust-region + line-search loop against the numpy restatement: same termination, iteration,
    accepted-step and contraction counts, depths to 1e-7 (relative to the largest depth)."""
    cs = synthetic.full_rt(n, seed=seed)
    start = np.full((n, 2), d0)
    d, s, rc = oracle.depth_solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, start, lam=lam, c=c)
    d_np, info = rd.solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, start, lam=lam, c=c)
    assert rc == 0
    assert (TERM[s.termination], s.num_iterations, s.num_successful_steps, s.num_line_search_steps) == \
        (info["termination"], info["iterations"], info["successful"], info["line_search_steps"])
    assert np.abs(d - d_np).max() <= 1e-7 * max(1.0, np.abs(d_np).max())
    assert abs(s.final_cost - info["cost"]) <= 1e-10 * info["cost"]
    assert (d >= 0).all()


def test_line_search_changes_the_trajectory(oracle):
    """Where the old restatement (no line search: max_num_line_search_step_size_iterations = 0) and Ceres' default (20)
    part ways: when the full projected step fails the sufficient-decrease test, Ceres contracts it and (usually) accepts,
    where a bare trust-region loop rejects the step and shrinks the radius."""
    cs = synthetic.full_rt(500, seed=6)
    start = np.ones((500, 2))
    d_ls, s_ls, _ = oracle.depth_solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, start)
    d_no, s_no, _ = oracle.depth_solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, start,
                                       options=oracle.default_options(max_num_line_search_step_size_iterations=0))
    assert s_ls.num_line_search_steps >= 1 and s_no.num_line_search_steps == 0
    assert s_ls.num_iterations < s_no.num_iterations                     # 8 against 23 on this problem
    assert s_no.num_iterations - s_no.num_successful_steps >= 3          # the bare loop burns rejected steps instead
    assert np.abs(d_ls - d_no).max() > 1e-6                              # different stopping points (function tolerance)
    # ... of the same minimisation: costs agree to the function tolerance
    assert abs(s_ls.final_cost - s_no.final_cost) <= 1e-5 * s_no.final_cost
    # a start from which every full step passes Armijo: the two restatements coincide exactly
    start = np.full((500, 2), 3.0)
    a = oracle.depth_solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, start)
    b = oracle.depth_solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, start,
                           options=oracle.default_options(max_num_line_search_step_size_iterations=0))
    if a[1].num_line_search_steps == 0:
        assert np.array_equal(a[0], b[0]) and a[1].num_iterations == b[1].num_iterations


def test_depth_stage_with_line_search_reaches_the_bounded_minimum(oracle):
    """Run to tight tolerances from a start that triggers the line search AND puts depths on the bound: the result
    is the per-match bounded optimum scipy's trust-region-reflective solver finds on independently written residuals."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation

    n = 80
    cs = synthetic.full_rt(n, seed=6)
    o = oracle.default_options(function_tolerance=1e-16, parameter_tolerance=1e-13, gradient_tolerance=1e-13,
                               max_num_iterations=500)
    d, s, rc = oracle.depth_solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, np.full((n, 2), 0.05), options=o)
    assert rc == 0 and s.num_line_search_steps >= 1
    R = Rotation.from_rotvec(cs.rot_init).as_matrix()
    on_bound = 0
    for i in range(n):
        q = R @ cs.x1[i]

        def res(dd):
            return np.concatenate([dd[1] * cs.x2[i] - dd[0] * q + cs.tran_init, [np.exp(-dd[0]), np.exp(-dd[1])]])
        best = min((least_squares(res, st, bounds=(0.0, np.inf), xtol=1e-15, ftol=1e-15, gtol=1e-15)
                    for st in ([3.0, 3.0], [0.05, 0.05], np.maximum(d[i], 1e-3))), key=lambda r_: r_.cost)
        mine = 0.5 * np.sum(res(d[i]) ** 2)
        assert mine <= best.cost * (1 + 1e-9) + 1e-15, (i, mine, best.cost)
        assert np.abs(d[i] - best.x).max() <= 1e-5 * max(1.0, np.abs(best.x).max()), (i, d[i], best.x)
        on_bound += int((d[i] == 0.0).any())
    assert (d >= 0).all()


def test_committed_depth_fixture_cases_and_line_search(oracle):
    """Which of the d-only cases the other tests use go through a contraction (documented, so a change shows up)."""
    seen = {}
    for n, seed, d0, lam, c in CASES:
        cs = synthetic.full_rt(n, seed=seed)
        _, s, _ = oracle.depth_solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, np.full((n, 2), d0), lam=lam, c=c)
        seen[(n, seed)] = s.num_line_search_steps
    assert seen[(500, 6)] == 1 and seen[(400, 9)] == 1 and seen[(300, 11)] == 1 and seen[(300, 12)] == 0


# ---- hand-derived cases: pinned by pencil and paper, not by another restatement ----------------------------------
def test_hand_derived_cubic_interpolation_step(oracle):
    """phi(a) = a^3 + a^2 - a + 3: phi(0) = 3, phi'(0) = -1; first trial a = 1: phi = 4, phi' = 4.  The interpolant through
    value and slope at 0 and 1 is that very cubic (4 conditions, degree 3); phi' = 3a^2 + 2a - 1 = (3a - 1)(a + 1), so its
    minimiser is a = 1/3 EXACTLY, inside Ceres' bracket [1e-3, 0.6].  The full step fails Armijo (4 > 3 - 1e-4), the next
    trial must be 1/3, where phi = 3 - 5/27 passes (3 - 5/27 <= 3 - 1e-4 / 3): one contraction, step size 1/3."""
    samples = [(0.0, 3.0, -1.0), (1.0, 4.0, 4.0)]
    assert abs(product_hermite_argmin(samples, 1e-3, 0.6) - 1.0 / 3.0) <= 1e-14
    poly = oracle.interpolating_polynomial(samples)
    assert np.abs(poly - np.array([1.0, 1.0, -1.0, 3.0])).max() <= 1e-13
    assert abs(oracle.minimize_polynomial(poly, 1e-3, 0.6)[0] - 1.0 / 3.0) <= 1e-12
    phi = lambda a: (a**3 + a**2 - a + 3.0, 3 * a**2 + 2 * a - 1.0)
    for search in (product_armijo, oracle.armijo):
        trials = []

        def rec(a):
            trials.append(a)
            return phi(a)
        ok, step, contractions = search(rec, 3.0, -1.0)
        assert ok and contractions == 1 and abs(step - 1.0 / 3.0) <= 1e-12, (search.__name__, ok, step, contractions)
        assert trials[0] == 1.0 and abs(trials[1] - 1.0 / 3.0) <= 1e-12 and len(trials) == 2
    # the bracket clamps: phi(a) = (a - 0.9)^2 has its interpolated minimiser at 0.9 > 0.6 -> the next trial is 0.6 exactly
    assert product_hermite_argmin([(0.0, 0.81, -1.8), (1.0, 0.01, 0.2)], 1e-3, 0.6) == 0.6


def test_hand_derived_bound_active_case_where_the_projection_changes_the_slope(oracle):
    """One bounded parameter, d >= 0: d = 1, step delta = -4, cost f(d) = (d - 0.5)^2.  Ceres line-searches along the
    PROJECTED path, phi(a) = f(P(1 - 4a)) with P(d) = max(d, 0), and feeds the interpolation the slope g(P(d + a delta)) . delta
    -- the gradient at the projected point times the UN-projected direction (LineSearchFunction::Evaluate), not d phi / d a.
      a = 0   : phi = 0.25, slope = 2 (1 - 0.5) (-4) = -4
      a = 1   : P(-3) = 0, phi = 0.25 > 0.25 - 4e-4 -> rejected; slope fed = 2 (0 - 0.5) (-4) = +4 (the true d phi / d a is 0)
    Cubic through (0, 0.25, -4), (1, 0.25, +4): 0.25 - 4a + 4a^2, minimiser a = 1/2 exactly -> second trial 0.5.
      a = 0.5 : P(-1) = 0 again, phi = 0.25 -> rejected, slope fed = +4.
    From then on the quintic through three samples decides; every later trial lies in Ceres' bracket of the trial before
    it, and the accepted one satisfies Armijo on the projected path.  Product and oracle walk the same trials."""
    def phi(a):
        d = max(1.0 - 4.0 * a, 0.0)
        return (d - 0.5) ** 2, 2.0 * (d - 0.5) * -4.0
    walks = []
    for search in (product_armijo, oracle.armijo):
        trials = []

        def rec(a):
            trials.append(a)
            return phi(a)
        ok, step, contractions = search(rec, 0.25, -4.0)
        assert ok and trials[0] == 1.0 and abs(trials[1] - 0.5) <= 1e-14, (search.__name__, trials)
        for prev, cur in zip(trials[1:], trials[2:]):
            assert 1e-3 * prev - 1e-15 <= cur <= 0.6 * prev + 1e-15
        assert step == trials[-1] and contractions == len(trials) - 1
        assert phi(step)[0] <= 0.25 + 1e-4 * step * -4.0                       # Armijo on the projected path
        assert all(phi(a)[0] > 0.25 + 1e-4 * a * -4.0 for a in trials[:-1])    # ... and on no earlier trial
        walks.append(trials)
    assert len(walks[0]) == len(walks[1]) and np.abs(np.array(walks[0]) - np.array(walks[1])).max() <= 1e-9
