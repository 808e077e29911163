"""CPU: host side of the 8-point initial guess (csrc/sba_epipolar.hpp, reference .cpp:47-181) against numpy:
Jacobi eigen / 3x3 SVD / decomposeEssentialMat restatement, one trial against an explicit A + numpy SVD, and the
known answer on clean data.  The C-ABI entry sba_initial_guess_from_moments needs no device."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import ROOT
from spherical_bundle_adjuster_amd import api, synthetic

_h = None


def harness():
    global _h
    if _h is None:
        so = ROOT / "tests" / "harness" / "libepi_harness.so"
        src = ROOT / "tests" / "harness" / "epi_harness.cpp"
        hdr = ROOT / "spherical_bundle_adjuster_amd" / "csrc" / "sba_epipolar.hpp"
        if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(so), str(src)], check=True)
        _h = C.CDLL(str(so))
    return _h


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def group_moments(x1, x2):
    """numpy reference of the device pass: per group (i//2) % 64 the upper triangle of A^T A, A rows = kron(l, r)."""
    n = len(x1)
    A = (x1[:, :, None] * x2[:, None, :]).reshape(n, 9)            # .cpp:59-67
    grp = (np.arange(n) // 2) % 64
    iu = np.triu_indices(9)
    out = np.zeros((64, 45))
    for g in range(64):
        a = A[grp == g]
        out[g] = (a.T @ a)[iu]
    return out, A, grp


def euler_of(R):
    sy = np.hypot(R[0, 0], R[1, 0])
    return np.array([np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], sy), np.arctan2(R[1, 0], R[0, 0])])


def test_jacobi_and_svd3_vs_numpy():
    rng = np.random.default_rng(0)
    for n in (3, 9):
        for _ in range(10):
            M = rng.standard_normal((n + 4, n)); S = M.T @ M
            w, V = np.zeros(n), np.zeros((n, n))
            harness().harness_jacobi(C.c_int(n), _p(S), _p(w), _p(V))
            wn = np.linalg.eigvalsh(S)
            assert np.abs(w - wn).max() <= 1e-12 * wn.max()
            assert np.abs(V.T @ V - np.eye(n)).max() < 1e-13 and np.abs(S @ V - V * w).max() <= 1e-12 * wn.max()
    for _ in range(20):
        E = rng.standard_normal((3, 3))
        U, w, Vt = np.zeros((3, 3)), np.zeros(3), np.zeros((3, 3))
        harness().harness_svd3(_p(E), _p(U), _p(w), _p(Vt))
        assert np.abs(U @ np.diag(w) @ Vt - E).max() < 1e-13 and np.abs(w - np.linalg.svd(E, compute_uv=False)).max() < 1e-13
        assert np.abs(U.T @ U - np.eye(3)).max() < 1e-13 and np.abs(Vt @ Vt.T - np.eye(3)).max() < 1e-13 and w[0] >= w[1] >= w[2]


def test_smallest_eigvec_fast_path_vs_numpy():
    """The verified shortcut for the smallest eigenpair (what a trial needs from A^T A): whenever it reports success it
    must agree with LAPACK; inconclusive cases (clustered smallest eigenvalues, non-PSD input) must say so, because the
    caller then falls back to the full Jacobi decomposition."""
    rng = np.random.default_rng(5)
    h = harness()
    h.harness_smallest_eigvec.restype = C.c_int
    ok = 0
    for trial in range(300):
        n = 9 if trial % 3 else 3
        rows = n - 1 if trial % 7 == 0 else n + 6                  # rank-deficient (exact null vector) and full rank
        M = rng.standard_normal((rows, n)) * 10.0 ** rng.uniform(-3, 3)
        S = M.T @ M
        v, lam = np.zeros(n), np.zeros(1)
        if h.harness_smallest_eigvec(C.c_int(n), _p(S), _p(v), _p(lam)):
            ok += 1
            wn, Vn = np.linalg.eigh(S)
            assert abs(lam[0] - wn[0]) <= 1e-12 * wn[-1]
            assert abs(np.linalg.norm(v) - 1) < 1e-14 and v[np.argmax(np.abs(v))] > 0
            gap = (wn[1] - wn[0]) / wn[-1]
            assert np.abs(v - Vn[:, 0] * np.sign(Vn[:, 0] @ v)).max() <= 1e-13 / max(gap, 1e-13)
            assert np.abs(S @ v - lam[0] * v).max() <= 1e-13 * wn[-1]
    assert ok >= 280                                               # the fast path is the normal case
    # exactly repeated smallest eigenvalue: any vector of the eigenspace is acceptable if success is reported
    S = np.diag([1.0, 1.0, 5.0])
    v, lam = np.zeros(3), np.zeros(1)
    if h.harness_smallest_eigvec(C.c_int(3), _p(S), _p(v), _p(lam)):
        assert abs(lam[0] - 1.0) < 1e-12 and abs(v[2]) < 1e-6
    # not positive semi-definite / all-zero: must decline
    for S in (np.diag([1.0, -2.0, 3.0]), np.zeros((3, 3))):
        assert h.harness_smallest_eigvec(C.c_int(3), _p(np.ascontiguousarray(S)), _p(v), _p(lam)) == 0


def test_decompose_essential_recovers_motion():
    rng = np.random.default_rng(1)
    for _ in range(10):
        w = synthetic._true_rotation(rng); R = synthetic.rodrigues(w)
        t = synthetic._unit(rng.standard_normal(3))
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        E = tx @ R
        R1, R2, tt = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros(3)
        harness().harness_decompose(_p(E), _p(R1), _p(R2), _p(tt))
        assert min(np.abs(R1 - R).max(), np.abs(R2 - R).max()) < 1e-12
        assert min(np.abs(tt - t).max(), np.abs(tt + t).max()) < 1e-12
        for Rk in (R1, R2):
            assert abs(np.linalg.det(Rk) - 1) < 1e-12 and np.abs(Rk @ Rk.T - np.eye(3)).max() < 1e-12
        e = np.zeros(3, dtype=np.float32)
        harness().harness_euler(_p(R), _p(e))
        assert np.abs(e - euler_of(R)).max() < 1e-6          # single precision like rot2euler (.cpp:25-45)


def test_one_trial_vs_explicit_A_and_numpy_svd():
    """A trial from summed group moments == the reference recipe on the explicit subset matrix A (numpy SVD)."""
    c = synthetic.full_rt(4096, seed=12, sigma=5e-4, outlier_fraction=0.0)
    G, A, grp = group_moments(c.x1, c.x2)
    sel = np.zeros(16, dtype=np.int32)
    harness().harness_trial_groups(C.c_ulonglong(5), C.c_int(3), C.c_int(16), _p(sel))
    assert len(set(sel.tolist())) == 16 and sel.min() >= 0 and sel.max() < 64
    mom = G[np.sort(sel)].sum(0)
    e1, e2, tv = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    v1, v2, Ec = C.c_int(0), C.c_int(0), np.zeros((3, 3))
    harness().harness_trial(_p(mom), _p(e1), _p(e2), _p(tv), C.byref(v1), C.byref(v2), _p(Ec))
    # reference recipe (.cpp:53-85) on the explicit rows
    As = A[np.isin(grp, sel)]
    E = np.linalg.svd(As)[2][-1].reshape(3, 3)
    U, s, Vt = np.linalg.svd(E)
    Ec_ref = U @ np.diag([s[0], s[1], 0.0]) @ Vt
    assert min(np.abs(Ec - Ec_ref).max(), np.abs(Ec + Ec_ref).max()) < 1e-9          # null vector sign is free
    if np.linalg.det(U) < 0: U = -U
    if np.linalg.det(Vt) < 0: Vt = -Vt
    W = np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]])
    cands = [euler_of(U @ W @ Vt), euler_of(U @ W.T @ Vt)]
    for e in (e1, e2):      # {R1, R2} as a set
        assert min(np.abs(e - cands[0]).max(), np.abs(e - cands[1]).max()) < 1e-5
    assert min(np.abs(tv - U[:, 2]).max(), np.abs(tv + U[:, 2]).max()) < 1e-5


def test_initial_guess_known_answer_and_determinism():
    c = synthetic.full_rt(20000, seed=13, sigma=0.0, outlier_fraction=0.0)
    G, _, _ = group_moments(c.x1, c.x2)
    e, t, ncand = api.initial_guess_from_moments(G, 80, 0.25, seed=1)
    # left^T E right = 0 with x2 ~ R x1 - t  =>  the recovered rotation is R^T and T = +-R^T t
    Rt = synthetic.rodrigues(c.rot_true).T
    assert np.abs(e - euler_of(Rt)).max() < 1e-5 and ncand >= 80
    tt = Rt @ c.tran_true
    assert min(np.abs(t - tt).max(), np.abs(t + tt).max()) < 1e-5
    # the reference starts from init_rot = -Euler (.cpp:330): close to the true angle-axis for moderate rotations
    assert np.abs(-e - c.rot_true).max() < 0.15
    e2, t2, n2 = api.initial_guess_from_moments(G, 80, 0.25, seed=1)
    assert np.array_equal(e, e2) and np.array_equal(t, t2) and n2 == ncand
    # noisy data with outliers: the consensus still lands near the truth
    c = synthetic.full_rt(20000, seed=14)
    G, _, _ = group_moments(c.x1, c.x2)
    e, t, _ = api.initial_guess_from_moments(G, 80, 0.25, seed=2)
    assert np.abs(e - euler_of(synthetic.rodrigues(c.rot_true).T)).max() < 0.1
    # host threads (the reference's set_omp): the trials are collected in trial order, so the count never matters
    try:
        for threads in (2, 5, 8, 0):
            api.set_host_threads(threads)
            et, tt_, nt = api.initial_guess_from_moments(G, 80, 0.25, seed=2)
            assert np.array_equal(et, e) and np.array_equal(tt_, t), threads
            e7, t7, n7 = api.initial_guess_from_moments(G, 1100, 0.25, seed=2)   # enough trials for several threads
            api.set_host_threads(1)
            e7s, t7s, n7s = api.initial_guess_from_moments(G, 1100, 0.25, seed=2)
            assert np.array_equal(e7, e7s) and np.array_equal(t7, t7s) and n7 == n7s
        with pytest.raises(api.SbaError):
            api.set_host_threads(-1)
    finally:
        api.set_host_threads(1)
    with pytest.raises(api.SbaError):
        api.initial_guess_from_moments(G, 0, 0.25)


def _reference_recipe_guess(x1, x2, trials, rng):
    """The reference's initial_guess (.cpp:118-181) in numpy with PER-MATCH sampling: `trials` x (random floor(n/4)
    matches -> explicit A -> SVD -> rank 2 -> R1/R2 -> Euler, valid if all |angles| < 1.57) and the 20-80 % trimmed-mean
    consensus pick."""
    n = len(x1)
    A = (x1[:, :, None] * x2[:, None, :]).reshape(n, 9)
    W = np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]])
    cands = []
    for _ in range(trials):
        idx = rng.permutation(n)[: int(n * 0.25)]
        E = np.linalg.svd(A[idx], full_matrices=True)[2][-1].reshape(3, 3)
        U, s, Vt = np.linalg.svd(E)
        U, s, Vt = np.linalg.svd(U @ np.diag([s[0], s[1], 0.0]) @ Vt)
        if np.linalg.det(U) < 0: U = -U
        if np.linalg.det(Vt) < 0: Vt = -Vt
        for R in (U @ W @ Vt, U @ W.T @ Vt):
            e = euler_of(R)
            if np.abs(e).max() < 1.57:
                cands.append(e)
    c = np.array(cands)
    if len(c) == 0:
        return None
    d = np.sort(np.linalg.norm(c[:, None, :] - c[None, :, :], axis=2), axis=1)
    lo, hi = int(len(c) * 0.2), int(len(c) * 0.8)
    return c[np.argmin(d[:, lo:hi].mean(axis=1))] if hi > lo else c[0]


@pytest.mark.parametrize("n", [40, 100, 200])
def test_initial_guess_with_few_matches(n):
    """Realistic SURF match counts sit far below 64 x 4: many of the 64 interleaved groups are then empty.  The trials
    draw among the NON-EMPTY groups and never solve from fewer than 8 correspondences, so the consensus is as good as
    the reference's per-match sampling (numpy restatement above) -- not polluted by rank-deficient trials."""
    errs_ours, errs_ref = [], []
    for seed in range(12):
        c = synthetic.full_rt(n, seed=500 + seed, sigma=2e-4, outlier_fraction=0.0)
        G, _, grp = group_moments(c.x1, c.x2)
        assert np.allclose(G[:, [0, 9, 17, 24, 30, 35, 39, 42, 44]].sum(1), np.bincount(grp, minlength=64))   # trace = count
        truth = euler_of(synthetic.rodrigues(c.rot_true).T)
        e, t, ncand = api.initial_guess_from_moments(G, 80, 0.25, seed=seed)
        errs_ours.append(np.abs(e - truth).max())
        r = _reference_recipe_guess(c.x1, c.x2, 80, np.random.default_rng(seed))
        errs_ref.append(np.abs(r - truth).max() if r is not None else np.inf)
    ours, ref = np.median(errs_ours), np.median(errs_ref)
    # n = 40: the reference itself solves from 10 matches per trial -- both are rough; what matters is "not worse"
    assert ours <= max(2.0 * ref, 5e-3), (n, ours, ref, errs_ours, errs_ref)
    if n >= 100:
        assert ours < 0.02


def test_trial_subsets_with_few_matches_hold_at_least_eight():
    """White box: with n = 19 there are 10 non-empty groups (9 full pairs + 1 with one match); every trial's subset
    must hold at least 8 matches, drawn from non-empty groups only."""
    h = harness()
    nonempty = np.arange(10, dtype=np.int32)
    for trial in range(50):
        sel = np.full(10, -1, dtype=np.int32)
        h.harness_trial_groups_from(C.c_ulonglong(3), C.c_int(trial), C.c_int(10), _p(sel), _p(nonempty), C.c_int(10))
        assert sorted(sel.tolist()) == list(range(10))                         # a permutation of the non-empty groups
        short = np.full(2, -1, dtype=np.int32)
        h.harness_trial_groups_from(C.c_ulonglong(3), C.c_int(trial), C.c_int(2), _p(short), _p(nonempty), C.c_int(10))
        assert short.tolist() == sel[:2].tolist()                               # a longer draw extends a shorter one
    # all 64 occupied: identical to the historical draw over 0..63
    full = np.arange(64, dtype=np.int32)
    a, b = np.zeros(16, dtype=np.int32), np.zeros(16, dtype=np.int32)
    h.harness_trial_groups(C.c_ulonglong(5), C.c_int(3), C.c_int(16), _p(a))
    h.harness_trial_groups_from(C.c_ulonglong(5), C.c_int(3), C.c_int(16), _p(b), _p(full), C.c_int(64))
    assert np.array_equal(a, b)


# ---- the reference's own subsets (random_array / std::random_shuffle on the process's rand() stream) --------------------------
def test_glibc_rand_stream_is_the_one_the_reference_runs_on(oracle):
    """Known answers of glibc's never-seeded rand() (= srand(1), C standard 7.22.2.2): what `random_array` consumes in a
    reference process that has drawn nothing yet.  If this fails the C library is not the reference's and the subsets
    below are still self-consistent (product == std::random_shuffle) but no longer the reference's."""
    oracle.c_srand(1)
    assert [oracle.c_rand() for _ in range(5)] == [1804289383, 846930886, 1681692777, 1714636915, 1957747793]


@pytest.mark.parametrize("seed", [1, 0, 2, 12345, 2**31 - 1, 2**32 - 1])
def test_library_copy_of_glibc_rand_equals_the_real_one(oracle, seed):
    """The product draws the reference's subsets from its OWN copy of glibc's generator (csrc/sba_epipolar.hpp: GlibcRand) --
    a process that has initialised HIP no longer owns rand() (libhsa-runtime64 imports srand and rand).  The copy must be
    the real thing: 100 000 consecutive values equal to the C library's rand() for several seeds, seed 1 being the
    never-seeded state the reference runs in."""
    oracle.c_srand(seed)
    api.reference_rand_seed(seed)
    real = [oracle.c_rand() for _ in range(100_000)]
    mine = [api.reference_rand_next() for _ in range(100_000)]
    assert real == mine


@pytest.mark.parametrize("n", [4, 5, 40, 333, 2048])
def test_reference_subsets_equal_libstdcxx_random_shuffle(oracle, n):
    """Product (hand-written loop over its copy of glibc's rand(), csrc/sba_epipolar.hpp) == the oracle's call of the REAL
    libstdc++ std::random_shuffle on the REAL rand() -- the function the reference itself calls
    (spherical_bundle_adjuster.hpp:209) -- element for element over all 80 trials, from the same stream state; and both
    consume the same number of draws."""
    ref = oracle.reference_trial_subsets(n, 80, reseed=True)
    after_ref = oracle.c_rand()
    api.reference_rand_seed(1)
    got = api.reference_trial_subsets(n, 80, 0.25)
    after_got = api.reference_rand_next()
    assert ref.shape == got.shape == (80, int(n * 0.25)) and np.array_equal(ref, got) and after_ref == after_got
    for row in got:                                   # each a prefix of a permutation of 0..n-1
        assert len(set(row.tolist())) == len(row) and (row >= 0).all() and (row < n).all()
    if n >= 40:
        assert len({tuple(r) for r in got.tolist()}) == 80        # a FRESH permutation per trial
    # the stream runs on from call to call (a second image pair in the same process): still the same lists as the real thing
    ref2 = oracle.reference_trial_subsets(n, 3, reseed=False)
    got2 = api.reference_trial_subsets(n, 3, 0.25)
    assert np.array_equal(ref2, got2) and (n < 40 or not np.array_equal(ref2, ref[:3]))


def test_reference_subsets_from_the_process_rand_on_request(oracle, monkeypatch):
    """SBA_GUESS_RAND=libc: draw from the process's own rand() instead -- the matcher's draws (FLANN's kd-trees call rand())
    then count exactly as in the reference's process."""
    monkeypatch.setenv("SBA_GUESS_RAND", "libc")
    oracle.c_srand(1)
    burn = [oracle.c_rand() for _ in range(1234)]                       # "the matcher"
    ref = oracle.reference_trial_subsets(333, 3, reseed=False)
    oracle.c_srand(1)
    assert [oracle.c_rand() for _ in range(1234)] == burn
    assert np.array_equal(api.reference_trial_subsets(333, 3, 0.25), ref)
    assert not np.array_equal(ref, oracle.reference_trial_subsets(333, 3, reseed=True))


def _recipe_on_subsets(x1, x2, subsets):
    """The reference's initial_guess in numpy on GIVEN subsets (explicit A, LAPACK SVD): oracle/oracle_py.py."""
    from oracle import oracle_py
    return oracle_py.initial_guess_recipe(x1, x2, subsets)


def subset_moments(x1, x2, subsets):
    A = (x1[:, :, None] * x2[:, None, :]).reshape(len(x1), 9)
    iu = np.triu_indices(9)
    return np.stack([(A[idx].T @ A[idx])[iu] for idx in subsets])


@pytest.mark.parametrize("n", [20, 40, 2048])
def test_guess_from_the_reference_subsets_equals_the_numpy_recipe(oracle, n):
    """Host half of sba_problem_initial_guess_reference (per-trial A^T A -> eigenvector -> rank 2 -> R1 / R2 -> consensus)
    against the reference recipe in numpy (explicit A, LAPACK SVD) on the SAME subsets -- the ones libstdc++'s
    std::random_shuffle draws.  n = 20: 5 rows per trial, fewer than 9 -- the last row of cv::SVDecomp's vt is then the
    5th singular vector, not a null vector of A.  Euler angles are float32 in the reference (cv::Vec3f)."""
    c = synthetic.full_rt(n, seed=900 + n, sigma=2e-4, outlier_fraction=0.0)
    subsets = oracle.reference_trial_subsets(n, 80, reseed=True)
    rows = subsets.shape[1]
    mom = np.ascontiguousarray(subset_moments(c.x1, c.x2, subsets))
    e, t, nc = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_int(0)
    h = harness()
    h.harness_guess_from_trial_moments.restype = C.c_int
    picked = h.harness_guess_from_trial_moments(_p(mom), C.c_int(80), C.c_int(rows), _p(e), _p(t), C.byref(nc))
    e_ref, t_ref, nc_ref = _recipe_on_subsets(c.x1, c.x2, subsets)
    assert picked >= 0 and nc.value == nc_ref, (nc.value, nc_ref)
    assert np.abs(e - e_ref).max() < 2e-6 * max(1.0, np.abs(e_ref).max()) + (1e-4 if n == 20 else 0.0), (e, e_ref)
    assert min(np.abs(t - t_ref).max(), np.abs(t + t_ref).max()) < 1e-5 + (1e-3 if n == 20 else 0.0)
    if n >= 2048:        # enough matches per trial for a usable guess: near the true rotation (left^T E right = 0 => R^T)
        assert np.abs(e - euler_of(synthetic.rodrigues(c.rot_true).T)).max() < 0.05


def test_smallest_eigvec_tiers_against_jacobi(tmp_path):
    """csrc/sba_epipolar.hpp smallest_eigvec (the 9 x 9 eigenvector every trial needs; also compiled for the device, where a
    fall-back to the cyclic Jacobi decomposition costs ~0.7 ms on one lane): on matrices with a prescribed spectrum it must
    agree with Jacobi -- eigenvalue to 1e-15 trace, vector to 1e-11 -- WITHOUT falling back when the smallest eigenvalues are
    merely close (ratio 1.6 ... 1.01: what subsets with outliers produce; the bisection tier), and may fall back only when
    they are degenerate to 1e-6 (the direction is then ill-defined to ~1e-7 anyway)."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "eig_harness"
    subprocess.run([gxx, "-O1", "-std=c++17", "-Wall", "-I", os.path.join(root, "spherical_bundle_adjuster_amd", "csrc"),
                    os.path.join(root, "tests", "harness", "eig_harness.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    rows = {r.split()[0]: r.split()[1:] for r in out if r.strip()}
    assert set(rows) == {"separated", "clustered_1.6", "clustered_1.05", "clustered_1.01", "near_degenerate_1e-6", "rank_deficient"}
    for name, (count, fallbacks, worst_v, worst_l) in rows.items():
        assert float(worst_l) <= 1e-15, (name, worst_l)
        if name == "near_degenerate_1e-6":
            assert float(worst_v) <= 1e-6
        else:
            assert int(fallbacks) == 0 and float(worst_v) <= 1e-11, (name, fallbacks, worst_v)
