"""CPU: host-side SO(3) math of the product (csrc/sba_rotation.hpp) against the oracle / numpy:
R and dR/dw (explicit kernel), the factored frame (B, J_l) and the moment -> normal-equation map."""
import ctypes as C
import subprocess

import numpy as np
import pytest

import ref_numpy as rn
from helpers import ROOT, pack_from_eval

_h = None


def harness():
    global _h
    if _h is None:
        so = ROOT / "tests" / "harness" / "librot_harness.so"
        src = ROOT / "tests" / "harness" / "rot_harness.cpp"
        hdr = ROOT / "spherical_bundle_adjuster_amd" / "csrc" / "sba_rotation.hpp"
        if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(so), str(src)], check=True)
        _h = C.CDLL(str(so))
    return _h


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


ROTS = [np.zeros(3), np.array([3e-9, -4e-9, 1e-9]), np.array([1.2e-8, 0.9e-8, 0.0]), np.array([1e-6, 2e-6, -1e-6]),
        np.array([1e-3, -2e-3, 5e-4]), np.array([0.06, 0.05, -0.06]),          # th^2 just below the series switch
        np.array([0.06, 0.06, -0.06]), np.array([0.21, -0.35, 0.11]), np.array([1.0, -1.2, 0.7]),
        np.array([1.8, -1.9, 1.7]) * (3.1 / np.linalg.norm([1.8, -1.9, 1.7]))]


def test_so3_coefficients_series_vs_mpmath_style_reference():
    # high-precision reference through numpy longdouble closed forms where they are well conditioned,
    # and through the series itself (in longdouble, more terms) where they are not
    for x in [0.0, 1e-16, 1e-10, 1e-6, 1e-3, 9.99e-3, 1.0001e-2, 0.2499, 0.2501, 0.5, 2.0, 9.0]:
        out = np.zeros(5)
        harness().harness_coeffs(C.c_double(x), _p(out))
        xl = np.longdouble(x)
        if x < 0.4:
            k = np.arange(0, 14)
            fact = np.array([np.prod(np.arange(1, m + 1, dtype=np.longdouble)) for m in range(0, 30)])
            a = sum((-1) ** i * xl ** i / fact[2 * i + 1] for i in k)
            b = sum((-1) ** i * xl ** i / fact[2 * i + 2] for i in k)
            c = sum((-1) ** i * xl ** i / fact[2 * i + 3] for i in k)
            bp = sum((-1) ** i * 2 * i * xl ** (i - 1) / fact[2 * i + 2] for i in k[1:])
        else:
            th = np.sqrt(xl)
            a, b, c = np.sin(th) / th, (1 - np.cos(th)) / xl, (th - np.sin(th)) / (xl * th)
            bp = (a - 2 * b) / xl
        ref = np.array([a, b, c, c - b, bp], dtype=np.float64)
        tol = np.full(5, 4e-16) if x < 0.25 else np.array([1e-15, 1e-15, 1e-15, 2e-15, 4.5e-16 / x**2 + 1e-15])
        assert (np.abs(out - ref) <= tol).all(), (x, out - ref)


@pytest.mark.parametrize("idx", range(len(ROTS)))
def test_rotation_and_derivatives_vs_oracle(oracle, idx):
    w = ROTS[idx]
    R, G = np.zeros(9), np.zeros(27)
    harness().harness_rotation(_p(w), _p(R), _p(G))
    R, G = R.reshape(3, 3), G.reshape(3, 3, 3)
    rng = np.random.default_rng(idx)
    th2 = float(w @ w)
    for _ in range(3):
        p = rng.standard_normal(3)
        assert np.abs(R @ p - oracle.rotate(w, p)).max() <= 4e-16 * 3
        dRp = np.stack([G[j] @ p for j in range(3)], axis=1)
        # oracle Jacobian of e = x2 - (R x1 - t) w.r.t. w is -d(R x1)/dw
        _, J = oracle.point(0, p, np.zeros(3), w, np.zeros(3), 1.0, 1.0)
        if th2 <= np.finfo(float).eps:
            assert np.array_equal(dRp, -J[:, :3])                     # small-angle branch: -[p]x exactly
            continue
        # truth: 50-digit differentiation of Rodrigues' formula
        assert np.abs(dRp - rn.d_rotated_d_w_mp(w, p)).max() <= 3e-15, (w, np.abs(dRp - rn.d_rotated_d_w_mp(w, p)).max())
        # The oracle follows Ceres' formula through dual numbers; just above its small-angle threshold
        # that formula multiplies the cancellation-prone (1 - cos th) by 1/th, so ITS Jacobian carries
        # ~eps/th of noise (3.5e-11 at th = 1.5e-8).  The product's series is the accurate one.
        noise = np.finfo(float).eps / th2 ** 0.5 * np.abs(p).max()
        assert np.abs(dRp + J[:, :3]).max() <= 3e-15 + noise, (w, np.abs(dRp + J[:, :3]).max())


@pytest.mark.parametrize("idx", range(len(ROTS)))
@pytest.mark.parametrize("mode", [0, 2])
def test_factored_moments_reproduce_normal_equations(oracle, idx, mode):
    """Build the device's moment pack in numpy (v = -d1 R x1 with the Ceres-branch R), push it through
    moments_to_normal_pack and compare with the oracle's dual-number normal equations."""
    w = ROTS[idx]
    rng = np.random.default_rng(100 + idx)
    n = 200
    x1 = rng.standard_normal((n, 3)); x1 /= np.linalg.norm(x1, axis=1, keepdims=True)
    x2 = rng.standard_normal((n, 3)); x2 /= np.linalg.norm(x2, axis=1, keepdims=True)
    d12 = rng.uniform(0.5, 6.0, (n, 2))
    t = np.array([0.3, -0.2, 0.9])
    R = rn.rotmat(w)
    v = -(d12[:, :1] * (x1 @ R.T))
    e = t + d12[:, 1:] * x2 + v
    s = (e * e).sum(1)
    wgt = np.where(s > 1.0, 1.0 / np.sqrt(s), 1.0)
    rho = np.where(s > 1.0, 2 * np.sqrt(s) - 1.0, s)
    mom = np.zeros(24)
    M = np.einsum("i,ik,il->kl", wgt, v, v)
    mom[0:6] = [M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2]]
    mom[6:15] = np.einsum("i,ik,il->kl", wgt, v, e).reshape(-1)
    mom[15] = wgt.sum()
    mom[16:19] = (wgt[:, None] * v).sum(0)
    mom[19:22] = (wgt[:, None] * e).sum(0)
    mom[22] = 0.5 * rho.sum()
    mom[23] = (s > 1.0).sum()
    B, J = np.zeros(9), np.zeros(9)
    harness().harness_frame(_p(w), _p(B), _p(J))
    pack = np.zeros(24)
    harness().harness_moments_to_pack(C.c_int(1), C.c_int(1 if mode == 2 else 0), _p(B), _p(J), _p(mom), _p(pack))
    ref = pack_from_eval(mode, oracle.evaluate(mode, x1, x2, w, t, delta=1.0, d12=d12, threads=1))
    if mode == 0:
        pack[15] = 0.0; pack[19:22] = 0.0          # rot-only device kernel does not produce these
    scale = np.abs(ref[:15]).max()
    th = float(np.sqrt(w @ w))
    noise = 0.0 if th * th <= np.finfo(float).eps else np.finfo(float).eps / th   # oracle's own Ceres-formula noise
    assert np.abs(pack[:15] - ref[:15]).max() <= (1e-13 + noise * 4) * scale, (w, np.abs(pack - ref).max() / scale)
    assert np.abs(pack[16:22] - ref[16:22]).max() <= (1e-13 + noise * 4) * max(np.abs(ref[16:22]).max(), scale)
    assert abs(pack[22] - ref[22]) <= 1e-13 * ref[22] and pack[23] == ref[23]
