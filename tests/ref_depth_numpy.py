"""Independent numpy restatement of the reference's d-only stage as Ceres solves it (tests/ only).

What it restates (reference spherical_bundle_adjuster.cpp:1004-1063, solved by `Solve(opt, &problem_d, ...)` at
.cpp:197 with the options of .cpp:334-338): ONE bounded trust-region problem over all depth pairs, Levenberg-Marquardt
strategy with Jacobi scaling, candidate projected onto d >= 0, and -- because the problem is bounds-constrained and
Solver::Options::max_num_line_search_step_size_iterations defaults to 20 -- Ceres' projected ARMIJO line search with
CUBIC interpolation on every trust-region step.

Deliberately written differently from oracle/sba_oracle.cpp + oracle/ceres_line_search.hpp so that agreement is a
real cross-check: closed-form residual derivatives instead of dual numbers, vectorised numpy instead of loops,
numpy.linalg.solve for the interpolating polynomial, numpy.roots (companion-matrix eigenvalues, as Ceres itself uses)
for the critical points.
"""
import numpy as np
from scipy.spatial.transform import Rotation


class DepthProblem:
    def __init__(self, x1, x2, rot, tran, lam=1.0, c=1.0):
        self.q = np.asarray(x1, dtype=np.float64) @ Rotation.from_rotvec(rot).as_matrix().T   # R x1
        self.u = np.asarray(x2, dtype=np.float64)
        self.t = np.asarray(tran, dtype=np.float64)
        self.lam, self.c = lam, c

    def residuals(self, d):
        e = d[:, 1:2] * self.u - d[:, 0:1] * self.q + self.t              # .cpp:1008-1027
        reg = self.lam * np.exp(-self.c * d)                               # .cpp:1028-1029
        return e, reg

    def cost(self, d):
        e, reg = self.residuals(d)
        return 0.5 * (np.sum(e * e) + np.sum(reg * reg))

    def gradient(self, d):
        e, reg = self.residuals(d)
        g = np.empty_like(d)
        g[:, 0] = -np.sum(self.q * e, axis=1) - self.c * reg[:, 0] ** 2
        g[:, 1] = np.sum(self.u * e, axis=1) - self.c * reg[:, 1] ** 2
        return g

    def hessian_blocks(self, d):
        _, reg = self.residuals(d)
        h11 = np.sum(self.q * self.q, axis=1) + (self.c * reg[:, 0]) ** 2
        h22 = np.sum(self.u * self.u, axis=1) + (self.c * reg[:, 1]) ** 2
        h12 = -np.sum(self.q * self.u, axis=1)
        return h11, h12, h22


def hermite_polynomial(samples):
    """samples: list of (x, f, f'); the polynomial of degree 2k-1 through all values and slopes (highest first)."""
    m = 2 * len(samples)
    A = np.zeros((m, m))
    b = np.zeros(m)
    for k, (x, f, fp) in enumerate(samples):
        for j in range(m):
            p = m - 1 - j
            A[2 * k, j] = x ** p
            A[2 * k + 1, j] = p * x ** (p - 1) if p > 0 else 0.0
        b[2 * k], b[2 * k + 1] = f, fp
    return np.linalg.solve(A, b)


def minimize_on_interval(poly, lo, hi):
    """Ceres MinimizePolynomial: mid point, both ends, and the real parts of ALL roots of the derivative that lie inside."""
    xs = [(lo + hi) / 2.0, lo, hi]
    xs += [r.real for r in np.roots(np.polyder(poly)) if lo <= r.real <= hi]
    best_x, best_v = xs[0], np.polyval(poly, xs[0])
    for x in xs[1:]:
        v = np.polyval(poly, x)
        if v < best_v:
            best_x, best_v = x, v
    return best_x


def armijo(phi, f0, g0, dir_inf_norm, max_iter=20, c1=1e-4, max_contraction=1e-3, min_contraction=0.6, min_step=1e-9):
    """phi(a) -> (value, directional derivative).  Returns (success, step size, contractions)."""
    prev = None
    cur = (1.0,) + phi(1.0)
    iters = 0
    while cur[1] > f0 + c1 * g0 * cur[0]:
        iters += 1
        if iters >= max_iter:
            return False, 1.0, iters
        samples = [(0.0, f0, g0), cur] + ([prev] if prev is not None else [])
        a = minimize_on_interval(hermite_polynomial(samples), max_contraction * cur[0], min_contraction * cur[0])
        if a * dir_inf_norm < min_step:
            return False, 1.0, iters
        prev = cur
        cur = (a,) + phi(a)
    return True, cur[0], iters


def solve(x1, x2, rot, tran, d0, lam=1.0, c=1.0, max_iter=50, radius=1e4, line_search_iterations=20,
          ftol=1e-6, gtol=1e-10, ptol=1e-8, min_rel_decrease=1e-3, min_diag=1e-6, max_diag=1e32):
    P = DepthProblem(x1, x2, rot, tran, lam, c)
    d = np.array(d0, dtype=np.float64).reshape(-1, 2).copy()
    project = lambda z: np.maximum(z, 0.0)
    cost, g = P.cost(d), P.gradient(d)
    h11, h12, h22 = P.hessian_blocks(d)
    s = np.stack([1.0 / (1.0 + np.sqrt(h11)), 1.0 / (1.0 + np.sqrt(h22))], axis=1)    # Jacobi scaling, iteration 0
    info = dict(iterations=0, successful=0, line_search_steps=0, termination=None, trace=[])
    nu, reuse, D = 2.0, False, None
    it = 0
    while True:
        if it >= max_iter:
            info["termination"] = "no_convergence"; break
        if np.abs(d - project(d - g)).max(initial=0.0) <= gtol:
            info["termination"] = "gradient"; break
        it += 1
        info["iterations"] = it
        H11, H12, H22 = s[:, 0] ** 2 * h11, s[:, 0] * s[:, 1] * h12, s[:, 1] ** 2 * h22
        G = s * g
        if not reuse:
            D = np.stack([np.clip(H11, min_diag, max_diag), np.clip(H22, min_diag, max_diag)], axis=1)
        A11, A22 = H11 + D[:, 0] / radius, H22 + D[:, 1] / radius
        det = A11 * A22 - H12 * H12
        y = np.stack([(-G[:, 0] * A22 + G[:, 1] * H12) / det, (-G[:, 1] * A11 + G[:, 0] * H12) / det], axis=1)
        model = -np.sum(G * y) - 0.5 * np.sum(H11 * y[:, 0] ** 2 + 2 * H12 * y[:, 0] * y[:, 1] + H22 * y[:, 1] ** 2)
        if not model > 0:
            radius /= nu; nu *= 2; reuse = True
            continue
        delta = s * y
        alpha = 1.0
        if line_search_iterations > 0:
            phi = lambda a: (P.cost(project(d + a * delta)), float(np.sum(P.gradient(project(d + a * delta)) * delta)))
            ok, a, contractions = armijo(phi, cost, float(np.sum(g * delta)), np.abs(delta).max(),
                                         max_iter=line_search_iterations)
            info["line_search_steps"] += contractions
            if ok:
                alpha = a
        cand = project(d + alpha * delta)
        cand_cost = P.cost(cand)
        info["trace"].append((it, alpha, cand_cost))
        if np.linalg.norm(cand - d) <= ptol * (np.linalg.norm(d) + ptol):
            info["termination"] = "parameter"; break
        if abs(cost - cand_cost) <= ftol * cost:
            info["termination"] = "function"; break
        rho = (cost - cand_cost) / model
        if rho > min_rel_decrease:
            d, cost, g = cand, cand_cost, P.gradient(cand)
            h11, h12, h22 = P.hessian_blocks(d)
            info["successful"] += 1
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            nu, reuse = 2.0, False
        else:
            radius /= nu; nu *= 2; reuse = True
    info["cost"] = cost
    return d, info
