"""Independent numpy restatement of the trust-region Levenberg-Marquardt loop Ceres runs for the reference's rot-only /
tran-only stages (`Solve(opt, &problem_rot|tran, &summary)`, spherical_bundle_adjuster.cpp:203, 209, options :334-338) and
for the joint R|t problem -- tests/ only.

Deliberately NOT derived from oracle/sba_oracle.cpp or csrc/sba_lm.hpp: the normal equations come from the closed-form
numpy Jacobian (tests/ref_numpy.py), the damped system is solved with numpy.linalg.solve (not a hand-written Cholesky),
and the loop is written from the description of Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy:
  * Jacobi scaling s_i = 1 / (1 + sqrt(H_ii)) fixed at the start;
  * step: (H_s + D / radius) y = -g_s with D = clamp(diag(H_s), 1e-6, 1e32), kept while steps are rejected;
  * model decrease  -g_s.y - y.H_s.y / 2;  step quality rho = (cost - cost_new) / model decrease;
  * stop inside an iteration on |step| <= 1e-8 (|x| + 1e-8) or |cost change| <= 1e-6 cost; after it on the iteration
    limit, on max|g| <= 1e-10, on radius < 1e-32;
  * accept if rho > 1e-3: radius /= max(1/3, 1 - (2 rho - 1)^3); else radius /= nu, nu *= 2.
The parameter update is additive on the angle-axis vector and on t (no local parameterisation anywhere in the reference).
"""
import numpy as np

import ref_numpy as rn


def solve(mode, x1, x2, rot, tran, d1=1.0, d2=1.0, d12=None, delta=1.0, max_iter=50):
    rot, tran = np.array(rot, dtype=np.float64), np.array(tran, dtype=np.float64)
    free = {0: [0, 1, 2], 1: [3, 4, 5], 2: [0, 1, 2, 3, 4, 5]}[mode]

    def evaluate(r, t):
        H, g, cost, _, _ = rn.normal_equations(mode, x1, x2, r, t, d1, d2, delta, d12)
        return H[np.ix_(free, free)], g[free], cost

    H, g, cost = evaluate(rot, tran)
    scale = 1.0 / (1.0 + np.sqrt(np.diag(H)))
    radius, nu, D = 1e4, 2.0, None
    info = dict(iterations=0, successful=0, termination=None, initial_cost=cost)
    it = 0
    while True:
        if it >= max_iter:
            info["termination"] = "no_convergence"; break
        if np.abs(g).max() <= 1e-10:
            info["termination"] = "gradient"; break
        if radius < 1e-32:
            info["termination"] = "min_radius"; break
        it += 1
        info["iterations"] = it
        Hs = H * np.outer(scale, scale)
        gs = g * scale
        if D is None:
            D = np.clip(np.diag(Hs), 1e-6, 1e32)
        y = np.linalg.solve(Hs + np.diag(D / radius), -gs)
        model = -gs @ y - 0.5 * y @ Hs @ y
        if not model > 0:
            radius /= nu; nu *= 2
            continue
        step = np.zeros(6)
        step[free] = scale * y
        r_new, t_new = rot + step[:3], tran + step[3:]
        Hn, gn, cost_new = evaluate(r_new, t_new)
        x = np.concatenate([rot, tran])[free]
        if np.linalg.norm(step[free]) <= 1e-8 * (np.linalg.norm(x) + 1e-8):
            info["termination"] = "parameter"; break
        if abs(cost - cost_new) <= 1e-6 * cost:
            info["termination"] = "function"; break
        rho = (cost - cost_new) / model
        if rho > 1e-3:
            rot, tran, H, g, cost = r_new, t_new, Hn, gn, cost_new
            info["successful"] += 1
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            nu, D = 2.0, None
        else:
            radius /= nu; nu *= 2
    info["cost"] = cost
    return rot, tran, info
