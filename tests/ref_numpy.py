"""Independent numpy restatement (analytic Jacobian) used to pin the C++ oracle.

Deliberately written from the closed-form derivative of Rodrigues' formula rather than from dual
numbers, so that agreement with oracle/sba_oracle.cpp is a real cross-check:
    d(R p)/dw = u w^T - a [p]x + b s I + b w p^T,   u = -a p + a'(w x p) + b' s w,   s = w.p
    a = sin(th)/th, b = (1-cos th)/th^2, a' = (cos th - a)/th^2, b' = (a - 2b)/th^2
"""
import numpy as np


def skew(p):
    return np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]], dtype=np.float64)


def rotmat(w):
    w = np.asarray(w, dtype=np.float64)
    th2 = w @ w
    if th2 > np.finfo(np.float64).eps:
        th = np.sqrt(th2)
        K = skew(w)
        return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th2 * (K @ K)
    return np.eye(3) + skew(w)


def d_rotated_d_w(w, p):
    """3x3 Jacobian of R(w) p with respect to w."""
    w = np.asarray(w, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    th2 = w @ w
    if th2 <= np.finfo(np.float64).eps:
        return -skew(p)
    th = np.sqrt(th2)
    a = np.sin(th) / th
    b = (1 - np.cos(th)) / th2
    ap = (np.cos(th) - a) / th2
    bp = (a - 2 * b) / th2
    s = w @ p
    u = -a * p + ap * np.cross(w, p) + bp * s * w
    return np.outer(u, w) - a * skew(p) + b * s * np.eye(3) + b * np.outer(w, p)


def residual_jacobian(mode, x1, x2, rot, tran, d1, d2):
    """e (3,), J (3,6) over [rot|tran]; frozen blocks zero.  mode: 0 rot, 1 tran, 2 joint."""
    R = rotmat(rot)
    e = d2 * np.asarray(x2) - (R @ (d1 * np.asarray(x1)) - np.asarray(tran))
    J = np.zeros((3, 6))
    if mode in (0, 2):
        J[:, :3] = -d1 * d_rotated_d_w(rot, x1)
    if mode in (1, 2):
        J[:, 3:] = np.eye(3)
    return e, J


def huber(delta, s):
    if delta > 0 and s > delta * delta:
        r = np.sqrt(s)
        return 2 * delta * r - delta * delta, delta / r
    return s, 1.0


def normal_equations(mode, x1, x2, rot, tran, d1=1.0, d2=1.0, delta=1.0, d12=None):
    H = np.zeros((6, 6))
    g = np.zeros(6)
    cost = 0.0
    sw = 0.0
    nout = 0.0
    for i in range(len(x1)):
        a, b = (d12[i] if d12 is not None else (d1, d2))
        e, J = residual_jacobian(mode, x1[i], x2[i], rot, tran, a, b)
        s = e @ e
        rho, w = huber(delta, s)
        H += w * (J.T @ J)
        g += w * (J.T @ e)
        cost += 0.5 * rho
        sw += w
        nout += 1.0 if (delta > 0 and s > delta * delta) else 0.0
    return H, g, cost, sw, nout


def d_rotated_d_w_mp(w, p, digits=50):
    """50-digit reference of d(R(w) p)/dw (Rodrigues on the unit axis, differentiated by mpmath)."""
    import mpmath as mp
    mp.mp.dps = digits
    w = [mp.mpf(float(x)) for x in w]
    p = [mp.mpf(float(x)) for x in p]

    def rotated(*ww):
        th = mp.sqrt(sum(x * x for x in ww))
        k = [x / th for x in ww]
        kxp = [k[1] * p[2] - k[2] * p[1], k[2] * p[0] - k[0] * p[2], k[0] * p[1] - k[1] * p[0]]
        kp = sum(a * b for a, b in zip(k, p))
        return [p[i] * mp.cos(th) + kxp[i] * mp.sin(th) + k[i] * kp * (1 - mp.cos(th)) for i in range(3)]
    J = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            order = [0, 0, 0]
            order[j] = 1
            J[i, j] = float(mp.diff(lambda a, b, c: rotated(a, b, c)[i], tuple(w), tuple(order)))
    return J
