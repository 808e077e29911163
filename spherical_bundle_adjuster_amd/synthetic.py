"""Seeded synthetic unit-sphere correspondences.

The reference ships no images or fixtures, so every workload is synthetic:

* ``rotation_only``  (configs C1 twin / C2 / C4): x1 uniform on S^2, ground-truth rotation with
  angle U(5deg, 40deg) about a uniform axis, x2 = normalize(R x1 + sigma n), a fraction of
  outliers replaced by uniform S^2 points (exercises the Huber branch); t = 0, d1 = d2 = 1.
* ``full_rt``        (config C3): 3-D points X with depth |X| ~ U(2, 20); x1 = X/|X|, d1 = |X|,
  Y = R X - t, x2 = Y/|Y|, d2 = |Y| so that e = d2 x2 - d1 R x1 + t = 0 exactly at the truth
  (before noise); |t| = 1 like the output of decomposeEssentialMat
  (reference spherical_bundle_adjuster.cpp:83-85).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

BASE_SEED = 20240601


@dataclass
class Correspondences:
    x1: np.ndarray            # (n, 3) f64 left unit vectors  (key_point_left_rect)
    x2: np.ndarray            # (n, 3) f64 right unit vectors (key_point_right_rect)
    d12: np.ndarray | None    # (n, 2) per-match depths or None
    rot_true: np.ndarray      # (3,) angle-axis
    tran_true: np.ndarray     # (3,)
    rot_init: np.ndarray      # perturbed start
    tran_init: np.ndarray


def _unit(v: np.ndarray) -> np.ndarray:
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def rodrigues(w: np.ndarray) -> np.ndarray:
    """Rotation matrix of an angle-axis vector (numpy; used for data generation only)."""
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * (K @ K)


def _sphere(rng: np.random.Generator, n: int) -> np.ndarray:
    return _unit(rng.standard_normal((n, 3)))


def _true_rotation(rng: np.random.Generator) -> np.ndarray:
    axis = _unit(rng.standard_normal(3))
    return axis * np.deg2rad(rng.uniform(5.0, 40.0))


def _perturb(rng: np.random.Generator, w: np.ndarray, deg: float) -> np.ndarray:
    return w + _unit(rng.standard_normal(3)) * np.deg2rad(deg)


def rotation_only(n: int, seed: int = BASE_SEED, sigma: float = 1e-3, outlier_fraction: float = 0.05,
                  perturb_deg: float = 5.0, shard: int = 0) -> Correspondences:
    """`seed` fixes the ground truth and the start point; (`seed`, `shard`) fixes the points, so ranks of a
    sharded run draw independent correspondences of the SAME two-view geometry."""
    rng = np.random.default_rng(seed)
    w = _true_rotation(rng)
    w0 = _perturb(rng, w, perturb_deg)
    if shard:
        rng = np.random.default_rng([seed, shard])
    R = rodrigues(w)
    x1 = _sphere(rng, n)
    x2 = x1 @ R.T
    if sigma > 0:
        x2 = x2 + sigma * rng.standard_normal((n, 3))
    x2 = _unit(x2) if n else x2
    if outlier_fraction > 0 and n:
        out = rng.random(n) < outlier_fraction
        x2[out] = _sphere(rng, int(out.sum()))
    return Correspondences(np.ascontiguousarray(x1), np.ascontiguousarray(x2), None, w, np.zeros(3), w0, np.zeros(3))


def full_rt(n: int, seed: int = BASE_SEED + 2, sigma: float = 1e-3, outlier_fraction: float = 0.05,
            perturb_deg: float = 5.0, depth_noise: float = 0.0, shard: int = 0) -> Correspondences:
    rng = np.random.default_rng(seed)
    w = _true_rotation(rng)
    R = rodrigues(w)
    t = _unit(rng.standard_normal(3))
    w0 = _perturb(rng, w, perturb_deg)
    t0 = _unit(t + 0.1 * rng.standard_normal(3))
    if shard:
        rng = np.random.default_rng([seed, shard])
    X = _sphere(rng, n) * rng.uniform(2.0, 20.0, size=(n, 1))
    d1 = np.linalg.norm(X, axis=1)
    x1 = X / d1[:, None] if n else X
    Y = X @ R.T - t
    d2 = np.linalg.norm(Y, axis=1)
    x2 = Y / d2[:, None] if n else Y
    if sigma > 0 and n:
        x2 = _unit(x2 + sigma * rng.standard_normal((n, 3)))
    if outlier_fraction > 0 and n:
        out = rng.random(n) < outlier_fraction
        x2[out] = _sphere(rng, int(out.sum()))
    d12 = np.stack([d1, d2], axis=1)
    if depth_noise > 0 and n:
        d12 = d12 * (1.0 + depth_noise * rng.standard_normal((n, 2)))
    return Correspondences(np.ascontiguousarray(x1), np.ascontiguousarray(x2), np.ascontiguousarray(d12), w, t, w0, t0)


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of n correspondences for `rank` of `world`."""
    per = -(-n // world)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)
