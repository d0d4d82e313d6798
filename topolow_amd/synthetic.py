"""Synthetic dissimilarity matrices of BASELINE.json's configs 3 and 4.

The generator follows the recipe of the reference's "Handling Large and Sparse Data"
example (README.md:316-342 of the reference): a 5-cluster Gaussian mixture (sigma 0.8,
cluster weights .25/.25/.20/.20/.10), Euclidean distances with 5 % multiplicative noise,
floored at 0.1, symmetric, zero diagonal.  Unlike that example the missing pairs are drawn
uniformly at random and their number is exact.  The cluster centres of the 3-D example are
extended to `latent_dim` coordinates by repeating their pattern.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_CENTRES_3D = np.array([[0.0, 0.0, 0.0],
                        [5.0, 0.0, 0.0],
                        [0.0, 5.0, 0.0],
                        [5.0, 5.0, 0.0],
                        [2.5, 2.5, 3.0]])
_WEIGHTS = np.array([0.25, 0.25, 0.20, 0.20, 0.10])


def cluster_centres(latent_dim: int) -> np.ndarray:
    reps = -(-latent_dim // 3)
    return np.tile(_CENTRES_3D, (1, reps))[:, :latent_dim].copy()


@dataclass
class SyntheticProblem:
    n: int
    latent_dim: int
    dissimilarity: np.ndarray      # n x n float64, NaN = missing, symmetric, diag 0
    true_coordinates: np.ndarray   # n x latent_dim
    missing_fraction: float


def latent_points(n: int, latent_dim: int, rng: np.random.Generator) -> np.ndarray:
    centres = cluster_centres(latent_dim)
    assign = rng.choice(len(_WEIGHTS), size=n, p=_WEIGHTS)
    return centres[assign] + rng.normal(0.0, 0.8, size=(n, latent_dim))


def make_problem(n: int, latent_dim: int = 5, missing: float = 0.7, seed: int = 12345,
                 noise_cv: float = 0.05, floor: float = 0.1) -> SyntheticProblem:
    """Dense synthetic problem (use for n up to a few 10^4: memory is O(n^2))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = latent_points(n, latent_dim, rng)
    iu, ju = np.triu_indices(n, k=1)
    npairs = iu.shape[0]
    # distances of the upper triangle, block-wise to bound temporaries
    d = np.empty(npairs, dtype=np.float64)
    step = 4_000_000
    for s in range(0, npairs, step):
        e = min(npairs, s + step)
        diff = x[iu[s:e]] - x[ju[s:e]]
        d[s:e] = np.sqrt(np.einsum("ij,ij->i", diff, diff))
    d *= 1.0 + rng.normal(0.0, noise_cv, size=npairs)
    np.maximum(d, floor, out=d)
    n_missing = int(round(missing * npairs))
    if n_missing > 0:
        u = rng.random(npairs, dtype=np.float32).astype(np.float64)
        u += rng.random(npairs) * 2.0 ** -24  # break float32 ties
        kth = np.partition(u, n_missing - 1)[n_missing - 1]
        d[u <= kth] = np.nan
    D = np.zeros((n, n), dtype=np.float64)
    D[iu, ju] = d
    D[ju, iu] = d
    return SyntheticProblem(n, latent_dim, D, x, missing)


def initial_positions(D: np.ndarray, ndim: int, seed: int) -> np.ndarray:
    """The reference's random-walk start (R/core.R:407-415) from a NumPy stream:
    row 0 is the origin, row r = row r-1 + U(0, 2*max(D)/n) per coordinate."""
    n = D.shape[0]
    rng = np.random.Generator(np.random.PCG64(seed))
    step = np.nanmax(D) / n
    steps = rng.uniform(0.0, 2.0 * step, size=(n - 1, ndim))
    pos = np.zeros((n, ndim), dtype=np.float64)
    pos[1:] = np.cumsum(steps, axis=0)
    return pos
