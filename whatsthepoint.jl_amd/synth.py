"""Synthetic workloads of SURVEY.md §8d.  The counter-based generator is implemented three
times with identical output: here (numpy), csrc/wtp_hash.hip gen_uniform_kernel (device) and
oracle/wtp_oracle.c (C)."""
from __future__ import annotations

import numpy as np

SEED = 20260821


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform(n: int, dim: int = 3, dtype=np.float32, seed: int = SEED, first: int = 0):
    """n points uniform in [0,1)^dim: (splitmix64(seed*2^40 + 3*i + axis) >> 40) * 2^-24."""
    with np.errstate(over="ignore"):
        i = np.arange(first, first + n, dtype=np.uint64)[:, None] * np.uint64(3) + np.arange(dim, dtype=np.uint64)[None, :]
        h = _splitmix64((np.uint64(seed) << np.uint64(40)) + i)
    return ((h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(dtype)


def graded(n: int, h_ratio: float = 4.0, delta: float = 0.2, dtype=np.float32, seed: int = SEED):
    """Graded cloud (config C5): thinning of the uniform stream with acceptance (h_w/h(x))^3,
    h = BoundaryLayerSpacing sigmoid (src/discretization/spacings.jl:121-133) of the distance to
    the nearest cube face, h_bulk/h_wall = h_ratio.  Returns the first n accepted points."""
    out, got, first = [], 0, 0
    while got < n:
        m = max(4 * (n - got), 1 << 16)
        x = uniform(m, 3, np.float64, seed, first)
        u = uniform(m, 1, np.float64, seed + 1, first)[:, 0]
        first += m
        d = np.minimum(x, 1.0 - x).min(axis=1)
        sig = 1.0 / (1.0 + np.exp(-(d - delta / 2) / (delta / 6)))
        h = 1.0 + (h_ratio - 1.0) * sig  # in units of h_wall
        keep = u < (1.0 / h) ** 3
        out.append(x[keep])
        got += int(keep.sum())
    return np.concatenate(out)[:n].astype(dtype)
