"""Synthetic candidate pools for the recombination benchmark and the parity tests.

The reference draws its pools from a torch prior (``BASQ/_sampler.py:31-34``:
``pts_rec = prior.sample([n_rec]); pts_nys = pts_rec[:int(n_rec * nys_ratio)];
w = ones(n_rec) / n_rec``) and its demo likelihood is a Gaussian mixture
(``BASQ/experiment/gmm.py:10-56``).  Here the pool itself is a Gaussian-mixture
sample, generated so that *every machine produces the same bits*: golden
fixtures store only seeds and expected outputs, so the pool must not depend on
libm, SIMD width or the torch build.

Construction (integer draws + exactly-representable float arithmetic only):

* all randomness comes from ``numpy.random.Generator(PCG64(seed)).integers`` on
  32-bit words (a pure integer stream);
* a standard normal deviate is the Irwin-Hall sum of twelve U[0,1) variates
  minus 6 (mean 0, variance 1, support [-6, 6]); each variate is ``k * 2**-32``
  so the twelve-term sum is exact in float64;
* component means are ``3 * (2u - 1)`` with ``u = k * 2**-32``;
* a point is ``mean[component] + z`` -- one IEEE add per coordinate.
"""
from __future__ import annotations

import hashlib

import numpy as np
import torch

_TWO_M32 = 2.0 ** -32
_CHUNK_ROWS = 1 << 16


def gmm_pool(n: int, d: int, seed: int, n_components: int = 8) -> torch.Tensor:
    """Return an ``[n, d]`` float64 CPU tensor of Gaussian-mixture samples.

    Bit-reproducible across hosts for a given ``(n, d, seed, n_components)``.
    """
    if n < 0 or d <= 0 or n_components <= 0:
        raise ValueError("gmm_pool: need n >= 0, d > 0, n_components > 0")
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    u = rng.integers(0, 1 << 32, size=(n_components, d), dtype=np.uint32)
    means = 3.0 * (2.0 * (u.astype(np.float64) * _TWO_M32) - 1.0)
    comp = rng.integers(0, n_components, size=n, dtype=np.int64)
    out = np.empty((n, d), dtype=np.float64)
    for lo in range(0, n, _CHUNK_ROWS):
        hi = min(n, lo + _CHUNK_ROWS)
        k = rng.integers(0, 1 << 32, size=(hi - lo, d, 12), dtype=np.uint32)
        z = k.sum(axis=2, dtype=np.uint64).astype(np.float64) * _TWO_M32 - 6.0
        out[lo:hi] = means[comp[lo:hi]] + z
    return torch.from_numpy(out)


def prior_sampler_split(pts_rec: torch.Tensor, nys_ratio: float = 1e-2, n_nys: int | None = None):
    """Mirror of ``PriorSampler.__call__`` (``BASQ/_sampler.py:21-34``).

    ``pts_nys`` is a *prefix* of the pool and the importance weights are uniform.
    """
    n = pts_rec.shape[0]
    m = int(n * nys_ratio) if n_nys is None else int(n_nys)
    w_is = torch.ones(n, dtype=pts_rec.dtype, device=pts_rec.device) / n
    return pts_rec[:m], pts_rec, w_is


def pool_digest(t: torch.Tensor) -> str:
    """sha256 of the raw little-endian float64 bytes (fixture integrity check)."""
    a = np.ascontiguousarray(t.detach().cpu().numpy().astype("<f8", copy=False))
    return hashlib.sha256(a.tobytes()).hexdigest()
