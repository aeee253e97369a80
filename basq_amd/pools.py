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


def _host_stationary(x: torch.Tensor, y: torch.Tensor, family: str, lengthscale: float, outputscale: float) -> torch.Tensor:
    """``ScaleKernel(RBF | Matern)`` on HOST tensors, for the few hundred synthetic observations below only (the candidates'
    kernel values are the HIP library's business): squared distances of the points centred on the first operand's mean from one
    augmented product ``[-2a, |a|^2, 1] [b, 1, |b|^2]^T``, clamped at zero -- the formulation gpytorch publishes for its
    stationary kernels (``BASQ/_parameters.py:192-208`` selects them)."""
    def sq_dist(a, b):
        shift = a.mean(-2, keepdim=True)
        a, b = a - shift, b - shift
        an, bn = a.pow(2).sum(-1, keepdim=True), b.pow(2).sum(-1, keepdim=True)
        lhs = torch.cat([-2.0 * a, an, torch.ones_like(an)], dim=-1)
        rhs = torch.cat([b, torch.ones_like(bn), bn], dim=-1)
        return lhs.matmul(rhs.transpose(-2, -1)).clamp_min_(0)

    if family == "rbf":
        base = sq_dist(x.div(lengthscale), y.div(lengthscale)).div_(-2).exp_()
    else:
        mean = x.mean(dim=-2, keepdim=True)
        r = sq_dist((x - mean).div(lengthscale), (y - mean).div(lengthscale)).clamp_min_(1e-30).sqrt_()
        c5, c3 = 5.0 ** 0.5, 3.0 ** 0.5
        if family == "matern52":
            base = (c5 * r).add(1).add(5.0 / 3.0 * r ** 2) * torch.exp(-c5 * r)
        elif family == "matern32":
            base = (c3 * r).add(1) * torch.exp(-c3 * r)
        else:
            raise ValueError(family)
    return base.mul(outputscale)


def synthetic_gp_state(Xobs: torch.Tensor, family: str, lengthscale: float, outputscale: float, noise: float, seed: int):
    """A stand-in for a FITTED exact GP on the observations ``Xobs`` (host tensor), as the posterior / WSABI kernels of BASELINE
    configs 1 and 5 need one and gpytorch is not in the image: targets from a fixed smooth positive function of the inputs (no
    random numbers, no libm beyond ``exp``), then the caches gpytorch's prediction strategy would hold --

        W = (K(X, X) + noise I)^-1   (``BASQ/_gp.py:246-255``: ``S S^T`` of ``covar_cache``),   mean_cache = W (y - mean(y)).

    -> ``(W [n_obs, n_obs], mean_const, mean_cache [n_obs], y)``.  The synthetic-input twin of ``gmm_pool``: ``bench.py``'s
    ``configs`` leg and the tools build their posterior kernels from it; ``tests/test_pools.py`` holds it bit-identical to the
    generator the reference-generated goldens were made with."""
    n = Xobs.shape[0]
    K = _host_stationary(Xobs, Xobs, family, float(lengthscale), float(outputscale)) + noise * torch.eye(n, dtype=Xobs.dtype)
    proj = torch.arange(1, Xobs.shape[1] + 1, dtype=Xobs.dtype) / Xobs.shape[1]
    t = Xobs @ proj
    y = 1.0 + 0.5 * t * t / (1.0 + t * t) + 0.01 * (seed % 7)
    mean_const = float(y.mean())
    W = torch.cholesky_inverse(torch.linalg.cholesky(K))
    return W, mean_const, W @ (y - mean_const), y


def kernel_for_case(c: dict):
    """The structured kernel object of a parity case ``c`` (``tests/cases.py``: the dict stored verbatim in every golden fixture)
    built from seeds alone -- pool-style observations + ``synthetic_gp_state``."""
    from . import kernels as BK

    k = c["kernel"]
    base = BK.StationaryKernel(k["family"], k["lengthscale"], k["outputscale"])
    p = k.get("posterior")
    if p is None:
        return base
    Xobs = gmm_pool(p["n_obs"], c["d"], p["obs_seed"])
    W, mean_const, mean_cache, _ = synthetic_gp_state(Xobs, k["family"], k["lengthscale"], k["outputscale"], p["noise"],
                                                      p["obs_seed"])
    post = BK.PosteriorKernel(base, Xobs, W, p.get("diag_noise", p["noise"]))
    if k.get("warp", "none") == "none":
        return post
    return BK.WsabiKernel(post, mean_const, mean_cache, k["warp"])
