"""Gaussian mixtures with ONE shared covariance, on the HIP pairwise kernel.

    pdf(x_i) = sum_k w_k N(x_i; mu_k, Sigma)

is what the reference evaluates at ``BASQ/_acquisition_function.py:64-97`` (``joint_pdf``), ``:167-188``
(``joint_pdf_mean``) and ``BASQ/experiment/gmm.py:46-56`` (the demo likelihood) by materialising all
``n_x * n_k`` differences and calling ``MultivariateNormal.log_prob``.  With ``Sigma = L L^T``,

    N(x; mu, Sigma) = (2 pi)^(-d/2) |L|^-1 exp(-1/2 |L^-1 (x - mu)|^2)

is an RBF kernel (lengthscale 1) between whitened points, so the whole mixture is ONE kernel mat-vec
(``basq_kernel_matvec_f64``): nothing of size ``n_x * n_k`` is stored.

``SharedCovMixture`` is the object the acquisition and the samplers are built from: weights, means, one covariance;
``pdf`` (device mat-vec) and ``draw`` (per-component sample counts ``int(n w_k)``, components in order).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from ._lib import ROLE_A, ROLE_B
from .kernels import StationaryKernel


def mixture_pdf(ops, x, means, weights, cov):
    """``x [n, d]``, ``means [k, d]``, ``weights [k]``, ``cov [d, d]`` (SPD) -> ``pdf [n]`` (float64, on the device)."""
    x = ops.to_device(x, torch.float64)
    means = ops.to_device(means, torch.float64)
    weights = ops.to_device(weights, torch.float64).reshape(-1).contiguous()
    cov = ops.to_device(cov, torch.float64)
    n, d = x.shape
    if means.shape[0] == 0:
        return ops.zeros(n)
    L = torch.linalg.cholesky(cov)
    # whitened coordinates: rows of X L^-T
    xw = torch.linalg.solve_triangular(L, x.T, upper=False).T.contiguous()
    mw = torch.linalg.solve_triangular(L, means.T, upper=False).T.contiguous()
    spec = StationaryKernel("rbf", 1.0, 1.0).spec(d)
    center = ops.col_mean(mw)
    pa = ops.pack(spec, xw, center, ROLE_A, pad_rows_to=64)
    pb = ops.pack(spec, mw, center, ROLE_B)
    s = ops.matvec(spec, pa, n, pb, means.shape[0], weights, 0.0)
    norm = (2.0 * math.pi) ** (-0.5 * d) / torch.diagonal(L).prod()
    return s * norm


def gauss_logpdf(x, mean, cov):
    """``log N(x_i; mean, cov)`` for rows of ``x`` ([n, d]); ``mean`` [d] or [n, d].  Small (n_obs-sized) inputs: plain torch."""
    L = torch.linalg.cholesky(cov)
    z = torch.linalg.solve_triangular(L, (x - mean).T, upper=False)
    d = x.shape[-1]
    return -0.5 * (z * z).sum(0) - torch.log(torch.diagonal(L)).sum() - 0.5 * d * math.log(2.0 * math.pi)


def mvn_draw(mean, cov, count, device, generator_parity=True):
    """``count`` samples of N(mean, cov) -> ``[count, d]`` on ``device``.

    ``generator_parity`` (default): drawn with ``MultivariateNormal.sample`` from the CPU global generator -- where and
    how the reference draws when it runs on the CPU, so ``torch.manual_seed`` reproduces its pools -- then moved.
    Otherwise drawn on the device (torch's device generator: Philox; no host work, no H2D copy of the pool)."""
    count = int(count)
    if generator_parity:
        from torch.distributions.multivariate_normal import MultivariateNormal

        mvn = MultivariateNormal(mean.detach().to("cpu", torch.float64), cov.detach().to("cpu", torch.float64))
        return mvn.sample(torch.Size([count])).to(device)
    mean = mean.detach().to(device, torch.float64)
    L = torch.linalg.cholesky(cov.detach().to(device, torch.float64))
    z = torch.randn(count, mean.shape[-1], dtype=torch.float64, device=device)
    return mean + z @ L.T


@dataclass
class SharedCovMixture:
    weights: torch.Tensor        # [k]  (sum to the mixture's total mass; may be empty)
    means: torch.Tensor          # [k, d]
    cov: torch.Tensor            # [d, d]

    def __len__(self):
        return int(self.means.shape[0])

    def pdf(self, ops, x):
        return mixture_pdf(ops, x, self.means, self.weights, self.cov)

    def counts(self, n):
        """Samples per component when ``n`` are requested: ``int(n w_k)`` each (truncation, as the reference)."""
        return (n * self.weights).type(torch.int).tolist()

    def draw(self, n, device, generator_parity=True):
        """Component by component, in order -> ``[sum_k int(n w_k), d]``."""
        parts = [mvn_draw(self.means[k], self.cov, c, device, generator_parity) for k, c in enumerate(self.counts(n))]
        if not parts:
            return torch.zeros(0, self.means.shape[-1], dtype=torch.float64, device=device)
        return torch.cat(parts)
