"""Gaussian-mixture density with a shared covariance, on the HIP pairwise kernel.

    pdf(x_i) = sum_k w_k N(x_i; mu_k, Sigma)

is what the reference evaluates at ``BASQ/_acquisition_function.py:64-97`` (``joint_pdf``), ``:167-188``
(``joint_pdf_mean``) and ``BASQ/experiment/gmm.py:46-56`` (the demo likelihood) by materialising all
``n_x * n_k`` differences and calling ``MultivariateNormal.log_prob``.  With ``Sigma = L L^T``,

    N(x; mu, Sigma) = (2 pi)^(-d/2) |L|^-1 exp(-1/2 |L^-1 (x - mu)|^2)

is an RBF kernel (lengthscale 1) between whitened points, so the whole mixture is ONE kernel mat-vec
(``basq_kernel_matvec_f64``): nothing of size ``n_x * n_k`` is stored.
"""
from __future__ import annotations

import math

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
