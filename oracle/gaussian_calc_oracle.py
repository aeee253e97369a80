"""CPU oracle for ``BASQ/_gaussian_calc.py`` (``unimodal_approximation``, :53-82) -- TEST INFRASTRUCTURE ONLY.

Restates the reference's direct formulation: all n_obs^2 pairwise Gaussian densities, normalised weights,
mixture mean and the ``[n_obs^2, d, d]`` covariance sum.  Pinned against the imported reference by
``tests/test_gaussian_calc.py::test_oracle_vs_imported_reference`` (here) and by the committed fixture
``tests/golden/gaussian_calc.json`` (everywhere).
"""
import math

import torch
from torch.distributions.multivariate_normal import MultivariateNormal


def unimodal_approximation_oracle(Xobs, woodbury_vector, lengthscale, outputscale, alpha):
    n, d = Xobs.shape
    W = torch.eye(d, dtype=Xobs.dtype) * lengthscale ** 2                       # :50
    v = outputscale * math.sqrt(float(torch.linalg.det(2 * math.pi * W)))        # :51
    x = (Xobs.unsqueeze(1) - Xobs.unsqueeze(0)).reshape(n * n, d)                # :67
    Npdfs = MultivariateNormal(torch.zeros(d, dtype=Xobs.dtype), 2 * W).log_prob(x).exp().reshape(n, n)   # :68-72
    w_raw = 0.5 * (v ** 2) * (woodbury_vector.unsqueeze(1) * woodbury_vector.unsqueeze(0)) * Npdfs       # :74
    w_m = w_raw / w_raw.sum()                                                    # :75
    xbar = (Xobs.unsqueeze(1) + Xobs.unsqueeze(0)) / 2
    mu = alpha + (w_m.unsqueeze(2) * xbar).sum(axis=0).sum(axis=0)               # :77
    Xij2 = xbar.reshape(n * n, d) - mu                                           # :78
    Wm = w_m.reshape(n * n, 1)
    cov = (Wm.unsqueeze(1) * Xij2.unsqueeze(2) @ Xij2.unsqueeze(1)).sum(axis=0) + W / 2   # :79-80
    return mu, _safe_cov(cov)                                                     # :81


def _is_psd(mat):                                                                # _utils.py:45-57
    try:
        torch.linalg.cholesky(mat)
        return bool((mat == mat.T).all() and (torch.linalg.eig(mat)[0].real >= 0).all())
    except Exception:
        return False


def _safe_cov(cov):
    """Covariance that ``Utils.safe_mvn_register`` ends up registering (_utils.py:59-81)."""
    if _is_psd(cov):
        return cov
    cov = torch.nan_to_num(cov)
    cov = torch.sqrt(cov * cov.T)
    if not _is_psd(cov):
        n = cov.size(0)
        jitter = torch.ones(n, dtype=cov.dtype) * 1e-5
        while not _is_psd(cov):
            cov[range(n), range(n)] += jitter
            jitter *= 2
    return cov
