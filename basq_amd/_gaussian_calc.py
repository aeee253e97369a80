"""Closed-form Gaussian integrals of an RBF GP -- mirror of ``BASQ/_gaussian_calc.py``.

``GaussianCalc(prior, device)`` keeps the reference's method names and return values:

* ``parameters_extraction(model)``                      ``_gaussian_calc.py:44-51``
* ``unimodal_approximation(model, alpha) -> MultivariateNormal``   ``:53-82``
* ``uniform_transformation(model, Y_unwarp)``           ``:84-109``

The reference evaluates the n_obs^2 pairwise Gaussian densities ``N(x_i - x_j; 0, 2W)`` and then builds an
``[n_obs^2, d, d]`` tensor for the mixture covariance (0.8 GB at n_obs = 1000, d = 10).  Here the pairwise
part runs through the HIP kernel mat-vec (``basq_kernel_matvec_f64``: the same fused pairwise kernel as
the block sums) and the moments are assembled from mat-vec products:

    k_ij = exp(-|x_i - x_j|^2 / (4 l^2))            (the pdf's constant and 1/2 v^2 cancel in the weights)
    g = K a,  Z = K (a . X)                          a = woodbury_vector
    S0 = a.g,   S1 = X^T (a . g),   S2 = 1/2 [ X^T diag(a . g) X + (a . X)^T Z ]
    w-mean of (x_i + x_j)/2 = S1/S0,   w-mean of its outer product = S2/S0

so nothing of size n_obs^2 is ever stored.
"""
from __future__ import annotations

import copy
import math
import warnings

import torch
from torch.distributions.multivariate_normal import MultivariateNormal
from torch.distributions.uniform import Uniform

from ._lib import ROLE_A, ROLE_B
from .kernels import StationaryKernel


def _accepts_as_covariance(mat) -> bool:
    """The acceptance test behind ``Utils.is_psd`` (``BASQ/_utils.py:45-57``), in this repository's own terms (as
    ``_basis.make_cov_psd``): exactly symmetric, a Cholesky factor exists, no negative eigenvalue.  The spectrum comes from
    the symmetric solver -- the matrix IS exactly symmetric once the first test has passed, so the reference's general
    ``eig`` sees the same eigenvalues up to round-off.  Non-finite entries fail the test (the reference's ``try`` catches the
    LAPACK error they raise)."""
    if not bool(torch.isfinite(mat).all()) or not torch.equal(mat, mat.T):
        return False
    if int(torch.linalg.cholesky_ex(mat).info.item()) != 0:
        return False
    return bool((torch.linalg.eigvalsh(mat) >= 0).all())


def safe_mvn_register(mu, cov, max_doublings: int = 60):
    """``Utils.safe_mvn_register`` (``BASQ/_utils.py:59-81``): a ``MultivariateNormal`` from an estimated covariance that may
    fail the test above.  Repair in the reference's order and arithmetic: (1) NaNs -> 0, (2) the elementwise geometric mean
    with the transpose (symmetrises; ``sqrt`` of a negative product -> NaN, which step 3 then never cures -- the
    reference loops forever there, this raises after ``max_doublings``), (3) a diagonal jitter that starts at 1e-5 and doubles
    until the matrix is accepted."""
    if _accepts_as_covariance(cov):
        return MultivariateNormal(mu, cov)
    warnings.warn("basq_amd: the estimated covariance is not positive semi-definite; repairing it (symmetrise, then diagonal "
                  "jitter), as BASQ/_utils.py:59-81 does")
    cov = torch.nan_to_num(cov)
    cov = torch.sqrt(cov * cov.T)
    jitter, tries = 1e-5, 0
    while not _accepts_as_covariance(cov):
        if tries >= max_doublings:
            raise ValueError("safe_mvn_register: the covariance estimate cannot be repaired by diagonal jitter")
        cov = cov + jitter * torch.eye(cov.shape[0], dtype=cov.dtype, device=cov.device)
        jitter, tries = 2.0 * jitter, tries + 1
    return MultivariateNormal(mu, cov)


class GaussianCalc:
    def __init__(self, prior, device, ops=None):
        self.prior = prior
        self.device = torch.device(device)
        self._ops = ops

    def _get_ops(self):
        if self._ops is None:
            from ._ops import HipOps

            self._ops = HipOps(self.device)
        return self._ops

    def get_cache(self, model):
        """``:18-42`` -> (woodbury_vector, woodbury_inv = S S^T).  When the caches do not exist yet (a freshly trained
        model: ``prediction_strategy`` is None) the reference evaluates the model once at the prior mean to build
        them (``:35-38``); same here."""
        from .kernels import _prediction_caches

        warm = None
        if getattr(model, "prediction_strategy", None) is None or \
                not hasattr(model.prediction_strategy, "covar_cache"):
            warm = self.prior.loc.view(-1).unsqueeze(0)                       # :36
        mean_cache, S = _prediction_caches(model, warm)
        return mean_cache, S @ S.T

    def parameters_extraction(self, model):
        ops = self._get_ops()
        self.Xobs = ops.to_device(copy.deepcopy(model.train_inputs[0]).detach(), torch.float64)
        self.n_data, self.n_dims = self.Xobs.size()
        wv, winv = self.get_cache(model)
        self.woodbury_vector = ops.to_device(wv.detach(), torch.float64)
        self.woodbury_inv = ops.to_device(winv.detach(), torch.float64)
        self.outputscale = float(model.covar_module.outputscale.detach().reshape(-1)[0])
        self.lengthscale = float(model.covar_module.base_kernel.lengthscale.detach().reshape(-1)[0])
        self.W = torch.eye(self.n_dims, dtype=torch.float64, device=self.Xobs.device) * self.lengthscale ** 2
        self.v = self.outputscale * math.sqrt((2 * math.pi * self.lengthscale ** 2) ** self.n_dims)

    def unimodal_approximation(self, model, alpha):
        """Moment-matched Gaussian of the GP-modelled likelihood (``:53-82``)."""
        self.parameters_extraction(model)
        ops = self._get_ops()
        X, a = self.Xobs, self.woodbury_vector
        n, d = X.shape
        # N(x_i - x_j; 0, 2 l^2 I)  is an RBF kernel with lengthscale sqrt(2) l (up to a constant)
        spec = StationaryKernel("rbf", math.sqrt(2.0) * self.lengthscale, 1.0).spec(d)
        center = ops.col_mean(X)
        pa = ops.pack(spec, X, center, ROLE_A, pad_rows_to=64)
        pb = ops.pack(spec, X, center, ROLE_B)
        g = ops.matvec(spec, pa, n, pb, n, a.contiguous(), 0.0)               # K a
        Y = a.unsqueeze(1) * X                                                 # [n, d]
        Z = torch.stack([ops.matvec(spec, pa, n, pb, n, Y[:, k].contiguous(), 0.0) for k in range(d)], 1)   # K Y
        ag = a * g
        S0 = ag.sum()
        m0 = (X * ag.unsqueeze(1)).sum(0) / S0                                 # sum_ij w_ij (x_i + x_j)/2   (:77)
        second = 0.5 * ((X * ag.unsqueeze(1)).T @ X + Y.T @ Z) / S0            # sum_ij w_ij xbar xbar^T
        mu = float(alpha) + m0 if not torch.is_tensor(alpha) else alpha.to(m0) + m0
        cov = second - torch.outer(m0, mu) - torch.outer(mu, m0) + torch.outer(mu, mu) + self.W / 2   # :78-80
        cov = 0.5 * (cov + cov.T)
        if d >= 2:
            # Reference behaviour, reproduced on purpose: its covariance sum (:80) is asymmetric at round-off
            # level, `is_psd` demands exact symmetry (_utils.py:55), so `safe_mvn_register` always takes its
            # repair branch  cov <- sqrt(cov * cov.T)  (_utils.py:72-73), i.e. the element-wise absolute value.
            cov = torch.sqrt(torch.nan_to_num(cov) * torch.nan_to_num(cov).T)
            jitter = 1e-5
            while not _accepts_as_covariance(cov):               # the reference's doubling diagonal jitter (_utils.py:75-80)
                cov = cov + jitter * torch.eye(d, dtype=cov.dtype, device=cov.device)
                jitter *= 2.0
            return MultivariateNormal(mu, cov)
        return safe_mvn_register(mu, cov)

    def uniform_transformation(self, model, Y_unwarp):
        """``:84-109`` (elementwise torch ops on the device; nothing pairwise)."""
        self.parameters_extraction(model)
        uni_min = self.Xobs.min(0)[0]
        uni_max = self.Xobs.max(0)[0]
        n_dims = self.n_dims

        def uni_sampler(N):
            return torch.stack([Uniform(uni_min[i], uni_max[i]).sample(torch.Size([N])) for i in range(n_dims)]).T

        def uni_logpdf(Xq):
            return torch.ones(Xq.size(0), dtype=torch.float64, device=Xq.device) * torch.sum(-torch.log(uni_max - uni_min))

        eps = -math.sqrt(torch.finfo(torch.float64).max)
        Y = Y_unwarp.to(self.Xobs.device, torch.float64).clone()
        Y[Y.isnan()] = eps
        Y[Y.isinf()] = eps
        Y[Y < eps] = eps
        inside = ((self.Xobs >= uni_min) & (self.Xobs <= uni_max)).all(1)
        return self.Xobs[inside], Y[inside], uni_sampler, uni_logpdf
