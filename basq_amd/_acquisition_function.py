"""The sparse-GMM square-root acquisition of ``BASQ/_acquisition_function.py``, rebuilt around
two mixture constructions and the device mat-vec density.

An RBF GP with ``W = l^2 I`` makes everything here a product of Gaussians.  With ``k(x, X_i) = v N(x; X_i, W)``
(``v = s2 sqrt(det 2 pi W)``, ``_gaussian_calc.py:50-51``), prior ``N(m0, S0)`` and ``P = K(X,X)^-1`` (woodbury_inv):

* variance x prior (``sparseGMM``, reference ``:33-62``).  ``prior(x) [s2 - sum_ij P_ij k(x,X_i) k(x,X_j)]``; a pair
  with ``P_ij < 0`` ADDS mass ``|v^2 P_ij|`` along ``N(x; X_i, W) N(x; X_j, W) prior(x)``, a Gaussian in ``x`` with
      precision  2/l^2 I + S0^-1          mean  cov ((X_i + X_j)/l^2 + S0^-1 m0).
  The reference keeps the ``n_gaussians`` heaviest ordered pairs above ``threshold`` next to the prior itself
  (weights renormalised over what is kept) -- pair order and truncation are observable through the sampling counts,
  so they are kept exactly; the algebra is done in the isotropic form above (no matrix inverses of ``W``).
* mean x prior (``sparseGMM_mean``, ``:122-165``).  ``(m(x) - c) prior(x) = sum_i a_i v N(x; X_i, W) prior(x)``:
  component i has precision ``1/l^2 I + S0^-1``, mean ``cov (X_i/l^2 + S0^-1 m0)`` and signed mass
  ``a_i N(X_i; m0, W + S0)``.  The reference ranks by mass x ``N(mean_i; m0, cov)``, splits its budget between the
  positive and negative parts in proportion to their totals, takes absolute masses and drops those below
  ``threshold`` of the total.

``joint_pdf`` / ``joint_pdf_mean`` (the reference builds ``[n_x * k, d]`` difference tensors for them, ``:64-97``,
``:167-188``) are kernel mat-vecs on whitened points (:mod:`basq_amd._mixture`); ``sampling`` / ``sampling_mean``
draw ``int(n w_k)`` points per component, in component order.  Pinned by ``tests/golden/acquisition.json`` and
``sampler.json``, produced by the reference's own classes.
"""
from __future__ import annotations

import torch

from ._gaussian_calc import GaussianCalc
from ._mixture import SharedCovMixture, gauss_logpdf, mixture_pdf


def _spd_inverse(A):
    return torch.cholesky_inverse(torch.linalg.cholesky(A))


def variance_proposal(Xobs, P, lengthscale, outputscale, v, m0, S0, cap, floor):
    """-> ``(prior_share, SharedCovMixture)`` of the variance-times-prior acquisition (see the module docstring)."""
    inv_l2 = 1.0 / (lengthscale * lengthscale)
    S0inv = _spd_inverse(S0)
    eye = torch.eye(Xobs.shape[1], dtype=torch.float64, device=Xobs.device)
    cov = _spd_inverse(2.0 * inv_l2 * eye + S0inv)
    rows, cols = torch.nonzero(P < 0, as_tuple=True)                 # ordered pairs, row-major
    mass = ((v * v) * P[rows, cols]).abs()
    total = outputscale + mass.sum()
    share0, share = outputscale / total, mass / total
    ranked = share.argsort(descending=True)[:cap]
    chosen = ranked[share[ranked] > floor]
    pair_sum = Xobs[rows[chosen]] + Xobs[cols[chosen]]
    means = (pair_sum * inv_l2 + S0inv @ m0) @ cov                   # cov is symmetric: rows of (cov @ rhs^T)^T
    kept = share0 + share[chosen].sum()
    return share0 / kept, SharedCovMixture(share[chosen] / kept, means, cov)


def mean_proposal(Xobs, alpha, lengthscale, m0, S0, cap, floor):
    """-> ``SharedCovMixture`` of the |mean|-times-prior proposal (see the module docstring)."""
    d = Xobs.shape[1]
    inv_l2 = 1.0 / (lengthscale * lengthscale)
    S0inv = _spd_inverse(S0)
    eye = torch.eye(d, dtype=torch.float64, device=Xobs.device)
    cov = _spd_inverse(inv_l2 * eye + S0inv)
    centres = (Xobs * inv_l2 + S0inv @ m0) @ cov
    signed = alpha * gauss_logpdf(Xobs, m0, (lengthscale * lengthscale) * eye + S0).exp()
    signed = signed / signed.sum()
    score = signed * gauss_logpdf(centres, m0, cov).exp()
    up, down = score > 0, score < 0
    mass_up, mass_down = score[up].sum(), score[down].sum().abs()
    n_up = int(mass_up / (mass_up + mass_down) * cap)
    take_up = score[up].argsort(descending=True)[:n_up]
    take_down = score[down].argsort()[:cap - n_up]
    w = torch.cat([signed[up][take_up], signed[down][take_down].abs()])
    mu = torch.cat([centres[up][take_up], centres[down][take_down]])
    heavy = w > floor * w.sum()
    w, mu = w[heavy], mu[heavy]
    return SharedCovMixture(w / w.sum(), mu, cov)


class SquareRootAcquisitionFunction(GaussianCalc):
    """Reference-facing names (``update``, ``sparseGMM``, ``joint_pdf``, ``sampling``, ``sparseGMM_mean``, ``joint_pdf_mean``,
    ``sampling_mean``; attributes ``wA, wAA, mu_AA, sigma_AA, d_AA, w_mean, mu_mean, sig_mean, d_mean``) over the two
    mixtures above.  ``generator_parity=False`` draws every sample on the device instead of the CPU generator."""

    def __init__(self, prior, model, device, n_gaussians=100, threshold=1e-5, ops=None, generator_parity=True):
        super().__init__(prior, device, ops=ops)
        self.n_gaussians = n_gaussians
        self.threshold = threshold
        self.generator_parity = generator_parity
        self.update(model)

    def _prior_on_device(self):
        dev = self.Xobs.device
        return self.prior.loc.to(dev, torch.float64), self.prior.covariance_matrix.to(dev, torch.float64)

    def update(self, model):
        self.parameters_extraction(model)
        self.wA, self._mix_A = self._build_variance_mixture()
        self.wAA, self.mu_AA, self.sigma_AA, self.d_AA = self._mix_A.weights, self._mix_A.means, self._mix_A.cov, len(self._mix_A)
        self._mix_mean = self._build_mean_mixture()
        self.w_mean, self.mu_mean, self.sig_mean, self.d_mean = (self._mix_mean.weights, self._mix_mean.means,
                                                               self._mix_mean.cov, len(self._mix_mean))

    def _build_variance_mixture(self):
        m0, S0 = self._prior_on_device()
        s2 = torch.as_tensor(self.outputscale, dtype=torch.float64, device=self.Xobs.device)
        return variance_proposal(self.Xobs, self.woodbury_inv, self.lengthscale, s2, self.v, m0, S0, self.n_gaussians,
                                 self.threshold)

    def _build_mean_mixture(self):
        m0, S0 = self._prior_on_device()
        return mean_proposal(self.Xobs, self.woodbury_vector, self.lengthscale, m0, S0, self.n_gaussians, self.threshold)

    # ---- the reference's method names ----------------------------------------------------------------------------
    def sparseGMM(self):
        w0, mix = self._build_variance_mixture()
        return w0, mix.weights, mix.means, mix.cov

    def sparseGMM_mean(self):
        mix = self._build_mean_mixture()
        return mix.weights, mix.means, mix.cov

    def joint_pdf(self, x):
        ops = self._get_ops()
        m0, S0 = self._prior_on_device()
        unit = torch.ones(1, dtype=torch.float64, device=self.Xobs.device)
        out = self.wA * mixture_pdf(ops, x, m0.reshape(1, -1), unit, S0)
        return out if len(self._mix_A) == 0 else out + self._mix_A.pdf(ops, x)

    def joint_pdf_mean(self, x):
        return self._mix_mean.pdf(self._get_ops(), x)

    def _prior_draw(self, count):
        if self.generator_parity:
            return self.prior.sample(torch.Size([int(count)])).to(self.Xobs.device)
        from ._mixture import mvn_draw

        m0, S0 = self._prior_on_device()
        return mvn_draw(m0, S0, count, self.Xobs.device, generator_parity=False)

    def sampling(self, n):
        from_prior = self._prior_draw((n * self.wA).type(torch.int))
        if len(self._mix_A) == 0:
            return from_prior
        rest = self._mix_A.draw(n, self.Xobs.device, self.generator_parity)
        return torch.cat([from_prior.to(rest.dtype), rest])

    def sampling_mean(self, n):
        return self._mix_mean.draw(n, self.Xobs.device, self.generator_parity)
