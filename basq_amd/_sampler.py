"""Pool samplers behind the recombination (``BASQ/_sampler.py``): ``PriorSampler`` (:7-34) and
``UncertaintySampler`` (:37-280) -- same constructors, method names and return triples
``(pts_nys, pts_rec, w_IS)``; the bodies are organised around three device primitives:

* ``_GPView``: posterior mean / variance of the GP at a batch of points (``predict``, ``BASQ/_gp.py:213-230``) through
  ``basq_kernel_matvec_f64`` and row-chunked ``basq_gram_f64`` blocks;
* the two proposal mixtures of :mod:`basq_amd._acquisition_function` (densities = kernel mat-vecs);
* ``_importance_resample``: weights ``exp(log target - log proposal)`` + one multinomial draw.

Every sampler of the reference is then a small recipe over log-terms evaluated once per pool:

    target "mean"      log|m(x)| + log prior(x)            proposal: the mean mixture   (``SIR_from_mean``)
    target "variance"  log C(x)  + log prior(x)            proposal: the variance mixture (``SIR_from_AF``)

Two quirks of the reference's weights are reproduced on purpose, because they are visible in its pools: the mean
recipe subtracts the proposal DENSITY, not its logarithm (``_sampler.py:160``), and so does the ``approx`` recipe
(``:131``).  Random draws: with ``generator_parity=True`` (default) every draw -- ``prior.sample``, the per-component
Gaussian draws, ``torch.multinomial`` -- is made on the CPU global generator in the reference's order, so
``torch.manual_seed(k)`` reproduces the reference's pools (``tests/golden/sampler.json``); ``generator_parity=False``
draws everything on the device (no host RNG work, no H2D copy of the pool: SURVEY f4's purpose).

One stated deviation: the reference's ``predict`` evaluates variances under ``gpytorch.settings.fast_pred_var()``
(LOVE, a low-rank approximation); here the variance is exact.
"""
from __future__ import annotations

import torch

from ._acquisition_function import SquareRootAcquisitionFunction
from ._mixture import gauss_logpdf, mvn_draw
from .kernels import PosteriorKernel, StationaryKernel


class PriorSampler:
    """``pts_rec ~ prior``, ``pts_nys`` = its first ``n_rec * nys_ratio`` rows, uniform weights (``_sampler.py:21-34``).

    ``generator_parity=False``: the pool is generated ON the device (``prior`` must be a MultivariateNormal): nothing
    is drawn on the host and nothing crosses PCIe."""

    def __init__(self, prior, n_rec, nys_ratio, device, generator_parity=True):
        self.prior = prior
        self.n_rec = n_rec
        self.nys_ratio = nys_ratio
        self.device = torch.device(device)
        self.generator_parity = generator_parity

    def __call__(self, n_rec):
        if self.generator_parity:
            pool = self.prior.sample(sample_shape=torch.Size([n_rec])).to(self.device)
        else:
            pool = mvn_draw(self.prior.loc, self.prior.covariance_matrix, n_rec, self.device, generator_parity=False)
        head = pool[:int(self.n_rec * self.nys_ratio)]
        return head, pool, torch.full((n_rec,), 1.0 / n_rec, dtype=pool.dtype, device=self.device)


class _GPView:
    """Posterior mean and (exact) predictive variance of the fitted GP on the device."""

    ROWS_PER_BLOCK = 1 << 17         # variance: rows per Gram block (block x n_obs doubles)

    def __init__(self, post: PosteriorKernel):
        self.post = post

    def mean(self, ops, x):
        x = ops.to_device(x, torch.float64).contiguous()
        return self.post.gp_mean(ops, x, ops.col_mean(x))

    def mean_var(self, ops, x):
        x = ops.to_device(x, torch.float64).contiguous()
        c = ops.col_mean(x)
        mean = self.post.gp_mean(ops, x, c)
        if x.shape[0] == 0:
            return mean, ops.zeros(0)
        blocks = [self.post.gp_variance(ops, x[r0:r0 + self.ROWS_PER_BLOCK].contiguous(), c)
                  for r0 in range(0, x.shape[0], self.ROWS_PER_BLOCK)]
        return mean, torch.cat(blocks)


class UncertaintySampler(SquareRootAcquisitionFunction):
    def __init__(self, prior, model, n_rec, nys_ratio, device, sampling_method="approx", ratio=0.5, ratio_super=100,
                 n_gaussians=100, threshold=1e-5, ops=None, generator_parity=True):
        super().__init__(prior, model, device, n_gaussians=n_gaussians, threshold=threshold, ops=ops,
                         generator_parity=generator_parity)
        if sampling_method not in ("approx", "exact"):
            raise Exception("The given sampling method is undefined.")
        self.model = model
        self.ratio = ratio
        self.nys_ratio = nys_ratio
        self.ratio_super = ratio_super
        self.sampling_method = sampling_method

    def update(self, model):
        super().update(model)
        self.model = model
        noise = float(model.likelihood.noise.detach().reshape(-1)[0])
        const = float(model.mean_module.constant.detach().reshape(-1)[0])
        rbf = StationaryKernel("rbf", self.lengthscale, self.outputscale)
        self._gp = _GPView(PosteriorKernel(rbf, self.Xobs, self.woodbury_inv, noise, const, self.woodbury_vector))

    # ---- building blocks ---------------------------------------------------------------------------------------------
    def _log_prior(self, X):
        m0, S0 = self._prior_on_device()
        return gauss_logpdf(X.to(m0), m0, S0)

    def _draw_index(self, weights, k):
        """``torch.multinomial(weights, k)`` -- on the CPU generator under generator parity (where the reference draws)."""
        if self.generator_parity:
            return torch.multinomial(weights.detach().to("cpu"), k).to(weights.device)
        return torch.multinomial(weights, k)

    def _importance_resample(self, X, log_target, minus, k):
        """``k`` rows of ``X`` drawn with probabilities ``exp(log_target - minus)`` (unnormalised)."""
        return X[self._draw_index(torch.exp(log_target - minus), k)]

    def _from_mean_mixture(self, n_super, k):
        X = self.sampling_mean(n_super)
        log_target = self._gp.mean(self._get_ops(), X).abs().log() + self._log_prior(X)
        # reference quirk (:160): the proposal's density is subtracted as is, not its logarithm
        return self._importance_resample(X, log_target, torch.nan_to_num(self.joint_pdf_mean(X)), k)

    def _from_variance_mixture(self, n_super, k):
        X = self.sampling(n_super)
        _, var = self._gp.mean_var(self._get_ops(), X)
        log_target = var.log() + self._log_prior(X)
        return self._importance_resample(X, log_target, torch.nan_to_num(self.joint_pdf(X)).log(), k)

    def _proposal_pool(self, n):
        """``ratio`` of the pool from the variance mixture, the rest from the prior (mixture part first)."""
        if self.ratio == 0:
            return self._prior_draw(n)
        if self.ratio == 1:
            return self.sampling(n)
        a = self.sampling(int(self.ratio * n))
        return torch.cat([a, self._prior_draw(int((1 - self.ratio) * n)).to(a.dtype)])

    # ---- the reference's methods ---------------------------------------------------------------------------------------
    def pdf(self, X):
        """Density of the ``approx`` proposal relative to the prior (``:72-88``; the prior itself for ratio 0)."""
        if self.ratio == 0:
            return self._log_prior(X).exp()
        mix = self.joint_pdf(X)
        if self.ratio == 1:
            return mix
        pri = self._log_prior(X).exp()
        return ((1 - self.ratio) * pri + self.ratio * mix) / pri

    def SIR(self, X, weights, n_return):
        return X[self._draw_index(weights, n_return)]

    def SIR_from_mean(self, n_super, n):
        return self._from_mean_mixture(n_super, n)

    def SIR_from_AF(self, n_super, n):
        return self._from_variance_mixture(n_super, n)

    def approx(self, n):
        pool = self._proposal_pool(n)
        log_target = self._gp.mean(self._get_ops(), pool).abs().log() + self._log_prior(pool)
        w = torch.nan_to_num(torch.exp(log_target - torch.nan_to_num(self.pdf(pool))))    # (:131: density, not log)
        total = w.sum()
        w = w / total if total != 0 else torch.full_like(w, 1.0 / len(w))
        return self.SIR(pool, w, int(n * self.nys_ratio)), pool, w

    def calc_weights(self, pts_rec):
        """Normalised ``|m| prior / g`` with ``g = (r C + (1 - r) |m|) prior`` (``r C prior`` for r = 1), ``:193-216``."""
        mean, var = self._gp.mean_var(self._get_ops(), pts_rec)
        lp = self._log_prior(pts_rec)
        num = torch.exp(mean.abs().log() + lp)
        if self.ratio < 1:
            den = torch.exp(torch.log(self.ratio * var + (1 - self.ratio) * mean.abs()) + lp)
        else:
            den = torch.exp(torch.log(torch.tensor(float(self.ratio), dtype=torch.float64, device=lp.device)) + var.log() + lp)
        q = num / den
        return q / q.sum()

    def exact(self, n):
        k_nys = int(n * self.nys_ratio)
        r, big = self.ratio, self.ratio_super
        if r == 0:                                               # everything from |m| prior: uniform weights
            pool = self._from_mean_mixture(int(big * n), n)
            return pool[:k_nys], pool, torch.full((n,), 1.0 / n, dtype=torch.float64, device=self.Xobs.device)
        if r == 1:                                               # pure uncertainty sampling; Nystrom points from |m| prior
            pool = self._from_variance_mixture(int(big * n), n)
            weights = self.calc_weights(pool)
            return self._from_mean_mixture(n, k_nys), pool, weights
        from_mean = self._from_mean_mixture(int(big * (1 - r) * n), int((1 - r) * n))
        from_var = self._from_variance_mixture(int(big * r * n), int(r * n))
        pool = torch.cat([from_var, from_mean])
        return from_mean[:k_nys], pool, self.calc_weights(pool)

    def __call__(self, n):
        return self.approx(n) if self.sampling_method == "approx" else self.exact(n)
