"""Samplers of ``BASQ/_sampler.py`` on the device: ``PriorSampler`` (:7-34) and ``UncertaintySampler``
(:37-280).

``UncertaintySampler`` keeps the reference's constructor and methods (``pdf``, ``SIR``, ``approx``, ``SIR_from_mean``,
``SIR_from_AF``, ``calc_weights``, ``exact``, ``__call__``).  The heavy parts -- GP predictions and mixture densities
over ``ratio_super * n`` super-samples -- run on the HIP pairwise kernel (``basq_kernel_matvec_f64`` through
``PosteriorKernel.gp_mean`` and :mod:`basq_amd._mixture`, ``basq_gram_f64`` for the variances, in row chunks).

Random draws are made where the reference makes them when it runs on the CPU -- ``prior.sample``, the per-component
``MultivariateNormal.sample`` calls and ``torch.multinomial`` consume the CPU global generator in the same order
and with the same shapes -- so ``torch.manual_seed(k)`` reproduces the reference's pools
(``tests/test_sampler.py`` against ``tests/golden/sampler.json``).

One stated deviation: the reference's ``predict`` evaluates variances under ``gpytorch.settings.fast_pred_var()``
(LOVE, a low-rank approximation); here the variance is exact.
"""
from __future__ import annotations

import torch
from torch.distributions.multivariate_normal import MultivariateNormal

from ._acquisition_function import SquareRootAcquisitionFunction
from .kernels import PosteriorKernel, StationaryKernel


class PriorSampler:
    def __init__(self, prior, n_rec, nys_ratio, device):
        self.prior = prior
        self.n_rec = n_rec
        self.nys_ratio = nys_ratio
        self.device = torch.device(device)

    def __call__(self, n_rec):
        """-> ``(pts_nys, pts_rec, w_IS)``: Nystrom points are a prefix of the pool, weights uniform (:31-34)."""
        pts_rec = self.prior.sample(sample_shape=torch.Size([n_rec])).to(self.device)
        pts_nys = pts_rec[:int(self.n_rec * self.nys_ratio)]
        w = torch.ones(n_rec, dtype=pts_rec.dtype, device=self.device) / n_rec
        return pts_nys, pts_rec, w


class UncertaintySampler(SquareRootAcquisitionFunction):
    PREDICT_CHUNK = 1 << 17          # rows per variance chunk (chunk x n_obs doubles of Gram block)

    def __init__(self, prior, model, n_rec, nys_ratio, device, sampling_method="approx", ratio=0.5, ratio_super=100,
                 n_gaussians=100, threshold=1e-5, ops=None):
        super().__init__(prior, model, device, n_gaussians=n_gaussians, threshold=threshold, ops=ops)
        self.model = model
        self.ratio = ratio
        self.nys_ratio = nys_ratio
        self.ratio_super = ratio_super
        self.sampling_method = sampling_method

    # ---- GP predictions (``predict``, BASQ/_gp.py:213-230) -------------------------------------------------------
    def update(self, model):
        super().update(model)
        self.model = model
        noise = float(model.likelihood.noise.detach().reshape(-1)[0])
        const = float(model.mean_module.constant.detach().reshape(-1)[0])
        base = StationaryKernel("rbf", self.lengthscale, self.outputscale)
        self._post = PosteriorKernel(base, self.Xobs, self.woodbury_inv, noise, const, self.woodbury_vector)

    def _predict(self, x, want_var=True):
        """-> ``(mean, var)`` of ``model.likelihood(model(x))``; the variance in row chunks of the Gram block."""
        ops = self._get_ops()
        x = ops.to_device(x, torch.float64).contiguous()
        center = ops.col_mean(x)
        mean = self._post.gp_mean(ops, x, center)
        if not want_var:
            return mean, None
        var = torch.cat([self._post.gp_variance(ops, x[lo:lo + self.PREDICT_CHUNK].contiguous(), center)
                         for lo in range(0, x.shape[0], self.PREDICT_CHUNK)]) if x.shape[0] else ops.zeros(0)
        return mean, var

    def _prior_log_prob(self, X):
        loc, Sigma = self._prior_on_device()
        return MultivariateNormal(loc, Sigma).log_prob(X.to(loc))

    # ---- the reference's methods -------------------------------------------------------------------------------
    def pdf(self, X):                                                          # :72-88
        if self.ratio == 0:
            return self._prior_log_prob(X).exp()
        if self.ratio == 1:
            return self.joint_pdf(X)
        g_pdf = self.joint_pdf(X)
        f_pdf = self._prior_log_prob(X).exp()
        return ((1 - self.ratio) * f_pdf + self.ratio * g_pdf) / f_pdf

    def SIR(self, X, weights, n_return):                                       # :90-104
        """``torch.multinomial`` on the CPU global generator (where the reference's CPU run draws it)."""
        draw = torch.multinomial(weights.detach().to("cpu"), n_return)
        return X[draw.to(X.device)]

    def approx(self, n):                                                       # :106-141
        dev = self.Xobs.device
        if self.ratio == 0:
            pts_rec = self.prior.sample(torch.Size([n])).to(dev)
        elif self.ratio == 1:
            pts_rec = self.sampling(n)
        else:
            first = self.sampling(int(self.ratio * n))
            second = self.prior.sample(torch.Size([int((1 - self.ratio) * n)])).to(dev)
            pts_rec = torch.cat([first, second.to(first.dtype)])
        mean, _ = self._predict(pts_rec, want_var=False)
        w = torch.exp(torch.log(torch.abs(mean)) + self._prior_log_prob(pts_rec) - torch.nan_to_num(self.pdf(pts_rec)))
        w = torch.nan_to_num(w)
        if torch.sum(w) == 0:
            weights = torch.ones(len(w), dtype=w.dtype, device=w.device) / len(w)
        else:
            weights = w / torch.sum(w)
        n_nys = int(n * self.nys_ratio)
        pts_nys = self.SIR(pts_rec, weights, n_nys)
        return pts_nys, pts_rec, weights

    def SIR_from_mean(self, n_super, n):                                       # :143-166
        X_pi = self.sampling_mean(n_super)
        mean, _ = self._predict(X_pi, want_var=False)
        mean_log = mean.abs().log()
        prior_log = self._prior_log_prob(X_pi).exp().log()                     # safe_mvn_prob(...).log()
        sampler_log = torch.nan_to_num(self.joint_pdf_mean(X_pi))              # (sic: no log in the reference)
        w_mpi_B = torch.exp(mean_log + prior_log - sampler_log)
        return self.SIR(X_pi, w_mpi_B, n)

    def SIR_from_AF(self, n_super, n):                                         # :168-191
        X_A = self.sampling(n_super)
        _, var_A = self._predict(X_A)
        prior_log = self._prior_log_prob(X_A).exp().log()
        sampler_log = torch.nan_to_num(self.joint_pdf(X_A)).log()
        w_C_A = torch.exp(var_A.log() + prior_log - sampler_log)
        return self.SIR(X_A, w_C_A, n)

    def calc_weights(self, pts_rec):                                           # :193-216
        mean_rec, var_rec = self._predict(pts_rec)
        lp = self._prior_log_prob(pts_rec)
        f_rec = torch.exp(torch.abs(mean_rec).log() + lp)
        if self.ratio < 1:
            g_rec = torch.exp(torch.log(self.ratio * var_rec + (1 - self.ratio) * torch.abs(mean_rec)) + lp)
        else:
            g_rec = torch.exp(torch.log(torch.tensor(float(self.ratio), dtype=torch.float64, device=lp.device))
                              + var_rec.log() + lp)
        w_IC = f_rec / g_rec
        return w_IC / w_IC.sum()

    def exact(self, n):                                                        # :218-263
        n_nys = int(n * self.nys_ratio)
        dev = self.Xobs.device
        if self.ratio == 0:
            n_super = int(self.ratio_super * n)
            pts_rec = self.SIR_from_mean(n_super, n)
            return pts_rec[:n_nys], pts_rec, torch.ones(n, dtype=torch.float64, device=dev) / n
        if self.ratio == 1:
            n_super = int(self.ratio_super * n)
            pts_rec = self.SIR_from_AF(n_super, n)
            w_IC = self.calc_weights(pts_rec)
            pts_nys = self.SIR_from_mean(n, n_nys)
            return pts_nys, pts_rec, w_IC
        n_super = int(self.ratio_super * (1 - self.ratio) * n)
        n_pi = int((1 - self.ratio) * n)
        X_f = self.SIR_from_mean(n_super, n_pi)
        pts_nys = X_f[:n_nys]
        n_super = int(self.ratio_super * self.ratio * n)
        n_rec = int(self.ratio * n)
        X_rec = self.SIR_from_AF(n_super, n_rec)
        pts_rec = torch.cat([X_rec, X_f])
        return pts_nys, pts_rec, self.calc_weights(pts_rec)

    def __call__(self, n):                                                     # :265-280
        if self.sampling_method == "approx":
            return self.approx(n)
        if self.sampling_method == "exact":
            return self.exact(n)
        raise Exception("The given sampling method is undefined.")
