"""Mirror of ``BASQ/_acquisition_function.py``: the sparse-GMM square-root acquisition.

``SquareRootAcquisitionFunction(prior, model, device)`` keeps the reference's methods.  The small, n_obs-sized
algebra of ``sparseGMM`` / ``sparseGMM_mean`` is restated with torch on the device (line references inline);
the two density evaluations over a large ``x`` -- ``joint_pdf`` (:64-97) and ``joint_pdf_mean`` (:167-188), which
the uncertainty sampler calls on ~100 x n_rec points -- run as kernel mat-vecs (:mod:`basq_amd._mixture`).
"""
from __future__ import annotations

import torch
from torch.distributions.multivariate_normal import MultivariateNormal

from ._gaussian_calc import GaussianCalc
from ._mixture import mixture_pdf


class SquareRootAcquisitionFunction(GaussianCalc):
    def __init__(self, prior, model, device, n_gaussians=100, threshold=1e-5, ops=None):
        super().__init__(prior, device, ops=ops)
        self.n_gaussians = n_gaussians
        self.threshold = threshold
        self.update(model)

    def _prior_on_device(self):
        dev = self.Xobs.device
        return self.prior.loc.to(dev, torch.float64), self.prior.covariance_matrix.to(dev, torch.float64)

    def update(self, model):                                                   # :22-31
        self.parameters_extraction(model)
        self.wA, self.wAA, self.mu_AA, self.sigma_AA = self.sparseGMM()
        self.d_AA = len(self.mu_AA)
        self.w_mean, self.mu_mean, self.sig_mean = self.sparseGMM_mean()
        self.d_mean = len(self.mu_mean)

    def sparseGMM(self):                                                       # :33-62
        loc, Sigma = self._prior_on_device()
        i, j = torch.where(self.woodbury_inv < 0)
        _w1_ = torch.as_tensor(self.outputscale, dtype=torch.float64, device=self.Xobs.device)
        _w2_ = torch.abs((self.v ** 2) * self.woodbury_inv[i, j])
        _Z = _w1_ + torch.sum(_w2_)
        _w1, _w2 = _w1_ / _Z, _w2_ / _Z
        Winv = self.W.inverse()
        Sinv = Sigma.inverse()
        sigma2 = (2 * Winv + Sinv).inverse()
        _idx = _w2.argsort(descending=True)[:self.n_gaussians]
        idx = _idx[_w2[_idx] > self.threshold]
        Xi = self.Xobs[i[idx]]
        Xj = self.Xobs[j[idx]]
        w2 = _w2[idx]
        mu2 = (sigma2 @ Winv @ (Xi + Xj).T).T + sigma2 @ Sinv @ loc
        zA = _w1 + torch.sum(w2)
        return _w1 / zA, w2 / zA, mu2, sigma2

    def joint_pdf(self, x):                                                    # :64-97
        ops = self._get_ops()
        loc, Sigma = self._prior_on_device()
        one = torch.ones(1, dtype=torch.float64, device=self.Xobs.device)
        first = self.wA * mixture_pdf(ops, x, loc.reshape(1, -1), one, Sigma)
        if len(self.wAA) == 0:
            return first
        return first + mixture_pdf(ops, x, self.mu_AA, self.wAA, self.sigma_AA)

    def _mvn_draw(self, loc, cov, count):
        """``MultivariateNormal(loc, cov).sample([count])`` drawn from the CPU global generator (as the reference
        does when it runs on the CPU -- same shapes, same order, so ``torch.manual_seed`` reproduces its samples),
        then moved to the device."""
        mvn = MultivariateNormal(loc.detach().to("cpu", torch.float64), cov.detach().to("cpu", torch.float64))
        return mvn.sample(torch.Size([int(count)])).to(self.Xobs.device)

    def sampling(self, n):                                                     # :99-120
        cntA = (n * self.wA).type(torch.int)
        samplesA = self.prior.sample(torch.Size([int(cntA)])).to(self.Xobs.device)
        if len(self.wAA) == 0:
            return samplesA
        cntAA = (n * self.wAA).type(torch.int).tolist()
        samplesAA = torch.cat([self._mvn_draw(self.mu_AA[i], self.sigma_AA, cnt) for i, cnt in enumerate(cntAA)])
        return torch.cat([samplesA.to(samplesAA.dtype), samplesAA])

    def sparseGMM_mean(self):                                                  # :122-165
        loc, Sigma = self._prior_on_device()
        Winv = self.W.inverse()
        Sinv = Sigma.inverse()
        sig_prime = (Winv + Sinv).inverse()
        mu_prime = (sig_prime @ ((Winv @ self.Xobs.T).T + Sinv @ loc).T).T
        npdfs = MultivariateNormal(loc, self.W + Sigma).log_prob(self.Xobs).exp()
        omega_prime = self.woodbury_vector * npdfs
        weights = omega_prime / omega_prime.sum()
        W_prime = weights * MultivariateNormal(loc, sig_prime).log_prob(mu_prime).exp()
        W_pos = W_prime[W_prime > 0].sum()
        W_neg = W_prime[W_prime < 0].sum().abs()
        N_pos = int(W_pos / (W_pos + W_neg) * self.n_gaussians)
        N_neg = self.n_gaussians - N_pos
        idx_pos = W_prime[W_prime > 0].argsort(descending=True)[:N_pos]
        idx_neg = W_prime[W_prime < 0].argsort()[:N_neg]
        weights_pos = weights[W_prime > 0][idx_pos]
        weights_neg = weights[W_prime < 0][idx_neg].abs()
        weights = torch.cat([weights_pos, weights_neg])
        mu_mean = torch.cat([mu_prime[W_prime > 0][idx_pos], mu_prime[W_prime < 0][idx_neg]])
        idx_weights = weights > (self.threshold * weights.sum())
        weights = weights[idx_weights]
        mu_mean = mu_mean[idx_weights]
        return weights / weights.sum(), mu_mean, sig_prime

    def joint_pdf_mean(self, x):                                               # :167-188
        return mixture_pdf(self._get_ops(), x, self.mu_mean, self.w_mean, self.sig_mean)

    def sampling_mean(self, n):                                                # :190-206
        cnts = (n * self.w_mean).type(torch.int).tolist()
        return torch.cat([self._mvn_draw(self.mu_mean[i], self.sig_mean, cnt) for i, cnt in enumerate(cnts)])
