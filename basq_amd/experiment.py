"""Demo likelihood of the reference's driver: ``BASQ/experiment/gmm.py`` (``GMM``), evaluated on the device.

Construction consumes the torch RNG exactly as the reference does (``torch.randperm`` for the number of
components, ``torch.rand`` for every mean and for the diagonal covariance), so ``torch.manual_seed(k)`` builds
the same mixture; ``__call__`` is one kernel mat-vec (:mod:`basq_amd._mixture`)."""
from __future__ import annotations

import torch
from torch.distributions.multivariate_normal import MultivariateNormal

from ._mixture import mixture_pdf


class GMM:
    def __init__(self, dim, mu_pi, cov_pi, device, ops=None):
        self.dim = dim
        self.device = torch.device(device)
        self.mu_pi = mu_pi
        self.cov_pi = cov_pi
        self._ops = ops
        # gmm.py:18-22 (same RNG draws, in the same order)
        self.n_comp = int(torch.arange(10, 16)[torch.randperm(6)[:1]].item())
        self.means = torch.stack([3 * (2 * torch.rand(self.dim) - 1) for _ in range(self.n_comp)])
        self.cov = torch.diag(3 * torch.rand(self.dim) + 1)
        npdfs = MultivariateNormal(self.mu_pi.cpu(), self.cov_pi.cpu() + self.cov).log_prob(self.means).exp()
        self.weights = (1 / npdfs) / self.n_comp                                   # :40-44

    def _get_ops(self):
        if self._ops is None:
            from ._ops import HipOps

            self._ops = HipOps(self.device)
        return self._ops

    def __call__(self, x):
        """``sum_k w_k N(mu_k - x; 0, cov)`` (:46-56)."""
        if x.dim() == 1:
            x = x.unsqueeze(1)
        return mixture_pdf(self._get_ops(), x, self.means, self.weights, self.cov)
