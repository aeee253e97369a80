"""``PriorSampler`` (``BASQ/_sampler.py:7-34``) with the pool created on the device."""
from __future__ import annotations

import torch


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
