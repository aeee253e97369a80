"""Batch-selection API of the reference, kept verbatim at the boundary.

* :class:`BASQ` -- ``run_rchq(pts_nys, pts_rec, w_IS, kernel) -> (x, w)``  (``BASQ/_basq.py:59-80``;
  note the argument order differs from ``recombination``).  The reference's ``BASQ`` also owns GP
  fitting, samplers and defaults (``BASQ/_parameters.py``); those stay in the user's PyTorch/GPyTorch
  loop -- this class carries only what the recombination path reads: ``batch_size`` and ``device``.
* :class:`KernelQuadrature` -- ``rchq`` (``BASQ/_quadrature.py:29-51``) and ``quadrature`` (``:53-64``):
  ``EZy = w . mean_predict(X)``, ``VarZy = w^T K(X, X) w``.
"""
from __future__ import annotations

import torch

from ._rchq import recombination


class BASQ:
    def __init__(self, batch_size: int = 100, device=None, kernel=None):
        self.batch_size = int(batch_size)      # _parameters.py:40
        self.device = torch.device(device if device is not None else "cuda")
        self.kernel = kernel

    def run_rchq(self, pts_nys, pts_rec, w_IS, kernel):
        """-> ``(x, w)``: the selected batch ``pts_rec[idx]`` and its positive quadrature weights."""
        idx, w = recombination(
            pts_rec,
            pts_nys,
            self.batch_size,
            kernel,
            self.device,
            init_weights=w_IS,
        )
        x = pts_rec.to(idx.device)[idx]
        return x, w


class KernelQuadrature:
    def __init__(self, n_rec, n_nys, n_quad, batch_size, sampler, kernel, device, mean_predict=None):
        self.n_rec = n_rec
        self.n_nys = n_nys
        self.n_quad = n_quad
        self.batch_size = batch_size
        self.sampler = sampler
        self.kernel = kernel
        self.device = torch.device(device)
        # reference: ``self.predict_mean`` of the GP wrapper (_parameters.py:174-190); structured kernels carry it
        self.mean_predict = mean_predict if mean_predict is not None else getattr(kernel, "predict_mean", None)

    def rchq(self, pts_nys, pts_rec, w_IS, batch_size, kernel):
        idx, w = recombination(pts_rec, pts_nys, batch_size, kernel, self.device, init_weights=w_IS)
        x = pts_rec.to(idx.device)[idx]
        return x, w

    def quadrature(self):
        """-> ``(EZy, VarZy)`` (``_quadrature.py:53-64``)."""
        pts_nys, pts_rec, w_IS = self.sampler(self.n_quad)
        X, w = self.rchq(pts_nys, pts_rec, w_IS, self.batch_size, self.kernel)
        EZy = (w @ self.mean_predict(X)).item()
        VarZy = (w @ self.kernel(X, X) @ w).item()
        return EZy, VarZy
