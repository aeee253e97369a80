"""Batch-selection API of the reference, kept verbatim at the boundary.

* :class:`BASQ` -- ``run_rchq(pts_nys, pts_rec, w_IS, kernel) -> (x, w)``  (``BASQ/_basq.py:59-80``;
  note the argument order differs from ``recombination``).  The reference's ``BASQ`` also owns GP
  fitting, samplers and defaults (``BASQ/_parameters.py``); those stay in the user's PyTorch/GPyTorch
  loop -- this class carries only what the recombination path reads: ``batch_size`` and ``device``.
* :class:`KernelQuadrature` -- ``rchq`` (``BASQ/_quadrature.py:29-51``), ``quadrature`` (``:53-64``):
  ``EZy = w . mean_predict(X)``, ``VarZy = w^T K(X, X) w``; ``prior_max`` (``:66-84``: the same estimate over a pool drawn
  from an optimised Gaussian prior) and ``uniform_trans`` (``:86-107``: over a uniform pool, with the PRIOR kernel of an
  importance-weighted GP model and its predictive mean).
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

    def _estimate(self, X, w, mean, kernel):
        """``(w . mean, w^T K(X, X) w)`` as Python floats (``_quadrature.py:61-62``)."""
        mean = mean.to(device=w.device, dtype=w.dtype)
        K = kernel(X, X).to(device=w.device, dtype=w.dtype)
        return (w @ mean).item(), (w @ K @ w).item()

    def quadrature(self):
        """-> ``(EZy, VarZy)`` (``_quadrature.py:53-64``)."""
        pts_nys, pts_rec, w_IS = self.sampler(self.n_quad)
        X, w = self.rchq(pts_nys, pts_rec, w_IS, self.batch_size, self.kernel)
        return self._estimate(X, w, self.mean_predict(X), self.kernel)

    def prior_max(self, mvn_max):
        """-> ``(EZy, VarZy)`` when the prior is the optimised Gaussian ``mvn_max`` (``_quadrature.py:66-84``): the pool is
        ``mvn_max.sample([n_quad])`` (CPU global generator, as in the reference), its first ``n_nys`` points are the
        Nystrom sample, weights uniform."""
        pts_rec = mvn_max.sample(sample_shape=torch.Size([self.n_quad]))
        pts_nys = pts_rec[:self.n_nys]
        w_IS = torch.ones(self.n_quad) / self.n_quad
        X, w = self.rchq(pts_nys, pts_rec, w_IS, self.batch_size, self.kernel)
        return self._estimate(X, w, self.mean_predict(X), self.kernel)

    def uniform_trans(self, model_IS, uni_sampler):
        """-> ``(EZy, VarZy)`` when the prior is transformed into a uniform distribution (``_quadrature.py:86-107``).

        ``model_IS``: the GP model fitted to the importance-weighted observations -- a gpytorch ``ExactGP`` (read by
        attribute access, ``kernels.from_gpytorch_model``) or a ``kernels.PosteriorKernel`` carrying the GP mean.  The
        recombination runs with the model's PRIOR kernel (``model_IS.covar_module.forward``, ``:101``), the mean is the
        GP's predictive mean at the selected points (``predict(X, model_IS)[0]``, ``:102``)."""
        from .kernels import from_gpytorch_model

        post = model_IS if hasattr(model_IS, "gp_mean") else from_gpytorch_model(model_IS, kind="predictive")
        prior_kernel = post.base
        pts_rec = uni_sampler(self.n_quad)
        pts_nys = pts_rec[:self.n_nys]
        w_IS = torch.ones(self.n_quad) / self.n_quad
        X, w = self.rchq(pts_nys, pts_rec, w_IS, self.batch_size, prior_kernel)
        return self._estimate(X, w, post.predict_mean(X), prior_kernel)
