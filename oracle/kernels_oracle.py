"""CPU oracle: the kernel callables the reference hands to ``recombination``.

TEST INFRASTRUCTURE ONLY.  Nothing under ``basq_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the timed CPU baseline.

What is restated here
---------------------
* The reference builds its recombination kernel from gpytorch objects chosen at
  ``BASQ/_parameters.py:192-208`` (``ScaleKernel(RBFKernel())``,
  ``ScaleKernel(MaternKernel(nu=2.5|1.5))``) and calls them as
  ``model.covar_module.forward(x, y)`` (``BASQ/_gp.py:270-272``,
  ``BASQ/_quadrature.py:101,104``).  gpytorch is a third-party dependency that is
  NOT vendored under ``/root/reference`` and is not installed in this image; the
  reference does not pin a version (``README.md:26-29`` lists names only) and has
  no tests or golden vectors at that boundary.  The functions below restate
  gpytorch's *published* algorithm (``gpytorch/kernels/kernel.py`` ``sq_dist`` /
  ``dist``, ``rbf_kernel.py`` ``postprocess_rbf``, ``matern_kernel.py``,
  ``scale_kernel.py``, gpytorch 1.x series): mean-centred augmented-GEMM squared
  distance, clamp at zero, ``exp(-d2/2)`` / Matern polynomial times exponential,
  multiplied by the output scale.  => **parity unpinned at the gpytorch
  boundary** (kernel *values*); everything downstream of the callable
  (``BASQ/_rchq.py``) is pinned against the imported reference itself, see
  ``oracle/make_golden.py``.
* ``predictive_covariance`` (``BASQ/_gp.py:259-277``) including its quirk of
  adding the likelihood noise to the leading diagonal of *rectangular* blocks.
* ``WsabiGP.wsabil_kernel`` / ``wsabim_kernel`` (``BASQ/_wsabi.py:205-249``) with
  the GP posterior mean of ``predict`` (``BASQ/_gp.py:213-230``) written as
  ``const + k(x, Xobs) @ mean_cache`` (gpytorch exact prediction strategy).

The reference never wraps kernel calls in ``torch.no_grad`` so gpytorch's
"x1 is x2 -> zero the diagonal exactly" shortcut is *not* taken there (its
inputs carry ``requires_grad`` through the lengthscale); we restate that branch
faithfully: no diagonal fill.
"""
from __future__ import annotations

import math

import torch


def _sq_dist(x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """gpytorch ``sq_dist`` (requires-grad branch): centred [-2x, |x|^2, 1]·[y, 1, |y|^2]^T, clamp>=0."""
    shift = x1.mean(-2, keepdim=True)
    a = x1 - shift
    b = x2 - shift
    a_n = a.pow(2).sum(dim=-1, keepdim=True)
    b_n = b.pow(2).sum(dim=-1, keepdim=True)
    lhs = torch.cat([-2.0 * a, a_n, torch.ones_like(a_n)], dim=-1)
    rhs = torch.cat([b, torch.ones_like(b_n), b_n], dim=-1)
    res = lhs.matmul(rhs.transpose(-2, -1))
    return res.clamp_min_(0)


class StationaryOracle:
    """``ScaleKernel(base)`` with a single shared lengthscale (no ARD), as ``_parameters.py:200-205`` builds it."""

    def __init__(self, family: str, lengthscale: float, outputscale: float = 1.0):
        if family not in ("rbf", "matern52", "matern32"):
            raise ValueError(family)
        self.family = family
        self.lengthscale = float(lengthscale)
        self.outputscale = float(outputscale)

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if self.family == "rbf":
            d2 = _sq_dist(x.div(self.lengthscale), y.div(self.lengthscale))
            base = d2.div_(-2).exp_()
        else:
            mean = x.mean(dim=-2, keepdim=True)
            xs = (x - mean).div(self.lengthscale)
            ys = (y - mean).div(self.lengthscale)
            r = _sq_dist(xs, ys).clamp_min_(1e-30).sqrt_()
            if self.family == "matern52":
                e = torch.exp(-math.sqrt(5.0) * r)
                c = (math.sqrt(5.0) * r).add(1).add(5.0 / 3.0 * r ** 2)
            else:
                e = torch.exp(-math.sqrt(3.0) * r)
                c = (math.sqrt(3.0) * r).add(1)
            base = c * e
        return base.mul(self.outputscale)


class PosteriorOracle:
    """``predictive_covariance(x, y, model)`` -- ``BASQ/_gp.py:259-277`` (bound at ``_vbq.py:119-128``).

    ``woodbury_inv`` is the matrix the reference forms as ``S @ S.T`` from gpytorch's
    ``covar_cache`` (``_gp.py:246-255``); here it is an explicit input.
    """

    def __init__(self, base: StationaryOracle, Xobs: torch.Tensor, woodbury_inv: torch.Tensor, noise: float):
        self.base = base
        self.Xobs = Xobs
        self.W = woodbury_inv
        self.noise = float(noise)

    def __call__(self, x, y):
        Kxy = self.base(x, y)
        KxX = self.base(x, self.Xobs)
        KXy = self.base(self.Xobs, y)
        cov = Kxy - KxX @ self.W @ KXy
        k = min(len(x), len(y))
        r = torch.arange(k)
        cov[r, r] = cov[r, r] + self.noise
        return cov


class WsabiOracle:
    """``wsabil_kernel`` / ``wsabim_kernel`` -- ``BASQ/_wsabi.py:205-249`` (jitter = 0, ``:56``)."""

    def __init__(self, post: PosteriorOracle, mean_const: float, mean_cache: torch.Tensor, label: str = "wsabil",
                 jitter: float = 0.0):
        if label not in ("wsabil", "wsabim"):
            raise ValueError(label)
        self.post = post
        self.mean_const = float(mean_const)
        self.mean_cache = mean_cache
        self.label = label
        self.jitter = float(jitter)

    def mean(self, x):
        """Posterior mean of the warped GP: ``predict(x, model)[0]`` (``_gp.py:213-230``)."""
        return self.mean_const + self.post.base(x, self.post.Xobs) @ self.mean_cache

    def __call__(self, x, y):
        mu_x = self.mean(x)
        mu_y = self.mean(y)
        cov = self.post(x, y)
        if mu_x.dim() == 1 and mu_y.dim() == 1:
            out = mu_x.unsqueeze(1) * cov * mu_y.unsqueeze(0)
        else:                                                   # batched y [nb, S, d] (SOBER/_rchq.py:124): _wsabi.py:219-222
            out = mu_x.unsqueeze(1) * cov * mu_y.unsqueeze(1)
        if self.label == "wsabim":
            out = out + 0.5 * (cov ** 2)
        k = min(len(x), len(y))
        r = torch.arange(k)
        out[r, r] = out[r, r] + self.jitter
        return out


def direct_rbf(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """``exp(-|x-y|^2/2)`` by explicit differences: the kernel of SURVEY §8c's known-answer test."""
    return torch.exp(-0.5 * ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1))


def synthetic_gp_state(Xobs: torch.Tensor, base: StationaryOracle, noise: float, seed: int):
    """A stand-in for a fitted GP: targets from a fixed smooth function, exact caches.

    Returns ``(woodbury_inv, mean_const, mean_cache, y)`` with
    ``woodbury_inv = (K + noise I)^-1`` and ``mean_cache = woodbury_inv @ (y - mean_const)``.
    Used to drive the posterior / WSABI kernels without gpytorch.
    """
    n = Xobs.shape[0]
    K = base(Xobs, Xobs) + noise * torch.eye(n, dtype=Xobs.dtype)
    # Deterministic positive targets (a warped-likelihood look-alike), no RNG or libm dependence.
    proj = torch.arange(1, Xobs.shape[1] + 1, dtype=Xobs.dtype) / Xobs.shape[1]
    t = Xobs @ proj
    y = 1.0 + 0.5 * t * t / (1.0 + t * t) + 0.01 * (seed % 7)
    mean_const = float(y.mean())
    L = torch.linalg.cholesky(K)
    W = torch.cholesky_inverse(L)
    mean_cache = W @ (y - mean_const)
    return W, mean_const, mean_cache, y


def predict_oracle(test_x: torch.Tensor, model):
    """``predict(test_x, model)`` -- ``BASQ/_gp.py:213-230``: ``(pred.mean, pred.variance)`` of
    ``model.likelihood(model(test_x))`` for an exact GP with a constant mean and a ``ScaleKernel(RBF)``:

        mean = c + k(x, X) mean_cache,   var = s2 - diag(k(x, X) W k(X, x)) + noise,   W = S S^T (covar_cache)

    ``model`` is duck-typed like the stub of ``oracle/make_golden_sampler.py`` (same attributes the reference reads
    at ``_gp.py:233-256`` / ``_gaussian_calc.py:44-51`` plus ``mean_module.constant`` and ``likelihood.noise``).
    Deviation, stated: the reference evaluates the variance under ``gpytorch.settings.fast_pred_var()`` (LOVE, a
    low-rank approximation inside gpytorch, which is not installed here); this is the exact variance.  Parity
    unpinned at the gpytorch boundary, as for the kernels above.
    """
    Xobs = model.train_inputs[0]
    ls = float(model.covar_module.base_kernel.lengthscale.reshape(-1)[0])
    s2 = float(model.covar_module.outputscale)
    base = StationaryOracle("rbf", ls, s2)
    S = model.prediction_strategy.covar_cache
    W = S @ S.T
    KxX = base(test_x, Xobs)
    mean = float(model.mean_module.constant) + KxX @ model.prediction_strategy.mean_cache
    var = s2 - ((KxX @ W) * KxX).sum(1) + float(model.likelihood.noise)
    return mean, var
