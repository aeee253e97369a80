"""Kernel objects accepted by :func:`basq_amd.recombination` in place of the reference's callables.

The reference passes ``kernel`` as an opaque Python callable ``(X[a,d], Y[b,d]) -> Tensor[a,b]``
(``BASQ/_rchq.py:4-25``): a bound method of a GP wrapper (``VanillaGP.predictive_kernel``
``BASQ/_vbq.py:119-128``, ``WsabiGP.wsabil_kernel`` / ``wsabim_kernel`` ``BASQ/_wsabi.py:205-249``) or
``ScaleKernel.forward`` (``BASQ/_quadrature.py:101``).  A GPU kernel cannot call back into Python per
block, so the native path needs the kernel's *structure*.  The classes here are still callables with
the reference's dense semantics (``k(X, Y)`` returns the ``[a, b]`` matrix, computed by the HIP Gram
kernel) -- so code such as ``VarZy = w @ kernel(X, X) @ w`` (``BASQ/_quadrature.py:62``) keeps working --
and additionally expose what the fused recombination kernels need:

=====================  =====================================================================
``StationaryKernel``   ``ScaleKernel(RBFKernel | MaternKernel(nu))`` (``_parameters.py:192-208``)
``PosteriorKernel``    ``predictive_covariance`` (``BASQ/_gp.py:259-277``), incl. the noise it adds
                       to the leading diagonal of every block
``WsabiKernel``        ``wsabil_kernel`` / ``wsabim_kernel`` (``BASQ/_wsabi.py:205-249``)
=====================  =====================================================================

``from_gpytorch_model`` builds them from a fitted gpytorch model by attribute access only (gpytorch
itself is not imported).

``CallableKernel`` keeps the reference's own contract for everything else: ANY callable ``(X, Y) -> Tensor[a, b]``
(tutorial 02, "BayesQuad with arbitrary kernel"; BASELINE config 4 names this path).  The callable is evaluated on
the device in chunks of candidates and the block sums are taken by ``basq_dense_blocksum_f64`` -- the unfused,
HBM-bound form (8 bytes per pair) -- so the structured classes remain the fast path; ``recombination`` wraps a
bare callable automatically.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from ._lib import FAMILY, MAX_DIM, ROLE_A, ROLE_B


@dataclass(frozen=True)
class KernelSpec:
    family: str
    d: int
    lengthscale: float
    outputscale: float
    accurate_exp: bool = False      # fused block sums with the 1e-17 exponential (GP posteriors: include/basq_hip.h)


def _ops_for(x):
    from ._ops import HipOps

    return HipOps(x.device)


def _col_mean(ops, x):
    """Column means of ``x`` (gpytorch centres on the first operand): ``basq_col_mean_f64``, or a tensor reduction for
    points wider than the library's packed rows."""
    return ops.col_mean(x) if x.shape[1] <= MAX_DIM else x.mean(0)


def _mm(ops, A, B):
    """``A @ B`` on the library's own f64 MFMA GEMM (``basq_gemm_f64``) where the ops object has one."""
    g = getattr(ops, "gemm", None)
    if g is None or A.dtype != torch.float64 or B.dtype != torch.float64 or A.dim() != 2 or B.dim() != 2:
        return A @ B
    return g(A.contiguous(), B.contiguous())


class StationaryKernel:
    """``outputscale * base(|x - y| / lengthscale)``, base in {rbf, matern52, matern32}, shared lengthscale."""

    def __init__(self, family: str, lengthscale: float, outputscale: float = 1.0):
        if family not in FAMILY:
            raise ValueError(f"unknown kernel family {family!r}; expected one of {sorted(FAMILY)}")
        if not lengthscale > 0:
            raise ValueError("lengthscale must be positive")
        self.family = family
        self.lengthscale = float(lengthscale)
        self.outputscale = float(outputscale)

    def spec(self, d: int) -> KernelSpec:
        if not 1 <= d <= MAX_DIM:
            raise ValueError(f"dimension {d} outside 1..{MAX_DIM}")
        return KernelSpec(self.family, int(d), self.lengthscale, self.outputscale)

    def fits_packed_rows(self, d: int) -> bool:
        """The fused kernels hold a point as a packed row of at most 40 doubles (d <= 38); beyond that the kernel is
        evaluated by device tensor ops (``_dense_wide``) and the recombination takes the dense path."""
        return 1 <= d <= MAX_DIM

    def dense(self, ops, x, y, center=None, diag_offset=0):
        """``diag_offset``: ``x`` is rows ``diag_offset..`` of the first operand of a square block (the structured
        kernels add their diagonal terms on the TRUE diagonal then); no effect for a stationary kernel."""
        if center is None:
            center = _col_mean(ops, x)          # gpytorch centres on the first operand
        if not self.fits_packed_rows(x.shape[1]):
            return self._dense_wide(x, y, center)
        spec = self.spec(x.shape[1])
        pa = ops.pack(spec, x, center, ROLE_A)
        pb = ops.pack(spec, y, center, ROLE_B)
        return ops.gram(spec, pa, x.shape[0], pb, y.shape[0])

    def _dense_wide(self, x, y, center):
        """The same kernel values for d > 38, by tensor operations on the device (one GEMM for the scaled squared
        distances of the centred points, as gpytorch forms them, then the family's closed form): the envelope path --
        correct for any d, unfused (the [len(x), len(y)] matrix is materialised)."""
        xs = (x - center) * (1.0 / self.lengthscale)
        ys = (y - center) * (1.0 / self.lengthscale)
        D = xs @ ys.T - 0.5 * (xs * xs).sum(1, keepdim=True) - 0.5 * (ys * ys).sum(1).unsqueeze(0)   # -|x - y|^2 / (2 l^2)
        if self.family == "rbf":
            return self.outputscale * torch.exp(D.clamp_max(0.0))
        r2 = (-2.0 * D).clamp_min(1e-30)
        r = r2.sqrt()
        if self.family == "matern52":
            a = (5.0 ** 0.5) * r
            return self.outputscale * (((a + 1.0) + (5.0 / 3.0) * r2) * torch.exp(-a))
        a = (3.0 ** 0.5) * r
        return self.outputscale * ((a + 1.0) * torch.exp(-a))

    def __call__(self, x, y):
        x = x.to(torch.float64).contiguous()
        y = y.to(torch.float64).contiguous()
        return self.dense(_ops_for(x), x, y)

    # structure queried by the engine
    base = property(lambda self: self)
    posterior = None
    warp = "none"


class PosteriorKernel:
    """GP posterior covariance ``k(x,y) - k(x,X) W k(X,y)`` + ``noise`` on the leading diagonal."""

    warp = "none"

    def __init__(self, base: StationaryKernel, Xobs, woodbury_inv, noise: float, mean_const: float | None = None,
                 mean_cache=None):
        self.base = base
        self.Xobs = Xobs
        self.W = woodbury_inv
        self.noise = float(noise)
        self.mean_const = None if mean_const is None else float(mean_const)   # for predict_mean (quadrature)
        self.mean_cache = mean_cache

    @property
    def posterior(self):
        return self

    COND_WARN = 1e7

    def condition_number(self):
        """2-norm condition number of the Woodbury matrix ``W = (K(X, X) + noise I)^-1`` -- that of the observation Gram
        -- computed once per kernel object (an ``n_obs x n_obs`` symmetric eigenvalue problem on the host).  Above
        ~1e7 the posterior covariance ``k - k(.,X) W k(X,.)`` is a catastrophic cancellation: the reference's own selection
        then changes when its base kernel moves by one ulp, and no implementation can reproduce it."""
        c = self.__dict__.get("_cond")
        if c is None:
            Wh = self.W.detach().to("cpu", torch.float64)
            ev = torch.linalg.eigvalsh(0.5 * (Wh + Wh.T)).abs()
            lo = float(ev.min())
            c = self.__dict__["_cond"] = float(ev.max()) / lo if lo > 0 else float("inf")
        return c

    def dense(self, ops, x, y, center=None, diag_offset=0):
        Xo = ops.to_device(self.Xobs, torch.float64)
        W = ops.to_device(self.W, torch.float64)
        if center is None:
            center = _col_mean(ops, x)
        cov = self.base.dense(ops, x, y, center)
        KxX = self.base.dense(ops, x, Xo, center)
        KXy = self.base.dense(ops, Xo, y, center)
        cov = cov - _mm(ops, _mm(ops, KxX, W), KXy)
        k = min(x.shape[0], y.shape[0] - diag_offset)
        if k > 0:
            cov.diagonal(diag_offset)[:k] += self.noise       # _gp.py:275-276
        return cov

    def __call__(self, x, y):
        x = x.to(torch.float64).contiguous()
        y = y.to(torch.float64).contiguous()
        return self.dense(_ops_for(x), x, y)

    def gp_mean(self, ops, x, center=None):
        """GP posterior mean ``predict(x, model)[0]`` (``BASQ/_gp.py:213-230``) through the HIP kernel mat-vec."""
        if self.mean_cache is None or self.mean_const is None:
            raise ValueError("PosteriorKernel was built without the GP mean (mean_const / mean_cache)")
        Xo = ops.to_device(self.Xobs, torch.float64)
        v = ops.to_device(self.mean_cache, torch.float64)
        if center is None:
            center = _col_mean(ops, x)
        if not self.base.fits_packed_rows(x.shape[1]):
            return self.mean_const + self.base.dense(ops, x, Xo, center) @ v
        spec = self.base.spec(x.shape[1])
        pa = ops.pack(spec, x, center, ROLE_A, pad_rows_to=64)
        pb = ops.pack(spec, Xo, center, ROLE_B)
        return ops.matvec(spec, pa, x.shape[0], pb, Xo.shape[0], v, self.mean_const)

    def gp_variance(self, ops, x, center=None):
        """Predictive variance of ``predict`` (``model.likelihood(model(x)).variance``): diag of the posterior + noise."""
        Xo = ops.to_device(self.Xobs, torch.float64)
        W = ops.to_device(self.W, torch.float64)
        if center is None:
            center = _col_mean(ops, x)
        KxX = self.base.dense(ops, x, Xo, center)
        return self.base.outputscale - (_mm(ops, KxX, W) * KxX).sum(1) + self.noise

    def predict_mean(self, x):
        """``VanillaGP.predict_mean`` (``BASQ/_vbq.py:141-151``): the ``mean_predict`` of the quadrature."""
        x = x.to(torch.float64).contiguous()
        return self.gp_mean(_ops_for(x), x)


class WsabiKernel:
    """``mu(x) cov(x,y) mu(y)`` (+ ``0.5 cov^2`` for WSABI-M), ``mu`` = warped-GP posterior mean."""

    def __init__(self, post: PosteriorKernel, mean_const: float, mean_cache, label: str = "wsabil",
                 jitter: float = 0.0, alpha: float = 0.0):
        if label not in ("wsabil", "wsabim"):
            raise ValueError(label)
        self.posterior = post
        self.base = post.base
        self.mean_const = float(mean_const)
        self.mean_cache = mean_cache
        self.warp = label
        self.jitter = float(jitter)
        self.alpha = float(alpha)          # WSABI offset: l = alpha + 0.5 l~^2  (BASQ/_wsabi.py:102-121)

    def mean(self, ops, x, center=None):
        """``predict(x, model)[0]`` (``BASQ/_gp.py:213-230``) = const + k(x, Xobs) @ mean_cache, via the HIP mat-vec."""
        Xo = ops.to_device(self.posterior.Xobs, torch.float64)
        v = ops.to_device(self.mean_cache, torch.float64)
        if center is None:
            center = _col_mean(ops, x)
        if not self.base.fits_packed_rows(x.shape[1]):
            return self.mean_const + self.base.dense(ops, x, Xo, center) @ v
        spec = self.base.spec(x.shape[1])
        pa = ops.pack(spec, x, center, ROLE_A, pad_rows_to=64)
        pb = ops.pack(spec, Xo, center, ROLE_B)
        return ops.matvec(spec, pa, x.shape[0], pb, Xo.shape[0], v, self.mean_const)

    def dense(self, ops, x, y, center=None, diag_offset=0):
        if center is None:
            center = _col_mean(ops, x)
        cov = self.posterior.dense(ops, x, y, center, diag_offset)
        out = self.mean(ops, x, center).unsqueeze(1) * cov * self.mean(ops, y, center).unsqueeze(0)
        if self.warp == "wsabim":
            out = out + 0.5 * cov * cov
        k = min(x.shape[0], y.shape[0] - diag_offset)
        if k > 0:
            out.diagonal(diag_offset)[:k] += self.jitter
        return out

    def __call__(self, x, y):
        x = x.to(torch.float64).contiguous()
        y = y.to(torch.float64).contiguous()
        return self.dense(_ops_for(x), x, y)

    def predict_mean(self, x):
        """``wsabil_mean_predict`` / ``wsabim_mean_predict`` (``BASQ/_wsabi.py:278-300``)."""
        x = x.to(torch.float64).contiguous()
        ops = _ops_for(x)
        center = _col_mean(ops, x)
        mu_w = self.mean(ops, x, center)
        if self.warp == "wsabil":
            return self.alpha + 0.5 * mu_w ** 2
        return self.alpha + 0.5 * (mu_w ** 2 + self.posterior.gp_variance(ops, x, center))


class CallableKernel:
    """An opaque kernel callable, the reference's ``kernel`` argument as is (``BASQ/_rchq.py:8,16``).

    ``fn(X[a, d], Y[b, d]) -> Tensor[a, b]`` is called with tensors on the HIP device (``input_dtype``, float64 by
    default) under ``torch.no_grad()``; its result is cast to float64.  Two evaluation modes:

    * ``block_exact=True``: exactly the reference's calls -- one ``fn(pts_nys, block of 2n points)`` per block
      (``_rchq.py:81-86``) and one for the ragged tail (``:91-99``); N/(2n) Python calls per round.  Needed for callables
      whose value depends on the block they are asked for, like the reference's DEFAULT kernel ``predictive_covariance``,
      which adds the likelihood noise to entries ``[k][k]`` of EVERY block (``BASQ/_gp.py:275-276``);
    * ``block_exact=False``: ``fn(pts_nys, chunk)`` for chunks of up to ``chunk_bytes / (8 m)`` consecutive candidates --
      far fewer, larger calls, but only equivalent for callables that evaluate column by column.  The default chunk is 1 GB
      of kernel values (the callable's own temporaries are a small multiple of that; the part has 288 GB): every chunk
      costs one read-modify-write of the [m, 2n] block sums, 6 % of the chunk's bytes at that size, 25 % at 256 MB.

    ``block_exact=None`` (the default, and what a bare callable handed to ``recombination`` gets): decided once per batch
    by a probe (``resolve_mode``) -- the callable is asked for two blocks at once and one by one; unless the answers
    agree, the exact mode is taken.  Correct by default; the chunked mode is an optimisation the probe has to earn.
    """

    opaque = True
    base = None
    posterior = None
    warp = "none"
    PROBE_RTOL = 1e-13           # block dependence below this (relative to max |K|) cannot move a selection (SURVEY finding 3)

    def __init__(self, fn, block_exact: bool | None = None, chunk_bytes: int = 1 << 30, input_dtype=torch.float64):
        if not callable(fn):
            raise TypeError("CallableKernel needs a callable (X, Y) -> Tensor")
        self.fn = fn
        self.block_exact = None if block_exact is None else bool(block_exact)
        self.chunk_bytes = int(chunk_bytes)
        self.input_dtype = input_dtype

    def dense(self, ops, x, y, center=None, diag_offset=0):
        with torch.no_grad():
            K = self.fn(x.to(self.input_dtype), y.to(self.input_dtype))
        K = K.to_dense() if hasattr(K, "to_dense") else K        # gpytorch lazy tensors / linear operators
        if not torch.is_tensor(K) or tuple(K.shape) != (x.shape[0], y.shape[0]):
            raise ValueError("kernel callable must return a dense [len(X), len(Y)] tensor; got %r"
                             % (getattr(K, "shape", type(K)),))
        return ops.to_device(K.detach(), torch.float64)

    def resolve_mode(self, ops, pts_nys, S: int) -> bool:
        """-> True when the batch must make the reference's own block-by-block calls.

        Probe (``block_exact=None``): ``fn(pts_nys, [Y1; Y2])`` against ``[fn(pts_nys, Y1), fn(pts_nys, Y2)]`` for two
        blocks of S points.  The blocks are cut from the Nystrom points themselves (repeated cyclically), so every rank of
        a multi-GPU run probes the same data and reaches the same decision.  ``predictive_covariance`` fails the probe
        through its noise on entries ``[k][S + k]`` of the second block (unless the noise is exactly negligible); a
        callable that is a function of the pair alone passes."""
        if self.block_exact is not None:
            return self.block_exact
        m = pts_nys.shape[0]
        if m == 0 or S < 1:
            return True
        rows = torch.arange(2 * S, device=pts_nys.device) % m
        Y = pts_nys[rows]
        both = self.dense(ops, pts_nys, Y)
        one = torch.cat([self.dense(ops, pts_nys, Y[:S]), self.dense(ops, pts_nys, Y[S:])], 1)
        scale = float(both.abs().max().item())
        dev = float((both - one).abs().max().item())
        return not (dev <= self.PROBE_RTOL * scale)              # (NaNs fail the probe: exact mode)

    def __call__(self, x, y):
        return self.fn(x, y)


def _prediction_caches(model, warm_x):
    """``(mean_cache, covar_cache)`` of a gpytorch ExactGP, warming them up the way the reference does when they do
    not exist yet (``BASQ/_gp.py:247-253``, ``BASQ/_gaussian_calc.py:32-38``): a freshly trained model, or one put
    back into train mode, has ``prediction_strategy = None`` -> ``model.eval(); model(one point)`` builds it."""
    try:
        ps = model.prediction_strategy
        return ps.mean_cache, ps.covar_cache
    except AttributeError:
        model.eval()
        with torch.no_grad():
            model(warm_x)
        ps = model.prediction_strategy
        return ps.mean_cache, ps.covar_cache


def from_gpytorch_model(model, kind: str = "predictive", wsabi_label: str = "wsabil", wsabi_alpha: float = 0.0):
    """Build a kernel object from a fitted gpytorch ``ExactGP`` (duck-typed; gpytorch is not imported).

    ``kind``: ``"prior"`` -> ``model.covar_module.forward`` (``_quadrature.py:101``);
    ``"predictive"`` -> ``predictive_covariance`` (``_vbq.py:119-128``); ``"wsabi"`` -> WSABI-L/M.
    Reads the same attributes the reference reads at ``_gp.py:233-256`` and ``_gaussian_calc.py:44-51``.
    """
    cm = model.covar_module
    bk = cm.base_kernel
    name = type(bk).__name__
    if "RBF" in name:
        family = "rbf"
    elif "Matern" in name:
        nu = float(getattr(bk, "nu"))
        family = {2.5: "matern52", 1.5: "matern32"}.get(nu)
        if family is None:
            raise ValueError(f"Matern nu={nu} not supported")
    else:
        raise ValueError(f"unsupported base kernel {name}")
    ls = bk.lengthscale
    if getattr(ls, "numel", lambda: 1)() != 1:
        raise ValueError("ARD lengthscales are not supported (the reference uses a single shared lengthscale)")
    base = StationaryKernel(family, float(ls.detach().reshape(-1)[0]), float(cm.outputscale.detach().reshape(-1)[0]))
    if kind == "prior":
        return base
    Xobs = model.train_inputs[0].detach()
    noise = float(model.likelihood.noise.detach().reshape(-1)[0])
    mean_cache, S = _prediction_caches(model, Xobs[0].unsqueeze(0))      # _gp.py:247-253 (incl. the warm-up call)
    mean_cache, S = mean_cache.detach().reshape(-1), S.detach()
    const = float(model.mean_module.constant.detach().reshape(-1)[0])
    post = PosteriorKernel(base, Xobs, S @ S.T, noise, const, mean_cache)       # _gp.py:255
    if kind == "predictive":
        return post
    if kind == "wsabi":
        return WsabiKernel(post, const, mean_cache, wsabi_label, alpha=wsabi_alpha)
    raise ValueError(kind)


SOBER_MODES = ("predictive_covariance", "weighted_predictive_covariance", "kernel")


def from_sober_kernel(kernel_or_model, mode: str | None = None):
    """The structured equivalent of SOBER's ``Kernel(model, mode)`` wrapper (``SOBER/_kernel.py:4-45``) -- what the reference's
    tutorials hand to ``SOBER/_rchq.py:recombination`` -- so that it takes the fused path instead of the dense one.

    ``kernel_or_model``: a ``Kernel`` object (duck-typed: anything with ``.model`` and ``.mode``) or the gpytorch model itself
    (then ``mode`` is required).  The three modes of ``Kernel.__call__`` (``:16-29``):

    * ``"predictive_covariance"`` -> ``PosteriorKernel`` with NO noise diagonal: SOBER's ``predictive_covariance``
      (``SOBER/_gp.py:281-305``) has the ``+ lik_var`` lines of ``BASQ/_gp.py:275-276`` commented out;
    * ``"weighted_predictive_covariance"`` -> ``mu(x) cov(x, y) mu(y)`` (``:32-45``; ``mu = predict_mean`` =
      ``SOBER/_gp.py:240-253``): ``WsabiKernel(..., "wsabil")`` over the same noise-free covariance;
    * ``"kernel"`` -> ``model.covar_module.forward`` (``:27``): the ``StationaryKernel``.

    The model's attributes are read exactly as ``from_gpytorch_model`` reads them (``SOBER/_gp.py:255-278`` is
    ``BASQ/_gp.py:233-256`` verbatim, warm-up call included)."""
    model = getattr(kernel_or_model, "model", None)
    if model is not None and hasattr(kernel_or_model, "mode"):
        mode = kernel_or_model.mode if mode is None else mode
    else:
        model = kernel_or_model
    if mode not in SOBER_MODES:
        raise ValueError(f'mode should be from {list(SOBER_MODES)}')                  # (the reference's own message, :29)
    if mode == "kernel":
        return from_gpytorch_model(model, "prior")
    post = from_gpytorch_model(model, "predictive")
    post = PosteriorKernel(post.base, post.Xobs, post.W, 0.0, post.mean_const, post.mean_cache)
    if mode == "predictive_covariance":
        return post
    return WsabiKernel(post, post.mean_const, post.mean_cache, "wsabil")


def looks_like_sober_kernel(obj) -> bool:
    """Duck test for SOBER's ``Kernel`` wrapper: a callable carrying ``.model`` (a GP with a ``covar_module``) and ``.mode``."""
    return (callable(obj) and hasattr(obj, "mode") and hasattr(getattr(obj, "model", None), "covar_module")
            and not hasattr(obj, "base"))

