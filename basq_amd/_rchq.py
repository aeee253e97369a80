"""Drop-in for ``BASQ/_rchq.py``: ``recombination(pts_rec, pts_nys, num_pts, kernel, device, init_weights)``.

Same name, argument order, return convention and RNG consumption as the reference
(``BASQ/_rchq.py:4-25``):

* returns ``(idx, w)``: ``idx`` int64, ascending, ``len <= num_pts``; ``w`` float64 aligned with ``idx``;
* consumes exactly one ``torch.randn(len(pts_nys), num_pts - 1)`` from the **CPU global generator**
  (inside the reference's ``torch.svd_lowrank``, ``_rchq.py:29``), so ``torch.manual_seed(k)`` before the
  call selects the same Nystrom basis as it does for the reference on CPU;
* ``init_weights`` is accepted and ignored, as in the reference (``_rchq.py:53`` overwrites it with 1/N).

Differences that are part of the design:

* ``kernel``: the objects of :mod:`basq_amd.kernels` (``StationaryKernel`` / ``PosteriorKernel`` /
  ``WsabiKernel``, or ``from_gpytorch_model(model, ...)``) take the fused GPU path, which needs the kernel's
  structure; any OTHER callable ``(X[a,d], Y[b,d]) -> Tensor[a,b]`` -- the reference's contract, tutorial 02 --
  is evaluated on the device chunk by chunk and block-summed by ``basq_dense_blocksum_f64``
  (``kernels.CallableKernel``: correct for every kernel, HBM-bound instead of fused);
* arithmetic is float64 whatever the default dtype (SURVEY finding 3: the reference's selection is only
  reproducible in float64);
* ``device`` must be a HIP device (``torch.device('cuda', i)``); there is no CPU path.
"""
from __future__ import annotations

import torch

from ._engine import EngineTrace, LocalComm, RecombinationEngine, TorchDistComm
from ._ops import HipOps


def _as_kernel_object(kernel):
    """Structured kernels pass through (fused path); any other callable is the reference's opaque ``kernel``
    argument (``BASQ/_rchq.py:8,16``) and runs through the chunked dense path (``kernels.CallableKernel``)."""
    if all(hasattr(kernel, a) for a in ("base", "posterior", "warp", "dense")):
        return kernel
    if callable(kernel):
        from .kernels import CallableKernel

        return CallableKernel(kernel)
    raise TypeError("kernel must be a basq_amd.kernels object or a callable (X[a,d], Y[b,d]) -> Tensor[a,b]; got %r"
                    % (type(kernel),))


def recombination(
    pts_rec,          # random samples for recombination          [N, d]
    pts_nys,          # samples for the Nystrom approximation       [m, d]
    num_pts,          # number of samples finally returned (batch size)
    kernel,           # basq_amd.kernels object (fused path) or any callable (X, Y) -> Tensor (chunked dense path)
    device,           # HIP device
    init_weights=0,   # ignored, as in the reference
    *,
    trace: EngineTrace | None = None,
):
    kernel = _as_kernel_object(kernel)
    ops = HipOps(device)
    eng = RecombinationEngine(ops, LocalComm())
    N = pts_rec.shape[0]
    return eng.run(pts_rec, 0, N, pts_nys, int(num_pts), kernel, trace)


def recombination_sharded(pts_local, gid0, n_total, pts_nys, num_pts, kernel, device, group=None,
                          trace: EngineTrace | None = None):
    """Multi-GPU entry: every rank passes its contiguous slice ``pts_rec[gid0 : gid0 + len(pts_local)]``.

    One process per GPU, ``torch.distributed`` initialised by the caller (backend ``nccl`` = RCCL).
    Slices must tile ``0..n_total`` in rank order; ``pts_nys`` identical on all ranks.  The result is
    identical on every rank and equal (indices) to the single-GPU result.
    """
    kernel = _as_kernel_object(kernel)
    ops = HipOps(device)
    eng = RecombinationEngine(ops, TorchDistComm(group))
    return eng.run(pts_local, int(gid0), int(n_total), pts_nys, int(num_pts), kernel, trace)
