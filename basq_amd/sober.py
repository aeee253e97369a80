"""SOBER-flavoured entry point (``SOBER/_rchq.py:6-31``) -- what the reference's tutorials call
through ``SOBER/BASQ/_basq.py:19-36``.

    recombination(pts_rec, pts_nys, num_pts, kernel, device, dtype, init_weights=None, calc_obj=None) -> (idx, w)

Same argument order as the reference.  ``kernel``: a ``basq_amd.kernels`` object, SOBER's own ``Kernel(model, mode)`` wrapper
(``SOBER/_kernel.py``: adapted to its structured equivalent, ``kernels.from_sober_kernel``) or any other callable (dense path).
Differences from :func:`basq_amd.recombination` are those of the two
reference files: importance weights are honoured (zero-weight candidates are dropped), the Nystrom Gram goes
through ``make_cov_psd``, the ragged remainder of every round is additionally added to sets ``0..N_rest-1``,
and an elimination that runs out of positive entries stops early.  ``dtype`` is accepted for signature
compatibility; arithmetic is float64 (the reference's default, ``SOBER/_settings.py:11``).  ``calc_obj`` (the
optional objective row, ``SOBER/_rchq.py:66-69``) is honoured exactly as far as the reference can execute it: for
pools of at most ``2 * num_pts`` points (its single-reduction branch, ``:77-111``).  For larger pools the reference
raises at ``:140-142`` (shape error in its own objective sums) and this entry raises ``RuntimeError`` as well.
"""
from __future__ import annotations

import torch

from ._engine import EngineTrace, LocalComm, RecombinationEngine
from ._ops import HipOps
from ._rchq import _as_kernel_object


def _adapt_sober_kernel(kernel):
    """SOBER's own ``Kernel(model, mode)`` wrapper (``SOBER/_kernel.py:4-45``: what the tutorials pass) becomes its structured
    equivalent (``kernels.from_sober_kernel``) and takes the fused path; a wrapper over a GP the fused kernels do not cover
    (ARD lengthscales, another base kernel) stays the opaque callable it is -- the dense path, with a warning saying so."""
    from . import kernels as BK

    if not BK.looks_like_sober_kernel(kernel):
        return kernel
    try:
        return BK.from_sober_kernel(kernel)
    except (ValueError, AttributeError, TypeError) as e:
        import warnings

        warnings.warn(f"basq_amd.sober.recombination: SOBER Kernel(model, mode={getattr(kernel, 'mode', None)!r}) has no fused "
                      f"equivalent ({e}); it is evaluated densely on the device (8 bytes per pair)", RuntimeWarning, stacklevel=3)
        return kernel


def recombination(pts_rec, pts_nys, num_pts, kernel, device, dtype=torch.float64, init_weights=None, calc_obj=None, *,
                  trace: EngineTrace | None = None):
    kernel = _as_kernel_object(_adapt_sober_kernel(kernel))
    eng = RecombinationEngine(HipOps(device), LocalComm())
    objective = None if calc_obj is None else -1 * calc_obj(pts_rec)            # :67-69
    return eng.run(pts_rec, 0, pts_rec.shape[0], pts_nys, int(num_pts), kernel, trace, variant="sober",
                   init_weights=init_weights, objective=objective)
