"""SOBER-flavoured entry point (``SOBER/_rchq.py:6-31``) -- what the reference's tutorials call
through ``SOBER/BASQ/_basq.py:19-36``.

    recombination(pts_rec, pts_nys, num_pts, kernel, device, dtype, init_weights=None, calc_obj=None) -> (idx, w)

Same argument order as the reference.  Differences from :func:`basq_amd.recombination` are those of the two
reference files: importance weights are honoured (zero-weight candidates are dropped), the Nystrom Gram goes
through ``make_cov_psd``, the ragged remainder of every round is additionally added to sets ``0..N_rest-1``,
and an elimination that runs out of positive entries stops early.  ``dtype`` is accepted for signature
compatibility; arithmetic is float64 (the reference's default, ``SOBER/_settings.py:11``).  ``calc_obj`` (the
optional objective row, ``SOBER/_rchq.py:66-68``) is not built: passing one raises.
"""
from __future__ import annotations

import torch

from ._engine import EngineTrace, LocalComm, RecombinationEngine
from ._ops import HipOps
from ._rchq import _require_structured


def recombination(pts_rec, pts_nys, num_pts, kernel, device, dtype=torch.float64, init_weights=None, calc_obj=None, *,
                  trace: EngineTrace | None = None):
    if calc_obj is not None:
        raise NotImplementedError("calc_obj (objective-aware recombination, SOBER/_rchq.py:66-68) is not built")
    _require_structured(kernel)
    eng = RecombinationEngine(HipOps(device), LocalComm())
    return eng.run(pts_rec, 0, pts_rec.shape[0], pts_nys, int(num_pts), kernel, trace, variant="sober",
                   init_weights=init_weights)
