"""basq_amd -- MI355X-native kernel recombination for BASQ (hot path of ma921/BASQ ``BASQ/_rchq.py``).

Public surface (mirrors the reference's names):

    recombination(pts_rec, pts_nys, num_pts, kernel, device, init_weights=0) -> (idx, w)
    recombination_many([(pts_rec, pts_nys, num_pts, kernel), ...], device, in_flight=2) -> [(idx, w), ...]
    BASQ(batch_size, device).run_rchq(pts_nys, pts_rec, w_IS, kernel) -> (x, w)
    KernelQuadrature(...).rchq / .quadrature
    kernels.StationaryKernel / PosteriorKernel / WsabiKernel / from_gpytorch_model
    GaussianCalc(prior, device).unimodal_approximation / uniform_transformation
    SquareRootAcquisitionFunction / PriorSampler / UncertaintySampler(prior, model, n_rec, nys_ratio, device, ...)
    sober.recombination(pts_rec, pts_nys, num_pts, kernel, device, dtype, init_weights)   (SOBER/_rchq.py flavour)

Importing the package does not touch the GPU; the HIP library is loaded on first use and its
absence is an error (there is no CPU fallback).
"""
import os as _os


def configure_hw_queues(n: int = 16) -> bool:
    """Opt in to ``n`` hardware queues for batches in flight (``recombination_many``): the ROCm runtime maps HIP streams onto
    ``GPU_MAX_HW_QUEUES`` hardware queues -- FOUR by default, so with more streams than that two batches' launches queue behind one
    another (sixteen queues: N = 2e4 with four batches in flight 413 -> 660 batches/s, the headline size 72 -> 85,
    profiles/r07_x_hw_queues_batches_in_flight.txt; one batch at a time is unaffected).  The runtime reads the variable ONCE, at
    the process's first HIP call, and it applies to every HIP user of the process (torch, RCCL): that is why importing this
    package no longer sets it -- call this (or export the variable) BEFORE anything touches the GPU.  A value already in the
    environment is kept.  -> False (with a warning) when HIP is initialised already and the call can have no effect."""
    import warnings

    import torch

    if torch.cuda.is_initialized():
        if _os.environ.get("GPU_MAX_HW_QUEUES") is None:
            warnings.warn("basq_amd.configure_hw_queues(): the HIP runtime is initialised already -- GPU_MAX_HW_QUEUES has no effect "
                          "now; export it (or call this) before the first GPU call", RuntimeWarning, stacklevel=2)
        return False
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(n)))
    return True


from . import kernels, pools, sober                           # noqa: F401,E402
from ._basq import BASQ, KernelQuadrature                      # noqa: F401,E402
from ._engine import EngineTrace                               # noqa: F401,E402
from ._acquisition_function import SquareRootAcquisitionFunction   # noqa: F401
from ._gaussian_calc import GaussianCalc                       # noqa: F401,E402
from ._sampler import PriorSampler, UncertaintySampler         # noqa: F401
from ._rchq import (SlotPool, recombination, recombination_many, recombination_many_sharded,   # noqa: F401,E402
                    recombination_sharded, release_slots)

__all__ = ["recombination", "recombination_sharded", "recombination_many", "recombination_many_sharded", "BASQ", "KernelQuadrature", "GaussianCalc", "SquareRootAcquisitionFunction", "PriorSampler", "UncertaintySampler", "EngineTrace", "kernels",
           "pools", "sober", "SlotPool", "release_slots", "configure_hw_queues"]
