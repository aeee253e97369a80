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
  is evaluated on the device and block-summed by ``basq_dense_blocksum_f64`` (``kernels.CallableKernel``: correct for
  every kernel, HBM-bound instead of fused).  A bare callable is asked for exactly the blocks the reference asks for
  (``_rchq.py:81-99``) unless a probe shows that its values do not depend on the block (then: large chunks);
* arithmetic is float64 whatever the default dtype (SURVEY finding 3: the reference's selection is only
  reproducible in float64);
* ``device`` must be a HIP device (``torch.device('cuda', i)``); there is no CPU path.
"""
from __future__ import annotations

import torch

from ._engine import EngineTrace, Job, LocalComm, RecombinationEngine, TorchDistComm
from ._ops import HipOps


def _as_kernel_object(kernel):
    """Structured kernels pass through (fused path); any other callable is the reference's opaque ``kernel``
    argument (``BASQ/_rchq.py:8,16``) and runs through the dense path (``kernels.CallableKernel``, mode decided by its
    probe: the reference's own block-by-block calls unless the callable provably does not depend on the block)."""
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
    kernel,           # basq_amd.kernels object (fused path) or any callable (X, Y) -> Tensor (dense path)
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
                          trace: EngineTrace | None = None, comm=None):
    """Multi-GPU entry: every rank passes its contiguous slice ``pts_rec[gid0 : gid0 + len(pts_local)]``.

    One process per GPU, ``torch.distributed`` initialised by the caller (backend ``nccl`` = RCCL).
    Slices must tile ``0..n_total`` in rank order; ``pts_nys`` identical on all ranks.  The result is
    identical on every rank and equal (indices) to the single-GPU result.

    ``comm``: a communicator object to use instead of ``TorchDistComm(group)`` -- anything with its interface (``rank``,
    ``world``, ``all_gather``, ``broadcast``, ``for_slot``); the two-processes-on-one-GPU test passes a host-staged one.
    """
    kernel = _as_kernel_object(kernel)
    ops = HipOps(device)
    eng = RecombinationEngine(ops, comm if comm is not None else TorchDistComm(group))
    return eng.run(pts_local, int(gid0), int(n_total), pts_nys, int(num_pts), kernel, trace)


# ---- several independent recombinations in flight ------------------------------------------------------------------------
class SlotPool:
    """The per-batch-in-flight resources of ``recombination_many``: one ``HipOps`` -- a HIP stream plus that stream's pinned
    staging buffers, cluster-kernel workspace and transposed-basis cache -- per (device, slot).  Slots are created on first
    use and reused by later calls (creating a stream and pinning host memory costs milliseconds); ``release()`` drops them.
    A pool serves one caller at a time: ``lease`` holds its lock for the duration of a ``recombination_many`` call, so two
    host threads never drive the same slot's stream and staging buffers concurrently."""

    def __init__(self):
        import threading

        self._slots = {}                 # (device index, slot) -> HipOps
        self._lock = threading.Lock()

    def lease(self, device, n):
        """Context manager -> the first ``n`` slots of ``device`` (exclusive until the ``with`` block ends)."""
        import contextlib

        @contextlib.contextmanager
        def _lease():
            with self._lock:
                dev = torch.device(device)
                idx = dev.index if dev.index is not None else torch.cuda.current_device()
                out = []
                for k in range(n):
                    ops = self._slots.get((idx, k))
                    if ops is None:
                        ops = self._slots[(idx, k)] = HipOps(torch.device("cuda", idx), stream=torch.cuda.Stream(device=idx))
                    out.append(ops)
                yield out

        return _lease()

    def release(self):
        """Synchronise and drop every slot (streams, pinned buffers, workspaces); the next call re-creates what it needs."""
        with self._lock:
            for ops in self._slots.values():
                ops.synchronize()
                ops.release_caches()
            self._slots.clear()


_DEFAULT_POOL = SlotPool()


def release_slots(comms: bool = False):
    """Free the streams and cached buffers ``recombination_many`` / ``recombination_many_sharded`` keep between calls.  LOCAL:
    a rank may call it on its own (memory pressure on one rank).  ``comms=True`` additionally destroys the per-slot process
    groups of ``recombination_many_sharded`` -- a COLLECTIVE: every rank of the group must make the same call, or the next
    sharded call rebuilds sub-groups on some ranks only and hangs in ``new_group`` (``_engine.release_slot_comms``)."""
    _DEFAULT_POOL.release()
    if comms:
        from ._engine import release_slot_comms

        release_slot_comms()


def _run_many(jobs, device, comm, in_flight, timings=None, pool: SlotPool | None = None):
    jobs = list(jobs)
    if not jobs:
        return []
    with (pool or _DEFAULT_POOL).lease(device, max(1, min(int(in_flight), len(jobs)))) as slots:
        cur = torch.cuda.current_stream(slots[0].device)
        for ops in slots:
            ops.stream.wait_stream(cur)              # the inputs were produced on the caller's stream
        # (the inputs are consumed on the slots' streams without record_stream: run_many synchronises every slot before
        # it returns, so the caller cannot free or overwrite them while a slot still reads them)
        res = RecombinationEngine(slots[0], comm).run_many(jobs, slots)
        for ops in slots:
            cur.wait_stream(ops.stream)
        for idx, w in res:                           # allocated on a slot's stream, handed to the caller's
            idx.record_stream(cur)
            w.record_stream(cur)
    if timings is not None:                          # host clock at each batch's first launch and at its result
        timings.extend((j.times.get("start"), j.times.get("done")) for j in jobs)
    return res


def recombination_many(calls, device, in_flight: int = 2, seeds=None, traces=None, timings=None, pool: SlotPool | None = None):
    """Several INDEPENDENT recombinations with ``in_flight`` of them on the GPU at a time -> ``[(idx, w), ...]``.

    ``calls``: ``(pts_rec, pts_nys, num_pts, kernel)`` per recombination -- e.g. the two calls every BASQ iteration makes,
    the batch selection (``BASQ/_basq.py:82-88``) and the quadrature (``:104-106`` -> ``_quadrature.py:59-60``).  Each batch
    runs on its own HIP stream with its own workspaces; while one batch is in its chain of single-work-group reductions
    (null space + elimination, one CU busy), the block sums and GEMMs of the other fill the chip.

    Results are bit-identical to sequential ``recombination`` calls in the same order: the CPU global generator is
    consumed in call order (one ``torch.randn(m, num_pts - 1)`` each, as in the reference), and ``seeds[k]`` (optional)
    is what ``torch.manual_seed(seeds[k])`` right before call k would be.  ``timings`` (a list) receives one
    ``(start, done)`` pair of host clock readings per call: the per-batch latency with company on the GPU.
    ``pool``: the ``SlotPool`` whose streams and staging buffers the batches use (default: a module-level pool, freed by
    ``basq_amd.release_slots()``).
    """
    jobs = []
    for k, (pts_rec, pts_nys, num_pts, kernel) in enumerate(calls):
        jobs.append(Job(pts_rec, 0, pts_rec.shape[0], pts_nys, int(num_pts), _as_kernel_object(kernel),
                        trace=None if traces is None else traces[k], seed=None if seeds is None else seeds[k]))
    return _run_many(jobs, device, LocalComm(), in_flight, timings, pool)


def recombination_many_sharded(calls, device, group=None, in_flight: int = 4, seeds=None, timings=None, comm=None):
    """``recombination_many`` with every pool sharded over the ranks of ``group``: ``calls`` holds
    ``(pts_local, gid0, n_total, pts_nys, num_pts, kernel)`` per recombination (see ``recombination_sharded``).

    On G GPUs the wide kernels of a batch shrink by G while its chain of single-work-group reductions does not
    (replicated on every rank): keeping several batches in flight is what keeps the GPUs busy.  All ranks must pass the
    same number of calls; the scheduler resumes batches in a fixed order, so the ranks issue their collectives in the
    same sequence.
    """
    comm = comm if comm is not None else TorchDistComm(group)     # (``comm``: see ``recombination_sharded``)
    jobs = []
    for k, (pts_local, gid0, n_total, pts_nys, num_pts, kernel) in enumerate(calls):
        jobs.append(Job(pts_local, int(gid0), int(n_total), pts_nys, int(num_pts), _as_kernel_object(kernel),
                        seed=None if seeds is None else seeds[k]))
    return _run_many(jobs, device, comm, in_flight, timings)
