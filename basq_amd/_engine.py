"""Recombination engine: communicators, the trace record, and the scheduler that drives batches (one process per GPU).

A batch (``_batch.Batch``) is a step generator that never blocks: wherever the host must wait for the GPU it yields the
event.  ``RecombinationEngine.run`` drives one batch and blocks on each event -- the reference's synchronous call.
``RecombinationEngine.run_many`` keeps several batches in flight from ONE host thread, each on its own HIP stream with its
own workspaces: while batch A's single-work-group reductions (null space + elimination: ~30 % of a batch's GPU time on one
of 256 CUs) run, batch B's block sums and GEMMs fill the other CUs.  The reference calls the path twice per BASQ
iteration, independently (selection ``BASQ/_basq.py:82-88`` and quadrature ``:104-106``): that pair is the use case.

Multi-GPU: the candidate pool is sharded in contiguous id ranges; per round every rank block-sums and
projects its shard, the ``(q+1) x S`` messages are all-gathered and added in rank order, and every rank runs the
(deterministic) reduction on the same message; re-weighting/compaction are local.  The exchange is stream-ordered, so the
descriptor-driven rounds need no host wait on any rank count, and several batches can be in flight on every rank (the
scheduler then resumes them in a fixed order, so that all ranks issue their collectives in the same sequence).
"""
from __future__ import annotations

import contextlib
import time
from collections import deque
from dataclasses import dataclass, field

import torch

from . import _config as cfg
from ._basis import drive
from ._batch import Batch


# ----------------------------------------------------------------------------------------------------
# communicators
# ----------------------------------------------------------------------------------------------------
class LocalComm:
    rank, world = 0, 1

    def all_gather(self, t):
        return t.unsqueeze(0)

    def broadcast(self, t, src=0):
        return t

    def for_slot(self, i):
        return self


class TorchDistComm:
    """``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo`` in the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_gather(self, t):
        """-> ``[world, *t.shape]`` (rank order).  Messages are tiny ((q+1) x S doubles): latency-bound."""
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        # ONE flat output buffer (RCCL: a single ncclAllGather, no per-rank output list / copy-out kernels)
        self.dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=self.group)
        return out

    def broadcast(self, t, src=0):
        """``src``: rank WITHIN this communicator's group."""
        g_src = src if self.group is None else self.dist.get_global_rank(self.group, src)
        self.dist.broadcast(t, src=g_src, group=self.group)
        return t

    def for_slot(self, i):
        """A communicator over the same ranks on a process group OF ITS OWN (created on first use, cached; every rank must
        ask for the slots in the same order -- ``run_many`` does, up front).  Collectives of one process group execute in
        issue order on one internal stream: with owner-rank reductions a batch's broadcast waits for the owner's chain of
        single-work-group kernels, and on a shared group every other batch's all-gather would queue behind it.  One group
        (= one RCCL communicator) per batch in flight keeps the batches' exchanges independent.

        The cache is keyed on the parent process-group OBJECT (held by the entry, so its id cannot be reused while the entry
        lives; the default group resolves to the object of the CURRENT initialisation) and an entry whose sub-group no longer
        answers -- ``destroy_process_group()`` + re-init in one process: test suites, long-lived services -- is rebuilt."""
        parent = self.group if self.group is not None else self.dist.group.WORLD
        key = (id(parent), i)
        hit = _SLOT_COMMS.get(key)
        if hit is not None and hit[0] is parent:
            try:
                self.dist.get_rank(hit[1].group)                 # raises once the sub-group has been destroyed
                return hit[1]
            except Exception:
                pass
        for k in [k for k, (_, c) in _SLOT_COMMS.items() if not _group_alive(self.dist, c.group)]:
            del _SLOT_COMMS[k]                                   # communicators of a destroyed world
        ranks = list(range(self.dist.get_world_size())) if self.group is None else self.dist.get_process_group_ranks(self.group)
        comm = TorchDistComm(self.dist.new_group(ranks=ranks))
        _SLOT_COMMS[key] = (parent, comm)
        return comm


def _group_alive(dist, group) -> bool:
    try:
        dist.get_rank(group)
        return True
    except Exception:
        return False


_SLOT_COMMS = {}                 # (id of the parent group, slot) -> (parent group, TorchDistComm on its own process group)


def release_slot_comms():
    """Destroy the per-slot process groups ``TorchDistComm.for_slot`` created (collective: every rank calls it) and forget them."""
    import torch.distributed as dist

    for _, comm in list(_SLOT_COMMS.values()):
        try:
            if dist.is_initialized():
                dist.destroy_process_group(comm.group)
        except Exception:
            pass                                                 # (already gone with its parent)
    _SLOT_COMMS.clear()


# ----------------------------------------------------------------------------------------------------
# trace (tests / profiling)
# ----------------------------------------------------------------------------------------------------
@dataclass
class EngineTrace:
    """``host_sync=True`` (default) synchronises around every phase to attribute host timers and takes the round-by-round
    loop; ``host_sync=False`` leaves the batch on the path an untraced call takes (descriptor-driven rounds) and fills
    ``rounds`` / ``kernel_events`` after the fact."""
    rounds: list = field(default_factory=list)      # dicts: R, S, nb, n_tail, kept, tot, XcarT (optional)
    U: torch.Tensor | None = None
    timers: dict = field(default_factory=dict)
    keep_tensors: bool = False
    time_kernels: bool = False                      # record HIP events around every block-sum launch
    kernel_events: list = field(default_factory=list)   # (start_evt, end_evt, dict(pairs=, R=, m=, S=))
    chain_events: list = field(default_factory=list)    # time_kernels: (start_evt, end_evt) around each round's null space + elimination
    side_pairs: float = 0.0                         # kernel values evaluated on the side stream (message columns of an epoch's irregular candidates: not in kernel_events)
    sample_clock: object = None                     # time_kernels: a HipOps on a SECOND stream -> shader clock beside each class launch
    host_sync: bool = True                          # synchronise around phases to attribute host timers

    def add_time(self, key, dt):
        self.timers[key] = self.timers.get(key, 0.0) + dt


# ----------------------------------------------------------------------------------------------------
# engine
# ----------------------------------------------------------------------------------------------------
@dataclass
class Job:
    """One recombination for ``RecombinationEngine.run_many`` (the arguments of ``run``)."""
    pts_local: torch.Tensor
    gid0: int
    n_total: int
    pts_nys: torch.Tensor
    num_pts: int
    kernel: object
    trace: EngineTrace | None = None
    variant: str = "basq"
    init_weights: object = None
    objective: object = None
    seed: int | None = None                         # ``torch.manual_seed(seed)`` right before this batch's draw
    times: dict = field(default_factory=dict)       # filled by run_many: host clock at "start" and "done" (latency)


class RecombinationEngine:
    def __init__(self, ops, comm=None):
        self.ops = ops
        self.comm = comm or LocalComm()

    # ------------------------------------------------------------------------------------------------
    def run(self, pts_local, gid0: int, n_total: int, pts_nys, num_pts: int, kernel, trace: EngineTrace | None = None,
            variant: str = "basq", init_weights=None, objective=None):
        """One recombination batch (``_batch.Batch``), synchronously: ``(idx int64[<=num_pts] ascending, w float64)`` on the
        ops device, identical on every rank.  The launch stream is looked up once for the whole batch."""
        batch = Batch(self.ops, self.comm, pts_local, gid0, n_total, pts_nys, num_pts, kernel, trace, variant,
                      init_weights, objective)
        pin = getattr(self.ops, "pin_stream", None)
        if pin is None:
            return drive(batch.steps())
        pin()
        try:
            return drive(batch.steps())
        finally:
            self.ops.unpin_stream()

    # ------------------------------------------------------------------------------------------------
    def run_many(self, jobs, slot_ops, ordered: bool | None = None):
        """Several independent batches in flight, one per entry of ``slot_ops`` (``HipOps`` objects bound to their own
        streams) -> the list of ``(idx, w)`` in job order.

        Batches start in job order, and each one's first segment -- operands, round-1 block sums, the Gaussian draw
        from the CPU global generator, the range finder's launches -- runs to its first wait before the next batch is
        looked at: the generator is consumed in job order, exactly as by sequential calls (``Job.seed`` re-seeds it right
        before that batch's draw).  After that a batch is resumed whenever the event it waits for has fired; with
        ``ordered`` (default on several ranks) strictly in FIFO order, so that every rank enqueues its collectives in the
        same sequence.  Results equal the sequential runs' bit for bit: a batch's arithmetic does not depend on what else
        is in flight."""
        jobs = list(jobs)
        if ordered is None:
            ordered = self.comm.world > 1
        results = [None] * len(jobs)
        pending = deque(enumerate(jobs))
        active = deque()                                        # [job index, generator, (ops, comm), event it waits for]
        pipelined = len(slot_ops) > 1
        # Several ranks, several batches in flight: batch k's per-round reductions run on rank k mod G only (the outcome is
        # broadcast), so every GPU carries 1/G of the chains instead of all of them; each batch in flight talks on a process
        # group of its own (``TorchDistComm.for_slot``).
        owner_mode = cfg.OWNER_RANK_REDUCTION and self.comm.world > 1 and pipelined
        # (ops, communicator) PAIRS: the process group belongs to the slot a batch actually takes -- "one group per batch in
        # flight" holds whichever batch finishes first (with FIFO resumption the slots free up in one order on every rank)
        free = deque((ops, self.comm.for_slot(i) if owner_mode else self.comm) for i, ops in enumerate(slot_ops))
        # every job re-seeds the generator: the Gaussian draw of a batch (2.2 ms of host time at the headline size) then runs
        # on the batch's owner alone instead of on every rank; the generators are put back in step after the last job
        draw_on_owner = owner_mode and bool(jobs) and all(j.seed is not None for j in jobs)
        last_drew = True

        def advance(entry):
            """Resume a batch until its next wait (-> True) or its end (-> False, result stored, slot freed)."""
            k, gen, slot, _ = entry
            with _stream_of(slot[0]):
                try:
                    entry[3] = next(gen)
                    return True
                except StopIteration as stop:
                    results[k] = stop.value
                    jobs[k].times["done"] = time.perf_counter()
                    free.append(slot)
                    return False

        while pending or active:
            while pending and free:
                k, job = pending.popleft()
                slot = free.popleft()
                ops, comm = slot
                if job.seed is not None:
                    torch.manual_seed(job.seed)
                job.times["start"] = time.perf_counter()
                batch = Batch(ops, comm, job.pts_local, job.gid0, job.n_total, job.pts_nys, job.num_pts, job.kernel,
                              job.trace, job.variant, job.init_weights, job.objective, pipelined=pipelined,
                              owner=(k % self.comm.world) if owner_mode else None, draw_on_owner=draw_on_owner)
                entry = [k, batch.steps(), slot, None]
                alive = advance(entry)
                if k == len(jobs) - 1:                           # (the draw belongs to a batch's first segment)
                    last_drew = getattr(batch, "drew_test_matrix", True) is not False
                if alive:
                    active.append(entry)
            if not active:
                continue
            pick = 0
            if not ordered:
                for i, e in enumerate(active):
                    if _fired(e[3]):
                        pick = i
                        break
            entry = active[pick]
            del active[pick]
            entry[3].synchronize()
            if advance(entry):
                active.append(entry)
        if draw_on_owner and not last_drew:
            # sequential calls would leave every rank's generator at "last seed + one draw": this rank skipped that draw
            from ._basis import _skip_test_matrix_draw

            last = jobs[-1]
            torch.manual_seed(last.seed)
            _skip_test_matrix_draw(slot_ops[0], int(last.pts_nys.shape[0]), int(last.num_pts) - 1)
        for ops in slot_ops:                                     # the results are valid for every stream once this returns
            sync = getattr(ops, "synchronize", None)
            if sync is not None:
                sync()
        return results


def _fired(ev):
    q = getattr(ev, "query", None)
    return True if q is None else bool(q())


def _stream_of(ops):
    """Context in which torch's allocations and library calls go to the stream ``ops`` is bound to."""
    s = getattr(ops, "stream", None)
    return torch.cuda.stream(s) if s is not None else contextlib.nullcontext()
