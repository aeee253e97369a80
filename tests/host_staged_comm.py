"""TEST-ONLY communicator: the engine's collectives between processes that SHARE one GPU.

RCCL refuses two ranks on one device, and a test box has one MI355X.  ``HostStagedComm`` keeps ``TorchDistComm``'s interface
(``rank``, ``world``, ``all_gather``, ``broadcast``, ``for_slot``) but moves every message device -> host -> ``gloo`` -> host ->
device.  Each call waits for the CURRENT stream before it reads the message (the engine enqueues its collectives stream-ordered;
here the host stands in for the stream order), so the kernels on either side of an exchange -- shard-aware ``*_geo`` launches,
owner-rank broadcasts, the per-slot groups of batches in flight -- run exactly as under RCCL, on real HIP streams, in two
processes.  Nothing under ``basq_amd/`` imports this module.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class HostStagedComm:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.calls = {"all_gather": 0, "broadcast": 0}
        self._slots = {}

    def all_gather(self, t):
        """-> ``[world, *t.shape]`` on ``t``'s device, rank order."""
        self.calls["all_gather"] += 1
        host = t.detach().contiguous().cpu()                    # (waits for the current stream: the message is complete)
        out = torch.empty((self.world,) + tuple(host.shape), dtype=host.dtype)
        dist.all_gather_into_tensor(out.view(-1), host.view(-1), group=self.group)
        return out.to(t.device)

    def broadcast(self, t, src=0):
        """In place on ``t``; ``src`` = rank within this communicator's group."""
        self.calls["broadcast"] += 1
        g_src = src if self.group is None else dist.get_global_rank(self.group, src)
        host = t.detach().contiguous().cpu()
        dist.broadcast(host, src=g_src, group=self.group)
        if self.rank != src:
            t.copy_(host.to(t.device))
        return t

    def for_slot(self, i):
        comm = self._slots.get(i)
        if comm is None:
            ranks = list(range(dist.get_world_size())) if self.group is None else dist.get_process_group_ranks(self.group)
            comm = self._slots[i] = HostStagedComm(dist.new_group(ranks=ranks, backend="gloo"))
        return comm
