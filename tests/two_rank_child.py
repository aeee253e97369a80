"""Child of tests/test_two_ranks_one_gpu.py: ONE rank of a two-rank world whose ranks share ``cuda:0``.

    python tests/two_rank_child.py RANK WORLD PORT OUT.pt CASE [CASE ...]

A fresh process (it initialises HIP itself), real ``HipOps`` on real streams, collectives through ``HostStagedComm`` (gloo).
Writes what the parent asserts on: per case the sharded call's indices / weights / per-round kept sets, the results of four
batches in flight on two slots (owner-rank reductions), and digests of the CPU generator's state after each.
"""
import hashlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rng_digest():
    return hashlib.sha256(torch.get_rng_state().numpy().tobytes()).hexdigest()[:16]


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    names = sys.argv[5:]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import basq_amd
    import basq_amd._config as cfg
    from basq_amd._partition import initial_shards
    from tests.cases import BY_NAME, build_pool, build_product_kernel
    from tests.host_staged_comm import HostStagedComm

    assert cfg.ASYNC_ROUNDS and cfg.OWNER_RANK_REDUCTION and cfg.SHARDED_BASIS
    dev = torch.device("cuda", 0)
    comm = HostStagedComm()
    res = {"rank": rank, "device": str(torch.cuda.get_device_properties(dev).name), "cases": {}}
    shards = {}
    for name in names:
        c = BY_NAME[name]
        pts, nys = build_pool(c)
        off, n = initial_shards(c["N"], world)[rank]
        shards[name] = (pts[off:off + n].to(dev), off, c["N"], nys.to(dev), c["n"], build_product_kernel(c))
    # ---- one batch at a time: descriptor-driven rounds (no host wait per round), reduction replicated on both ranks ----
    for name in names:
        c = BY_NAME[name]
        tr = basq_amd.EngineTrace(host_sync=False)              # (stays on the path an untraced call takes)
        torch.manual_seed(c["torch_seed"])
        before = dict(comm.calls)
        idx, w = basq_amd.recombination_sharded(*shards[name], dev, trace=tr, comm=comm)
        res["cases"][name] = dict(idx=idx.cpu().tolist(), w=w.cpu(), kept=[r["kept"] for r in tr.rounds],
                                  rounds=[(r["R"], r["S"]) for r in tr.rounds], rng=rng_digest(),
                                  all_gathers=comm.calls["all_gather"] - before["all_gather"],
                                  broadcasts=comm.calls["broadcast"] - before["broadcast"])
    # ---- four batches, two in flight: owner-rank reductions (batch k's chain on rank k mod 2, outcome broadcast on the slot's group)
    jobs = [names[k % len(names)] for k in range(4)]
    calls = [shards[nm] for nm in jobs]
    seeds = [BY_NAME[nm]["torch_seed"] for nm in jobs]
    many = basq_amd.recombination_many_sharded(calls, dev, in_flight=2, seeds=seeds, comm=comm)
    res["many"] = [dict(name=nm, idx=i.cpu().tolist(), w=w.cpu()) for nm, (i, w) in zip(jobs, many)]
    res["many_rng"] = rng_digest()
    res["slot_broadcasts"] = sorted((k, s.calls["broadcast"]) for k, s in comm._slots.items())
    torch.cuda.synchronize()
    torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()
    print("TWO-RANK-OK", rank, flush=True)


if __name__ == "__main__":
    main()
