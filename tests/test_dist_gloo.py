"""The N > 1 path on the CPU: world_size 2 and 3 over gloo, stand-in device ops, against the golden
vectors.  Exercises sharding by global position, the per-round all-gather + ordered sum on rank 0,
the broadcast of the reduction result and the closed-form local compaction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.cases import BY_NAME, build_pool, build_product_kernel, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sober_worker(rank, world, port, i, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from basq_amd._engine import RecombinationEngine, TorchDistComm
        from basq_amd._partition import initial_shards
        from basq_amd.kernels import StationaryKernel
        from basq_amd.pools import gmm_pool
        from oracle.make_golden_sober import CASES, case_weights
        from tests.cpu_stand_in import CpuStandInOps

        c = CASES[i]
        pts = gmm_pool(c["N"], c["d"], c["pool_seed"])
        w0 = case_weights(c)
        off, n = initial_shards(c["N"], world)[rank]
        torch.manual_seed(1)
        idx, w = RecombinationEngine(CpuStandInOps(), TorchDistComm()).run(
            pts[off:off + n].clone(), off, c["N"], pts[: c["m"]], c["n"], StationaryKernel(c["family"], c["lengthscale"], 1.0),
            variant="sober", init_weights=None if w0 is None else w0[off:off + n].clone())
        q.put((rank, idx.tolist(), w.tolist()))
    finally:
        dist.destroy_process_group()


def _timeout_worker(rank, world, port, name, q, bad_rank):
    """One rank's cluster kernels 'time out' (status 2) now and then; the others never do."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings

        from basq_amd._engine import RecombinationEngine, TorchDistComm
        from basq_amd._partition import initial_shards
        from tests.cpu_stand_in import CpuStandInOps

        class TimingOut(CpuStandInOps):
            def car_eliminate(self, PhiT, mu, M, s, cluster=True, out=None):
                res = super().car_eliminate(PhiT, mu, M, s, cluster, out)
                if cluster and rank == bad_rank and self.calls["car"] in (2, 5):
                    res[3][1] = 2
                return res

        c = BY_NAME[name]
        pts, nys = build_pool(c)
        off, n = initial_shards(c["N"], world)[rank]
        torch.manual_seed(c["torch_seed"])
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            idx, w = RecombinationEngine(TimingOut(), TorchDistComm()).run(pts[off:off + n].clone(), off, c["N"], nys, c["n"],
                                                                           build_product_kernel(c))
        q.put((rank, idx.tolist(), w.tolist(), sum("timed out" in str(x.message) or "repeated" in str(x.message) for x in wlist)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,bad_rank", [("rbf_ragged", 2, 1), ("cfg1_posterior_1e4", 3, 2)])
def test_cluster_timeout_on_one_rank_is_a_collective_decision(name, world, bad_rank):
    """with the reduction replicated on every rank, a cluster kernel's time-out (status 2) is local to ONE rank.
    The retry -- in the descriptor-driven rounds: the fall-back to the round-by-round loop -- must be taken by ALL ranks or
    their all-gathers pair up across different rounds.  The status travels with the next exchange (maximum over the ranks);
    every rank returns the golden batch and every rank reports the retry."""
    fx = load_golden(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timeout_worker, args=(r, world, port, name, q, bad_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    for rank, idx, w, n_warn in res:
        assert idx == fx["idx"], f"rank {rank}"
        assert ((torch.tensor(w, dtype=torch.float64) - gw).abs() / gw).max().item() <= 1e-6
        assert n_warn >= 1, f"rank {rank} did not report the retry"


def _worker(rank, world, port, name, q, rank0_only=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import basq_amd._config as E
        from basq_amd._engine import EngineTrace, RecombinationEngine, TorchDistComm

        if rank0_only:                                          # basis + per-round reduction on rank 0, results broadcast
            E.SHARDED_BASIS = False
            E.REPLICATED_REDUCTION = False
        from basq_amd._partition import initial_shards
        from tests.cpu_stand_in import CpuStandInOps

        c = BY_NAME[name]
        pts, nys = build_pool(c)
        off, n = initial_shards(c["N"], world)[rank]
        tr = EngineTrace()
        torch.manual_seed(c["torch_seed"] if rank == 0 else 12345 + rank)   # only rank 0's draw may matter
        idx, w = RecombinationEngine(CpuStandInOps(), TorchDistComm()).run(
            pts[off:off + n].clone(), off, c["N"], nys, c["n"], build_product_kernel(c), tr)
        q.put((rank, idx.tolist(), w.tolist(), [r["kept"] for r in tr.rounds]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,rank0_only", [("rbf_ragged", 2, False), ("rbf_1e4", 2, False),
                                                   ("cfg1_posterior_1e4", 2, False), ("wsabil_2e4", 2, False),
                                                   ("rbf_ragged", 3, False), ("rbf_tiny_final", 2, False),
                                                   ("wsabim_1e4", 2, False), ("posterior_noise_ragged", 3, False),
                                                   ("wsabil_noise_ragged", 2, False), ("wsabim_noise_ragged", 2, False),
                                                   ("rbf_ragged", 3, True),
                                                   ("cfg1_posterior_1e4", 2, True)])
def test_sharded_engine_matches_golden(name, world, rank0_only):
    """Default multi-rank mode (sharded range finder, reduction replicated on every rank) and the rank-0-only mode
    (basis and reduction on rank 0, results broadcast): same indices as the reference either way."""
    fx = load_golden(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q, rank0_only)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    for rank, idx, w, kept in res:
        assert idx == fx["idx"], f"rank {rank}: indices differ"
        assert ((torch.tensor(w, dtype=torch.float64) - gw).abs() / gw).max().item() <= 1e-6
        assert kept == [r["kept"] for r in fx["rounds"]]
    # every rank returns the identical result
    assert all(r[1] == res[0][1] and r[2] == res[0][2] for r in res)


@pytest.mark.parametrize("i,world", [(1, 2), (2, 3)])
def test_sharded_sober_variant_matches_golden(i, world):
    """SOBER flavour (importance weights with zeros, remainder double count) sharded over ranks."""
    import json

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sober.json")) as f:
        fx = json.load(f)[i]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sober_worker, args=(r, world, port, i, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    for rank, idx, w in res:
        assert idx == fx["idx"], f"rank {rank}"
        assert ((torch.tensor(w, dtype=torch.float64) - gw).abs() / gw).max().item() <= 1e-6


def _sober_tutorial_worker(rank, world, port, i, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from basq_amd._engine import RecombinationEngine, TorchDistComm
        from basq_amd._partition import initial_shards
        from oracle.make_golden_sober import TUTORIAL_CASES, tutorial_inputs
        from tests.cases import build_product_kernel
        from tests.cpu_stand_in import CpuStandInOps

        c = TUTORIAL_CASES[i]
        pts, nys = tutorial_inputs(c)
        off, n = initial_shards(c["N"], world)[rank]
        ops = CpuStandInOps()
        torch.manual_seed(1)
        idx, w = RecombinationEngine(ops, TorchDistComm()).run(pts[off:off + n].clone(), off, c["N"], nys, c["n"],
                                                               build_product_kernel(c), variant="sober")
        q.put((rank, idx.tolist(), w.tolist(), ops.calls.get("regroup", 0)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("i,world", [(1, 2), (4, 3), (3, 4)])
def test_sharded_sober_tutorial_cases_on_the_class_paths(i, world):
    """Round 4: the SOBER variant at the tutorials' size on several ranks -- residue-class sums (regrouped rounds), the
    remainder's first count as one more irregular chunk (shards that end inside the remainder), WSABI-M's squared covariance
    per class with its per-candidate noise terms (case 4) -- against the goldens of the imported ``SOBER/_rchq.py``."""
    import json

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sober_tutorial.json")) as f:
        fx = json.load(f)[i]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sober_tutorial_worker, args=(r, world, port, i, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    rtol = 1e-5 if fx["case"]["kernel"]["warp"] == "wsabim" else 1e-6        # (tests/test_sober.py::_tut_rtol)
    for rank, idx, w, n_regroup in res:
        assert idx == fx["idx"], f"rank {rank}"
        assert ((torch.tensor(w, dtype=torch.float64) - gw).abs() / gw).max().item() <= rtol
        assert n_regroup > 0                                                  # the class path was taken
    assert all(r[1] == res[0][1] and r[2] == res[0][2] for r in res)


def _basis_worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from basq_amd._basis import _ShardedProducts, nystrom_basis
        from basq_amd._engine import TorchDistComm
        from basq_amd._partition import initial_shards
        from tests.cpu_stand_in import CpuStandInOps

        c = BY_NAME[name]
        _, nys = build_pool(c)
        ops, comm = CpuStandInOps(), TorchDistComm()
        kern = build_product_kernel(c)
        center = ops.col_mean(nys)
        shards = initial_shards(c["m"], world)
        r0, mr = shards[rank]
        rows = kern.dense(ops, nys[r0:r0 + mr].contiguous(), nys, center, diag_offset=r0)
        torch.manual_seed(7 if rank == 0 else 99)
        U = nystrom_basis(ops, _ShardedProducts(ops, comm, rows, shards, c["m"]), c["n"] - 1)
        full = kern.dense(ops, nys, nys, center)
        q.put((rank, U, (rows - full[r0:r0 + mr]).abs().max().item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("cfg1_posterior_1e4", 2), ("wsabil_2e4", 3)])
def test_sharded_range_finder_equals_dense(name, world):
    """Row-sharded Gram products (+ the symmetry A ~ A^T) give the same Nystrom basis as the single-process range
    finder, up to the sign of each row and rounding; the row blocks carry the structured kernels' diagonal terms on the
    true diagonal (``diag_offset``); every rank ends up with the same basis; only rank 0's generator is consumed."""
    from basq_amd._basis import nystrom_basis
    from tests.cpu_stand_in import CpuStandInOps

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_basis_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    c = BY_NAME[name]
    _, nys = build_pool(c)
    ops = CpuStandInOps()
    kern = build_product_kernel(c)
    torch.manual_seed(7)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)        # as in the workers: a BLAS team that resizes under load re-associates the products
    try:
        Ud = nystrom_basis(ops, kern.dense(ops, nys, nys, ops.col_mean(nys)), c["n"] - 1)
    finally:
        torch.set_num_threads(nthreads)
    for rank, U, blk_err in res:
        assert blk_err == 0.0                                   # same entries as the full Gram, diagonal terms included
        assert torch.equal(U, res[0][1])                        # replicated steps agree bit for bit across ranks
        sign = torch.sign((U * Ud).sum(1, keepdim=True))
        assert (U * sign - Ud).abs().max().item() <= 1e-7


# ---- world sizes 1 / 2 / 3 / 4 / 8 against each other (multi-rank result equality) --------------------------
def _plain_worker(rank, world, port, name, q, two_batches=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from basq_amd._engine import EngineTrace, RecombinationEngine, TorchDistComm
        from basq_amd._partition import initial_shards
        from tests.cpu_stand_in import CpuStandInOps

        c = BY_NAME[name]
        out = []
        if not two_batches:
            pts, nys = build_pool(c)
            off, n = initial_shards(c["N"], world)[rank]
            tr = EngineTrace()
            torch.manual_seed(c["torch_seed"])
            idx, w = RecombinationEngine(CpuStandInOps(), TorchDistComm()).run(
                pts[off:off + n].clone(), off, c["N"], nys, c["n"], build_product_kernel(c), tr)
            out.append((idx.tolist(), w.tolist(), [r["kept"] for r in tr.rounds]))
        else:
            # a BASQ-style loop: every rank draws EVERY pool from its own global generator (as PriorSampler / bench.py
            # do) and keeps its slice -- only correct while the generators of all ranks stay in lock-step
            torch.manual_seed(77)
            for _ in range(2):
                pool = torch.randn(c["N"], c["d"], dtype=torch.float64) * 2.0
                nys = pool[: c["m"]]
                off, n = initial_shards(c["N"], world)[rank]
                idx, w = RecombinationEngine(CpuStandInOps(), TorchDistComm()).run(
                    pool[off:off + n].clone(), off, c["N"], nys, c["n"], build_product_kernel(c))
                out.append((idx.tolist(), w.tolist(), float(pool.sum())))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _run_world(name, world, two_batches=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plain_worker, args=(r, world, port, name, q, two_batches)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return [res[r] for r in range(world)]


@pytest.mark.parametrize("name", ["rbf_ragged", "cfg1_posterior_1e4"])
def test_world_sizes_1_2_3_4_8_agree(name):
    """Same batch on 1, 2, 3, 4 and 8 ranks: every rank of every run returns the golden indices and the same per-round
    survivor sets; within a run all ranks are BIT-identical (replicated fixed-order reduction); between rank counts the
    weights agree to <= 1e-10 relative (measured 1e-15 .. 2e-12: the per-rank partial messages are associated
    differently and the sharded range finder multiplies by A where one rank multiplies by A^T, both at rounding level,
    amplified ~100x by the reduction -- so bit-equality ACROSS rank counts is not promised)."""
    fx = load_golden(name)
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    ref_w = None
    for world in (1, 2, 3, 4, 8):
        res = _run_world(name, world)
        idx0, w0, kept0 = res[0][0]
        assert idx0 == fx["idx"] and kept0 == [r["kept"] for r in fx["rounds"]], f"world {world}"
        for r in range(1, world):
            assert res[r][0] == res[0][0], f"world {world}: rank {r} differs from rank 0 (must be bit-identical)"
        w = torch.tensor(w0, dtype=torch.float64)
        assert ((w - gw).abs() / gw).max().item() <= 1e-6
        if ref_w is None:
            ref_w = w
        assert ((w - ref_w).abs() / ref_w).max().item() <= 1e-10, f"world {world} vs world 1"


def test_global_rng_stays_in_lock_step_across_ranks():
    """Two consecutive sharded batches with pools drawn from the GLOBAL generator on every rank: rank 0
    alone consumes the svd_lowrank draw, so without the matching skip on the other ranks batch 2 would see a different
    pool on every rank.  All ranks must report the same pools and identical results, equal to a 1-rank run."""
    name = "rbf_1e4"
    one = _run_world(name, 1, two_batches=True)[0]
    for world in (2, 3):
        res = _run_world(name, world, two_batches=True)
        for r in range(world):
            for b in range(2):
                assert res[r][b][2] == one[b][2], f"world {world} rank {r} batch {b}: a different pool was drawn"
                assert res[r][b][0] == one[b][0], f"world {world} rank {r} batch {b}: indices differ from the 1-rank run"
                w, w1 = torch.tensor(res[r][b][1]), torch.tensor(one[b][1])
                assert ((w - w1).abs() / w1).max().item() <= 1e-9


# ---- descriptor-driven rounds on several ranks (no host wait per round; the exchange is stream-ordered) ----------------
def _async_worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from basq_amd._engine import EngineTrace, RecombinationEngine, TorchDistComm
        from basq_amd._partition import initial_shards
        from tests.cpu_stand_in import CpuStandInOps

        c = BY_NAME[name]
        pts, nys = build_pool(c)
        off, n = initial_shards(c["N"], world)[rank]
        tr = EngineTrace(host_sync=False)                       # the path an untraced call takes, rounds read back after
        ops = CpuStandInOps()
        torch.manual_seed(c["torch_seed"])
        idx, w = RecombinationEngine(ops, TorchDistComm()).run(pts[off:off + n].clone(), off, c["N"], nys, c["n"],
                                                               build_product_kernel(c), tr)
        q.put((rank, idx.tolist(), w.tolist(), [r["kept"] for r in tr.rounds], ops.calls.get("round_next", 0),
               ops.calls.get("epoch_turn", 0), ops.calls.get("compact_rounds", 0)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("rbf_ragged", 2), ("rbf_ragged", 4), ("rbf_ragged", 8),
                                        ("cfg1_posterior_1e4", 2), ("cfg1_posterior_1e4", 4),
                                        ("posterior_noise_ragged", 3), ("wsabil_noise_ragged", 2), ("wsabil_2e4", 4),
                                        ("rbf_2e4_defaults", 8), ("matern52_posterior", 2)])
def test_sharded_descriptor_driven_rounds(name, world):
    """The N > 1 fast path: every rank's shard of every round comes from the device-resident descriptor
    (``basq_round_next_i64`` evaluates ``next_shard`` in closed form), the per-round all-gather is enqueued like any other
    launch, and the host reads the descriptor table once.  Golden indices and per-round survivor sets on every rank,
    ranks bit-identical among themselves.  Round 6: inside an epoch every rank regroups the PARTIAL class messages and message
    columns of its shard (``epoch_turn`` is linear), and the compactions of an epoch walk the rank's shard through the descriptors."""
    fx = load_golden(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_async_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    for rank, idx, w, kept, n_next, n_turn, n_flush in res:
        assert n_next > 0, "the descriptor-driven rounds did not run"
        # round 6: several ranks take the column epochs too (partial class messages and columns per rank, basq_amd/_epochs.py)
        assert n_turn > 0 and n_flush > 0, "the column epochs did not run on this rank"
        assert idx == fx["idx"], f"rank {rank}: indices differ"
        assert kept == [r["kept"] for r in fx["rounds"]]
        assert ((torch.tensor(w, dtype=torch.float64) - gw).abs() / gw).max().item() <= 1e-6
    assert all(r[1] == res[0][1] and r[2] == res[0][2] for r in res)


# ---- the test-only host-staged communicator (tests/host_staged_comm.py) itself, on the CPU --------------------------------
def _host_staged_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from basq_amd._engine import Job, RecombinationEngine
        from basq_amd._partition import initial_shards
        from tests.cpu_stand_in import CpuStandInOps
        from tests.host_staged_comm import HostStagedComm

        comm = HostStagedComm()
        names = ["rbf_ragged", "cfg1_posterior_1e4"]
        out, jobs = [], []
        for name in names:
            c = BY_NAME[name]
            pts, nys = build_pool(c)
            off, n = initial_shards(c["N"], world)[rank]
            torch.manual_seed(c["torch_seed"])
            idx, w = RecombinationEngine(CpuStandInOps(), comm).run(pts[off:off + n].clone(), off, c["N"], nys, c["n"],
                                                                    build_product_kernel(c))
            out.append((idx.tolist(), w.tolist()))
            jobs.append(Job(pts[off:off + n].clone(), off, c["N"], nys, c["n"], build_product_kernel(c), seed=c["torch_seed"]))
        many = RecombinationEngine(CpuStandInOps(), comm).run_many(jobs + jobs, [CpuStandInOps(), CpuStandInOps()])
        q.put((rank, out, [(i.tolist(), w.tolist()) for i, w in many], sorted((k, s.calls["broadcast"]) for k, s in comm._slots.items())))
    finally:
        dist.destroy_process_group()


def test_host_staged_comm_drives_the_engine():
    """The communicator the two-processes-on-one-GPU test uses (``tests/test_two_ranks_one_gpu.py``) against the goldens on the CPU
    stand-in: sequential sharded calls and four batches on two slots (owner-rank reductions, one gloo group per slot)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_host_staged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, many, slot_b in res:
        for nm, (idx, w) in zip(["rbf_ragged", "cfg1_posterior_1e4"], out):
            assert idx == load_golden(nm)["idx"], f"rank {rank} {nm}"
        for nm, (idx, w) in zip(["rbf_ragged", "cfg1_posterior_1e4"] * 2, many):
            assert idx == load_golden(nm)["idx"], f"rank {rank} {nm} (in flight)"
        assert len(slot_b) == 2 and all(n > 0 for _, n in slot_b)
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]


# ---- per-slot process groups survive a destroy + re-init of the default group in one process -------------------
def _reinit_worker(rank, world, ports, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    torch.set_num_threads(2)
    from basq_amd._engine import Job, RecombinationEngine, TorchDistComm, _SLOT_COMMS, release_slot_comms
    from basq_amd._partition import initial_shards
    from tests.cpu_stand_in import CpuStandInOps

    names = ["rbf_ragged", "cfg1_posterior_1e4"]
    out = []
    try:
        for phase, port in enumerate(ports):
            os.environ["MASTER_PORT"] = str(port)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            jobs = []
            for name in names:
                c = BY_NAME[name]
                pts, nys = build_pool(c)
                off, n = initial_shards(c["N"], world)[rank]
                jobs.append(Job(pts[off:off + n].clone(), off, c["N"], nys, c["n"], build_product_kernel(c), seed=c["torch_seed"]))
            res = RecombinationEngine(CpuStandInOps(), TorchDistComm()).run_many(jobs, [CpuStandInOps(), CpuStandInOps()])
            out.append([i.tolist() for i, _ in res])
            n_cached = len(_SLOT_COMMS)
            if phase == 0:
                dist.destroy_process_group()             # (the cache still holds communicators of the destroyed world)
            else:
                release_slot_comms()                     # the documented way: sub-groups destroyed, cache empty
                assert len(_SLOT_COMMS) == 0
                dist.destroy_process_group()
            out.append(n_cached)
        q.put((rank, out))
    except Exception as e:                               # noqa: BLE001
        q.put((rank, repr(e)))


def test_slot_process_groups_survive_a_reinitialised_default_group():
    """The cache of per-slot process groups (``TorchDistComm.for_slot``) used to be keyed on ``id(group)`` -- ``id(None)`` for the
    default group -- so after ``destroy_process_group()`` and a second ``init_process_group()`` in the same process (test suites,
    long-lived services) it handed out communicators of the DESTROYED world.  Two initialisations in one process, owner-rank
    reductions on two slots in both: goldens both times."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ports = [_free_port(), _free_port()]
    procs = [ctx.Process(target=_reinit_worker, args=(r, world, ports, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [load_golden(nm)["idx"] for nm in ("rbf_ragged", "cfg1_posterior_1e4")]
    for rank, out in res:
        assert isinstance(out, list), f"rank {rank}: {out}"
        assert out[0] == want and out[2] == want, f"rank {rank}"
        assert out[1] == 2 and out[3] == 2               # two slot groups cached per initialisation (the stale ones replaced)
