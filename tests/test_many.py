"""Several recombinations in flight (``RecombinationEngine.run_many`` / ``basq_amd.recombination_many``).

The reference calls the hot path twice per BASQ iteration, independently: batch selection (``BASQ/_basq.py:82-88``) and
quadrature (``:104-106`` -> ``_quadrature.py:59-60``).  Keeping both on the GPU at once must not change either result:
every batch is bit-identical to its sequential run, and the CPU global generator is consumed in call order.

CPU part: the scheduler on the stand-in ops (interleaving, RNG order, seeds, traces, gloo world 2).
GPU part (``-m gpu``): the HIP path on separate streams.
"""
import os
import socket

import pytest
import torch

from basq_amd._engine import EngineTrace, Job, RecombinationEngine
from basq_amd.kernels import StationaryKernel
from basq_amd.pools import gmm_pool
from tests.cases import BY_NAME, build_pool, build_product_kernel, load_golden
from tests.cpu_stand_in import CpuStandInOps


def _jobs(specs, seeds=None, traces=None):
    out = []
    for k, (N, d, m, n, pool_seed, kern) in enumerate(specs):
        pts = gmm_pool(N, d, pool_seed)
        out.append(Job(pts, 0, N, pts[:m], n, kern, seed=None if seeds is None else seeds[k],
                       trace=None if traces is None else traces[k]))
    return out


SPECS = [(30_000, 3, 200, 20, 8, StationaryKernel("rbf", 1.5, 1.0)),        # several descriptor-driven rounds
         (9_000, 5, 120, 30, 5, StationaryKernel("matern52", 2.0, 1.3)),
         (700, 3, 60, 25, 6, StationaryKernel("rbf", 1.5, 1.0)),            # one asynchronous round at most
         (90, 2, 30, 20, 7, StationaryKernel("rbf", 1.0, 1.0)),             # nothing asynchronous
         (50_321, 3, 200, 16, 4, StationaryKernel("rbf", 1.5, 1.0))]


@pytest.mark.parametrize("in_flight", [1, 2, 3])
def test_run_many_equals_sequential_runs(in_flight):
    """Same results, bit for bit, and the same generator consumption as one call after the other."""
    torch.manual_seed(3)
    seq = [RecombinationEngine(CpuStandInOps()).run(j.pts_local, 0, j.n_total, j.pts_nys, j.num_pts, j.kernel)
           for j in _jobs(SPECS)]
    after_seq = torch.rand(1).item()
    torch.manual_seed(3)
    slots = [CpuStandInOps() for _ in range(in_flight)]
    many = RecombinationEngine(slots[0]).run_many(_jobs(SPECS), slots)
    after_many = torch.rand(1).item()
    assert after_seq == after_many
    for (ia, wa), (ib, wb) in zip(seq, many):
        assert torch.equal(ia, ib) and torch.equal(wa, wb)
    if in_flight > 1:
        assert all(o.calls.get("car", 0) > 0 for o in slots[:2])             # the batches really ran on different slots


def test_run_many_seeds_and_traces():
    """``Job.seed`` = ``torch.manual_seed(seed)`` right before that call; a trace per job sees its own rounds."""
    seeds = [11, 12, 13, 14, 15]
    seq = []
    for j, sd in zip(_jobs(SPECS), seeds):
        torch.manual_seed(sd)
        tr = EngineTrace()
        idx, w = RecombinationEngine(CpuStandInOps()).run(j.pts_local, 0, j.n_total, j.pts_nys, j.num_pts, j.kernel, tr)
        seq.append((idx, w, [r["kept"] for r in tr.rounds]))
    traces = [EngineTrace(host_sync=False) for _ in SPECS]
    slots = [CpuStandInOps(), CpuStandInOps()]
    many = RecombinationEngine(slots[0]).run_many(_jobs(SPECS, seeds, traces), slots)
    for (ia, wa, ka), (ib, wb), tr in zip(seq, many, traces):
        assert torch.equal(ia, ib)
        assert torch.allclose(wa, wb, rtol=1e-11, atol=0)                     # (the traced sequential run takes the sync loop)
        assert [r["kept"] for r in tr.rounds] == ka


def test_run_many_golden_pair():
    """The reference's own pair -- a selection-sized and a quadrature-sized pool -- against their goldens."""
    names = ["rbf_2e4_defaults", "cfg1_posterior_1e4", "wsabil_2e4"]
    jobs = []
    for nm in names:
        c = BY_NAME[nm]
        pts, nys = build_pool(c)
        jobs.append(Job(pts, 0, c["N"], nys, c["n"], build_product_kernel(c), seed=c["torch_seed"]))
    slots = [CpuStandInOps(), CpuStandInOps()]
    res = RecombinationEngine(slots[0]).run_many(jobs, slots)
    for nm, (idx, w) in zip(names, res):
        fx = load_golden(nm)
        assert idx.tolist() == fx["idx"]
        gw = torch.tensor(fx["w"], dtype=torch.float64)
        assert ((w - gw).abs() / gw).max().item() <= 1e-6


def test_trace_without_host_sync_takes_the_descriptor_path():
    """``EngineTrace(host_sync=False)`` leaves the batch on the path an untraced call takes and still reports every round."""
    c = BY_NAME["rbf_ragged"]
    pts, nys = build_pool(c)
    fx = load_golden("rbf_ragged")
    ops = CpuStandInOps()
    tr = EngineTrace(host_sync=False)
    torch.manual_seed(c["torch_seed"])
    idx, w = RecombinationEngine(ops).run(pts, 0, c["N"], nys, c["n"], build_product_kernel(c), tr)
    assert ops.calls.get("round_next", 0) > 0
    assert idx.tolist() == fx["idx"]
    assert [r["kept"] for r in tr.rounds] == [r["kept"] for r in fx["rounds"]]
    assert len(tr.rounds) == fx["n_rounds"] and all(r["S"] == r2["M"] for r, r2 in zip(tr.rounds, fx["rounds"]))


# ---- several batches in flight on several ranks (gloo) -----------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, opaque=(), owner_mode=True, n_slots=2, names=("rbf_ragged", "cfg1_posterior_1e4", "rbf_1e4")):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import basq_amd._config as cfg
        from basq_amd._engine import EngineTrace, TorchDistComm
        from basq_amd._partition import initial_shards

        cfg.OWNER_RANK_REDUCTION = owner_mode
        jobs = []
        for nm in names:
            c = BY_NAME[nm]
            pts, nys = build_pool(c)
            off, n = initial_shards(c["N"], world)[rank]
            kern = build_product_kernel(c)
            if nm in opaque:                                     # the same kernel as a bare callable (dense path)
                from basq_amd.kernels import CallableKernel
                from tests.cases import build_oracle_kernel

                ko, _ = build_oracle_kernel(c)
                kern = CallableKernel(lambda x, y, ko=ko: ko(x, y))
            jobs.append(Job(pts[off:off + n].clone(), off, c["N"], nys, c["n"], kern, seed=c["torch_seed"],
                            trace=EngineTrace(host_sync=False)))
        slots = [CpuStandInOps() for _ in range(n_slots)]
        res = RecombinationEngine(slots[0], TorchDistComm()).run_many(jobs, slots)
        import hashlib

        q.put((rank, [(i.tolist(), w.tolist()) for i, w in res], slots[0].calls.get("round_next", 0),
               [[r["kept"] for r in j.trace.rounds] for j in jobs],
               sum(sl.calls.get("car", 0) for sl in slots),
               hashlib.sha256(torch.get_rng_state().numpy().tobytes()).hexdigest()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_run_many_sharded_matches_goldens(world):
    """Two batches in flight on every rank of a gloo group: the ranks enqueue their collectives in the same order (the
    scheduler resumes batches FIFO on several ranks), every rank returns the golden batch of every job."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, n_next, _, _, _ in res:
        assert n_next > 0                                                     # descriptor-driven rounds on several ranks
        for nm, (idx, w) in zip(["rbf_ragged", "cfg1_posterior_1e4", "rbf_1e4"], out):
            fx = load_golden(nm)
            assert idx == fx["idx"], f"rank {rank} {nm}"
            gw = torch.tensor(fx["w"], dtype=torch.float64)
            assert ((torch.tensor(w, dtype=torch.float64) - gw).abs() / gw).max().item() <= 1e-6
    assert all(r[1] == res[0][1] for r in res)                                # bit-identical across the ranks


OWNER_NAMES = ("rbf_ragged", "cfg1_posterior_1e4", "rbf_1e4", "wsabil_noise_ragged", "matern52_posterior", "posterior_noise_ragged",
               "rbf_2e4_defaults", "wsabil_2e4")


@pytest.mark.parametrize("world,n_slots,owner_mode", [(2, 2, True), (4, 4, True), (8, 4, True), (3, 2, True), (2, 3, False)])
def test_run_many_owner_rank_reductions(world, n_slots, owner_mode):
    """Several ranks x several batches in flight with OWNER-RANK reductions: batch k's null space + elimination run on rank
    k mod G only, the outcome travels by a broadcast on the batch's own process group (``_config.OWNER_RANK_REDUCTION``).
    Eight jobs (stationary, posterior, WSABI-L; with and without a ragged remainder / a visible noise diagonal) on worlds
    2, 3, 4 and 8: every rank returns the golden indices AND the golden per-round kept sets of every job, all ranks are
    bit-identical, and the eliminations really were dealt out (each rank ran about 1/G of them -- in the replicated mode,
    ``owner_mode`` False, every rank runs all of them)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, (), owner_mode, n_slots, OWNER_NAMES)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, n_next, kept, _, _ in res:
        assert n_next > 0
        for nm, (idx, w), kr in zip(OWNER_NAMES, out, kept):
            fx = load_golden(nm)
            assert idx == fx["idx"], f"rank {rank} {nm}"
            gw = torch.tensor(fx["w"], dtype=torch.float64)
            assert ((torch.tensor(w, dtype=torch.float64) - gw).abs() / gw).max().item() <= 1e-6
            assert kr == [r["kept"] for r in fx["rounds"]], f"rank {rank} {nm}: per-round kept sets"
    assert all(r[1] == res[0][1] for r in res)
    # the CPU global generator ends where sequential calls would leave it -- last job's seed + ONE draw of its test matrix -- on
    # EVERY rank, although (owner mode, every job seeded) each draw was made by one rank only
    import hashlib

    c_last = BY_NAME[OWNER_NAMES[-1]]
    torch.manual_seed(c_last["torch_seed"])
    torch.randn(c_last["m"], c_last["n"] - 1, dtype=torch.float64)
    want = hashlib.sha256(torch.get_rng_state().numpy().tobytes()).hexdigest()
    assert all(r[5] == want for r in res)
    cars = sorted(r[4] for r in res)
    total_rounds = sum(load_golden(nm)["n_rounds"] for nm in OWNER_NAMES)
    if owner_mode:
        assert sum(cars) == total_rounds and cars[-1] <= total_rounds // world + 2 * max(load_golden(nm)["n_rounds"] for nm in OWNER_NAMES)
    else:
        assert all(c == total_rounds for c in cars)


@pytest.mark.parametrize("opaque", [("rbf_ragged",), ("cfg1_posterior_1e4",), ("rbf_ragged", "rbf_1e4")])
def test_run_many_sharded_with_opaque_callables_and_more_jobs_than_slots(opaque):
    """three jobs on two slots, some of them opaque callables -- whose basis only rank 0 computes (the other ranks
    wait for the broadcast).  Every rank must yield at the same points of such a batch, or the FIFO scheduler starts job 2
    on one rank while another still owes job 1's all-gather: collectives pair up across batches (a hang, or mixed buffers).
    All ranks return the goldens, bit-identical to each other."""
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, opaque)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, _, _, _, _ in res:
        for nm, (idx, w) in zip(["rbf_ragged", "cfg1_posterior_1e4", "rbf_1e4"], out):
            fx = load_golden(nm)
            assert idx == fx["idx"], f"rank {rank} {nm}"
            gw = torch.tensor(fx["w"], dtype=torch.float64)
            assert ((torch.tensor(w, dtype=torch.float64) - gw).abs() / gw).max().item() <= 1e-6
    assert all(r[1] == res[0][1] for r in res)


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("in_flight", [2, 3])
def test_recombination_many_gpu_bitwise_equals_sequential(in_flight):
    import basq_amd

    dev = torch.device("cuda", 0)
    calls, seq = [], []
    for k, (N, d, m, n, pool_seed, kern) in enumerate(SPECS + [(200_000, 10, 2_000, 100, 21, StationaryKernel("rbf", 2.0, 1.0)),
                                                                (120_000, 6, 600, 50, 22, StationaryKernel("matern32", 3.0, 2.0))]):
        pts = gmm_pool(N, d, pool_seed).to(dev)
        calls.append((pts, pts[:m].contiguous(), n, kern))
    seeds = list(range(40, 40 + len(calls)))
    for (pts, nys, n, kern), sd in zip(calls, seeds):
        torch.manual_seed(sd)
        seq.append(basq_amd.recombination(pts, nys, n, kern, dev))
    for _ in range(2):                                                        # twice: the slots' buffers are reused
        many = basq_amd.recombination_many(calls, dev, in_flight=in_flight, seeds=seeds)
        for (ia, wa), (ib, wb) in zip(seq, many):
            assert torch.equal(ia, ib) and torch.equal(wa, wb)


@pytest.mark.gpu
def test_recombination_many_gpu_goldens_structured_pair():
    """Selection + quadrature of one BASQ iteration (posterior-corrected kernel) in flight together, vs the goldens."""
    import basq_amd

    dev = torch.device("cuda", 0)
    names = ["cfg1_posterior_1e4", "wsabil_2e4", "cfg2_rbf_1e5", "posterior_noise_ragged"]
    calls, seeds = [], []
    for nm in names:
        c = BY_NAME[nm]
        pts, nys = build_pool(c)
        calls.append((pts.to(dev), nys.to(dev), c["n"], build_product_kernel(c)))
        seeds.append(c["torch_seed"])
    res = basq_amd.recombination_many(calls, dev, in_flight=2, seeds=seeds)
    for nm, (idx, w) in zip(names, res):
        fx = load_golden(nm)
        assert idx.cpu().tolist() == fx["idx"], nm
        gw = torch.tensor(fx["w"], dtype=torch.float64)
        assert ((w.cpu() - gw).abs() / gw).max().item() <= 1e-6


@pytest.mark.gpu
def test_recombination_many_gpu_mixed_kinds():
    """Batches of different kinds in flight together -- an opaque callable in the reference's block-by-block mode, WSABI-M
    (round-by-round loop with its squared-covariance block sums), a SOBER-free structured posterior on the descriptor path --
    each reproduces its golden: nothing is shared between batches in flight but the GPU."""
    import basq_amd
    from tests.cases import build_oracle_kernel

    dev = torch.device("cuda", 0)
    names = ["posterior_noise_ragged", "wsabim_1e4", "matern52_posterior", "rbf_ragged"]
    calls, seeds = [], []
    for i, nm in enumerate(names):
        c = BY_NAME[nm]
        pts, nys = build_pool(c)
        if i == 0:                                                # a bare lambda over the oracle's predictive_covariance
            ko, _ = build_oracle_kernel(c)
            for obj in (ko, getattr(ko, "post", None)):
                for attr in ("Xobs", "W", "mean_cache"):
                    t = getattr(obj, attr, None) if obj is not None else None
                    if torch.is_tensor(t):
                        setattr(obj, attr, t.to(dev))
            kern = lambda x, y, ko=ko: ko(x, y)                   # noqa: E731
        else:
            kern = build_product_kernel(c)
        calls.append((pts.to(dev), nys.to(dev), c["n"], kern))
        seeds.append(c["torch_seed"])
    for in_flight in (2, 4):
        res = basq_amd.recombination_many(calls, dev, in_flight=in_flight, seeds=seeds)
        for nm, (idx, w) in zip(names, res):
            fx = load_golden(nm)
            assert idx.cpu().tolist() == fx["idx"], (nm, in_flight)
            gw = torch.tensor(fx["w"], dtype=torch.float64)
            assert ((w.cpu() - gw).abs() / gw).max().item() <= 1e-6
