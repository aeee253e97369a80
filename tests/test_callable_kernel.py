"""The reference's own ``kernel`` contract: ANY callable ``(X[a,d], Y[b,d]) -> Tensor[a,b]`` (``BASQ/_rchq.py:8,16``;
tutorial 02 "BayesQuad with arbitrary kernel"; BASELINE config 4 "arbitrary-kernel path").

A bare callable is wrapped in ``kernels.CallableKernel`` and runs through the chunked dense path
(``basq_dense_blocksum_f64``).  The oracle's kernel classes, wrapped as plain lambdas, must hit the SAME golden
vectors as the structured (fused) path: indices bit-exact, weights <= 1e-6 relative.

CPU part: host logic on the stand-in ops (chunk offsets, shard offsets, block-exact calls, gloo world 2).
GPU part (``-m gpu``): the HIP kernel through the C ABI.
"""
import os
import socket

import pytest
import torch

from basq_amd._engine import EngineTrace, RecombinationEngine
from basq_amd.kernels import CallableKernel
from tests.cases import BY_NAME, build_oracle_kernel, build_pool, load_golden
from tests.cpu_stand_in import CpuStandInOps

W_RTOL = 1e-6


def _check(c, fx, idx, w, tr=None):
    assert idx.tolist() == fx["idx"], "selected indices differ from the reference"
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    if len(gw):
        assert ((w.cpu() - gw).abs() / gw).max().item() <= W_RTOL
    if tr is not None:
        assert [r["kept"] for r in tr.rounds] == [r["kept"] for r in fx["rounds"]]


def _to_device(kern, dev):
    """Move an oracle kernel's tensors (observations, Woodbury matrix, mean cache) to ``dev``."""
    for obj in (kern, getattr(kern, "post", None)):
        if obj is None:
            continue
        for name in ("Xobs", "W", "mean_cache"):
            t = getattr(obj, name, None)
            if torch.is_tensor(t):
                setattr(obj, name, t.to(dev))
    return kern


# name, block_exact, chunk_bytes (small chunks: several per round, starting mid-block)
CPU_CASES = [
    ("rbf_ragged", False, 123 * 8 * 1000),          # 1000 candidates per chunk: not a multiple of S = 74
    ("rbf_ragged", False, 256 << 20),               # one chunk per round
    ("matern32_8e3", False, 80 * 8 * 333),
    ("rbf_tiny_final", False, 256 << 20),           # n < N <= 2n: single reduction of the points
    ("rbf_direct_car", False, 256 << 20),
    ("rbf_all_kept", False, 256 << 20),
    ("cfg1_posterior_1e4", True, 256 << 20),        # predictive_covariance: per-block noise diagonal -> exact calls
    ("posterior_noise_ragged", True, 256 << 20),    # noise 1e-3: the per-block diagonal decides the selection
    ("wsabil_noise_ragged", True, 256 << 20),
    ("wsabim_noise_ragged", True, 256 << 20),       # WSABI-M as an opaque callable: no structure needed at all
    # block_exact=None (what a BARE callable gets): the probe must find the block dependence by itself ...
    ("posterior_noise_ragged", None, 256 << 20),
    ("cfg1_posterior_1e4", None, 256 << 20),        # ... even at the reference's default noise of 1e-10
    ("wsabim_noise_ragged", None, 256 << 20),
    ("rbf_ragged", None, 123 * 8 * 1000),           # ... and must let a plain kernel take the chunked mode
]


@pytest.mark.parametrize("name,block_exact,chunk_bytes", CPU_CASES)
def test_opaque_callable_host_logic_reproduces_golden(name, block_exact, chunk_bytes):
    c, fx = BY_NAME[name], load_golden(name)
    pts, nys = build_pool(c)
    ko, _ = build_oracle_kernel(c)
    kern = CallableKernel(lambda x, y: ko(x, y), block_exact=block_exact, chunk_bytes=chunk_bytes)
    tr = EngineTrace()
    torch.manual_seed(c["torch_seed"])
    ops = CpuStandInOps()
    idx, w = RecombinationEngine(ops).run(pts, 0, c["N"], nys, c["n"], kern, tr)
    _check(c, fx, idx, w, tr)
    if c["N"] > 2 * c["n"]:
        assert ops.calls.get("dense", 0) > 0 and ops.calls.get("blocksum", 0) == 0      # the dense path did the work
    if block_exact is None:
        post = c["kernel"]["posterior"] is not None
        assert kern.resolve_mode(ops, nys, 2 * min(c["n"], c["m"] + 1)) == post          # exact iff the kernel needs it


def test_probe_decisions():
    """``CallableKernel.resolve_mode``: exact calls unless the callable provably evaluates column by column."""
    ops = CpuStandInOps()
    g = torch.Generator().manual_seed(0)
    X = torch.randn(40, 3, generator=g, dtype=torch.float64)
    rbf = lambda x, y: torch.exp(-0.5 * torch.cdist(x, y) ** 2)                        # noqa: E731

    def noisy(noise):
        def fn(x, y):
            K = rbf(x, y)
            k = min(len(x), len(y))
            K[range(k), range(k)] += noise                                             # BASQ/_gp.py:275-276
            return K
        return fn

    assert CallableKernel(rbf).resolve_mode(ops, X, 16) is False
    assert CallableKernel(noisy(1e-3)).resolve_mode(ops, X, 16) is True
    assert CallableKernel(noisy(1e-10)).resolve_mode(ops, X, 16) is True               # the reference's default lik_var
    assert CallableKernel(noisy(0.0)).resolve_mode(ops, X, 16) is False
    assert CallableKernel(lambda x, y: rbf(x, y) * float("nan")).resolve_mode(ops, X, 16) is True   # NaNs: stay exact
    assert CallableKernel(lambda x, y: rbf(x, y) / len(y)).resolve_mode(ops, X, 16) is True         # depends on the block size
    assert CallableKernel(noisy(1e-3), block_exact=False).resolve_mode(ops, X, 16) is False         # explicit opt-in wins
    assert CallableKernel(rbf, block_exact=True).resolve_mode(ops, X, 16) is True


def test_callable_is_called_like_the_reference_in_block_exact_mode():
    """``block_exact``: one call per block of 2n candidates + one for the ragged tail, first operand always pts_nys."""
    c = BY_NAME["rbf_ragged"]
    pts, nys = build_pool(c)
    ko, _ = build_oracle_kernel(c)
    shapes = []

    def fn(x, y):
        shapes.append((tuple(x.shape), tuple(y.shape)))
        return ko(x, y)

    torch.manual_seed(c["torch_seed"])
    RecombinationEngine(CpuStandInOps()).run(pts, 0, c["N"], nys, c["n"], CallableKernel(fn, block_exact=True))
    S = 2 * c["n"]
    assert shapes[0] == ((c["m"], c["d"]), (c["m"], c["d"]))                 # the Nystrom Gram (:29)
    r1 = shapes[1:1 + c["N"] // S + 1]
    assert all(s == ((c["m"], c["d"]), (S, c["d"])) for s in r1[:-1])        # :81-86
    assert r1[-1] == ((c["m"], c["d"]), (c["N"] % S, c["d"]))                # :91-99 the remainder


def test_bare_callable_is_wrapped_and_rejects_bad_returns():
    from basq_amd._rchq import _as_kernel_object

    k = _as_kernel_object(lambda x, y: x @ y.T)
    assert isinstance(k, CallableKernel) and k.block_exact is None         # mode decided by the probe, per batch
    with pytest.raises(TypeError):
        _as_kernel_object(3.0)
    bad = CallableKernel(lambda x, y: (x @ y.T)[:, :-1])
    with pytest.raises(ValueError):
        bad.dense(CpuStandInOps(), torch.zeros(3, 2, dtype=torch.float64), torch.zeros(4, 2, dtype=torch.float64))


# ---- multi-rank (gloo, world 2): chunk offsets are GLOBAL positions --------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, q, block_exact=False):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from basq_amd._engine import TorchDistComm
        from basq_amd._partition import initial_shards

        c = BY_NAME[name]
        pts, nys = build_pool(c)
        ko, _ = build_oracle_kernel(c)
        off, n = initial_shards(c["N"], world)[rank]
        torch.manual_seed(c["torch_seed"])
        idx, w = RecombinationEngine(CpuStandInOps(), TorchDistComm()).run(
            pts[off:off + n].clone(), off, c["N"], nys, c["n"],
            CallableKernel(lambda x, y: ko(x, y), block_exact=block_exact, chunk_bytes=123 * 8 * 700))
        q.put((rank, idx.tolist(), w.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,block_exact", [
    ("rbf_ragged", 2, False),
    ("posterior_noise_ragged", 2, None),      # a bare predictive_covariance-style callable on two ranks: blocks that straddle
    ("posterior_noise_ragged", 3, True),      # the shard border are evaluated whole by the owner of their first point
    ("wsabim_noise_ragged", 4, None),
    ("rbf_tiny_final", 2, True),              # single reduction of the points, shards shorter than a block
])
def test_opaque_callable_sharded_matches_golden(name, world, block_exact):
    import torch.multiprocessing as mp

    fx = load_golden(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q, block_exact)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, idx, w in res:
        _check(BY_NAME[name], fx, torch.tensor(idx), torch.tensor(w, dtype=torch.float64))


# ---- GPU: the HIP path -------------------------------------------------------------------------------------------------
GPU_CASES = [
    ("rbf_ragged", False, 123 * 8 * 1000),
    ("matern52_3e4_d32", False, 64 << 20),          # tutorial 02 / BASELINE config 4 family: Matern-5/2 as a callable
    ("matern32_8e3", False, 256 << 20),
    ("rbf_tiny_final", False, 256 << 20),
    ("cfg1_posterior_1e4", True, 256 << 20),
    ("cfg1_posterior_1e4", False, 256 << 20),       # chunked: the 1e-10 noise diagonal lands elsewhere, same selection
    ("posterior_noise_ragged", True, 256 << 20),
    ("wsabim_noise_ragged", True, 256 << 20),
    ("cfg2_rbf_1e5", False, 256 << 20),
    ("posterior_noise_ragged", None, 256 << 20),    # a BARE lambda, no keyword: the probe selects the reference's calls
    ("wsabil_noise_ragged", None, 256 << 20),
    ("matern32_8e3", None, 256 << 20),              # a bare lambda over a plain kernel: whatever the probe picks, same batch
    # BASELINE config 4 through its named door, at full size: Matern-5/2, N=1e6, d=32, n=200 as a BARE lambda
    # (_rchq.py:81-86 with an arbitrary `kernel`; tutorial 02).  ~1 s of GPU time.
    ("cfg4_matern52_1e6_d32", None, 256 << 20),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,block_exact,chunk_bytes", GPU_CASES)
def test_opaque_callable_gpu_reproduces_golden(name, block_exact, chunk_bytes):
    import basq_amd

    dev = torch.device("cuda", 0)
    c, fx = BY_NAME[name], load_golden(name)
    pts, nys = build_pool(c)
    ko, _ = build_oracle_kernel(c)
    ko = _to_device(ko, dev)
    tr = basq_amd.EngineTrace()
    torch.manual_seed(c["torch_seed"])
    if block_exact is not None:
        kern = CallableKernel(lambda x, y: ko(x, y), block_exact=block_exact, chunk_bytes=chunk_bytes)
    else:
        kern = lambda x, y: ko(x, y)                # noqa: E731  a bare lambda, exactly what the reference accepts
    idx, w = basq_amd.recombination(pts, nys, c["n"], kern, dev, trace=tr)
    _check(c, fx, idx.cpu(), w, tr)


@pytest.mark.gpu
def test_opaque_callable_equals_fused_path_bitwise_on_indices():
    """Same pool, same seed: a structured kernel called as an opaque callable selects the fused path's points."""
    import basq_amd

    dev = torch.device("cuda", 0)
    pts = basq_amd.pools.gmm_pool(50_000, 6, 77)
    nys = pts[:500]
    sk = basq_amd.kernels.StationaryKernel("matern52", 3.0, 1.7)
    torch.manual_seed(5)
    i1, w1 = basq_amd.recombination(pts, nys, 64, sk, dev)
    torch.manual_seed(5)
    i2, w2 = basq_amd.recombination(pts, nys, 64, lambda x, y: sk(x, y), dev)
    assert i1.tolist() == i2.tolist()
    assert ((w1 - w2).abs() / w1).max().item() <= 1e-8
