"""Two REAL-KERNEL ranks on the one GPU a test box has (review r04, item 4).

No multi-GPU node has ever run this code: the sharded path was covered by ``gloo`` worlds on the CPU stand-in only.  Here two
fresh child processes share ``cuda:0`` and run the production path -- ``HipOps`` on HIP streams, shard-aware ``*_geo`` kernels,
descriptor-driven rounds, the sharded range finder, owner-rank reductions with per-slot groups -- exchanging their messages
through a test-only host-staged communicator (``tests/host_staged_comm.py``: RCCL refuses two ranks on one device).  Asserted
on BOTH ranks: golden indices, weights, per-round kept sets; identical generator states; the exchange counts of the design
(one all-gather per descriptor-driven round and no per-round broadcast with replicated reductions; broadcasts on every slot's group with
owner-rank reductions).  This is NOT a scaling measurement: both ranks share one device.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

from tests.cases import load_golden

pytestmark = pytest.mark.gpu

CASES = ["rbf_ragged", "cfg1_posterior_1e4", "cfg2_rbf_1e5", "wsabim_noise_ragged"]   # (the last: WSABI-M on two shards, r05)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_real_kernel_ranks_share_one_gpu(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port, world = _free_port(), 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs, outs = [], []
    for r in range(world):
        out = tmp_path / f"rank{r}.pt"
        outs.append(out)
        # fresh processes, started with Popen (never an exec from a process that has touched the GPU)
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "two_rank_child.py"), str(r), str(world),
                                       str(port), str(out)] + CASES, env=env, cwd=root, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0 and f"TWO-RANK-OK {r}" in logs[r], logs[r][-3000:]
    res = [torch.load(o, weights_only=False) for o in outs]
    for name in CASES:
        fx = load_golden(name)
        gw = torch.tensor(fx["w"], dtype=torch.float64)
        for r in range(world):
            got = res[r]["cases"][name]
            assert got["idx"] == fx["idx"], f"{name} rank {r}: indices differ from the reference"
            assert ((got["w"] - gw).abs() / gw).max().item() <= 1e-6, f"{name} rank {r}"
            assert len(got["kept"]) == fx["n_rounds"], f"{name} rank {r}: {len(got['kept'])} rounds"
            for mine, ref in zip(got["kept"], fx["rounds"]):
                assert mine == ref["kept"], f"{name} rank {r}: kept sets differ"
            # replicated reductions: ONE broadcast per batch (the range finder's Gaussian test matrix, drawn on rank 0), none per
            # round; at least one all-gather per round
            assert got["broadcasts"] == 1 and got["all_gathers"] >= fx["n_rounds"], (name, r, got["all_gathers"], got["broadcasts"])
        assert torch.equal(res[0]["cases"][name]["w"], res[1]["cases"][name]["w"]), f"{name}: weights differ BETWEEN the ranks"
        assert res[0]["cases"][name]["rng"] == res[1]["cases"][name]["rng"], f"{name}: generator states differ between the ranks"
    # four batches, two in flight, owner-rank reductions
    for r in range(world):
        for job in res[r]["many"]:
            fx = load_golden(job["name"])
            gw = torch.tensor(fx["w"], dtype=torch.float64)
            assert job["idx"] == fx["idx"], f"in flight, rank {r}, {job['name']}"
            assert ((job["w"] - gw).abs() / gw).max().item() <= 1e-6
        sb = res[r]["slot_broadcasts"]
        assert len(sb) == 2 and all(n > 0 for _, n in sb), f"rank {r}: owner broadcasts per slot group {sb}"
    for a, b in zip(res[0]["many"], res[1]["many"]):
        assert torch.equal(a["w"], b["w"])
    assert res[0]["many_rng"] == res[1]["many_rng"], "generator states differ between the ranks after the batches in flight"
