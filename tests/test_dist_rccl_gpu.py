"""The RCCL side of the multi-GPU path on the one GPU a test box has: a single-rank ``nccl`` group (RCCL on ROCm)
running the communicator's collectives on float64 device tensors and the sharded entry end to end.  The multi-rank
logic itself (offsets, ordered sums, broadcast result) is covered on the CPU with gloo (tests/test_dist_gloo.py)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, os.environ["BASQ_REPO"])
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    import basq_amd
    from basq_amd._engine import TorchDistComm
    from tests.cases import BY_NAME, build_pool, build_product_kernel, load_golden
    comm = TorchDistComm()
    assert (comm.rank, comm.world) == (0, 1)
    t = torch.arange(12, dtype=torch.float64, device=dev).reshape(3, 4) / 7
    g = comm.all_gather(t)
    assert g.shape == (1, 3, 4) and torch.equal(g[0], t)
    b = comm.broadcast(t.clone())
    assert torch.equal(b, t)
    c, fx = BY_NAME["rbf_ragged"], load_golden("rbf_ragged")
    pts, nys = build_pool(c)
    torch.manual_seed(c["torch_seed"])
    idx, w = basq_amd.recombination_sharded(pts, 0, c["N"], nys, c["n"], build_product_kernel(c), dev)
    assert idx.cpu().tolist() == fx["idx"]
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL-OK")
""")


def test_single_rank_rccl_group_runs_the_sharded_entry():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               BASQ_REPO=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
