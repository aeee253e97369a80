"""``bench.py`` on its multi-rank code path, on the one GPU a test box has (review r05, item 2c).

``BASQ_BENCH_FORCE_DIST=1 python bench.py --gpus 1`` with NO launcher in the environment: the parent starts its rank through
``torch.distributed.run`` before anything touches the GPU (the path ``python bench.py --gpus 8`` takes on an 8-GPU node), the rank
builds a real RCCL group, runs the sharded entry + the owner-rank batches in flight, and rank 0's JSON line comes back as the LAST
line of the parent's output with the parent's exit code 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launched_single_rank_rccl_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(BASQ_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", BASQ_BENCH_CONCURRENT_LIMIT_S="300")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)
    assert "torch.distributed.run" in r.stderr                   # the parent was the launcher
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0
    rccl = line["rccl"]
    assert rccl["world"] == 1 and rccl["backend"] == "nccl" and rccl["distinct_devices"] == 1
    par = line["parity_vs_golden"]
    assert par["pools_checked"] == 5 and par["indices_identical"] is True and par["max_rel_weight_error"] < 1e-5
    assert line["concurrent_timed_out"] is False
    assert all(c["bit_identical_to_sequential"] for c in line["concurrent"])
    # same schema as the plain one-GPU line
    for key in ("metric", "unit", "ms_per_step", "roofline", "cpu_baseline", "config", "scaling", "dtype", "configs", "value_incl_h2d"):
        assert key in line
