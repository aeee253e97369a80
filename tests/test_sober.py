"""SOBER-flavoured entry: oracle and product against goldens produced by SOBER/_rchq.py itself."""
import json
import os

import pytest
import torch

from oracle.make_golden_sober import CASES, case_objective, case_weights

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sober.json")


def _fx():
    with open(GOLD) as f:
        return json.load(f)


def _inputs(c):
    from basq_amd.pools import gmm_pool

    pts = gmm_pool(c["N"], c["d"], c["pool_seed"])
    return pts, pts[: c["m"]], case_weights(c)


@pytest.mark.parametrize("i", range(len(CASES)))
def test_sober_oracle_matches_golden(i):
    from oracle.kernels_oracle import StationaryOracle
    from oracle.rchq_oracle import recombination_sober_oracle

    c, fx = CASES[i], _fx()[i]
    pts, nys, w0 = _inputs(c)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(1)
        idx, w = recombination_sober_oracle(pts, nys, c["n"], StationaryOracle(c["family"], c["lengthscale"], 1.0), w0,
                                            calc_obj=case_objective(c))
    finally:
        torch.set_default_dtype(prev)
    assert idx.tolist() == fx["idx"]
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    assert ((w - gw).abs() / gw).max().item() <= 1e-9


@pytest.mark.parametrize("i", range(len(CASES)))
def test_sober_engine_host_logic_matches_golden(i):
    from basq_amd._engine import RecombinationEngine
    from basq_amd.kernels import StationaryKernel
    from tests.cpu_stand_in import CpuStandInOps

    c, fx = CASES[i], _fx()[i]
    pts, nys, w0 = _inputs(c)
    torch.manual_seed(1)
    f = case_objective(c)
    idx, w = RecombinationEngine(CpuStandInOps()).run(pts, 0, c["N"], nys, c["n"],
                                                       StationaryKernel(c["family"], c["lengthscale"], 1.0),
                                                       variant="sober", init_weights=w0,
                                                       objective=None if f is None else -1 * f(pts))
    assert idx.tolist() == fx["idx"]
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    assert ((w - gw).abs() / gw).max().item() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(CASES)))
def test_sober_gpu_matches_golden(i):
    from basq_amd import sober
    from basq_amd.kernels import StationaryKernel

    c, fx = CASES[i], _fx()[i]
    pts, nys, w0 = _inputs(c)
    torch.manual_seed(1)
    idx, w = sober.recombination(pts, nys, c["n"], StationaryKernel(c["family"], c["lengthscale"], 1.0),
                                 torch.device("cuda:0"), torch.float64, init_weights=w0, calc_obj=case_objective(c))
    assert idx.cpu().tolist() == fx["idx"]
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    assert ((w.cpu() - gw).abs() / gw).max().item() <= 1e-6
    assert abs(w.sum().item() - 1.0) < 1e-9


def test_reference_objective_branch_fails_for_large_pools():
    """With ``calc_obj`` the reference only works when the pool fits one reduction (N <= 2 num_pts): beyond that its own
    objective sums have mismatched shapes (``SOBER/_rchq.py:140-142``) and it raises.  The oracle restates that
    faithfully (it raises the same error), and the engine refuses the same inputs with a message that says so."""
    from basq_amd._engine import RecombinationEngine
    from basq_amd.kernels import StationaryKernel
    from basq_amd.pools import gmm_pool
    from oracle.kernels_oracle import StationaryOracle
    from oracle.rchq_oracle import recombination_sober_oracle
    from tests.cpu_stand_in import CpuStandInOps

    pts = gmm_pool(500, 3, 1)
    f = case_objective(dict(objective="bump"))
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(1)
        with pytest.raises(RuntimeError, match="broadcast shape"):
            recombination_sober_oracle(pts, pts[:40], 20, StationaryOracle("rbf", 2.0, 1.0), None, calc_obj=f)
        torch.manual_seed(1)
        with pytest.raises(RuntimeError, match="reference fails here too"):
            RecombinationEngine(CpuStandInOps()).run(pts, 0, 500, pts[:40], 20, StationaryKernel("rbf", 2.0, 1.0),
                                                     variant="sober", objective=-1 * f(pts))
    finally:
        torch.set_default_dtype(prev)


def test_objective_needs_the_sober_variant():
    from basq_amd._engine import RecombinationEngine
    from basq_amd.kernels import StationaryKernel
    from tests.cpu_stand_in import CpuStandInOps

    pts = torch.zeros(10, 2, dtype=torch.float64)
    with pytest.raises(ValueError):
        RecombinationEngine(CpuStandInOps()).run(pts, 0, 10, pts[:5], 3, StationaryKernel("rbf", 1.0), objective=torch.zeros(10))
