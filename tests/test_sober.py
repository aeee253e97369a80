"""SOBER-flavoured entry: oracle and product against goldens produced by SOBER/_rchq.py itself."""
import json
import os

import pytest
import torch

from oracle.make_golden_sober import CASES, case_weights

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
        idx, w = recombination_sober_oracle(pts, nys, c["n"], StationaryOracle(c["family"], c["lengthscale"], 1.0), w0)
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
    idx, w = RecombinationEngine(CpuStandInOps()).run(pts, 0, c["N"], nys, c["n"],
                                                       StationaryKernel(c["family"], c["lengthscale"], 1.0),
                                                       variant="sober", init_weights=w0)
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
                                 torch.device("cuda:0"), torch.float64, init_weights=w0)
    assert idx.cpu().tolist() == fx["idx"]
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    assert ((w.cpu() - gw).abs() / gw).max().item() <= 1e-6
    assert abs(w.sum().item() - 1.0) < 1e-9


def test_calc_obj_is_refused():
    from basq_amd import sober
    from basq_amd.kernels import StationaryKernel

    with pytest.raises(NotImplementedError):
        sober.recombination(torch.zeros(10, 2), torch.zeros(5, 2), 3, StationaryKernel("rbf", 1.0), "cuda", calc_obj=lambda x: x)
