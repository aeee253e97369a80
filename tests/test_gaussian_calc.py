"""GaussianCalc (BASQ/_gaussian_calc.py): oracle and product against the reference's
golden vectors (tests/golden/gaussian_calc.json, produced by oracle/make_golden_gaussian_calc.py)."""
import json
import os

import pytest
import torch

from oracle.make_golden_gaussian_calc import CASES, case_inputs, stub_model

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gaussian_calc.json")


def _fixtures():
    with open(GOLD) as f:
        return json.load(f)


@pytest.fixture(autouse=True)
def _f64():
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(prev)


@pytest.mark.parametrize("i", range(len(CASES)))
def test_oracle_matches_reference_golden(i):
    from oracle.gaussian_calc_oracle import unimodal_approximation_oracle

    c, fx = CASES[i], _fixtures()[i]
    Xobs, mc, _ = case_inputs(c)
    mu, cov = unimodal_approximation_oracle(Xobs, mc, c["lengthscale"], c["outputscale"], c["alpha"])
    assert (mu - torch.tensor(fx["mean"])).abs().max().item() <= 1e-13
    assert (cov - torch.tensor(fx["cov"])).abs().max().item() <= 1e-12


@pytest.mark.parametrize("i", range(len(CASES)))
def test_product_host_logic_matches_golden(i):
    """The mat-vec formulation (no n_obs^2 storage) on the CPU stand-in."""
    from basq_amd._gaussian_calc import GaussianCalc
    from tests.cpu_stand_in import CpuStandInOps

    c, fx = CASES[i], _fixtures()[i]
    Xobs, mc, S = case_inputs(c)
    mvn = GaussianCalc(None, "cpu", ops=CpuStandInOps()).unimodal_approximation(
        stub_model(Xobs, mc, S, c["lengthscale"], c["outputscale"]), c["alpha"])
    gm, gc = torch.tensor(fx["mean"]), torch.tensor(fx["cov"])
    assert (mvn.loc - gm).abs().max().item() <= 1e-10
    assert (mvn.covariance_matrix - gc).abs().max().item() <= 1e-9 * gc.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(CASES)))
def test_product_gpu_matches_golden(i):
    from basq_amd._gaussian_calc import GaussianCalc

    c, fx = CASES[i], _fixtures()[i]
    Xobs, mc, S = case_inputs(c)
    mvn = GaussianCalc(None, "cuda:0").unimodal_approximation(
        stub_model(Xobs, mc, S, c["lengthscale"], c["outputscale"]), c["alpha"])
    gm, gc = torch.tensor(fx["mean"]), torch.tensor(fx["cov"])
    assert (mvn.loc.cpu() - gm).abs().max().item() <= 1e-10
    assert (mvn.covariance_matrix.cpu() - gc).abs().max().item() <= 1e-9 * gc.abs().max().item()


@pytest.mark.reference
@pytest.mark.skipif(not os.path.isdir("/root/reference/BASQ"), reason="reference not mounted")
def test_oracle_vs_imported_reference():
    import sys
    import warnings

    sys.dont_write_bytecode = True
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from BASQ._gaussian_calc import GaussianCalc as RefGC

    from oracle.gaussian_calc_oracle import unimodal_approximation_oracle

    c = CASES[0]
    Xobs, mc, S = case_inputs(c)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mvn = RefGC(None, torch.device("cpu")).unimodal_approximation(
            stub_model(Xobs, mc, S, c["lengthscale"], c["outputscale"]), torch.tensor(c["alpha"]))
    mu, cov = unimodal_approximation_oracle(Xobs, mc, c["lengthscale"], c["outputscale"], c["alpha"])
    assert torch.equal(mu, mvn.loc) and torch.equal(cov, mvn.covariance_matrix)
