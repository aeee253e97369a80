"""f3/f4: sparse-GMM acquisition densities and the demo GMM likelihood as kernel mat-vecs, against
goldens produced by the reference's own classes (oracle/make_golden_acquisition.py)."""
import json
import os

import pytest
import torch

from oracle.make_golden_acquisition import CASE, prior_of, query_points
from oracle.make_golden_gaussian_calc import case_inputs, stub_model

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "acquisition.json")


def _fx():
    with open(GOLD) as f:
        return json.load(f)


def _check(ops, dev):
    from basq_amd._acquisition_function import SquareRootAcquisitionFunction
    from basq_amd.experiment import GMM

    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        fx, c = _fx(), CASE
        Xobs, mc, S = case_inputs(c)
        model = stub_model(Xobs, mc, S, c["lengthscale"], c["outputscale"])
        x = query_points(c)
        acq = SquareRootAcquisitionFunction(prior_of(c["d"]), model, dev, n_gaussians=40, ops=ops)
        assert acq.d_AA == fx["n_AA"] and acq.d_mean == fx["n_mean"]
        jp = acq.joint_pdf(x).cpu()
        jm = acq.joint_pdf_mean(x).cpu()
        g1, g2 = torch.tensor(fx["joint_pdf"]), torch.tensor(fx["joint_pdf_mean"])
        assert ((jp - g1).abs() / g1.abs().max()).max().item() <= 1e-10
        assert ((jm - g2).abs() / g2.abs().max()).max().item() <= 1e-10
        torch.manual_seed(fx["gmm_seed"])
        gmm = GMM(c["d"], torch.zeros(c["d"]), 4.0 * torch.eye(c["d"]), dev, ops=ops)
        assert gmm.n_comp == fx["gmm_n_comp"]
        lik = gmm(x).cpu()
        g3 = torch.tensor(fx["gmm_lik"])
        assert ((lik - g3).abs() / g3.abs().max()).max().item() <= 1e-10
    finally:
        torch.set_default_dtype(prev)


def test_acquisition_and_gmm_host_logic():
    from tests.cpu_stand_in import CpuStandInOps

    _check(CpuStandInOps(), "cpu")


@pytest.mark.gpu
def test_acquisition_and_gmm_gpu(hip_ops):
    _check(hip_ops, "cuda:0")


def test_prior_sampler_split():
    from torch.distributions.multivariate_normal import MultivariateNormal

    from basq_amd._sampler import PriorSampler

    prior = MultivariateNormal(torch.zeros(3), torch.eye(3))
    nys, rec, w = PriorSampler(prior, 1000, 1e-2, "cpu")(1000)
    assert rec.shape == (1000, 3) and nys.shape == (10, 3) and torch.equal(nys, rec[:10])
    assert torch.allclose(w, torch.full((1000,), 1e-3))
