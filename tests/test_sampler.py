"""``UncertaintySampler`` (``BASQ/_sampler.py:37-280``) against goldens produced by the reference's own class
(oracle/make_golden_sampler.py: the reference's sampler + acquisition code; only ``_gp.predict``, which needs gpytorch,
is the oracle's closed form).  Same seeds -> the same pools: every RNG draw is made where the reference makes it."""
import json
import os

import pytest
import torch

from oracle.make_golden_sampler import CASES, prior_of, query_points, sampler_model

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler.json")


def _fx():
    with open(GOLD) as f:
        return {r["case"]["name"]: r for r in json.load(f)}


def _rel(a, b):
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a.cpu().to(torch.float64) - b).abs().max() / b.abs().max()).item()


def _check(name, ops, dev):
    from basq_amd._sampler import UncertaintySampler

    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        fx = _fx()[name]
        c = next(cc for cc in CASES if cc["name"] == name)
        assert fx["case"] == c, "fixture was generated for different inputs: rerun oracle.make_golden_sampler"
        us = UncertaintySampler(prior_of(c["d"]), sampler_model(c), c["n"], c["nys_ratio"], dev, sampling_method=c["method"],
                                ratio=c["ratio"], ratio_super=c["ratio_super"], n_gaussians=c["n_gaussians"], ops=ops)
        assert us.d_AA == fx["n_AA"] and us.d_mean == fx["n_mean"]
        x = query_points(c)
        assert _rel(us.pdf(x), fx["pdf"]) <= 1e-9
        assert _rel(us.calc_weights(x), fx["calc_weights"]) <= 1e-8
        torch.manual_seed(c["torch_seed"])
        pts_nys, pts_rec, w = us(c["n"])
        g_rec = torch.tensor(fx["pts_rec"], dtype=torch.float64)
        assert tuple(pts_rec.shape) == tuple(g_rec.shape) and tuple(pts_nys.shape) == (len(fx["pts_nys"]), c["d"])
        assert _rel(pts_rec, g_rec) <= 1e-9, "pool differs: an RNG draw is out of order or a count changed"
        assert _rel(pts_nys, fx["pts_nys"]) <= 1e-9, "SIR picked other points"
        assert _rel(w, fx["w"]) <= 1e-7
        assert abs(float(w.sum()) - 1.0) <= 1e-12
    finally:
        torch.set_default_dtype(prev)


@pytest.mark.parametrize("name", [c["name"] for c in CASES])
def test_uncertainty_sampler_host_logic(name):
    from tests.cpu_stand_in import CpuStandInOps

    _check(name, CpuStandInOps(), "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", [c["name"] for c in CASES])
def test_uncertainty_sampler_gpu(hip_ops, name):
    _check(name, hip_ops, "cuda:0")


def test_predict_oracle_is_the_gp_posterior():
    """The shimmed ``predict`` is the exact GP posterior: interpolates the targets, variance ~ noise at the data."""
    from oracle.kernels_oracle import StationaryOracle, predict_oracle, synthetic_gp_state

    c = CASES[0]
    model = sampler_model(c)
    Xobs = model.train_inputs[0]
    _, const, _, y = synthetic_gp_state(Xobs, StationaryOracle("rbf", c["lengthscale"], c["outputscale"]), 1e-6, c["seed"])
    mean, var = predict_oracle(Xobs, model)
    assert (mean - y).abs().max().item() <= 1e-4
    assert (var > 0).all() and var.max().item() <= 1e-4


def test_device_generation_statistics_host_logic():
    """``generator_parity=False``: pools / mixture draws come from ``mvn_draw`` on the target device (here the CPU as a
    stand-in): right shapes, first two moments of the prior."""
    from torch.distributions.multivariate_normal import MultivariateNormal

    from basq_amd._sampler import PriorSampler

    cov = torch.tensor([[2.0, 0.3], [0.3, 0.5]], dtype=torch.float64)
    prior = MultivariateNormal(torch.tensor([1.0, -2.0], dtype=torch.float64), cov)
    torch.manual_seed(0)
    nys, rec, w = PriorSampler(prior, 40_000, 1e-2, "cpu", generator_parity=False)(40_000)
    assert rec.shape == (40_000, 2) and nys.shape == (400, 2) and torch.equal(nys, rec[:400])
    assert (rec.mean(0) - prior.loc).abs().max().item() < 0.03
    assert (torch.cov(rec.T) - cov).abs().max().item() < 0.05
    assert abs(float(w.sum()) - 1.0) < 1e-12


@pytest.mark.gpu
def test_device_generation_leaves_the_host_generator_alone(hip_ops):
    """SURVEY f4's purpose: with ``generator_parity=False`` the pool is generated on the GPU -- the CPU generator is not
    consumed (no host RNG work) and the tensors are born on the device (no H2D copy); the uncertainty sampler runs
    end to end the same way and returns a normalised weight vector."""
    from torch.distributions.multivariate_normal import MultivariateNormal

    from basq_amd._sampler import PriorSampler, UncertaintySampler

    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        c = CASES[0]
        prior = prior_of(c["d"])
        torch.manual_seed(123)
        before = torch.get_rng_state()
        nys, rec, w = PriorSampler(prior, 100_000, 1e-2, "cuda:0", generator_parity=False)(100_000)
        assert rec.is_cuda and nys.is_cuda and rec.shape == (100_000, c["d"])
        assert (rec.mean(0).cpu() - prior.loc).abs().max().item() < 0.05
        us = UncertaintySampler(prior, sampler_model(c), c["n"], c["nys_ratio"], "cuda:0", sampling_method="exact",
                                ratio=0.5, ratio_super=20, n_gaussians=c["n_gaussians"], ops=hip_ops, generator_parity=False)
        pts_nys, pts_rec, wts = us(c["n"])
        assert pts_rec.is_cuda and abs(float(wts.sum()) - 1.0) < 1e-12 and bool((wts >= 0).all())
        assert torch.equal(torch.get_rng_state(), before), "the CPU generator was consumed"
        assert isinstance(prior, MultivariateNormal)
    finally:
        torch.set_default_dtype(prev)
