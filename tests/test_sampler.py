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
