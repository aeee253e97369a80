"""Golden vectors for the uncertainty sampler (``BASQ/_sampler.py:37-280``).

``python -m oracle.make_golden_sampler`` runs the REFERENCE's own ``UncertaintySampler`` class (and, through it, its
``SquareRootAcquisitionFunction``) on a stub model.  ``BASQ/_sampler.py`` imports ``predict`` from ``BASQ/_gp.py``,
which needs gpytorch (not installed, no network): before the import, ``sys.modules['BASQ._gp']`` is given a module
whose only attribute is ``predict = oracle.kernels_oracle.predict_oracle`` -- the closed-form restatement of
``_gp.py:213-230`` (exact variance instead of gpytorch's LOVE approximation: stated there).  Everything else in the
fixture -- mixture construction, SIR, importance weights, the order and shape of every RNG draw -- is the
reference's code.  Parity of ``predict`` itself stays unpinned (no gpytorch here).
"""
import json
import os
import sys
import types
import warnings
from types import SimpleNamespace

import torch
from torch.distributions.multivariate_normal import MultivariateNormal

from oracle.make_golden_gaussian_calc import case_inputs

CASES = [
    dict(name="us_approx", method="approx", n_obs=80, d=4, lengthscale=1.3, outputscale=1.4, seed=41, n=400,
         nys_ratio=0.05, ratio=0.5, ratio_super=10, n_gaussians=40, torch_seed=5),
    dict(name="us_exact", method="exact", n_obs=80, d=4, lengthscale=1.3, outputscale=1.4, seed=41, n=400,
         nys_ratio=0.05, ratio=0.5, ratio_super=10, n_gaussians=40, torch_seed=6),
    dict(name="us_exact_r1", method="exact", n_obs=60, d=3, lengthscale=1.1, outputscale=1.0, seed=43, n=300,
         nys_ratio=0.1, ratio=1.0, ratio_super=8, n_gaussians=30, torch_seed=7),
]
NOISE = 1e-6


def prior_of(d):
    return MultivariateNormal(torch.zeros(d, dtype=torch.float64), 4.0 * torch.eye(d, dtype=torch.float64))


def sampler_model(c):
    """Stub of a fitted gpytorch ExactGP: the attributes ``_gp.py`` / ``_gaussian_calc.py`` read, nothing else."""
    from oracle.kernels_oracle import StationaryOracle, synthetic_gp_state

    Xobs, mean_cache, S = case_inputs(c)
    _, const, _, _ = synthetic_gp_state(Xobs, StationaryOracle("rbf", c["lengthscale"], c["outputscale"]), 1e-6, c["seed"])
    return SimpleNamespace(
        train_inputs=(Xobs,),
        prediction_strategy=SimpleNamespace(mean_cache=mean_cache, covar_cache=S),
        covar_module=SimpleNamespace(outputscale=torch.tensor(c["outputscale"], dtype=torch.float64),
                                     base_kernel=SimpleNamespace(
                                         lengthscale=torch.tensor([[c["lengthscale"]]], dtype=torch.float64))),
        mean_module=SimpleNamespace(constant=torch.tensor(const, dtype=torch.float64)),
        likelihood=SimpleNamespace(noise=torch.tensor([NOISE], dtype=torch.float64)),
        eval=lambda: None,
    )


def query_points(c):
    from basq_amd.pools import gmm_pool

    return gmm_pool(64, c["d"], c["seed"] + 1)


def main():
    from oracle.kernels_oracle import predict_oracle

    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    shim = types.ModuleType("BASQ._gp")
    shim.predict = predict_oracle
    sys.modules["BASQ._gp"] = shim
    from BASQ._sampler import UncertaintySampler as RefSampler

    torch.set_default_dtype(torch.float64)
    out = []
    for c in CASES:
        model = sampler_model(c)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            us = RefSampler(prior_of(c["d"]), model, c["n"], c["nys_ratio"], torch.device("cpu"),
                            sampling_method=c["method"], ratio=c["ratio"], ratio_super=c["ratio_super"],
                            n_gaussians=c["n_gaussians"])
            x = query_points(c)
            pdf = us.pdf(x)
            cw = us.calc_weights(x)
            torch.manual_seed(c["torch_seed"])
            pts_nys, pts_rec, w = us(c["n"])
        out.append(dict(case=c, pdf=[float(v) for v in pdf], calc_weights=[float(v) for v in cw],
                        pts_nys=pts_nys.tolist(), pts_rec=pts_rec.tolist(), w=[float(v) for v in w],
                        n_AA=int(us.d_AA), n_mean=int(us.d_mean)))
        print(c["name"], tuple(pts_nys.shape), tuple(pts_rec.shape), float(w.sum()))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sampler.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
