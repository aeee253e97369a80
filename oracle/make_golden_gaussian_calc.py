"""Golden vectors for the GaussianCalc path: runs the REFERENCE (``/root/reference/BASQ/_gaussian_calc.py``)
against a stub model exposing exactly the attributes it reads.  ``python -m oracle.make_golden_gaussian_calc``"""
import json
import os
import sys
from types import SimpleNamespace

import torch


def stub_model(Xobs, mean_cache, S, lengthscale, outputscale):
    return SimpleNamespace(
        train_inputs=(Xobs,),
        prediction_strategy=SimpleNamespace(mean_cache=mean_cache, covar_cache=S),
        covar_module=SimpleNamespace(outputscale=torch.tensor(outputscale, dtype=torch.float64),
                                     base_kernel=SimpleNamespace(lengthscale=torch.tensor([[lengthscale]], dtype=torch.float64))),
    )


def case_inputs(c):
    from basq_amd.pools import gmm_pool
    from oracle.kernels_oracle import StationaryOracle, synthetic_gp_state

    Xobs = gmm_pool(c["n_obs"], c["d"], c["seed"])
    base = StationaryOracle("rbf", c["lengthscale"], c["outputscale"])
    W, const, mean_cache, _ = synthetic_gp_state(Xobs, base, 1e-6, c["seed"])
    S = torch.linalg.cholesky(W)             # S S^T = woodbury_inv
    return Xobs, mean_cache, S


CASES = [dict(name="gc_small", n_obs=60, d=3, lengthscale=1.5, outputscale=1.2, alpha=0.3, seed=21),
         dict(name="gc_d10", n_obs=302, d=10, lengthscale=2.0, outputscale=1.0, alpha=0.8, seed=22)]


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from BASQ._gaussian_calc import GaussianCalc as RefGC

    torch.set_default_dtype(torch.float64)
    out = []
    for c in CASES:
        Xobs, mean_cache, S = case_inputs(c)
        model = stub_model(Xobs, mean_cache, S, c["lengthscale"], c["outputscale"])
        mvn = RefGC(None, torch.device("cpu")).unimodal_approximation(model, torch.tensor(c["alpha"]))
        out.append(dict(case=c, mean=[float(v) for v in mvn.loc], cov=[[float(v) for v in r] for r in mvn.covariance_matrix]))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gaussian_calc.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()
