"""Golden vectors for f3/f4: runs the REFERENCE's ``SquareRootAcquisitionFunction`` (``BASQ/_acquisition_function.py``)
and ``GMM`` (``BASQ/experiment/gmm.py``) on a stub model.  ``python -m oracle.make_golden_acquisition``"""
import json
import os
import sys

import torch
from torch.distributions.multivariate_normal import MultivariateNormal

from oracle.make_golden_gaussian_calc import case_inputs, stub_model

CASE = dict(name="acq_d4", n_obs=80, d=4, lengthscale=1.3, outputscale=1.4, seed=31, n_x=500)


def prior_of(d):
    return MultivariateNormal(torch.zeros(d, dtype=torch.float64), 4.0 * torch.eye(d, dtype=torch.float64))


def query_points(c):
    from basq_amd.pools import gmm_pool

    return gmm_pool(c["n_x"], c["d"], c["seed"] + 1)


def main():
    import warnings

    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from BASQ._acquisition_function import SquareRootAcquisitionFunction as RefAcq
    from BASQ.experiment.gmm import GMM as RefGMM

    torch.set_default_dtype(torch.float64)
    c = CASE
    Xobs, mc, S = case_inputs(c)
    model = stub_model(Xobs, mc, S, c["lengthscale"], c["outputscale"])
    x = query_points(c)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        acq = RefAcq(prior_of(c["d"]), model, torch.device("cpu"), n_gaussians=40)
        jp = acq.joint_pdf(x)
        jm = acq.joint_pdf_mean(x)
    torch.manual_seed(9)
    gmm = RefGMM(c["d"], torch.zeros(c["d"]), 4.0 * torch.eye(c["d"]), torch.device("cpu"))
    lik = gmm(x)
    out = dict(case=c, joint_pdf=[float(v) for v in jp], joint_pdf_mean=[float(v) for v in jm], n_AA=int(acq.d_AA),
               n_mean=int(acq.d_mean), gmm_seed=9, gmm_n_comp=int(gmm.n_comp), gmm_lik=[float(v) for v in lik])
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "acquisition.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path, out["n_AA"], out["n_mean"], out["gmm_n_comp"])


if __name__ == "__main__":
    main()
