"""``basq_amd.pools``: the synthetic-input generators the bench and the tools use must be the ones the goldens were made with."""
import torch

from basq_amd import pools
from tests.cases import BY_NAME, build_obs, build_oracle_kernel


def test_synthetic_gp_state_is_bit_identical_to_the_goldens_generator():
    for name in ("cfg1_posterior_1e4", "wsabil_2e4", "matern52_posterior", "wsabim_noise_ragged", "cfg5_wsabil_5e5"):
        c = BY_NAME[name]
        k, p = c["kernel"], c["kernel"]["posterior"]
        _, state = build_oracle_kernel(c)
        W, mc, cache, _ = pools.synthetic_gp_state(build_obs(c), k["family"], k["lengthscale"], k["outputscale"], p["noise"],
                                                   p["obs_seed"])
        assert torch.equal(W, state["W"]) and mc == state["mean_const"] and torch.equal(cache, state["mean_cache"]), name


def test_kernel_for_case_builds_the_structured_objects():
    from basq_amd import kernels as BK

    assert isinstance(pools.kernel_for_case(BY_NAME["cfg2_rbf_1e5"]), BK.StationaryKernel)
    kw = pools.kernel_for_case(BY_NAME["cfg5m_wsabim_5e5"])
    assert isinstance(kw, BK.WsabiKernel) and kw.warp == "wsabim"
    kp = pools.kernel_for_case(BY_NAME["posterior_noise_ragged"])
    assert isinstance(kp, BK.PosteriorKernel) and kp.noise == 1e-3
