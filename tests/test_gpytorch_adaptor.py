"""``kernels.from_gpytorch_model`` -- the adaptor INTEGRATION.md tells a maintainer to use -- on duck-typed gpytorch
models (gpytorch itself is not installed here; the adaptor only reads attributes, the same ones the reference reads at
``BASQ/_gp.py:233-256`` and ``BASQ/_gaussian_calc.py:32-51``).

Covered: kinds ``prior`` / ``predictive`` / ``wsabi`` (L and M) against ``tests.cases.build_product_kernel`` and the
oracle's callables; the ``[[l]]``-shaped lengthscale gpytorch stores; Matern ``nu``; a model whose
``prediction_strategy`` is still ``None`` (fresh from training / back in train mode), which the reference warms up with
``model.eval(); model(one point)`` (``_gp.py:247-253``, ``_gaussian_calc.py:32-38``).
"""
from types import SimpleNamespace

import pytest
import torch

from basq_amd import kernels as BK
from basq_amd._engine import RecombinationEngine
from tests.cases import BY_NAME, build_obs, build_oracle_kernel, build_pool, build_product_kernel, load_golden
from tests.cpu_stand_in import CpuStandInOps


class RBFKernel:        # the adaptor dispatches on the class NAME of covar_module.base_kernel
    def __init__(self, l):
        self.lengthscale = torch.tensor([[l]], dtype=torch.float64)      # gpytorch's [1, 1] shape


class MaternKernel:
    def __init__(self, l, nu):
        self.lengthscale = torch.tensor([[l]], dtype=torch.float64)
        self.nu = nu


class DuckGP:
    """A stand-in for a fitted ``ExactGPModel`` carrying a case's GP state.  ``S`` with ``S S^T = W`` plays
    ``covar_cache``.  ``lazy_caches``: ``prediction_strategy`` is None until the model is called (as gpytorch's is after
    training), and every ``eval()`` / call is logged."""

    def __init__(self, c, state, lazy_caches=False):
        k = c["kernel"]
        base = RBFKernel(k["lengthscale"]) if k["family"] == "rbf" else \
            MaternKernel(k["lengthscale"], 2.5 if k["family"] == "matern52" else 1.5)
        Wm = state["W"]
        S = torch.linalg.cholesky(Wm + 1e-18 * torch.eye(len(Wm), dtype=Wm.dtype))
        self._ps = SimpleNamespace(mean_cache=state["mean_cache"], covar_cache=S)
        self.train_inputs = (state["Xobs"],)
        self.covar_module = SimpleNamespace(outputscale=torch.tensor(k["outputscale"], dtype=torch.float64), base_kernel=base)
        self.mean_module = SimpleNamespace(constant=torch.tensor(state["mean_const"], dtype=torch.float64))
        self.likelihood = SimpleNamespace(noise=torch.tensor([state["noise"]], dtype=torch.float64))
        self.prediction_strategy = None if lazy_caches else self._ps
        self.calls = []

    def eval(self):
        self.calls.append("eval")

    def __call__(self, x):
        self.calls.append(("call", tuple(x.shape)))
        self.prediction_strategy = self._ps


def duck_model(c, state, lazy_caches=False):
    return DuckGP(c, state, lazy_caches)


def _as_callable(m):
    return m


def _state(c):
    _, st = build_oracle_kernel(c)
    return st


@pytest.mark.parametrize("name,kind", [("cfg1_posterior_1e4", "predictive"), ("matern52_posterior", "predictive"),
                                       ("wsabil_2e4", "wsabi"), ("wsabim_1e4", "wsabi")])
def test_adaptor_equals_hand_built_kernel(name, kind):
    c = BY_NAME[name]
    st = _state(c)
    k = BK.from_gpytorch_model(duck_model(c, st), kind, wsabi_label=c["kernel"]["warp"] if kind == "wsabi" else "wsabil")
    ref = build_product_kernel(c, st)
    assert type(k) is type(ref)
    assert (k.base.family, k.base.lengthscale, k.base.outputscale) == (ref.base.family, ref.base.lengthscale, ref.base.outputscale)
    post, rpost = k.posterior, ref.posterior
    assert post.noise == rpost.noise and torch.equal(post.Xobs, rpost.Xobs)
    assert (post.W - rpost.W).abs().max().item() <= 1e-12 * rpost.W.abs().max().item()      # S S^T == W
    if kind == "wsabi":
        assert k.warp == c["kernel"]["warp"] and k.mean_const == ref.mean_const
        assert torch.equal(k.mean_cache.reshape(-1), ref.mean_cache.reshape(-1))


def test_prior_kind_and_matern_nu():
    c = BY_NAME["matern32_8e3"]
    m = SimpleNamespace(covar_module=SimpleNamespace(outputscale=torch.tensor(2.0), base_kernel=MaternKernel(3.0, 1.5)))
    k = BK.from_gpytorch_model(m, "prior")
    assert isinstance(k, BK.StationaryKernel) and (k.family, k.lengthscale, k.outputscale) == ("matern32", 3.0, 2.0)
    m.covar_module.base_kernel = MaternKernel(3.0, 0.5)
    with pytest.raises(ValueError):
        BK.from_gpytorch_model(m, "prior")
    m.covar_module.base_kernel = SimpleNamespace(lengthscale=torch.ones(1, 3))       # ARD: unsupported, as documented
    with pytest.raises(ValueError):
        BK.from_gpytorch_model(m, "prior")
    assert c["kernel"]["family"] == "matern32"


@pytest.mark.parametrize("name,kind", [("cfg1_posterior_1e4", "predictive"), ("wsabil_noise_ragged", "wsabi")])
def test_adaptor_drives_the_engine_to_the_golden(name, kind):
    """End to end on the stand-in ops: model -> from_gpytorch_model -> recombination == the reference's golden."""
    c, fx = BY_NAME[name], load_golden(name)
    pts, nys = build_pool(c)
    k = BK.from_gpytorch_model(duck_model(c, _state(c)), kind, wsabi_label="wsabil")
    torch.manual_seed(c["torch_seed"])
    idx, w = RecombinationEngine(CpuStandInOps()).run(pts, 0, c["N"], nys, c["n"], k)
    assert idx.tolist() == fx["idx"]
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    assert ((w - gw).abs() / gw).max().item() <= 1e-6


def test_caches_are_warmed_up_like_the_reference():
    """``prediction_strategy is None`` (fresh model): the reference's get_cov_cache calls ``model.eval()`` and evaluates
    the model at ``Xobs[0]`` (``_gp.py:250-253``); the adaptor must do the same instead of raising AttributeError."""
    c = BY_NAME["cfg1_posterior_1e4"]
    st = _state(c)
    m = _as_callable(duck_model(c, st, lazy_caches=True))
    k = BK.from_gpytorch_model(m, "predictive")
    assert m.calls == ["eval", ("call", (1, c["d"]))]
    assert (k.W - build_product_kernel(c, st).W).abs().max().item() <= 1e-12 * st["W"].abs().max().item()
    # a model whose caches exist is not touched
    m2 = _as_callable(duck_model(c, st))
    BK.from_gpytorch_model(m2, "predictive")
    assert m2.calls == []


def test_gaussian_calc_get_cache_fallback():
    """``GaussianCalc.get_cache`` (``_gaussian_calc.py:32-38``): caches missing -> ``model.eval(); model(prior.loc)``."""
    from basq_amd._gaussian_calc import GaussianCalc

    c = BY_NAME["cfg1_posterior_1e4"]
    st = _state(c)
    m = _as_callable(duck_model(c, st, lazy_caches=True))
    prior = SimpleNamespace(loc=torch.zeros(c["d"], dtype=torch.float64))
    wv, winv = GaussianCalc(prior, "cpu", ops=CpuStandInOps()).get_cache(m)
    assert m.calls == ["eval", ("call", (1, c["d"]))]
    assert torch.equal(wv, st["mean_cache"])
    assert (winv - st["W"]).abs().max().item() <= 1e-12 * st["W"].abs().max().item()


def test_obs_builder_matches_state():
    c = BY_NAME["cfg1_posterior_1e4"]
    assert torch.equal(build_obs(c), _state(c)["Xobs"])


# ---- SOBER's Kernel(model, mode) wrapper (SOBER/_kernel.py:4-45) ---------------------------------------------------------
class DuckSoberKernel:
    """What ``SOBER/_kernel.py``'s ``Kernel`` looks like from outside: ``.model``, ``.mode``, callable."""

    def __init__(self, model, mode="predictive_covariance"):
        self.model, self.mode = model, mode

    def __call__(self, x, y):
        raise AssertionError("the structured equivalent must be used, not the wrapper's dense call")


def test_from_sober_kernel_maps_the_three_modes():
    c = BY_NAME["cfg1_posterior_1e4"]
    st = _state(c)
    ref = build_product_kernel(c, st)
    m = duck_model(c, st)
    k = BK.from_sober_kernel(DuckSoberKernel(m, "predictive_covariance"))
    assert isinstance(k, BK.PosteriorKernel) and k.noise == 0.0                  # SOBER/_gp.py:281-305: no noise diagonal
    assert (k.W - ref.W).abs().max().item() <= 1e-12 * ref.W.abs().max().item() and torch.equal(k.Xobs, ref.Xobs)
    k = BK.from_sober_kernel(DuckSoberKernel(m, "weighted_predictive_covariance"))
    assert isinstance(k, BK.WsabiKernel) and k.warp == "wsabil" and k.posterior.noise == 0.0 and k.jitter == 0.0
    assert k.mean_const == st["mean_const"] and torch.equal(k.mean_cache.reshape(-1), st["mean_cache"].reshape(-1))
    k = BK.from_sober_kernel(DuckSoberKernel(m, "kernel"))
    assert isinstance(k, BK.StationaryKernel) and (k.family, k.lengthscale, k.outputscale) == ("rbf", 2.0, 1.3)
    assert isinstance(BK.from_sober_kernel(m, "kernel"), BK.StationaryKernel)    # (the bare model + a mode)
    with pytest.raises(ValueError):
        BK.from_sober_kernel(DuckSoberKernel(m, "nonsense"))
    with pytest.raises(ValueError):
        BK.from_sober_kernel(m)                                                  # a bare model needs its mode
    assert BK.looks_like_sober_kernel(DuckSoberKernel(m)) and not BK.looks_like_sober_kernel(ref)
    assert not BK.looks_like_sober_kernel(lambda x, y: x @ y.T)


def test_sober_entry_adapts_the_wrapper_to_the_fused_path():
    """``sober.recombination(..., Kernel(model))`` -- the tutorials' call (``SOBER/BASQ/_basq.py:19-36``) -- must not take the
    dense door silently: the wrapper is replaced by its structured equivalent (a wrapper over an unsupported GP keeps the dense
    door, with a warning)."""
    import warnings

    from basq_amd import sober

    c = BY_NAME["cfg1_posterior_1e4"]
    m = duck_model(c, _state(c))
    got = sober._adapt_sober_kernel(DuckSoberKernel(m, "predictive_covariance"))
    assert isinstance(got, BK.PosteriorKernel) and got.noise == 0.0
    m.covar_module.base_kernel = SimpleNamespace(lengthscale=torch.ones(1, 3))    # ARD: no fused equivalent
    wrapper = DuckSoberKernel(m, "predictive_covariance")
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        assert sober._adapt_sober_kernel(wrapper) is wrapper
    assert any("evaluated densely" in str(x.message) for x in wl)
    f = lambda x, y: x @ y.T                                                      # noqa: E731
    assert sober._adapt_sober_kernel(f) is f


def test_sober_wrapper_drives_the_engine_to_the_sober_golden():
    """End to end on the stand-in ops: a tutorial-sized SOBER golden (RBF posterior, n_obs = 2) through the adapted wrapper."""
    import json
    import os

    from oracle.make_golden_sober import TUTORIAL_CASES, tutorial_inputs

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sober_tutorial.json")
    with open(path) as f:
        fx = [r for r in json.load(f) if r["case"]["name"] == "sober_tut01_rbf_nobs2"][0]
    c = [t for t in TUTORIAL_CASES if t["name"] == "sober_tut01_rbf_nobs2"][0]
    pts, nys = tutorial_inputs(c)
    _, st = build_oracle_kernel(c)
    k = BK.from_sober_kernel(DuckSoberKernel(duck_model(c, dict(st, noise=1e-10)), "predictive_covariance"))
    torch.manual_seed(c.get("torch_seed", 1))
    idx, w = RecombinationEngine(CpuStandInOps()).run(pts, 0, c["N"], nys, c["n"], k, variant="sober")
    assert idx.tolist() == fx["idx"]
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    assert ((w - gw).abs() / gw).max().item() <= 1e-5

