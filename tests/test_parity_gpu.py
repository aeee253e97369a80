"""GPU parity tests proper: the HIP path (through the C ABI) against the committed golden
fixtures (outputs of the reference itself, see oracle/make_golden.py) and against the oracle.

Bars (BASELINE.json north_star): selected indices identical to the reference; recombination
weights within 1e-5 relative (asserted here at 1e-6: the measured gap is ~1e-9).
"""
import json

import pytest
import torch

from tests.cases import BY_NAME, CASES, build_oracle_kernel, build_pool, build_product_kernel, has_golden, load_golden

pytestmark = pytest.mark.gpu

W_RTOL = 1e-6          # north_star bar is 1e-5
DEV = "cuda:0"


def _run(c, trace=None):
    import basq_amd

    pts, nys = build_pool(c)
    kern = build_product_kernel(c)
    torch.manual_seed(c["torch_seed"])
    idx, w = basq_amd.recombination(pts, nys, c["n"], kern, torch.device(DEV), trace=trace)
    return pts, idx.cpu(), w.cpu()


FAST = [c["name"] for c in CASES if not c["slow"]]
FULL = [c["name"] for c in CASES if c["slow"]]


@pytest.mark.parametrize("name", FAST + FULL)
def test_golden_parity(name):
    """idx bit-exact, weights <= 1e-6 rel, per-round surviving sets identical to the reference run."""
    import basq_amd

    if not has_golden(name):
        pytest.skip("fixture not generated")
    c = BY_NAME[name]
    fx = load_golden(name)
    from basq_amd.pools import pool_digest

    tr = basq_amd.EngineTrace()
    pts, idx, w = _run(c, tr)
    assert pool_digest(pts) == fx["pool_digest"], "regenerated pool differs from the fixture's inputs"
    gi = torch.tensor(fx["idx"], dtype=torch.int64)
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    assert idx.dtype == torch.int64 and w.dtype == torch.float64
    assert idx.tolist() == gi.tolist(), "selected indices differ from the reference"
    rel = ((w - gw).abs() / gw).max().item() if len(gw) else 0.0
    assert rel <= W_RTOL, f"weights off by {rel:.3e} relative"
    # every round kept the same sets as the reference
    assert len(tr.rounds) == fx["n_rounds"]
    for mine, ref in zip(tr.rounds, fx["rounds"]):
        assert mine["kept"] == ref["kept"]
    # reference-free invariants
    assert bool((w > 0).all())
    assert abs(w.sum().item() - 1.0) < 1e-12
    assert idx.tolist() == sorted(idx.tolist())
    assert len(idx) <= c["n"]


@pytest.mark.parametrize("name", ["rbf_1e4", "rbf_ragged", "matern52_posterior", "wsabil_2e4"])
def test_moment_matching(name):
    """Nystrom-moment match: sum_j w_j phi(x_j) == (1/N) sum_i phi(x_i), phi = U k(pt, .), and sum w = 1.

    Size-independent property of the algorithm; evaluated with the oracle's CPU kernel.
    """
    import basq_amd

    c = BY_NAME[name]
    tr = basq_amd.EngineTrace(keep_tensors=True)
    pts, idx, w = _run(c, tr)
    kern, _ = build_oracle_kernel(c)
    nys = pts[: c["m"]]
    U = tr.U.cpu()
    N = c["N"]
    full = torch.zeros(U.shape[0], dtype=torch.float64)
    for lo in range(0, N, 4096):
        full += (U @ kern(nys, pts[lo:lo + 4096])).sum(1) / N
    sel = U @ kern(nys, pts[idx]) @ w
    scale = full.abs().max().item()
    assert (sel - full).abs().max().item() <= 1e-9 * max(scale, 1e-30) + 1e-13


@pytest.mark.parametrize("name", ["rbf_2e4_defaults", "cfg1_posterior_1e4", "rbf_ragged"])
def test_golden_parity_with_the_reference_basis_svd(name, monkeypatch):
    """``_config.BASIS_SVD = True`` -- round 3's path: the range finder ends with the reference's own ``[q, m]`` SVD
    (``torch.svd_lowrank``, ``_rchq.py:29``) instead of stopping at the orthonormal ``Q`` -- is the documented A/B and fallback of
    the round-4 shortcut: it must keep reproducing the same goldens (traced AND untraced: both round loops)."""
    import basq_amd
    import basq_amd._config as eng

    monkeypatch.setattr(eng, "BASIS_SVD", True)
    c, fx = BY_NAME[name], load_golden(name)
    for trace in (basq_amd.EngineTrace(), None):
        _, idx, w = _run(c, trace)
        gw = torch.tensor(fx["w"], dtype=torch.float64)
        assert idx.tolist() == fx["idx"], "selected indices differ from the reference"
        assert ((w - gw).abs() / gw).max().item() <= W_RTOL
        if trace is not None:
            for mine, ref in zip(trace.rounds, fx["rounds"]):
                assert mine["kept"] == ref["kept"]


def test_repeatable_bitwise():
    """Same seed, same inputs -> bit-identical (deterministic reduction order everywhere)."""
    c = BY_NAME["rbf_2e4_defaults"]
    _, i1, w1 = _run(c)
    _, i2, w2 = _run(c)
    assert torch.equal(i1, i2) and torch.equal(w1, w2)


def test_run_rchq_api():
    """BASQ.run_rchq keeps the reference's (pts_nys, pts_rec, w_IS, kernel) -> (x, w) contract."""
    import basq_amd

    c = BY_NAME["rbf_1e4"]
    fx = load_golden("rbf_1e4")
    pts, nys = build_pool(c)
    basq = basq_amd.BASQ(batch_size=c["n"], device=DEV)
    torch.manual_seed(c["torch_seed"])
    x, w = basq.run_rchq(nys, pts, torch.ones(c["N"]) / c["N"], build_product_kernel(c))
    assert x.shape == (len(fx["idx"]), c["d"])
    assert torch.equal(x.cpu(), pts[torch.tensor(fx["idx"])])


def test_full_size_properties():
    """BASELINE config 3 size (N=1e6, d=10, n=100, m=1e4) through size-independent properties only."""
    import basq_amd
    from basq_amd.pools import gmm_pool

    N, d, n = 1_000_000, 10, 100
    pts = gmm_pool(N, d, seed=21)
    nys = pts[: N // 100]
    kern = basq_amd.kernels.StationaryKernel("rbf", 2.0)
    torch.manual_seed(3)
    tr = basq_amd.EngineTrace()
    idx, w = basq_amd.recombination(pts, nys, n, kern, torch.device(DEV), trace=tr)
    idx, w = idx.cpu(), w.cpu()
    assert len(idx) == n and idx.tolist() == sorted(set(idx.tolist()))
    assert bool((w > 0).all()) and abs(w.sum().item() - 1.0) < 1e-12
    # survivors halve every round: R_{r+1} = nb * n_keep (+ tail)
    for a, b in zip(tr.rounds[:-1], tr.rounds[1:]):
        assert b["R"] <= a["R"] // 2 + a["S"]


def test_differential_fuzz_gpu():
    """Random configurations through the HIP path against the oracle: odd sizes, every small-d kernel variant, tiny and
    ragged pools.  Indices identical, weights inside the 1e-5 bar; configurations whose Nystrom Gram is numerically
    rank-deficient for the requested q are skipped (the reference itself is round-off dependent there, see
    tests/test_host_logic.py::test_reference_is_unstable_when_q_exceeds_the_numerical_rank)."""
    import basq_amd
    from basq_amd.pools import gmm_pool
    from oracle.kernels_oracle import StationaryOracle
    from oracle.rchq_oracle import recombination_oracle

    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        g = torch.Generator().manual_seed(23)
        compared = 0
        for case in range(40):
            N = int(torch.randint(3, 4000, (1,), generator=g))
            d = int(torch.randint(1, 15, (1,), generator=g))
            n = int(torch.randint(2, 61, (1,), generator=g))
            m = int(torch.randint(1, min(N, 200) + 1, (1,), generator=g))
            fam = ["rbf", "matern52", "matern32"][case % 3]
            ls = 0.8 + 0.4 * (case % 5)
            pts = gmm_pool(N, d, 300 + case)
            nys = pts[:m]
            ko = StationaryOracle(fam, ls, 1.0)
            ev = torch.linalg.eigvalsh(ko(nys, nys))
            if int((ev > 1e-10 * ev.max()).sum()) < min(n - 1, m):
                continue
            torch.manual_seed(case)
            io, wo = recombination_oracle(pts, nys, n, ko)
            torch.manual_seed(case)
            ie, we = basq_amd.recombination(pts, nys, n, basq_amd.kernels.StationaryKernel(fam, ls, 1.0), torch.device(DEV))
            assert io.tolist() == ie.cpu().tolist(), f"case {case}: N={N} d={d} n={n} m={m} {fam}"
            if len(wo):
                assert ((we.cpu() - wo).abs() / wo).max().item() <= 1e-5
            compared += 1
        assert compared >= 25
    finally:
        torch.set_default_dtype(prev)


# the cases of ``structured_fuzz_cases(11, 150)`` outside the 1e-5 bar, each explained by the reference's own sensitivity (the
# first four are the ones of round 3's builder-run log, profiles/r06_x_fuzz_more_seeds_9_10_11.txt)
STRUCTURED_FUZZ_UNSTABLE = [2, 19, 21, 93, 135]


def test_structured_differential_fuzz_gpu():
    """Random configurations with STRUCTURED kernels -- stationary, GP posterior (``_gp.py:259-277``), WSABI-L and WSABI-M
    (``_wsabi.py:205-249``), likelihood noise 1e-10 / 1e-6 / 1e-3 -- through the HIP path against the oracle (``_rchq.py:81-99``
    with those callables): indices identical, weights inside the 1e-5 bar.

    Where the bar applies: everywhere the REFERENCE ITSELF stays inside it when its base-kernel values move by <= 1 ulp.
    A case that leaves the bar is accepted only if the oracle, run against itself with five such perturbation patterns,
    (a) changes its own indices, or (b) moves its own weights by at least a quarter of what the engine is off by -- then it
    is counted as ``unstable``, not as compared.  Two regimes produce such cases: posteriors whose
    observation Gram is ill-conditioned (catastrophic cancellation in ``k - k(.,X) W k(X,.)``) and batches with a nearly
    vanishing weight (the relative error of the smallest weight is what the bar measures).  The case list is the one of
    ``tools/fuzz_structured.py 11`` (round 3's builder-run log), cases 19 / 21 / 93 included."""
    import warnings

    import basq_amd
    from oracle.rchq_oracle import recombination_oracle
    from tests.cases import (build_oracle_kernel, build_perturbed_oracle_kernel, build_pool, build_product_kernel,
                             observation_gram_condition, structured_fuzz_cases)

    def deviation(ia, wa, ib, wb):
        same = ia.tolist() == ib.tolist()
        return same, (((wa - wb).abs() / wb).max().item() if same and len(wb) else (0.0 if same else float("inf")))

    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        compared = rank_deficient = unstable = 0
        kinds = [0, 0, 0, 0]
        unstable_log, unstable_cases = [], []
        for i, c in enumerate(structured_fuzz_cases(11, 150)):
            pts, nys = build_pool(c)
            ko, state = build_oracle_kernel(c)
            A = ko(nys, nys)
            ev = torch.linalg.eigvalsh(0.5 * (A + A.T))
            if int((ev > 1e-10 * ev.abs().max()).sum()) < min(c["n"] - 1, c["m"]):
                rank_deficient += 1                              # q beyond the numerical rank: see test_host_logic.py
                continue
            cond = observation_gram_condition(c, state)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                torch.manual_seed(c["torch_seed"])
                io, wo = recombination_oracle(pts, nys, c["n"], ko)
                torch.manual_seed(c["torch_seed"])
                ie, we = basq_amd.recombination(pts, nys, c["n"], build_product_kernel(c, state), torch.device(DEV))
            same, rel = deviation(ie.cpu(), we.cpu(), io, wo)
            if same and rel <= 1e-5:
                compared += 1
                kinds[i % 4] += 1
                continue
            label = f"case {i} ({c['kernel']['family']}, warp {c['kernel']['warp']}, N={c['N']} d={c['d']} n={c['n']} m={c['m']}, " \
                    f"posterior {c['kernel']['posterior']}, cond {cond:.1e}): idx equal {same}, rel {rel:.2e}"
            ref_idx_moves, ref_rel = False, 0.0
            for s in (1, 2, 3, 4, 5):                            # the reference against itself, kernel values +- 1 ulp
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    torch.manual_seed(c["torch_seed"])
                    ip, wp = recombination_oracle(pts, nys, c["n"], build_perturbed_oracle_kernel(c, s))
                same_p, rel_p = deviation(ip, wp, io, wo)
                ref_idx_moves = ref_idx_moves or not same_p
                ref_rel = max(ref_rel, rel_p if same_p else 0.0)
            explained = ref_idx_moves or (same and rel <= 4.0 * ref_rel)
            assert explained, f"outside the bar where the reference is stable (own weights move {ref_rel:.2e}): " + label
            unstable += 1
            unstable_cases.append(i)
            unstable_log.append(label + f"; reference vs itself: idx moves {ref_idx_moves}, rel {ref_rel:.2e}")
        print("\n".join(unstable_log))
        assert compared >= 100 and min(kinds) >= 20, (compared, kinds)
        # (review r04: "a regression from 6 to 12 passes silently"; ADVICE r5: an exact set breaks on any legitimate 1-ulp change.)
        # Every member has shown the reference's own instability above; the KNOWN members are the expected ones, at most two
        # further cases may cross the bar (another compiler, another exponential) before this is a regression worth a look,
        # and a known case coming back inside the bar is only reported
        new = sorted(set(unstable_cases) - set(STRUCTURED_FUZZ_UNSTABLE))
        back = sorted(set(STRUCTURED_FUZZ_UNSTABLE) - set(unstable_cases))
        if new or back:
            warnings.warn(f"structured fuzz: cases outside the bar changed -- new {new}, back inside {back}")
        assert len(new) <= 2, (unstable_cases, unstable_log)
    finally:
        torch.set_default_dtype(prev)


def test_fuzz_more_hard_cases_gpu():
    """The two hardest cases of the builder-run bug hunt (``tools/fuzz_more.py`` seeds 7-11, 2 000 cases,
    ``profiles/r06_x_fuzz_more_seeds_*.txt``) as named regression cases:

    * structured seed 8 case 33 (RBF posterior, cond 3.5e10, N=1246 n=38 m=232): the engine's indices differ from the oracle's; five
      1-ulp patterns do not show the reference's instability, forty do (it moves its OWN indices in 4 of them);
    * SOBER seed 10 ``sfz79`` (WSABI-M, noise 1e-3, m = 16 < n): indices equal, weights 9.8e-5 off -- within 4x of what the reference's
      own weights move under twenty patterns (8.5e-5).

    Pinned: the engine's outcome (which side of the bar) AND the explanation (the reference against itself)."""
    import warnings

    import basq_amd
    from basq_amd import sober
    from oracle.rchq_oracle import recombination_oracle, recombination_sober_oracle
    from tests.cases import (build_oracle_kernel, build_perturbed_oracle_kernel, build_pool, build_product_kernel,
                             structured_fuzz_cases)
    from tests.test_sober import _sober_fuzz_cases

    def deviation(ia, wa, ib, wb):
        same = ia.tolist() == ib.tolist()
        return same, (((wa - wb).abs() / wb).max().item() if same and len(wb) else (0.0 if same else float("inf")))

    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            # -- structured seed 8 case 33
            c = structured_fuzz_cases(8, 34)[33]
            assert (c["N"], c["n"], c["m"]) == (1246, 38, 232)
            pts, nys = build_pool(c)
            ko, state = build_oracle_kernel(c)
            torch.manual_seed(c["torch_seed"])
            io, wo = recombination_oracle(pts, nys, c["n"], ko)
            torch.manual_seed(c["torch_seed"])
            ie, we = basq_amd.recombination(pts, nys, c["n"], build_product_kernel(c, state), torch.device(DEV))
            same, rel = deviation(ie.cpu(), we.cpu(), io, wo)
            if not (same and rel <= 1e-5):
                moved = 0
                for s in range(1, 41):
                    torch.manual_seed(c["torch_seed"])
                    ip, wp = recombination_oracle(pts, nys, c["n"], build_perturbed_oracle_kernel(c, s))
                    moved += not deviation(ip, wp, io, wo)[0]
                assert moved >= 1, "seed 8 case 33: outside the bar, and forty 1-ulp patterns leave the reference's indices alone"
            # -- SOBER seed 10 sfz79
            c, w0 = _sober_fuzz_cases(80, seed=110)[79]
            assert (c["N"], c["n"], c["m"]) == (2947, 47, 16)
            pts, nys = build_pool(c)
            ko, state = build_oracle_kernel(c)
            torch.manual_seed(c["torch_seed"])
            io, wo = recombination_sober_oracle(pts, nys, c["n"], ko, None if w0 is None else w0.clone())
            torch.manual_seed(c["torch_seed"])
            ie, we = sober.recombination(pts, nys, c["n"], build_product_kernel(c, state), torch.device(DEV), torch.float64,
                                         init_weights=w0)
            same, rel = deviation(ie.cpu(), we.cpu(), io, wo)
            assert same, "sfz79: indices differ"
            if rel > 1e-5:
                ref_rel = 0.0
                for s in range(1, 21):
                    torch.manual_seed(c["torch_seed"])
                    ip, wp = recombination_sober_oracle(pts, nys, c["n"], build_perturbed_oracle_kernel(c, s),
                                                        None if w0 is None else w0.clone())
                    sp, rp = deviation(ip, wp, io, wo)
                    assert sp, "sfz79: the reference moves its own indices (it did not when the case was recorded)"
                    ref_rel = max(ref_rel, rp)
                assert rel <= 4.0 * ref_rel, f"sfz79: weights off by {rel:.2e}, the reference's own move by {ref_rel:.2e}"
    finally:
        torch.set_default_dtype(prev)


def test_index_mismatch_is_attributed_to_the_posterior_cancellation_not_to_the_shortcuts():
    """Review r05, item 3: WHICH part of the engine costs the parity in structured seed 8 case 33 (RBF posterior, observation Gram
    cond 3.5e10), the one known index mismatch?  Measured (``tools/attribute_mismatch.py`` ->
    ``profiles/r08_b_attribution_of_parity_misses_per_switch.txt``) and pinned here:

    * NOT the shortcuts: with the reference-form basis (``BASIS_SVD``: the final SVD of ``svd_lowrank``, ``_rchq.py:28-31``), with
      the host-LAPACK null space (``GPU_NULLSPACE = False``, ``:138-143``), with ``torch.svd_lowrank`` itself on the host
      (``GPU_RANGE_FINDER = False``) and with all of them together the engine selects the SAME indices as by default;
    * the arithmetic of the posterior covariance: the reference's OWN formulation (explicit ``k - k(.,X) W k(X,.)`` per block,
      ``_gp.py:259-277``) evaluated by device tensor operations and pushed through the engine's dense path reproduces the
      oracle's indices -- with weights that differ from the oracle's by orders of magnitude more than the 1e-5 bar (2e-2 when
      recorded): the same formula under another GEMM's rounding.  The fused path folds the correction into the contraction
      matrix by linearity (``U_ext``), a third rounding of the same cancellation, and lands on other indices."""
    import warnings

    import basq_amd
    import basq_amd._config as eng
    from oracle.kernels_oracle import PosteriorOracle, StationaryOracle
    from oracle.rchq_oracle import recombination_oracle
    from tests.cases import build_oracle_kernel, build_pool, build_product_kernel, structured_fuzz_cases

    c = structured_fuzz_cases(8, 34)[33]
    pts, nys = build_pool(c)
    ko, state = build_oracle_kernel(c)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    saved = (eng.BASIS_SVD, eng.GPU_NULLSPACE, eng.GPU_RANGE_FINDER)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.manual_seed(c["torch_seed"])
            io, wo = recombination_oracle(pts, nys, c["n"], ko)
            runs = {}
            for key in ((False, True, True), (True, True, True), (False, False, True), (True, False, False)):
                eng.BASIS_SVD, eng.GPU_NULLSPACE, eng.GPU_RANGE_FINDER = key
                try:
                    torch.manual_seed(c["torch_seed"])
                    ie, we = basq_amd.recombination(pts, nys, c["n"], build_product_kernel(c, state), torch.device(DEV))
                finally:
                    eng.BASIS_SVD, eng.GPU_NULLSPACE, eng.GPU_RANGE_FINDER = saved
                runs[key] = ie.cpu().tolist()
            default = runs[(False, True, True)]
            if default == io.tolist():
                return                                               # (a change made the case match: nothing left to attribute)
            for key, idx in runs.items():
                assert idx == default, f"BASIS_SVD, GPU_NULLSPACE, GPU_RANGE_FINDER = {key} changes the engine's selection"
            k = c["kernel"]
            dev = torch.device(DEV)
            kdev = PosteriorOracle(StationaryOracle(k["family"], k["lengthscale"], k["outputscale"]), state["Xobs"].to(dev),
                                   state["W"].to(dev), state["noise"])
            torch.manual_seed(c["torch_seed"])
            iq, wq = basq_amd.recombination(pts.to(dev), nys.to(dev), c["n"], kdev, dev)
            assert iq.cpu().tolist() == io.tolist(), "the reference's formulation on the device no longer reproduces the oracle's indices"
            rel = ((wq.cpu() - wo).abs() / wo).max().item()
            assert rel > 1e-4, f"the reference's own formulation now agrees to {rel:.1e}: the case is no longer ill-conditioned?"
    finally:
        torch.set_default_dtype(prev)


@pytest.mark.parametrize("N,d,n,m", [(1, 2, 2, 1), (2, 2, 2, 1), (2, 2, 3, 2), (3, 1, 2, 2), (5, 3, 2, 3), (4, 2, 2, 4),
                                     (7, 2, 3, 1), (10, 2, 10, 5), (6, 2, 2, 6), (0, 2, 3, 0)])
def test_tiny_and_degenerate_pools_gpu(N, d, n, m):
    """The HIP path on one-point pools, pools smaller than the batch, a single Nystrom feature, m = 1, the empty pool."""
    import basq_amd
    from basq_amd.pools import gmm_pool
    from oracle.kernels_oracle import StationaryOracle
    from oracle.rchq_oracle import recombination_oracle

    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        pts = gmm_pool(max(N, 1), d, 3)[:N]
        nys = pts[:m]
        torch.manual_seed(1)
        io, wo = recombination_oracle(pts, nys, n, StationaryOracle("rbf", 1.5, 1.0))
        torch.manual_seed(1)
        ie, we = basq_amd.recombination(pts, nys, n, basq_amd.kernels.StationaryKernel("rbf", 1.5, 1.0), torch.device(DEV))
        assert io.tolist() == ie.cpu().tolist()
        assert len(wo) == len(we) and (len(wo) == 0 or ((we.cpu() - wo).abs() / wo).max().item() <= 1e-9)
    finally:
        torch.set_default_dtype(prev)


@pytest.mark.parametrize("name", FAST + FULL)
def test_golden_parity_descriptor_driven_rounds(name):
    """The production path on one GPU (stationary, posterior, WSABI-L and -- round 5 -- WSABI-M kernels) -- rounds enqueued
    without a host wait, geometry in a device-resident descriptor -- against the same goldens (no trace: a trace selects the
    round-by-round loop)."""
    import basq_amd
    import basq_amd._config as eng

    if not has_golden(name):
        pytest.skip("fixture not generated")
    assert eng.ASYNC_ROUNDS
    c = BY_NAME[name]
    fx = load_golden(name)
    if c["N"] > 2 * c["n"]:                                      # (pools of a single reduction have no such rounds)
        from basq_amd._batch import Plan
        from basq_amd._engine import LocalComm
        from basq_amd._ops import HipOps

        plan = Plan.of(build_product_kernel(c), "basq", None, LocalComm(), HipOps(torch.device(DEV)), None, n_sets=2 * c["n"])
        assert plan.async_rounds, "this case does not take the descriptor-driven rounds"
    _, idx, w = _run(c, None)
    gi = torch.tensor(fx["idx"], dtype=torch.int64)
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    assert idx.tolist() == gi.tolist(), "selected indices differ from the reference"
    rel = ((w - gw).abs() / gw).max().item() if len(gw) else 0.0
    assert rel <= W_RTOL, f"weights off by {rel:.3e} relative"


@pytest.mark.parametrize("name", ["rbf_1e4", "rbf_2e4_defaults", "rbf_ragged", "matern52_3e4_d32", "cfg2_rbf_1e5", "cfg3_rbf_1e6",
                                  "cfg4_matern52_1e6_d32", "cfg1_posterior_1e4", "wsabil_2e4", "matern52_posterior",
                                  "posterior_noise_ragged", "wsabil_noise_ragged", "cfg5_wsabil_5e5"])
def test_golden_parity_column_epochs_and_round5_epochs(name):
    """Round 6: inside an epoch the candidates outside the residue classes are message columns (``basq_amd/_epochs.py``; the default
    for these cases) -- against the goldens with EVERY round's kept sets (a trace read after the fact), and against the round-5
    form of the same rounds (``IRR_COLUMNS = False``: irregular blocks evaluated, projected and compacted every round), which must
    select the same batch."""
    import basq_amd
    import basq_amd._config as eng

    if not has_golden(name):
        pytest.skip("fixture not generated")
    c, fx = BY_NAME[name], load_golden(name)
    pts, nys = build_pool(c)
    pts, nys = pts.to(DEV), nys.to(DEV)
    kern = build_product_kernel(c)
    out = {}
    for flag in (True, False):
        old = eng.IRR_COLUMNS
        eng.IRR_COLUMNS = flag
        try:
            tr = basq_amd.EngineTrace(host_sync=False)
            torch.manual_seed(c["torch_seed"])
            idx, w = basq_amd.recombination(pts, nys, c["n"], kern, torch.device(DEV), trace=tr)
            out[flag] = (idx.cpu(), w.cpu(), [r["kept"] for r in tr.rounds])
        finally:
            eng.IRR_COLUMNS = old
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    for flag in (True, False):
        idx, w, kept = out[flag]
        assert idx.tolist() == fx["idx"], f"IRR_COLUMNS={flag}: selected indices differ from the reference"
        assert ((w - gw).abs() / gw).max().item() <= W_RTOL
        assert kept == [r["kept"] for r in fx["rounds"]][:len(kept)] and len(kept) >= 1
    assert torch.allclose(out[True][1], out[False][1], rtol=1e-9, atol=0)


# ---- shape envelope: the reference accepts any d and any num_pts (_rchq.py:4-25) -------------------------------------------
def _oracle(c):
    from oracle.rchq_oracle import recombination_oracle

    pts, nys = build_pool(c)
    ko, _ = build_oracle_kernel(c)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(c["torch_seed"])
        return recombination_oracle(pts, nys, c["n"], ko)
    finally:
        torch.set_default_dtype(prev)


@pytest.mark.parametrize("kernel", [
    dict(family="rbf", lengthscale=4.0, outputscale=1.0, posterior=None, warp="none"),
    dict(family="matern52", lengthscale=6.0, outputscale=1.2, posterior=dict(n_obs=40, noise=1e-3, obs_seed=31), warp="none"),
])
def test_dimension_beyond_the_packed_rows_degrades_to_the_dense_path(kernel):
    """d = 45 > 38: the fused kernels cannot pack the points; the structured kernel then runs through the dense path (with
    a RuntimeWarning), the oracle's batch all the same."""
    from tests.cases import case

    c = case("wide_d45", 6_000, 45, 120, 30, kernel, pool_seed=61, torch_seed=3)
    idx_o, w_o = _oracle(c)
    with pytest.warns(RuntimeWarning, match="packed-row limit"):
        pts, idx, w = _run(c)
    assert idx.tolist() == idx_o.tolist()
    assert ((w - w_o).abs() / w_o).max().item() <= W_RTOL


def test_host_nullspace_route_reproduces_goldens():
    """``GPU_NULLSPACE = False``: the per-round null space from host LAPACK (the reference's own SVD) -- the route batches
    with 2 * num_pts > 1024 take -- selects the golden batches too."""
    import basq_amd._config as cfg

    old = cfg.GPU_NULLSPACE
    cfg.GPU_NULLSPACE = False
    try:
        for name in ("rbf_ragged", "cfg1_posterior_1e4", "matern52_3e4_d32"):
            c, fx = BY_NAME[name], load_golden(name)
            _, idx, w = _run(c)
            assert idx.tolist() == fx["idx"], name
            gw = torch.tensor(fx["w"], dtype=torch.float64)
            assert ((w - gw).abs() / gw).max().item() <= W_RTOL
    finally:
        cfg.GPU_NULLSPACE = old


def test_untraced_batch_beyond_the_reduction_kernels():
    """the same envelope WITHOUT a trace -- the call every user makes.  ``Plan.of`` must keep such a batch off the
    descriptor-driven rounds (they call the GPU null-space kernel directly, which rejects 2 * num_pts > 1024): it used to
    raise ``BasqHipError``; now it warns and returns the batch the traced run returns."""
    from tests.cases import K, case

    c = case("wide_n520", 6_000, 8, 1_500, 520, K("rbf", 1.0), pool_seed=62, torch_seed=4)
    with pytest.warns(RuntimeWarning, match="host LAPACK"):
        _, idx_t, w_t = _run(c, basq_amd_trace(keep_tensors=True))
    with pytest.warns(RuntimeWarning, match="host LAPACK"):
        _, idx, w = _run(c)
    assert idx.tolist() == idx_t.tolist() and torch.equal(w, w_t)


def basq_amd_trace(**kw):
    import basq_amd

    return basq_amd.EngineTrace(**kw)


@pytest.mark.parametrize("M,s,seed", [(200, 100, 0), (74, 37, 1), (400, 200, 2), (150, 60, 3)])
def test_wide_elimination_by_tensor_ops_equals_the_kernel(M, s, seed):
    """The envelope form of the elimination (``HipOps._car_eliminate_wide``: the reference's steps as device tensor
    operations, used for 2 * num_pts > 1024) against ``basq_car_eliminate_f64`` on shapes both can run: same survivors,
    same weights (the kernel is bit-exact with the reference's op order; the tensor ops may differ by an ulp)."""
    from basq_amd._ops import HipOps

    ops = HipOps(torch.device(DEV))
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(s, M, generator=g, dtype=torch.float64)
    X[0] = 1.0
    PhiT = ops.nullspace(ops.to_device(X), s, M)
    mu = ops.to_device(torch.rand(M, generator=g, dtype=torch.float64) + 0.1)
    kr1, kept1, w1, info1 = ops.car_eliminate(PhiT.clone(), mu.clone(), M, s)
    kr2, kept2, w2, info2 = ops._car_eliminate_wide(PhiT.clone(), mu.clone(), M, s)
    assert info1.tolist() == info2.tolist()
    n_keep = int(info1[0])
    assert kept1[:n_keep].tolist() == kept2[:n_keep].tolist() and kr1.tolist() == kr2.tolist()
    assert torch.allclose(w1[:n_keep], w2[:n_keep], rtol=1e-12, atol=0)


def test_batch_size_beyond_the_reduction_kernels_degrades_gracefully():
    """num_pts = 520 -> 2n = 1040 sets > 1024: host-LAPACK null space + tensor-op elimination, with a RuntimeWarning instead
    of an error.  At this size the reference's own selection moves under 1e-12 perturbations of the kernel values (measured
    with the oracle), so the check is the reference-free contract: positive weights summing to one, at most num_pts points,
    ascending indices, and the Nystrom moments of the pool reproduced."""
    from tests.cases import K, case

    import basq_amd

    c = case("wide_n520", 6_000, 8, 1_500, 520, K("rbf", 1.0), pool_seed=62, torch_seed=4)
    tr = basq_amd.EngineTrace(keep_tensors=True)
    with pytest.warns(RuntimeWarning, match="host LAPACK"):
        pts, idx, w = _run(c, tr)
    assert 0 < len(idx) <= c["n"] and idx.tolist() == sorted(set(idx.tolist()))
    assert bool((w > 0).all()) and abs(w.sum().item() - 1.0) < 1e-11
    ko, _ = build_oracle_kernel(c)
    nys, U = pts[: c["m"]], tr.U.cpu()
    full = (U @ ko(nys, pts)).mean(1)                             # the q = 519 Nystrom moments of the pool ...
    sel = U @ ko(nys, pts[idx]) @ w                               # ... reproduced by the weighted batch
    assert (sel - full).abs().max().item() <= 1e-9 * full.abs().max().item() + 1e-13
