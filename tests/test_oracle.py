"""The oracle against the reference's golden vectors (committed outputs of the reference run in the
build container, see oracle/make_golden.py) -- runs anywhere, no GPU, no /root/reference."""
import os

import pytest
import torch

from tests.cases import BY_NAME, CASES, build_oracle_kernel, build_pool, has_golden, load_golden

SMALL = [c["name"] for c in CASES if not c["slow"] and c["N"] <= 30_000]


@pytest.fixture(autouse=True)
def _f64_default():
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)       # the reference allocates in the default dtype
    yield
    torch.set_default_dtype(prev)


def test_known_answer_survey_8c():
    """SURVEY §8c known-answer test of the reference (recorded from BASQ/_rchq.py in fp64)."""
    from oracle.kernels_oracle import direct_rbf
    from oracle.rchq_oracle import recombination_oracle

    g = torch.Generator().manual_seed(1234)
    X = torch.randn(1000, 2, generator=g, dtype=torch.float64)
    torch.manual_seed(7)
    idx, w = recombination_oracle(X, X[:50], 10, direct_rbf)
    assert idx.tolist() == [27, 48, 147, 208, 266, 307, 348, 368, 528, 586]
    ref_w = torch.tensor([0.0652065411967, 0.119271503457, 0.0984233495763, 0.10017205585, 0.167531837628,
                          0.00661903613011, 0.186993262378, 0.0118420499484, 0.194040389823, 0.0498999740124])
    assert ((w - ref_w).abs() / ref_w).max().item() < 1e-9
    assert abs(w.sum().item() - 1.0) < 1e-14


@pytest.mark.parametrize("name", SMALL)
def test_oracle_reproduces_golden(name):
    """Bit-exact: indices, weights and the per-round surviving sets of the reference."""
    from basq_amd.pools import pool_digest
    from oracle.rchq_oracle import Trace, recombination_oracle

    if not has_golden(name):
        pytest.skip("fixture not generated")
    c, fx = BY_NAME[name], load_golden(name)
    pts, nys = build_pool(c)
    assert pool_digest(pts) == fx["pool_digest"], "pool generator is not bit-reproducible on this host"
    kern, _ = build_oracle_kernel(c)
    tr = Trace()
    torch.manual_seed(c["torch_seed"])
    idx, w = recombination_oracle(pts, nys, c["n"], kern, tr)
    assert idx.tolist() == fx["idx"]
    # torch.randn / MKL are the same binaries as in the build container; allow round-off if the CPU differs
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    assert ((w - gw).abs() / gw).max().item() <= 1e-9 if len(gw) else True
    assert len(tr.rounds) == fx["n_rounds"]
    for mine, ref in zip(tr.rounds, fx["rounds"]):
        assert mine.kept_sets is None or mine.kept_sets.tolist() == ref["kept"]


def test_selection_invariant_under_basis_sign_flips():
    """The property the GPU range finder relies on: flipping the sign of rows of U changes nothing, bit for bit."""
    from oracle.kernels_oracle import StationaryOracle
    from oracle.rchq_oracle import divide_and_recombine, nystrom_basis

    c = BY_NAME["rbf_ragged"]
    pts, nys = build_pool(c)
    k = StationaryOracle("rbf", 1.5, 0.7)
    torch.manual_seed(1)
    _, U = nystrom_basis(nys, c["n"] - 1, k)
    w0, i0 = divide_and_recombine(pts, U, nys, k)
    g = torch.Generator().manual_seed(3)
    D = (torch.randint(0, 2, (U.shape[0], 1), generator=g) * 2 - 1).double()
    w1, i1 = divide_and_recombine(pts, D * U, nys, k)
    assert torch.equal(i0, i1) and torch.equal(w0, w1)


@pytest.mark.parametrize("name", ["rbf_ragged", "rbf_1e4", "matern32_8e3", "cfg1_posterior_1e4"])
def test_selection_invariant_under_basis_rotations(name):
    """Round 4: the selection does not depend on WHICH orthonormal basis of the Nystrom feature space the rows of U are.

    Left-multiplying U by an orthogonal R turns every round's matrix X = [1 ; features] into diag(1, R) X: the same Gram X^T X and
    the same first row (the ones), hence the same Golub-Kahan right vectors -- LAPACK's right Householder reflectors, whose trailing
    rows are the null-space basis the elimination pivots on, are unchanged even by the sign flips that uniqueness leaves open
    (dlarfg: (alpha, x) -> (-alpha, -x) gives the same tau and v).  So the reference's own op sequence (the oracle) returns the same points
    and the same weights (to rounding) for U and for R U -- which is why the engine may stop at the range finder's orthonormal basis Q and
    skip the [q, m] SVD of ``torch.svd_lowrank`` (``_rchq.py:29``) altogether."""
    from oracle.rchq_oracle import divide_and_recombine, nystrom_basis
    from tests.cases import build_oracle_kernel

    c = BY_NAME[name]
    pts, nys = build_pool(c)
    k, _ = build_oracle_kernel(c)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(c["torch_seed"])
        _, U = nystrom_basis(nys, c["n"] - 1, k)
        w0, i0 = divide_and_recombine(pts, U, nys, k)
        g = torch.Generator().manual_seed(5)
        R = torch.linalg.qr(torch.randn(U.shape[0], U.shape[0], generator=g, dtype=torch.float64)).Q
        w1, i1 = divide_and_recombine(pts, R @ U, nys, k)
    finally:
        torch.set_default_dtype(prev)
    assert torch.equal(i0, i1)
    assert ((w0 - w1).abs() / w0).max().item() <= 1e-8


def test_tie_margins_recorded():
    """Golden cases are well separated from pivot ties (fp64 stability margin, SURVEY finding 3)."""
    from oracle.kernels_oracle import StationaryOracle
    from oracle.rchq_oracle import Trace, recombination_oracle

    c = BY_NAME["rbf_1e4"]
    pts, nys = build_pool(c)
    tr = Trace()
    torch.manual_seed(c["torch_seed"])
    recombination_oracle(pts, nys, c["n"], StationaryOracle("rbf", 2.0), tr)
    assert tr.tie_margin > 1e-7


@pytest.mark.reference
@pytest.mark.skipif(not os.path.isdir("/root/reference/BASQ"), reason="reference not mounted")
@pytest.mark.parametrize("name", ["kat_small", "rbf_ragged", "cfg1_posterior_1e4", "matern32_8e3", "wsabim_1e4"])
def test_oracle_vs_imported_reference(name):
    """Where the reference is mounted: oracle == imported BASQ._rchq.recombination, bit for bit."""
    import sys

    sys.dont_write_bytecode = True
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from BASQ._rchq import recombination as ref_recombination

    from oracle.rchq_oracle import recombination_oracle

    c = BY_NAME[name]
    pts, nys = build_pool(c)
    kern, _ = build_oracle_kernel(c)
    torch.manual_seed(c["torch_seed"])
    i0, w0 = ref_recombination(pts, nys, c["n"], kern, torch.device("cpu"))
    torch.manual_seed(c["torch_seed"])
    i1, w1 = recombination_oracle(pts, nys, c["n"], kern)
    assert torch.equal(i0, i1) and torch.equal(w0, w1)
