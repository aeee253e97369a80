"""Unit parity of each HIP entry point (through ctypes / the C ABI) against plain torch on the CPU.

The CPU side is ``tests/cpu_stand_in.py`` (contract restatement) and ``oracle/kernels_oracle.py``
(gpytorch-style kernel values).  Integer/index outputs and the elimination (same op order as the
reference) must match bit for bit; floating-point sums within 1e-12 relative.
"""
import os
import sys
import pytest
import torch

from tests.cpu_stand_in import CpuStandInOps

pytestmark = pytest.mark.gpu


def _spec(family, d, ell=1.7, os_=1.3):
    from basq_amd.kernels import StationaryKernel

    return StationaryKernel(family, ell, os_).spec(d)


def _rand(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, d, generator=g, dtype=torch.float64) * 1.5


@pytest.mark.parametrize("family", ["rbf", "matern52", "matern32"])
@pytest.mark.parametrize("d", [1, 2, 3, 7, 10, 14, 32, 38])
def test_gram_vs_oracle(hip_ops, family, d):
    from oracle.kernels_oracle import StationaryOracle

    x, y = _rand(77, d, 1), _rand(131, d, 2)
    spec = _spec(family, d)
    c = hip_ops.col_mean(hip_ops.to_device(x))
    pa = hip_ops.pack(spec, hip_ops.to_device(x), c, 0)
    pb = hip_ops.pack(spec, hip_ops.to_device(y), c, 1)
    K = hip_ops.gram(spec, pa, 77, pb, 131).cpu()
    ref = StationaryOracle(family, spec.lengthscale, spec.outputscale)(x, y)
    # values agree to fp64 round-off of the squared distance (absolute error ~1e-14 in the exponent)
    assert (K - ref).abs().max().item() <= 5e-13 * ref.abs().max().item()


def test_gram_extreme_distances(hip_ops):
    """exp underflow region: far-apart points give 0 (not NaN/inf), identical points give outputscale."""
    spec = _spec("rbf", 3, ell=0.01, os_=2.0)
    x = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.0, 0.0, 0.0]], dtype=torch.float64)
    xd = hip_ops.to_device(x)
    pa = hip_ops.pack(spec, xd, None, 0)
    pb = hip_ops.pack(spec, xd, None, 1)
    K = hip_ops.gram(spec, pa, 3, pb, 3).cpu()
    assert torch.isfinite(K).all()
    assert K[0, 0].item() == pytest.approx(2.0, rel=1e-14) and K[0, 2].item() == pytest.approx(2.0, rel=1e-14)
    assert K[0, 1].item() < 1e-300


@pytest.mark.parametrize("family,d,m,S,Rl,off,n_full,n_chunks,use_wx", [
    ("rbf", 10, 100, 200, 5000, 0, 5000, 1, False),          # exact blocks
    ("rbf", 10, 130, 200, 5077, 0, 5000, 3, False),          # tail + chunks, m not multiple of 64
    ("rbf", 2, 50, 22, 1000, 0, 990, 2, True),               # S not multiple of 16, weights
    ("matern52", 32, 64, 400, 3000, 0, 2800, 2, False),
    ("matern32", 5, 33, 102, 777, 0, 714, 1, False),
    ("rbf", 10, 100, 200, 2500, 1234, 5000, 2, False),       # a middle shard of a 2-rank split
    ("rbf", 10, 100, 200, 1311, 3766, 5000, 2, True),        # the last shard: block part + tail
    ("rbf", 3, 70, 150, 150, 0, 150, 1, False),              # final stage: one block, S = R
    ("rbf", 3, 70, 40, 37, 4003, 4000, 1, False),            # shard holding only tail positions
    ("rbf", 14, 600, 200, 4100, 100, 4000, 3, True),         # KP = 16, two row groups, shard starting mid-block, tail
    ("matern52", 18, 257, 74, 1000, 0, 962, 2, False),       # KP = 20 (largest LDS form), odd S, m = 256 + 1
    ("rbf", 10, 1000, 200, 1000, 0, 1000, 1, False),         # 5 blocks only: the tile pipeline's short trip counts
])
def test_blocksum_vs_standin(hip_ops, family, d, m, S, Rl, off, n_full, n_chunks, use_wx):
    cpu = CpuStandInOps()
    spec = _spec(family, d)
    nys, cand = _rand(m, d, 3), _rand(Rl, d, 4)
    g = torch.Generator().manual_seed(5)
    mu = torch.rand(Rl, generator=g, dtype=torch.float64) + 0.1
    wx = (torch.rand(Rl, generator=g, dtype=torch.float64) + 0.5) if use_wx else None
    center = nys.mean(0)
    A_c = cpu.pack(spec, nys, center, 0, pad_rows_to=64)
    B_c = cpu.pack(spec, cand, center, 1)
    Xc, tc = cpu.blocksum(spec, A_c, m, B_c, mu, wx, Rl, off, n_full, S, n_chunks)
    dev = hip_ops.to_device
    A_g = hip_ops.pack(spec, dev(nys), dev(center), 0, pad_rows_to=64)
    B_g = hip_ops.pack(spec, dev(cand), dev(center), 1)
    scale = Xc.abs().max().item()
    Xg, tg = hip_ops.blocksum(spec, A_g, m, B_g, dev(mu), dev(wx) if use_wx else None, Rl, off, n_full, S, n_chunks)
    Xg, tg = Xg.cpu(), tg.cpu()
    assert (Xg - Xc).abs().max().item() <= 1e-12 * scale
    assert (tg - tc).abs().max().item() <= 1e-13 * tc.abs().max().item()


@pytest.mark.parametrize("accurate", [False, True])
@pytest.mark.parametrize("family", ["rbf", "matern52", "matern32"])
def test_blocksum_single_values_over_the_whole_exponent_range(hip_ops, family, accurate):
    """The block sums' exponential takes its argument ALREADY SCALED from the matrix instruction (v_cvt_i32_f64 + v_fract_f64 of
    |y|; basq_pairwise.hip ``exp_scaled_k``): one candidate per set (S = R) turns every entry of the block sums into ONE kernel
    value, checked against float64 formulas evaluated on the raw points (direct differences, no expansion) over the whole range --
    coincident points (the product comes out as +-1e-16 |x|^2: both signs must give 1), values down to the denormals, exact zeros
    beyond -745 -- and against ``gpytorch``'s formulas (``BASQ/_parameters.py:192-208`` picks the kernels)."""
    import dataclasses

    d, ell = 3, 1.3
    spec = dataclasses.replace(_spec(family, d, ell=ell), accurate_exp=accurate)     # 2048-entry table + cubic (posterior kernels)
    g = torch.Generator().manual_seed(11)
    nys = torch.randn(64, d, generator=g, dtype=torch.float64) * 2.0
    dirs = torch.randn(96, d, generator=g, dtype=torch.float64)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    # distances (in lengthscales) from 0 to 60: -r^2/2 down to -1800 (RBF), -sqrt(5) r down to -134 (Matern)
    r = torch.cat([torch.zeros(32, dtype=torch.float64), torch.linspace(1e-9, 60.0, 64, dtype=torch.float64)])
    cand = nys[torch.arange(96) % 64] + dirs * (r * ell)[:, None]        # candidates 0..31 ARE Nystrom points 0..31
    R = S = 96
    mu = torch.ones(R, dtype=torch.float64)
    center = nys.mean(0)
    dev = hip_ops.to_device
    A_g = hip_ops.pack(spec, dev(nys), dev(center), 0, pad_rows_to=64)
    B_g = hip_ops.pack(spec, dev(cand), dev(center), 1)
    X, _ = hip_ops.blocksum(spec, A_g, 64, B_g, dev(mu), None, R, 0, R, S, 1)
    X = X.cpu().reshape(64, S)
    diff = (nys[:, None, :] - cand[None, :, :]) / ell
    r2 = (diff * diff).sum(-1)
    if family == "rbf":
        arg, want = -0.5 * r2, torch.exp(-0.5 * r2)
    else:
        rr = r2.clamp_min(1e-30).sqrt()
        c = 5.0 ** 0.5 if family == "matern52" else 3.0 ** 0.5
        poly = (1.0 + c * rr + (5.0 / 3.0) * r2) if family == "matern52" else (1.0 + c * rr)
        arg, want = -c * rr, poly * torch.exp(-c * rr)
    # what the expanded product x.y + h_x + h_y can lose: ~13 roundings at the size of its terms
    size = ((nys - center) / ell).pow(2).sum(1)[:, None] + ((cand - center) / ell).pow(2).sum(1)[None, :]
    tol = (2e-15 if accurate else 1e-13) + 4e-15 * size + 1e-15 * arg.abs()
    normal = want > 1e-290
    rel = ((X - want).abs() / want.clamp_min(1e-300))[normal]
    assert (rel <= tol[normal]).all(), (rel / tol[normal]).max().item()
    assert (X[~normal] <= 1e-289).all() and (X[~normal] >= 0).all()
    if family == "rbf":
        assert (X[arg < -746.0] == 0.0).all()                           # exact zeros, no denormal garbage
    # coincident points: exactly the pairs (j, j) for j < 32
    same = X[torch.arange(32), torch.arange(32)]
    assert (same - 1.0).abs().max().item() <= 1e-13


def test_matvec_vs_standin(hip_ops):
    cpu = CpuStandInOps()
    spec = _spec("rbf", 10)
    x, xo = _rand(1000, 10, 6), _rand(202, 10, 7)
    v = _rand(202, 1, 8).reshape(-1)
    center = x.mean(0)
    ref = cpu.matvec(spec, cpu.pack(spec, x, center, 0), 1000, cpu.pack(spec, xo, center, 1), 202, v, 0.7)
    dev = hip_ops.to_device
    out = hip_ops.matvec(spec, hip_ops.pack(spec, dev(x), dev(center), 0, pad_rows_to=64), 1000,
                         hip_ops.pack(spec, dev(xo), dev(center), 1), 202, dev(v), 0.7).cpu()
    assert (out - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()


@pytest.mark.parametrize("q,m,S,n_chunks", [(99, 1000, 200, 1), (99, 1030, 200, 3), (9, 50, 20, 2), (199, 333, 400, 1),
                                            (30, 77, 62, 1)])
def test_project_finalize_vs_standin(hip_ops, q, m, S, n_chunks):
    cpu = CpuStandInOps()
    U = _rand(q, m, 9)
    Xpart = _rand(n_chunks * m, S, 10).reshape(n_chunks, m, S)
    totpart = _rand(n_chunks, S, 11).abs() + 0.2
    ref = cpu.project(U, q, m, Xpart, totpart, n_chunks, S, 1.3)
    dev = hip_ops.to_device
    out = hip_ops.project(dev(U), q, m, dev(Xpart), dev(totpart), n_chunks, S, 1.3)
    assert (out.cpu() - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()
    # finalize: 2 "ranks", diagonal-noise term on
    parts_c = torch.stack([ref, 0.5 * ref])
    Xc, tc = cpu.finalize(parts_c, 2, q + 1, q, S, U, m, min(m, S), 1e-3, 0)
    Xg, tg = hip_ops.finalize(dev(parts_c), 2, q + 1, q, S, dev(U), m, min(m, S), 1e-3, 0)
    assert (Xg.cpu() - Xc).abs().max().item() <= 1e-13 * Xc.abs().max().item()
    assert torch.equal(tg.cpu(), tc)


# (dispatch of basq_car_eliminate_f64: rows in registers handed over in blocks of 7 -- 200/100, 180/80, 130/20, 70/5 -- or of
# 4 -- 150/100, 65/1, 62/31, 20/10, 101/100; several work-groups and a global ring -- 400/200 ...; rows in LDS -- 256/144, 230/130)
@pytest.mark.parametrize("M,s,seed", [(200, 100, 0), (200, 100, 1), (400, 200, 2), (150, 100, 3), (20, 10, 4),
                                      (62, 31, 5), (101, 100, 6), (180, 80, 7), (130, 20, 8), (70, 5, 9), (65, 1, 10),
                                      (256, 144, 11), (230, 130, 12), (256, 156, 13),
                                      # several work-groups, pivots through a ring of tagged granules in global memory
                                      (300, 150, 14), (320, 100, 15), (448, 248, 16), (260, 259, 17), (400, 150, 18),
                                      (512, 256, 19)])                                    # (8 column slots: the cluster kernel)
def test_car_eliminate_bit_exact(hip_ops, M, s, seed):
    """Same null-space basis in -> same pivots, and bit-identical weights (reference op order)."""
    cpu = CpuStandInOps()
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(s, M, generator=g, dtype=torch.float64)
    X[0] = 1.0
    Vh = torch.linalg.svd(X)[2]
    PhiT = Vh[-(M - s):, :].contiguous()
    mu = torch.rand(M, generator=g, dtype=torch.float64) + 0.05
    mu = mu / mu.sum()
    kr_c, kept_c, w_c, info_c = cpu.car_eliminate(PhiT.clone(), mu.clone(), M, s)
    mu_g = hip_ops.to_device(mu.clone())
    kr_g, kept_g, w_g, info_g = hip_ops.car_eliminate(hip_ops.to_device(PhiT.clone()), mu_g, M, s)
    n = int(info_c[0])
    assert info_g.cpu().tolist() == info_c.tolist()
    assert torch.equal(kr_g.cpu(), kr_c)
    assert torch.equal(kept_g.cpu()[:n], kept_c[:n])
    assert torch.equal(w_g.cpu()[:n], w_c[:n])          # bit-exact
    assert n <= s


# dispatch paths of basq_nullspace_f64: rows in registers+LDS (100x200 ...), rows in global memory with register
# rows (128x256), 8 columns per lane (200x400), 16 columns per lane / 8 waves (300x600), degenerate sizes.
NULLSPACE_SHAPES = [(100, 200), (100, 150), (100, 101), (37, 74), (31, 57), (10, 20), (1, 2), (2, 5), (33, 66),
                    (128, 256), (200, 400), (150, 500), (300, 600), (100, 1024)]


@pytest.mark.parametrize("s,M", NULLSPACE_SHAPES)
def test_nullspace_equals_lapack_svd_rows(hip_ops, s, M):
    """``basq_nullspace_f64`` returns the very rows ``torch.linalg.svd`` returns (``_rchq.py:140-143``): no sign or
    rotation freedom, agreement at rounding level."""
    from tests.cpu_stand_in import householder_nullspace

    g = torch.Generator().manual_seed(s * 1000 + M)
    X = torch.randn(s, M, generator=g, dtype=torch.float64)
    X[0] = 1.0
    ref = torch.linalg.svd(X)[2][s:]
    got = hip_ops.nullspace(hip_ops.to_device(X), s, M).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 1e-11
    assert (got - householder_nullspace(X)).abs().max().item() <= 1e-11     # same reflectors, other summation order
    assert (X @ got.T).abs().max().item() <= 1e-12 * M


def test_nullspace_zero_tail_row(hip_ops):
    """A row whose tail is exactly zero gives tau = 0 (dlarfg's early exit), not a NaN."""
    s, M = 4, 8
    X = torch.zeros(s, M, dtype=torch.float64)
    X[0, 0] = 2.0
    X[1, 1] = -3.0
    X[2, 2] = 1.0
    X[3, 3] = 5.0
    got = hip_ops.nullspace(hip_ops.to_device(X), s, M).cpu()
    assert torch.isfinite(got).all()
    assert torch.equal(got, torch.eye(M, dtype=torch.float64)[s:])


@pytest.mark.parametrize("M,s,seed", [(200, 100, 0), (400, 200, 2), (150, 100, 3), (62, 31, 5)])
def test_nullspace_then_eliminate_matches_host_svd_route(hip_ops, M, s, seed):
    """GPU null space + GPU elimination keeps the same sets as host LAPACK SVD + the CPU elimination."""
    cpu = CpuStandInOps()
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(s, M, generator=g, dtype=torch.float64)
    X[0] = 1.0
    PhiT = torch.linalg.svd(X)[2][-(M - s):, :].contiguous()
    mu = torch.rand(M, generator=g, dtype=torch.float64) + 0.05
    mu = mu / mu.sum()
    _, kept_c, w_c, info_c = cpu.car_eliminate(PhiT.clone(), mu.clone(), M, s)
    Pg = hip_ops.nullspace(hip_ops.to_device(X), s, M)
    _, kept_g, w_g, info_g = hip_ops.car_eliminate(Pg, hip_ops.to_device(mu.clone()), M, s)
    n = int(info_c[0])
    assert info_g.cpu().tolist() == info_c.tolist()
    assert torch.equal(kept_g.cpu()[:n], kept_c[:n])
    assert ((w_g.cpu()[:n] - w_c[:n]).abs() / w_c[:n]).max().item() <= 1e-8


def test_car_eliminate_degenerate_flag(hip_ops):
    """A null vector without a positive entry sets status=1 (the reference raises at _rchq.py:152)."""
    M, s = 8, 4
    PhiT = -torch.ones(M - s, M, dtype=torch.float64)
    mu = torch.full((M,), 1.0 / M, dtype=torch.float64)
    _, _, _, info = hip_ops.car_eliminate(hip_ops.to_device(PhiT), hip_ops.to_device(mu), M, s)
    assert int(info.cpu()[1]) == 1


@pytest.mark.parametrize("M,s", [(200, 100), (150, 100), (400, 200), (256, 144)])
def test_car_eliminate_zero_and_negative_weights_bit_exact(hip_ops, M, s):
    """Weights that are exactly zero (ratio +0: ties resolved by the first index) or slightly negative (a ratio below zero wins
    the test, as in the reference) -- the order keys of the register kernels' wave minimum must sort them like the IEEE compare."""
    cpu = CpuStandInOps()
    g = torch.Generator().manual_seed(1000 + M)
    X = torch.randn(s, M, generator=g, dtype=torch.float64)
    X[0] = 1.0
    PhiT = torch.linalg.svd(X)[2][-(M - s):, :].contiguous()
    mu = torch.rand(M, generator=g, dtype=torch.float64) + 0.05
    mu[torch.randperm(M, generator=g)[:M // 8]] = 0.0
    mu[torch.randperm(M, generator=g)[:M // 16]] = -1e-18
    kr_c, kept_c, w_c, info_c = cpu.car_eliminate(PhiT.clone(), mu.clone(), M, s)
    kr_g, kept_g, w_g, info_g = hip_ops.car_eliminate(hip_ops.to_device(PhiT.clone()), hip_ops.to_device(mu.clone()), M, s)
    n = int(info_c[0])
    assert info_g.cpu().tolist() == info_c.tolist()
    assert torch.equal(kr_g.cpu(), kr_c)
    assert torch.equal(kept_g.cpu()[:n], kept_c[:n])
    assert torch.equal(w_g.cpu()[:n], w_c[:n])


@pytest.mark.parametrize("M,s", [(200, 100), (300, 150), (400, 200)])
def test_car_eliminate_degenerate_flag_in_a_later_block(hip_ops, M, s):
    """The same flag when the null vector without a positive entry belongs to a LATER block of the register kernels (one
    work-group: M = 200; several work-groups and the global ring: M = 300, 400): the waves that only consume must see the
    producer's give-up and stop, and the wave that writes the outcome must report status 1."""
    g = torch.Generator().manual_seed(M)
    X = torch.randn(s, M, generator=g, dtype=torch.float64)
    X[0] = 1.0
    PhiT = torch.linalg.svd(X)[2][-(M - s):, :].contiguous()
    bad_row = (M - s) // 2 + 3
    PhiT[bad_row:] = -PhiT[bad_row:].abs() - 1.0         # (every later row: whichever of them the updates leave negative trips)
    mu = torch.full((M,), 1.0 / M, dtype=torch.float64)
    cpu = CpuStandInOps()
    _, _, _, info_c = cpu.car_eliminate(PhiT.clone(), mu.clone(), M, s)
    _, _, _, info = hip_ops.car_eliminate(hip_ops.to_device(PhiT), hip_ops.to_device(mu), M, s)
    assert int(info_c[1]) == 1
    assert info.cpu().tolist() == info_c.tolist()        # status 1 and the same number of positive weights at the stop


@pytest.mark.parametrize("Rl,off,n_full,S,kept", [
    (5000, 0, 5000, 200, list(range(0, 200, 2))),
    (5077, 0, 5000, 200, list(range(1, 200, 2))),            # last set kept -> tail survives
    (5077, 0, 5000, 200, list(range(0, 198, 2))),            # last set dropped
    (2500, 1234, 5000, 200, [3, 4, 50, 199]),
    (1311, 3766, 5000, 200, [0, 7, 199]),
])
def test_reweight_compact_vs_standin(hip_ops, Rl, off, n_full, S, kept):
    from basq_amd._partition import RoundGeometry, next_shard

    cpu = CpuStandInOps()
    kp = 12
    R = max(off + Rl, n_full)
    geo = RoundGeometry(R, S, n_full // S, n_full, R - n_full)
    new_off, new_Rl = next_shard(off, Rl, geo, kept)
    g = torch.Generator().manual_seed(1)
    cand = torch.randn(Rl, kp, generator=g, dtype=torch.float64)
    mu = torch.rand(Rl, generator=g, dtype=torch.float64) + 0.1
    wx = torch.rand(Rl, generator=g, dtype=torch.float64)
    gid = torch.arange(Rl, dtype=torch.int64) * 3 + 7
    keep_rank = torch.full((S,), -1, dtype=torch.int32)
    keep_rank[torch.tensor(kept)] = torch.arange(len(kept), dtype=torch.int32)
    w_star = torch.rand(len(kept), generator=g, dtype=torch.float64) + 0.1
    tot = torch.rand(S, generator=g, dtype=torch.float64) + 0.5
    ref = cpu.reweight_compact(cand, mu, gid, wx, Rl, off, n_full, S, kp, keep_rank, w_star, tot, len(kept), new_off, new_Rl)
    dev = hip_ops.to_device
    out = hip_ops.reweight_compact(dev(cand), dev(mu), dev(gid), dev(wx), Rl, off, n_full, S, kp, dev(keep_rank),
                                   dev(w_star), dev(tot), len(kept), new_off, new_Rl)
    for a, b in zip(out, ref):
        assert torch.equal(a.cpu()[:new_Rl], b[:new_Rl])     # bit-exact (multiply, then divide)


def test_gemm_f64(hip_ops):
    A, B = _rand(300, 257, 12), _rand(257, 99, 13)
    C = hip_ops.gemm(hip_ops.to_device(A), hip_ops.to_device(B), 0.5).cpu()
    ref = 0.5 * (A @ B)
    assert (C - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()


def test_no_cpu_path():
    """The product refuses CPU devices instead of silently computing there."""
    import basq_amd
    from basq_amd._lib import BasqHipError

    with pytest.raises(BasqHipError):
        basq_amd.recombination(torch.zeros(10, 2), torch.zeros(5, 2), 3, basq_amd.kernels.StationaryKernel("rbf", 1.0),
                               torch.device("cpu"))


@pytest.mark.parametrize("q", [1, 5, 8, 9, 30, 99, 100, 101, 137, 142, 143, 199, 200, 201, 300])
def test_chol_inv(hip_ops, q):
    """CholeskyQR building block: L L^T = G and (X W)^T (X W) = I.  Sizes cover the blocked inverse (8 <= q <= 100,
    incl. block sizes that do not divide q), the column-per-thread inverse in LDS (q <= 142), the packed-triangle
    factorisation + library triangular solve (143..200) and the global kernel."""
    X = _rand(4 * q + 7, q, 20 + q)
    G = X.T @ X
    Gd = hip_ops.to_device(G.clone())
    W, info = hip_ops.chol_inv(Gd)
    assert int(info.cpu()[0]) == 0
    L = torch.tril(Gd.cpu())
    assert (L @ L.T - G).abs().max().item() <= 1e-12 * G.abs().max().item()
    Q = X @ W.cpu()
    assert (Q.T @ Q - torch.eye(q, dtype=torch.float64)).abs().max().item() <= 1e-10
    assert torch.equal(torch.tril(W.cpu(), -1), torch.zeros(q, q, dtype=torch.float64))
    Winv = torch.linalg.inv(L).T                               # W = L^{-T}
    assert (W.cpu() - Winv).abs().max().item() <= 1e-9 * Winv.abs().max().item()


@pytest.mark.parametrize("q", [1, 5, 8, 9, 16, 30, 99, 100, 143, 199, 200])
@pytest.mark.parametrize("rows", [1000, 64, 37])
def test_chol_factor_and_trsm_rows(hip_ops, q, rows):
    """Panel Cholesky (packed triangle in LDS, 8-column panels) + row-parallel triangular solve: X L^-T is orthonormal
    and equals the host factorisation's result; the strict upper triangle of G is left alone."""
    rows = max(rows, q + 3)
    X = _rand(rows, q, 7 + q)
    G = X.T @ X
    Gd = hip_ops.to_device(G.clone())
    info = hip_ops.chol_factor(Gd)
    assert int(info.cpu()[0]) == 0
    L = torch.tril(Gd.cpu())
    Lref = torch.linalg.cholesky(G)
    assert (L - Lref).abs().max().item() <= 1e-11 * Lref.abs().max().item()
    assert torch.equal(torch.triu(Gd.cpu(), 1), torch.triu(G, 1))
    Q = hip_ops.trsm_rows(hip_ops.to_device(X), Gd).cpu()
    assert (Q.T @ Q - torch.eye(q, dtype=torch.float64)).abs().max().item() <= 1e-10
    Qref = torch.linalg.solve_triangular(Lref, X.T, upper=False).T
    assert (Q - Qref).abs().max().item() <= 1e-10 * Qref.abs().max().item()


def test_chol_factor_flags_rank_deficiency(hip_ops):
    X = _rand(50, 12, 3)
    X[:, 10] = X[:, 0] + X[:, 1]             # exactly dependent column (second panel)
    info = hip_ops.chol_factor(hip_ops.to_device(X.T @ X))
    assert int(info.cpu()[0]) == 11


def test_chol_inv_flags_rank_deficiency(hip_ops):
    X = _rand(50, 8, 3)
    X[:, 7] = X[:, 0] + X[:, 1]              # exactly dependent column
    _, info = hip_ops.chol_inv(hip_ops.to_device(X.T @ X))
    assert int(info.cpu()[0]) == 8


@pytest.mark.parametrize("m,nc,pg0,n_full,S", [(70, 1000, 0, 1000, 50), (70, 777, 1234, 1800, 60), (33, 150, 4003, 4000, 40),
                                               (10, 500, 100, 396, 22),
                                               # the 16-bytes-per-lane form (S and pg0 even, nc >= 4 S): slices, split last pair,
                                               # rows not a multiple of 8, odd row stride, long tail, one slice only (S = 2000)
                                               (19, 12345, 400, 12000, 200), (50, 5001, 800, 5600, 400), (9, 9000, 0, 8000, 2000),
                                               (17, 4097, 2, 3996, 6), (8, 1601, 10, 1600, 400), (23, 3000, 1234, 1200, 100)])
@pytest.mark.parametrize("square", [False, True])
def test_dense_blocksum_vs_standin(hip_ops, m, nc, pg0, n_full, S, square):
    """Dense block sums (opaque-callable path; squared: WSABI-M), incl. a chunk that starts mid-block, one that lies
    wholly in the tail, and a row-strided view of a wider buffer."""
    cpu = CpuStandInOps()
    Cm = _rand(m, nc, 31)
    g = torch.Generator().manual_seed(2)
    mu = torch.rand(nc, generator=g, dtype=torch.float64) + 0.1
    E0 = _rand(m, S, 32)
    Ec = E0.clone()
    cpu.dense_blocksum(Cm, mu, pg0, n_full, S, 0.5, Ec, square=square)
    Eg = hip_ops.to_device(E0.clone())
    wide = hip_ops.zeros(m, nc + 7)
    wide[:, :nc] = hip_ops.to_device(Cm)
    T0 = _rand(S, 1, 33).reshape(-1).contiguous()
    Tc = T0.clone()
    cpu.dense_blocksum(torch.ones(1, nc, dtype=torch.float64), mu, pg0, n_full, S, 1.0, Tc.reshape(1, -1))
    Tg = hip_ops.to_device(T0.clone())
    hip_ops.dense_blocksum(wide[:, :nc], hip_ops.to_device(mu), pg0, n_full, S, 0.5, Eg, square=square, tot=Tg)
    assert (Eg.cpu() - Ec).abs().max().item() <= 1e-12 * Ec.abs().max().item()
    assert (Tg.cpu() - Tc).abs().max().item() <= 1e-12 * Tc.abs().max().item()       # the set weights of the same launch


@pytest.mark.parametrize("family,d,m,n_obs,Rl,off,n_full,S,n_chunks,noise", [
    ("rbf", 10, 300, 202, 5000, 0, 4800, 200, 3, 0.0),          # WSABI-M shape: n_obs not a multiple of 4, ragged tail
    ("rbf", 4, 70, 50, 2321, 0, 2300, 50, 2, 1e-2),             # noise on [kappa][kappa] of blocks and tail
    ("matern52", 8, 130, 61, 1500, 700, 2800, 100, 1, 1e-3),    # a middle shard (no tail, starts mid-block)
    ("rbf", 6, 64, 8, 999, 1200, 2000, 40, 4, 1e-3),            # the last shard: blocks + the whole tail
    ("matern32", 24, 90, 33, 777, 0, 770, 70, 2, 0.0),          # KK = 7 (two row tiles per wave)
    ("rbf", 5, 40, 12, 37, 0, 0, 37, 1, 1e-2),                  # the final round: every point its own set, all tail
])
def test_blocksum_sq_vs_standin(hip_ops, family, d, m, n_obs, Rl, off, n_full, S, n_chunks, noise):
    """Fused WSABI-M term (basq_blocksum_sq_f64) vs the stand-in's dense covariance blocks."""
    cpu = CpuStandInOps()
    spec = _spec(family, d)
    nys, obs, cand = _rand(m, d, 51), _rand(n_obs, d, 52), _rand(Rl, d, 53)
    center = nys.mean(0)
    g = torch.Generator().manual_seed(7)
    mu = torch.rand(Rl, generator=g, dtype=torch.float64) + 0.05
    Bm = 0.1 * _rand(m, n_obs, 54)
    n4, mp = (n_obs + 3) // 4 * 4, (m + 63) // 64 * 64

    def run(ops):
        pa = ops.pack(spec, ops.to_device(torch.cat([nys, obs], 0)), ops.to_device(center), 0, pad_rows_to=64)
        pb = ops.pack(spec, ops.to_device(cand), ops.to_device(center), 1)
        bT = ops.zeros(n4, mp)
        bT[:n_obs, :m] = ops.to_device(Bm).t()
        kobs = ops.zeros(n4, Rl)
        ops.gram_into(spec, pa[m:m + n_obs], n_obs, pb, Rl, kobs)
        return ops.blocksum_sq(spec, pa, m, pb, ops.to_device(mu), Rl, off, n_full, S, n_chunks, bT, kobs, n_obs, noise)

    Ec = run(cpu).sum(0)
    Eg = run(hip_ops).sum(0).cpu()
    assert (Eg - Ec).abs().max().item() <= 2e-12 * Ec.abs().max().item()


@pytest.mark.parametrize("family,d,m,n_obs,nb,S,C,off_blocks,noise", [
    ("rbf", 10, 300, 202, 64, 200, 16, 0, 1e-3),               # an epoch start: 16 classes of a 64-block region
    ("matern52", 6, 130, 61, 37, 50, 8, 3, 1e-2),               # a shard that starts at block 3; blocks not a multiple of C
    ("rbf", 4, 70, 50, 9, 30, 2, 0, 0.0),                       # C = 2, odd block count
])
def test_blocksum_sq_classes_and_cov_diag_vs_standin(hip_ops, family, d, m, n_obs, nb, S, C, off_blocks, noise):
    """Round 4: WSABI-M's squared covariance per residue class (``basq_blocksum_sq_f64`` with class_mod: chunk c = the blocks
    b = c mod C) and the per-candidate noise cross terms (``basq_cov_diag_f64``) against the stand-in; the class sums add up to
    the contiguous-chunk result of the same range."""
    cpu = CpuStandInOps()
    spec = _spec(family, d)
    Rl, off = nb * S, off_blocks * S
    n_full = off + Rl
    nys, obs, cand = _rand(m, d, 61), _rand(n_obs, d, 62), _rand(Rl, d, 63)
    center = nys.mean(0)
    g = torch.Generator().manual_seed(9)
    mu = torch.rand(Rl, generator=g, dtype=torch.float64) + 0.05
    Bm = 0.1 * _rand(m, n_obs, 64)
    n4, mp = (n_obs + 3) // 4 * 4, (m + 63) // 64 * 64

    def run(ops):
        pa = ops.pack(spec, ops.to_device(torch.cat([nys, obs], 0)), ops.to_device(center), 0, pad_rows_to=64)
        pb = ops.pack(spec, ops.to_device(cand), ops.to_device(center), 1)
        bT = ops.zeros(n4, mp)
        bT[:n_obs, :m] = ops.to_device(Bm).t()
        kobs = ops.zeros(n4, Rl)
        ops.gram_into(spec, pa[m:m + n_obs], n_obs, pb, Rl, kobs)
        mud = ops.to_device(mu)
        Ecls = ops.blocksum_sq(spec, pa, m, pb, mud, Rl, off, n_full, S, C, bT, kobs, n_obs, 0.0, class_mod=C, class0=0)
        Eall = ops.blocksum_sq(spec, pa, m, pb, mud, Rl, off, n_full, S, 1, bT, kobs, n_obs, 0.0)
        val = ops.cov_diag(spec, pa, m, pb, Rl + 0, off, n_full, S, bT, kobs, n_obs, noise if noise else 1e-3)
        # a shard that ends inside the ragged remainder: the last S - 3 candidates renumbered as remainder points 0..
        tail = ops.cov_diag(spec, pa, m, pb[Rl - (S - 3):], S - 3, n_full, n_full, S, bT, kobs[:, Rl - (S - 3):], n_obs, 1e-3)
        return Ecls, Eall, val, tail

    Ec, Eca, vc, tc = run(cpu)
    Eg, Ega, vg, tg = (t.cpu() for t in run(hip_ops))
    scale = Eca.abs().max().item()
    assert (Eg - Ec).abs().max().item() <= 2e-12 * scale
    assert (Eg.sum(0) - Ega[0]).abs().max().item() <= 1e-11 * scale       # the classes tile the range
    assert (vg[:Rl] - vc[:Rl]).abs().max().item() <= 1e-12 * vc.abs().max().item()
    assert (tg[:S - 3] - tc[:S - 3]).abs().max().item() <= 1e-12 * tc.abs().max().item()


@pytest.mark.parametrize("R,S,reg_blocks,C,shard,sober", [
    (10_123, 200, 48, 16, None, False),                  # one rank, ragged remainder of 123
    (10_123, 200, 48, 16, (3_333, 3_400), True),         # a shard in the regular region; SOBER's extra remainder count (empty here)
    (10_123, 200, 48, 16, (6_100, 4_023), True),         # a shard that holds the end of the regular region and the remainder
    (4_060, 58, 64, 8, (4_001, 59), False),              # a shard inside the ragged remainder only
    (1_000, 50, 16, 4, (0, 1_000), False),               # no remainder (R a multiple of S)
])
def test_wsabim_descriptor_kernels_vs_host_geometry(hip_ops, R, S, reg_blocks, C, shard, sober):
    """Round 5: WSABI-M in the descriptor-driven rounds -- ``basq_blocksum_sq_geo_f64`` (modes 1-4), ``basq_cov_diag_geo_f64``
    and ``basq_sq_noise_part_geo_f64`` return what the host-geometry entries return for the same numbers (bit for bit where the
    same kernel runs), with buffers sized for an upper bound; the noise part against the CPU stand-in's closed form."""
    cpu = CpuStandInOps()
    d, m, n_obs, q = 5, 130, 41, 24
    spec = _spec("rbf", d)
    off, Rl = shard if shard is not None else (0, R)
    nys, obs, cand = _rand(m, d, 71), _rand(n_obs, d, 72), _rand(R, d, 73)[off:off + Rl]
    g = torch.Generator().manual_seed(11)
    mu = (torch.rand(R, generator=g, dtype=torch.float64) + 0.05)[off:off + Rl]
    Bm = 0.1 * _rand(m, n_obs, 74)
    U = _rand(q, m, 75)
    n4, mp = (n_obs + 3) // 4 * 4, (m + 63) // 64 * 64
    center = hip_ops.to_device(nys.mean(0))
    pa = hip_ops.pack(spec, hip_ops.to_device(torch.cat([nys, obs], 0)), center, 0, pad_rows_to=64)
    R_max = Rl + 333
    pb = hip_ops.zeros(R_max, hip_ops.kp(d))
    pb[:Rl] = hip_ops.pack(spec, hip_ops.to_device(cand), center, 1)
    mu_d = hip_ops.zeros(R_max)
    mu_d[:Rl] = hip_ops.to_device(mu)
    bT = hip_ops.zeros(n4, mp)
    bT[:n_obs, :m] = hip_ops.to_device(Bm).t()
    kobs = hip_ops.zeros(n4, R_max)
    hip_ops.gram_into(spec, pa[m:m + n_obs], n_obs, pb, R_max, kobs)
    nb = R // S
    n_full, reg_hi = nb * S, reg_blocks * S
    geo = hip_ops.geo_init(2, R, S, reg_hi, off, Rl)
    noise = 1e-2

    def host_range(lo, hi, n_ch, renumber=False, **kw):
        lo, hi = max(lo, off), min(hi, off + Rl)
        hi = max(hi, lo)
        sk = lo - off
        o, nf = (lo - n_full, S) if renumber else (lo, n_full)
        return hip_ops.blocksum_sq(spec, pa, m, pb[sk:], mu_d[sk:], hi - lo, o, nf, S, n_ch, bT, kobs[:, sk:], n_obs,
                                   kw.pop("noise", 0.0), **kw)

    Ea = hip_ops.blocksum_sq_geo(spec, pa, m, pb, mu_d, geo[0], 1, S, C, bT, kobs, n_obs, 0.0, class_mod=C)
    assert torch.equal(Ea, host_range(0, reg_hi, C, class_mod=C))
    Ea = hip_ops.blocksum_sq_geo(spec, pa, m, pb, mu_d, geo[0], 2, S, 1, bT, kobs, n_obs, 0.0)
    assert torch.equal(Ea, host_range(reg_hi, R, 1))
    Ea = hip_ops.blocksum_sq_geo(spec, pa, m, pb, mu_d, geo[0], 3, S, 3, bT, kobs, n_obs, noise)
    assert torch.equal(Ea, host_range(0, R, 3, noise=noise))
    Ea = hip_ops.blocksum_sq_geo(spec, pa, m, pb, mu_d, geo[0], 4, S, 1, bT, kobs, n_obs, noise)
    assert torch.equal(Ea, host_range(n_full, R, 1, renumber=True, noise=noise))
    # the per-candidate noise cross terms and the message part they form
    va = hip_ops.cov_diag_geo(spec, pa, m, pb, geo[0], R_max, S, bT, kobs, n_obs, noise)
    vb = hip_ops.cov_diag(spec, pa, m, pb, Rl, off, n_full, S, bT, kobs, n_obs, noise)
    assert torch.equal(va[:Rl], vb[:Rl])
    rows = q + 2
    part = hip_ops.zeros(rows, S) - 7.0                          # (poisoned: the kernel writes the whole part)
    hip_ops.sq_noise_part_geo(mu_d, va, geo[0], hip_ops.to_device(U), q, m, S, rows, sober, part)
    want = torch.zeros(rows, S, dtype=torch.float64)
    cpu.sq_noise_part_geo(mu, va[:max(Rl, 1)].cpu(), hip_ops.to_host(geo[0], "g0").clone(), U, q, m, S, rows, sober, want)
    assert (part.cpu() - want).abs().max().item() <= 1e-13 * max(want.abs().max().item(), 1e-300)
    assert float(part[0].abs().max()) == 0.0 and float(part[q + 1:].abs().max()) == 0.0


def test_quadrature_step_vs_oracle(hip_ops):
    """EZy = w . mean_predict(X), VarZy = w^T K(X, X) w with the structured kernels' own mean
    (BASQ/_quadrature.py:53-64), against the oracle's CPU kernels."""
    import basq_amd
    from basq_amd.pools import gmm_pool, prior_sampler_split
    from oracle.kernels_oracle import PosteriorOracle, StationaryOracle, WsabiOracle, synthetic_gp_state
    from oracle.rchq_oracle import recombination_oracle

    d, n_obs, N, n = 5, 60, 6000, 30
    Xobs = gmm_pool(n_obs, d, 41)
    base_o = StationaryOracle("rbf", 1.8, 1.1)
    W, const, mc, _ = synthetic_gp_state(Xobs, base_o, 1e-8, 3)
    alpha = 0.4
    K = basq_amd.kernels
    post = K.PosteriorKernel(K.StationaryKernel("rbf", 1.8, 1.1), Xobs, W, 1e-8, const, mc)
    for label in ("wsabil", "wsabim"):
        kern = K.WsabiKernel(post, const, mc, label, alpha=alpha)
        kern_o = WsabiOracle(PosteriorOracle(base_o, Xobs, W, 1e-8), const, mc, label)
        pool = gmm_pool(N, d, 42)

        def sampler(k):
            return prior_sampler_split(pool[:k], n_nys=60)

        kq = basq_amd.KernelQuadrature(N, 60, N, n, sampler, kern, "cuda:0")
        torch.manual_seed(5)
        EZy, VarZy = kq.quadrature()
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)
        try:
            torch.manual_seed(5)
            idx, w = recombination_oracle(pool, pool[:60], n, kern_o)
        finally:
            torch.set_default_dtype(prev)
        X = pool[idx]
        mu_w = kern_o.mean(X)
        if label == "wsabil":
            mean = alpha + 0.5 * mu_w ** 2
        else:
            kxX = base_o(X, Xobs)
            var = 1.1 - ((kxX @ W) * kxX).sum(1) + 1e-8
            mean = alpha + 0.5 * (mu_w ** 2 + var)
        assert EZy == pytest.approx(float(w @ mean), rel=1e-8)
        assert VarZy == pytest.approx(float(w @ kern_o(X, X) @ w), rel=1e-6, abs=1e-14)


def test_prior_max_and_uniform_trans_vs_oracle(hip_ops):
    """``KernelQuadrature.prior_max`` (BASQ/_quadrature.py:66-84) and ``uniform_trans`` (:86-107) against the oracle:
    same pools (drawn where the reference draws them, from the CPU global generator), same selection, same estimates."""
    import basq_amd
    from basq_amd.pools import gmm_pool
    from oracle.kernels_oracle import PosteriorOracle, StationaryOracle, synthetic_gp_state
    from oracle.rchq_oracle import recombination_oracle

    d, n_obs, N, m, n = 4, 50, 5000, 80, 24
    Xobs = gmm_pool(n_obs, d, 51)
    base_o = StationaryOracle("rbf", 1.6, 1.2)
    W, const, mc, _ = synthetic_gp_state(Xobs, base_o, 1e-6, 4)
    K = basq_amd.kernels
    post = K.PosteriorKernel(K.StationaryKernel("rbf", 1.6, 1.2), Xobs, W, 1e-6, const, mc)
    post_o = PosteriorOracle(base_o, Xobs, W, 1e-6)
    kq = basq_amd.KernelQuadrature(N, m, N, n, None, post, "cuda:0")

    def oracle_run(pool, kern_o):
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)
        try:
            return recombination_oracle(pool, pool[:m], n, kern_o)
        finally:
            torch.set_default_dtype(prev)

    def gp_mean(X):
        return const + base_o(X, Xobs) @ mc

    # prior_max: pool ~ mvn_max, posterior kernel, predictive mean
    mvn = torch.distributions.MultivariateNormal(torch.zeros(d, dtype=torch.float64), 1.5 * torch.eye(d, dtype=torch.float64))
    torch.manual_seed(9)
    EZy, VarZy = kq.prior_max(mvn)
    torch.manual_seed(9)
    pool = mvn.sample(sample_shape=torch.Size([N]))
    idx, w = oracle_run(pool, post_o)
    X = pool[idx]
    assert EZy == pytest.approx(float(w @ gp_mean(X)), rel=1e-8)
    assert VarZy == pytest.approx(float(w @ post_o(X, X) @ w), rel=1e-6, abs=1e-14)

    # uniform_trans: uniform pool, PRIOR kernel of the importance-weighted model, its predictive mean
    def uni_sampler(k):
        return torch.rand(k, d, dtype=torch.float64) * 6.0 - 3.0

    torch.manual_seed(10)
    EZu, VarZu = kq.uniform_trans(post, uni_sampler)
    torch.manual_seed(10)
    pool = uni_sampler(N)
    idx, w = oracle_run(pool, base_o)
    X = pool[idx]
    assert EZu == pytest.approx(float(w @ gp_mean(X)), rel=1e-8)
    assert VarZu == pytest.approx(float(w @ base_o(X, X) @ w), rel=1e-6, abs=1e-14)


@pytest.mark.parametrize("n", [990_000, 4_428, 16, 31])
def test_box_muller_vs_torch_randn(hip_ops, n):
    """Device Box-Muller of torch.rand's uniforms == torch.randn from the same generator state, to round-off."""
    torch.manual_seed(77)
    ref = torch.randn(n, dtype=torch.float64)
    torch.manual_seed(77)
    u = torch.rand(n, dtype=torch.float64)
    ut = torch.rand(16, dtype=torch.float64) if n % 16 else None
    out = hip_ops.box_muller(hip_ops.to_device(u), None if ut is None else hip_ops.to_device(ut)).cpu()
    assert (out - ref).abs().max().item() <= 1e-14


def test_sharded_products_on_device(hip_ops):
    """The row-sharded range finder's device code -- split-K GEMM on a row block, padded all-gather layout, and (round 4)
    CholeskyQR with the Gram product and the triangular solve on the rank's own rows -- against the dense range finder, on
    ONE GPU: two host threads play the ranks of a 2-rank group and exchange through a barrier (same device, same library)."""
    import threading

    import basq_amd._basis as E
    from basq_amd._ops import HipOps
    from basq_amd.kernels import StationaryKernel
    from basq_amd.pools import gmm_pool

    m, q = 1000, 99
    nys = hip_ops.to_device(gmm_pool(m, 10, 3))
    kern = StationaryKernel("rbf", 2.0)
    center = hip_ops.col_mean(nys)
    A = kern.dense(hip_ops, nys, nys, center)
    shards = [(0, 501), (501, 499)]
    torch.cuda.synchronize()

    class ThreadComm:
        world = 2
        slots, barrier = [None, None], threading.Barrier(2)

        def __init__(self, rank):
            self.rank = rank

        def all_gather(self, blk):
            torch.cuda.synchronize()
            ThreadComm.slots[self.rank] = blk
            ThreadComm.barrier.wait()
            out = torch.stack([ThreadComm.slots[0], ThreadComm.slots[1]], 0)
            ThreadComm.barrier.wait()
            return out

        def broadcast(self, t, src=0):
            torch.cuda.synchronize()
            if self.rank == src:
                ThreadComm.slots[src] = t
            ThreadComm.barrier.wait()
            t.copy_(ThreadComm.slots[src])
            torch.cuda.synchronize()
            ThreadComm.barrier.wait()
            return t

    results, errors = [None, None], []
    seed_state = torch.get_rng_state()

    def run(rank):
        try:
            ops = HipOps(hip_ops.device)
            if rank != 0:                                         # one process here: only "rank 0" may consume the generator
                ops.host_uniform = lambda n, tag: torch.zeros(n, dtype=torch.float64)
            r0, mr = shards[rank]
            prod = E._ShardedProducts(ops, ThreadComm(rank), A[r0:r0 + mr].contiguous(), shards, m)
            results[rank] = E.nystrom_basis(ops, prod, q)
            if rank == 0:
                assert torch.equal(prod.full(), A)
            else:
                prod.full()
        except Exception as exc:                                  # noqa: BLE001  (re-raised in the main thread)
            errors.append(exc)
            ThreadComm.barrier.abort()

    torch.manual_seed(5)                                          # rank 0 draws the test matrix; rank 1 only advances
    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    torch.manual_seed(5)
    Ud = E.nystrom_basis(hip_ops, A, q)
    assert torch.equal(results[0], results[1])                    # the same bits on both ranks
    Us = results[0]
    sign = torch.sign((Us * Ud).sum(1, keepdim=True))
    assert (Us * sign - Ud).abs().max().item() <= 1e-7
    del seed_state


def test_fuzz_single_workgroup_kernels(hip_ops):
    """Random shapes through the hand-synchronised kernels (tools/fuzz_reduction.py): LAPACK's null-space rows, bit-exact
    and bitwise-repeatable elimination, orthonormalising Cholesky -- a race would show up as a flaky last bit."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from fuzz_reduction import run

    assert run(30, 7, hip_ops).startswith("fuzz ok")


@pytest.mark.parametrize("s,M", [(200, 400), (100, 200), (150, 400), (256, 512), (101, 500)])
def test_cluster_reductions_repeatable_under_load(hip_ops, s, M):
    """The cluster kernels (one message per step through LDS, or through global memory between the four work-groups of
    a cluster) hand data between waves / CUs by hand: run null space + elimination repeatedly while a second stream
    keeps the chip busy with unrelated work (uneven load, warm caches) -- every repetition must return the LAPACK rows
    and bit-identical survivors.  A stale or torn hand-off shows up as a differing last bit or a wrong pivot."""
    g = torch.Generator().manual_seed(s + 7 * M)
    X = torch.randn(s, M, generator=g, dtype=torch.float64)
    X[0] = 1.0
    ref = torch.linalg.svd(X)[2][s:]
    mu = torch.rand(M, generator=g, dtype=torch.float64) + 0.05
    mu = mu / mu.sum()
    Xd, mud = hip_ops.to_device(X), hip_ops.to_device(mu)
    side = torch.cuda.Stream()
    A = torch.randn(4096, 4096, device=hip_ops.device, dtype=torch.float32)
    first = None
    for rep in range(12):
        with torch.cuda.stream(side):                          # uneven background load: a few large GEMMs per repetition
            for _ in range(1 + rep % 3):
                A = torch.tanh(A @ A * 1e-3)
        P = hip_ops.nullspace(Xd, s, M)
        kr, kept, w, info = hip_ops.car_eliminate(P.clone(), mud.clone(), M, s)
        out = (P.cpu(), kr.cpu(), w.cpu(), info.cpu().tolist())
        assert out[3][1] == 0, "status"
        assert (out[0] - ref).abs().max().item() <= 1e-11
        if first is None:
            first = out
        else:
            assert torch.equal(out[0], first[0]), f"repetition {rep}: null space differs bitwise"
            assert torch.equal(out[1], first[1]) and torch.equal(out[2][: out[3][0]], first[2][: first[3][0]])
    torch.cuda.synchronize()


@pytest.mark.parametrize("family,d,m,S,Rl,off,n_full,C,class0,n_ch", [
    ("rbf", 10, 100, 200, 6400, 0, 6400, 8, 0, 8),           # 32 blocks, all 8 classes in one launch
    ("rbf", 10, 100, 200, 6400, 0, 6400, 8, 7, 1),           # the last class alone (deferred launch of round 1)
    ("rbf", 10, 130, 200, 3000, 1300, 6400, 4, 0, 4),        # a middle shard starting and ending mid-block
    ("matern52", 32, 64, 400, 4000, 0, 4000, 2, 0, 2),       # d = 32 (2 row tiles per wave), S = 400
    ("rbf", 2, 50, 22, 880, 0, 880, 4, 1, 2),                # S not a multiple of 16; classes 1..2 of 4
])
def test_blocksum_residue_classes_vs_standin(hip_ops, family, d, m, S, Rl, off, n_full, C, class0, n_ch):
    """Residue-class chunks: chunk c = the blocks with (global block index) % C == class0 + c."""
    cpu = CpuStandInOps()
    spec = _spec(family, d)
    nys, cand = _rand(m, d, 3), _rand(Rl, d, 4)
    g = torch.Generator().manual_seed(5)
    mu = torch.rand(Rl, generator=g, dtype=torch.float64) + 0.1
    wx = torch.rand(Rl, generator=g, dtype=torch.float64) + 0.5
    center = nys.mean(0)
    A_c, B_c = cpu.pack(spec, nys, center, 0, pad_rows_to=64), cpu.pack(spec, cand, center, 1)
    Xc, tc = cpu.blocksum(spec, A_c, m, B_c, mu, wx, Rl, off, n_full, S, n_ch, class_mod=C, class0=class0)
    dev = hip_ops.to_device
    A_g = hip_ops.pack(spec, dev(nys), dev(center), 0, pad_rows_to=64)
    B_g = hip_ops.pack(spec, dev(cand), dev(center), 1)
    Xg, tg = hip_ops.blocksum(spec, A_g, m, B_g, dev(mu), dev(wx), Rl, off, n_full, S, n_ch, class_mod=C, class0=class0)
    scale = Xc.abs().max().item()
    assert (Xg.cpu() - Xc).abs().max().item() <= 1e-12 * scale
    assert (tg.cpu() - tc).abs().max().item() <= 1e-13 * tc.abs().max().item()
    if class0 == 0 and n_ch == C:                              # all classes together = the plain block sums
        X1, t1 = cpu.blocksum(spec, A_c, m, B_c, mu, wx, Rl, off, n_full, S, 1)
        assert (Xg.cpu().sum(0) - X1[0]).abs().max().item() <= 1e-12 * scale


@pytest.mark.parametrize("C,rows,S", [(8, 70, 200), (2, 33, 22), (16, 100, 400)])
def test_regroup_classes_vs_standin(hip_ops, C, rows, S):
    cpu = CpuStandInOps()
    T = _rand(C * rows, S, 11).reshape(C, rows, S).contiguous()
    g = torch.Generator().manual_seed(3)
    kept = torch.sort(torch.randperm(S, generator=g)[: S // 2]).values.to(torch.int32)
    kept_full = torch.zeros(S, dtype=torch.int32)
    kept_full[: S // 2] = kept
    w_star = torch.rand(S, generator=g, dtype=torch.float64) + 0.01
    tot = torch.rand(S, generator=g, dtype=torch.float64) + 0.5
    Tc = cpu.regroup_classes(T, kept_full, w_star, tot)
    dev = hip_ops.to_device
    Tg = hip_ops.regroup_classes(dev(T), dev(kept_full), dev(w_star), dev(tot))
    assert torch.equal(Tg.cpu(), Tc)                                          # mul then div, same order: bit-exact
    # round 4: the same regrouping fused with the next round's descriptor (basq_regroup_round_next_f64) -- both outputs
    # bit-identical to the two separate launches
    keep_rank = torch.full((S,), -1, dtype=torch.int32)
    keep_rank[kept.long()] = torch.arange(S // 2, dtype=torch.int32)
    info = torch.tensor([S // 2, 0], dtype=torch.int32)
    R = 37 * S + 11
    geo = hip_ops.geo_init(3, R, S, (37 // C) * C * S)
    hip_ops.round_next(geo[0], dev(info), dev(keep_rank), S, -1, True, geo[1])
    Tf = hip_ops.empty(C // 2, rows, S)
    hip_ops.regroup_round_next(dev(T), dev(kept_full), dev(w_star), dev(tot), Tf, geo[0], dev(info), dev(keep_rank), S, -1, True, geo[2])
    assert torch.equal(Tf.cpu(), Tc)
    assert torch.equal(geo[1].cpu(), geo[2].cpu())


@pytest.mark.parametrize("q,m,S,n_chunks", [(99, 1000, 200, 17), (9, 50, 20, 3), (199, 333, 400, 5), (100, 10_000, 200, 2)])
def test_project_chunks_and_sum_parts_vs_standin(hip_ops, q, m, S, n_chunks):
    """Per-chunk messages [tot ; U @ X_c] (the class messages of an epoch) and their ordered sum = the plain projection."""
    cpu = CpuStandInOps()
    U = _rand(q, m, 21)
    X = _rand(n_chunks * m, S, 22).reshape(n_chunks, m, S).contiguous()
    t = _rand(n_chunks, S, 23).abs() + 0.1
    Mc = cpu.project_chunks(U, q, m, X, t, n_chunks, S, 1.7)
    dev = hip_ops.to_device
    Mg = hip_ops.project_chunks(dev(U), q, m, dev(X), dev(t), n_chunks, S, 1.7)
    scale = Mc.abs().max().item()
    assert (Mg.cpu() - Mc).abs().max().item() <= 1e-12 * scale
    assert torch.equal(Mg.cpu()[:, 0, :], t)
    total = hip_ops.sum_parts(Mg).cpu()
    plain = hip_ops.project(dev(U), q, m, dev(X), dev(t), n_chunks, S, 1.7).cpu()
    assert (total - plain).abs().max().item() <= 1e-12 * scale
    assert (total - cpu.sum_parts(Mc)).abs().max().item() <= 1e-12 * scale


@pytest.mark.parametrize("M,K,N,trans,ksplit", [
    (1000, 3000, 99, False, None),       # the range finder's shape class: A Q
    (1000, 3000, 99, True, None),        # A^T Q from the stored A
    (99, 5000, 99, True, 31),            # Gram product X^T X of the orthonormalisation, many K slices
    (777, 1003, 199, False, 4),          # 13 column tiles, K not a multiple of 16, ragged rows
    (650, 333, 33, True, 1),             # few columns, one slice
    (130, 99, 99, False, 1),             # K shorter than the output (the final U = Q Ub)
    (64, 17, 208, False, 1),             # widest supported output, one partial trip
])
def test_skinny_gemm_vs_torch(hip_ops, M, K, N, trans, ksplit):
    """The tall-skinny f64 MFMA GEMM of the range finder against torch's matmul (row strides exercised)."""
    A = _rand(K if trans else M, M if trans else K, 71)
    B = _rand(K, N, 72)
    want = (A.t() if trans else A) @ B
    Aw = hip_ops.zeros(A.shape[0], A.shape[1] + 3)           # row-strided views of wider buffers
    Aw[:, :A.shape[1]] = hip_ops.to_device(A)
    Bw = hip_ops.zeros(K, N + 5)
    Bw[:, :N] = hip_ops.to_device(B)
    got = hip_ops.skinny_gemm(Aw[:, :A.shape[1]], Bw[:, :N], trans=trans, ksplit=ksplit).cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 1e-13 * K ** 0.5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M,K,N,pad,ksplit", [
    (256, 4096, 99, 0, 4),       # the engine's layout: tight, 16-byte aligned rows -> the pipelined trips
    (256, 4096, 99, 5, 3),       # NaN beyond column N of every B row: the 16 NT-wide row reads must not leak them
    (100, 640, 37, 0, 2),        # three column tiles on the four-tile kernel: tiles past N are read from the next rows
    (70, 96, 3, 0, 1),           # a row of B shorter than one tile: the wide reads span many rows, the last ones run masked
    (512, 2048, 199, 1, 2),      # 13 column tiles
    (96, 160, 112, 0, 1),        # N a whole number of tiles
    (256, 2048, 100, 0, 2),      # 6 tiles + 1 column group, all four columns of the group in use
    (130, 512, 97, 3, 1),        # the same kernel with one column of the group (and NaN right behind it), ragged rows
    (128, 1024, 200, 0, 2),      # 12 tiles + 2 column groups
    (70, 512, 197, 1, 1),        # ... with one column in the second group
])
@pytest.mark.parametrize("trans", [False, True])
def test_skinny_gemm_pipelined_path(hip_ops, M, K, N, pad, ksplit, trans):
    """Tight / aligned operands take the kernel's pipelined trips, whose B fragments are read 16 NT columns wide: whatever
    lies beyond column N (the next rows, or NaN padding) must never reach a stored column, and nothing may be read past
    the last row of B."""
    A = _rand(K if trans else M, M if trans else K, 73)
    B = _rand(K, N, 74)
    want = (A.t() if trans else A) @ B
    Ad = hip_ops.to_device(A)
    Bw = hip_ops.zeros(K, N + pad)
    Bw[:] = float("nan")
    Bw[:, :N] = hip_ops.to_device(B)
    got = hip_ops.skinny_gemm(Ad, Bw[:, :N], trans=trans, ksplit=ksplit).cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 1e-13 * K ** 0.5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("R,S,n_keep,reg_blocks,class_mod,drop_last,shard", [
    (10_000, 200, 100, 48, 16, False, None),     # regular region of 48 blocks in 16 classes + 2 blocks irregular, no tail
    (10_123, 200, 100, 48, 16, True, None),      # ragged tail, last set dropped
    (10_123, 200, 100, 0, 0, False, None),       # no classes: everything through mode 3
    (4_321, 100, 37, 40, 8, False, None),        # an elimination that kept fewer than half of the sets
    (10_123, 200, 100, 48, 16, False, (3_333, 3_456)),   # a rank's shard in the middle: mid-block start and end
    (10_123, 200, 100, 48, 16, True, (9_550, 573)),      # the last rank: end of the regular region, irregular blocks, tail
    (10_123, 200, 100, 48, 16, False, (10_050, 73)),     # a shard inside the ragged tail only
    (10_123, 200, 100, 0, 0, False, (0, 150)),           # a shard shorter than one block
])
def test_descriptor_driven_kernels_vs_host_geometry(hip_ops, R, S, n_keep, reg_blocks, class_mod, drop_last, shard):
    """basq_blocksum_geo_f64 / basq_reweight_compact_geo_f64 / basq_round_next_i64: the same results as the
    host-geometry entries given the same numbers, with launches and buffers sized for an upper bound -- for the whole
    pool on one rank and for a rank's shard ``[off, off + Rl)`` of it (multi-GPU rounds without a host wait)."""
    from basq_amd._partition import RoundGeometry, next_shard

    cpu = CpuStandInOps()
    d, m = 5, 130
    spec = _spec("rbf", d)
    off, Rl = shard if shard is not None else (0, R)
    nys, cand = _rand(m, d, 81), _rand(R, d, 82)[off:off + Rl]
    g = torch.Generator().manual_seed(9)
    mu = (torch.rand(R, generator=g, dtype=torch.float64) + 0.05)[off:off + Rl]
    center = hip_ops.to_device(nys.mean(0))
    pa = hip_ops.pack(spec, hip_ops.to_device(nys), center, 0, pad_rows_to=64)
    R_max = Rl + 777                                              # buffers and grids are sized for an upper bound
    pb = hip_ops.zeros(R_max, hip_ops.kp(d))
    pb[:Rl] = hip_ops.pack(spec, hip_ops.to_device(cand), center, 1)
    mu_d = hip_ops.zeros(R_max)
    mu_d[:Rl] = hip_ops.to_device(mu)
    nb = R // S
    n_full, reg_hi = nb * S, reg_blocks * S
    geo = hip_ops.geo_init(4, R, S, reg_hi, off, Rl)

    def host_range(lo, hi, n_ch, **kw):
        """The host-geometry entry over the global positions [lo, hi) restricted to the shard."""
        lo, hi = max(lo, off), min(hi, off + Rl)
        hi = max(hi, lo)
        return hip_ops.blocksum(spec, pa, m, pb[lo - off:], mu_d[lo - off:], None, hi - lo, lo, n_full, S, n_ch, **kw)

    # block sums: regular region (classes), the rest, everything
    if class_mod:
        Xa, ta = hip_ops.blocksum_geo(spec, pa, m, pb, mu_d, None, geo[0], 1, S, class_mod, class_mod=class_mod)
        Xb, tb = host_range(0, reg_hi, class_mod, class_mod=class_mod)
        assert torch.equal(Xa, Xb) and torch.equal(ta, tb)
        Xa, ta = hip_ops.blocksum_geo(spec, pa, m, pb, mu_d, None, geo[0], 2, S, 1)
        Xb, tb = host_range(reg_hi, R, 1)
        assert torch.equal(Xa, Xb) and torch.equal(ta, tb)
    Xa, ta = hip_ops.blocksum_geo(spec, pa, m, pb, mu_d, None, geo[0], 3, S, 3)
    Xb, tb = host_range(0, R, 3)
    assert torch.equal(Xa, Xb) and torch.equal(ta, tb)
    # an elimination outcome: n_keep sets kept (the last one or not), arbitrary positive weights
    kept_sets = sorted(torch.randperm(S - 1, generator=g)[:n_keep - (0 if drop_last else 1)].tolist() + ([] if drop_last else [S - 1]))
    keep_rank = torch.full((S,), -1, dtype=torch.int32)
    keep_rank[kept_sets] = torch.arange(len(kept_sets), dtype=torch.int32)
    w_star = torch.rand(S, generator=g, dtype=torch.float64) + 0.1
    tot = torch.rand(S, generator=g, dtype=torch.float64) + 0.1
    info = torch.tensor([len(kept_sets), 0], dtype=torch.int32)
    gid = torch.arange(R_max, dtype=torch.int64) * 3 + 1
    kr_d, ws_d, tot_d, info_d, gid_d = (hip_ops.to_device(t) for t in (keep_rank, w_star, tot, info, gid))
    new_R = nb * len(kept_sets) + (0 if drop_last else R - n_full)
    new_off, new_Rl = next_shard(off, Rl, RoundGeometry.of(R, S), kept_sets)
    # next descriptor (no expectation on the survivor count): the shard's closed form, then the compaction that reads it
    hip_ops.round_next(geo[0], info_d, kr_d, S, 0, False, geo[1])
    g1 = hip_ops.to_host(geo[1], "g1").clone()
    assert g1[0].item() == new_R and g1[3].item() == 0 and (g1[6].item(), g1[7].item()) == (new_off, new_Rl)
    out_rows = new_Rl + 55
    ca, ma, ga, _ = hip_ops.reweight_compact_geo(pb, mu_d, gid_d, None, geo[0], geo[1], info_d, R_max, S, hip_ops.kp(d), kr_d,
                                                 ws_d, tot_d, out_rows)
    cb, mb, gb, _ = hip_ops.reweight_compact(pb, mu_d, gid_d, None, Rl, off, n_full, S, hip_ops.kp(d), kr_d, ws_d, tot_d,
                                             len(kept_sets), new_off, new_Rl)
    assert torch.equal(ca[:new_Rl], cb[:new_Rl]) and torch.equal(ma[:new_Rl], mb[:new_Rl]) and torch.equal(ga[:new_Rl], gb[:new_Rl])
    # next descriptor with the host's expectation: fresh classes, inherited classes, none -- against the stand-in's closed form
    for mode in (8, -1, 0):
        hip_ops.round_next(geo[0], info_d, kr_d, S, mode, True, geo[2])
        want = torch.zeros(8, dtype=torch.int64)
        cpu.round_next(hip_ops.to_host(geo[0], "g0").clone(), info, keep_rank, S, mode, True, want)
        got = hip_ops.to_host(geo[2], "g2").clone()
        assert got.tolist() == want.tolist()
        viol = 2 * len(kept_sets) != S
        assert got[3].item() == (1 if viol else 0)
        assert got[0].item() == (0 if viol else new_R)           # after a violation the next round is EMPTY
        assert (got[6].item(), got[7].item()) == ((0, 0) if viol else (new_off, new_Rl))


def test_violated_descriptor_round_writes_nothing(hip_ops):
    """when an elimination keeps MORE sets than the host sized the next buffers for (early stop, status
    1) the descriptor-driven compaction must not write a single row, and every later ``*_geo`` launch must see an empty
    round.  The outputs are poisoned first; rows, weights and ids must come back untouched."""
    import ctypes as C

    from basq_amd._ops import _ptr

    d, S, R = 4, 40, 2_000
    spec = _spec("rbf", d)
    kp = hip_ops.kp(d)
    cand = hip_ops.pack(spec, hip_ops.to_device(_rand(R, d, 3)), hip_ops.to_device(torch.zeros(d, dtype=torch.float64)), 1)
    mu = hip_ops.zeros(R) + 1.0 / R
    gid = hip_ops.to_device(torch.arange(R, dtype=torch.int64))
    geo = hip_ops.geo_init(4, R, S, 0)
    keep_rank = torch.arange(S, dtype=torch.int32)                # EVERY set "kept": 2x what the host expects
    for info_v in ([S, 1], [S, 0], [S // 2, 2]):                  # early stop / too many sets / cluster time-out
        info = hip_ops.to_device(torch.tensor(info_v, dtype=torch.int32))
        kr = hip_ops.to_device(keep_rank)
        w_star, tot = hip_ops.zeros(S) + 0.5, hip_ops.zeros(S) + 1.0
        hip_ops.round_next(geo[0], info, kr, S, 0, True, geo[1])
        g1 = hip_ops.to_host(geo[1], "g1").tolist()
        assert g1[3] == 1 and g1[0] == 0 and g1[7] == 0
        out_rows = (R // S) * (S // 2) + S - 1                    # what the engine allocates for the next round
        co, mo, go = hip_ops.zeros(out_rows, kp) - 3.0, hip_ops.zeros(out_rows) - 3.0, hip_ops.zeros(out_rows, dtype=torch.int64) - 3
        rc = hip_ops.lib.basq_reweight_compact_geo_f64(_ptr(cand), _ptr(mu), _ptr(gid), None, _ptr(geo[0]), _ptr(geo[1]),
                                                       _ptr(info), R, S, kp, _ptr(kr), _ptr(w_star), _ptr(tot), out_rows,
                                                       S // 2, _ptr(co), _ptr(mo), _ptr(go), None,
                                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        assert bool((co == -3.0).all()) and bool((mo == -3.0).all()) and bool((go == -3).all())
        # launches already enqueued for the following rounds see an empty round: zero sums, nothing read past the buffers
        X, t = hip_ops.blocksum_geo(spec, hip_ops.pack(spec, hip_ops.to_device(_rand(64, d, 4)),
                                                       hip_ops.to_device(torch.zeros(d, dtype=torch.float64)), 0, pad_rows_to=64),
                                    64, co, mo, None, geo[1], 3, S, 2)
        assert float(X.abs().max()) == 0.0 and float(t.abs().max()) == 0.0


@pytest.mark.parametrize("d", [1, 10, 26, 27, 38])
@pytest.mark.parametrize("n", [1, 255, 257, 1000])
def test_pack_points_vs_standin(hip_ops, d, n):
    """Packed rows [(x - c) / l, 0.., h | 1, 1 | h] (LDS-staged kernel: 256 or 128 points per block, ragged last block)."""
    cpu = CpuStandInOps()
    spec = _spec("rbf", d)
    X = _rand(n, d, 90 + d)
    c = X.mean(0)
    for role in (0, 1):
        for center in (c, None):
            want = cpu.pack(spec, X, center, role)
            got = hip_ops.pack(spec, hip_ops.to_device(X), None if center is None else hip_ops.to_device(center), role).cpu()
            assert got.shape == want.shape
            assert (got - want).abs().max().item() <= 1e-14 * max(1.0, want.abs().max().item())


def test_shader_clock_sampler(hip_ops):
    """The measurement aid behind ``roofline.shader_clock_MHz_in_situ``: plausible MHz, one value per period."""
    clk = hip_ops.shader_clock_mhz(4, 50)
    torch.cuda.synchronize()
    assert clk.shape == (4,)
    assert bool(((clk > 500.0) & (clk < 3500.0)).all()), clk.tolist()
    with pytest.raises(Exception):
        hip_ops.shader_clock_mhz(0, 50)


_SPREAD_CHILD = r"""
import sys, time, torch
sys.path.insert(0, sys.argv[1])
from basq_amd._ops import HipOps
ops = HipOps(torch.device("cuda:0"))
out = {}
for s, M in ((200, 400), (150, 400)):
    g = torch.Generator().manual_seed(s + 7 * M)
    X = torch.randn(s, M, generator=g, dtype=torch.float64); X[0] = 1.0
    mu = torch.rand(M, generator=g, dtype=torch.float64) + 0.05; mu = mu / mu.sum()
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        P = ops.nullspace(ops.to_device(X), s, M)
        kr, kept, w, info = ops.car_eliminate(P.clone(), ops.to_device(mu), M, s)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[(s, M, rep)] = (P.cpu(), kr.cpu(), w.cpu(), info.cpu(), dt)
torch.save(out, sys.argv[2])
"""


def test_cluster_reductions_members_on_different_xcds(tmp_path):
    """The clusters write their granules with plain stores when all members share an XCD and write-through otherwise
    (``cluster_shares_xcd``).  ``BASQ_CLUSTER_SPREAD=1`` deals the members to eight DIFFERENT XCDs: the fall-back path must
    return the same bits as the same-XCD path (both sum in cluster order), three times in a row -- with status 0 (no member hit
    its spin limit: status 2) and within a fraction of the time a spin-limit hit takes (``BASQ_GRANULE_SPIN_LIMIT`` sweeps of
    ~1 us: 0.5 s).  The elimination's placement vote is taken by all 64 lanes of wave 0 (it used to run inside the
    lane-0 branch, so group 0 always chose plain stores)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for spread in ("0", "1"):
        out = tmp_path / f"spread{spread}.pt"
        env = dict(os.environ, BASQ_CLUSTER_SPREAD=spread)
        r = subprocess.run([sys.executable, "-c", _SPREAD_CHILD, root, str(out)], env=env, capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[spread] = torch.load(out)
    for key, a in res["0"].items():
        b = res["1"][key]
        assert a[3].tolist()[1] == 0 and b[3].tolist()[1] == 0, (key, a[3].tolist(), b[3].tolist())
        assert torch.equal(a[0], b[0]), f"{key}: null space differs between the same-XCD and the spread cluster"
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        if key[2] > 0:                                          # (rep 0 pays the library load and the first launches)
            assert a[4] < 0.1 and b[4] < 0.1, f"{key}: {a[4]:.3f} s / {b[4]:.3f} s -- a granule spin ran into its limit?"


@pytest.mark.parametrize("rows,q", [(10048, 99), (10000, 100), (1000, 30), (130, 7), (5000, 112), (64, 40), (10048, 199), (3000, 113), (700, 200)])
def test_fused_cholqr_equals_factor_plus_solve(hip_ops, rows, q):
    """``basq_cholqr_f64`` (factor and solve in one launch, the solvers following the factor panel by panel through a
    progress word) against ``basq_chol_factor_f64`` + ``basq_trsm_rows_f64``: same arithmetic, same bits -- repeated, with a
    second stream keeping the chip busy, so that a solver reading a panel too early would show."""
    g = torch.Generator().manual_seed(rows + q)
    X = torch.randn(rows, q, generator=g, dtype=torch.float64)
    Xd = hip_ops.to_device(X)
    G = (X.t() @ X)
    G1 = hip_ops.to_device(G.clone())
    info1 = hip_ops.chol_factor(G1)
    Q1 = hip_ops.trsm_rows(Xd, G1)
    side = torch.cuda.Stream()
    A = torch.randn(2048, 2048, device=hip_ops.device, dtype=torch.float32)
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 3):
                A = torch.tanh(A @ A * 1e-3)
        G2 = hip_ops.to_device(G.clone())
        Q2, info2 = hip_ops.cholqr(G2, Xd)
        assert int(info2.item()) == 0 and int(info1.item()) == 0
        assert torch.equal(torch.tril(G2), torch.tril(G1)), f"repetition {rep}: factor differs"
        assert torch.equal(Q2, Q1), f"repetition {rep}: Q differs bitwise"
    torch.cuda.synchronize()
    QtQ = (Q1.t() @ Q1).cpu()
    assert (QtQ - torch.eye(q, dtype=torch.float64)).abs().max().item() < 1e-9


def test_fused_cholqr_reports_a_failed_pivot(hip_ops):
    """A rank-deficient Gram: the factor flags the pivot, raises the abort word, and every solver returns (no hang)."""
    g = torch.Generator().manual_seed(3)
    X = torch.randn(4000, 40, generator=g, dtype=torch.float64)
    X[:, 17] = X[:, 3] + X[:, 5]                               # exact linear dependence
    Xd = hip_ops.to_device(X)
    G = hip_ops.to_device(X.t() @ X)
    info_ref = hip_ops.chol_factor(G.clone())
    Q, info = hip_ops.cholqr(G.clone(), Xd)
    torch.cuda.synchronize()
    assert int(info_ref.item()) > 0 and int(info.item()) == int(info_ref.item())


# ---------------------------------------------------------------------------------------------------------------------
# epochs without a pairwise evaluation inside (ABI 15): basq_blocksum_geo_f64 mode 5, basq_epoch_turn_f64,
# basq_reweight_compact_rounds_f64
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,S,reg_blocks,C", [(10_000, 200, 48, 16), (10_123, 200, 48, 16), (3_456, 64, 48, 8), (2_000, 200, 8, 8),
                                              (1_650, 200, 8, 2)])
def test_blocksum_geo_one_chunk_per_irregular_block(hip_ops, R, S, reg_blocks, C):
    """geo_mode 5: the full blocks behind the regular region, one chunk each, = the host-geometry class-mode entry over those
    blocks (bit for bit); chunks without a block are zeros; the ragged tail stays out (mode 4 carries it, point k in set k)."""
    d, m = 5, 130
    spec = _spec("rbf", d)
    nys, cand = _rand(m, d, 71), _rand(R, d, 72)
    g = torch.Generator().manual_seed(10)
    mu = torch.rand(R, generator=g, dtype=torch.float64) + 0.05
    center = hip_ops.to_device(nys.mean(0))
    pa = hip_ops.pack(spec, hip_ops.to_device(nys), center, 0, pad_rows_to=64)
    pb = hip_ops.pack(spec, hip_ops.to_device(cand), center, 1)
    mu_d = hip_ops.to_device(mu)
    nb = R // S
    n_full, reg_hi, e = nb * S, reg_blocks * S, nb - reg_blocks
    assert 0 <= e < C
    geo = hip_ops.geo_init(2, R, S, reg_hi)
    E = C - 1
    Xa, ta = hip_ops.blocksum_geo(spec, pa, m, pb, mu_d, None, geo[0], 5, S, E, class_mod=C)
    for b in range(E):
        if b < e:
            lo = reg_hi + b * S
            Xb, tb = hip_ops.blocksum(spec, pa, m, pb[lo:], mu_d[lo:], None, S, lo, n_full, S, 1)
            assert torch.equal(Xa[b], Xb[0]) and torch.equal(ta[b], tb[0])
            assert torch.equal(ta[b], mu_d[lo:lo + S])           # one point per set: the set weight IS the point's weight
        else:
            assert not Xa[b].any() and not ta[b].any()
    # tail: point k in set k, set weights = the points' weights
    Xt, tt = hip_ops.blocksum_geo(spec, pa, m, pb, mu_d, None, geo[0], 4, S, 1)
    t = R - n_full
    assert torch.equal(tt[0, :t], mu_d[n_full:]) and not tt[0, t:].any() and not Xt[0][:, t:].any()
    # all of it together = the ordinary irregular chunk (mode 2) up to the order of the sums
    X2, t2 = hip_ops.blocksum_geo(spec, pa, m, pb, mu_d, None, geo[0], 2, S, 1)
    tog = Xa.sum(0)
    tog[:, S - 1] += Xt[0].sum(1)
    assert (tog - X2[0]).abs().max().item() <= 1e-12 * X2[0].abs().max().item()


@pytest.mark.parametrize("C,E_in,e,t,rows,S,last_kept", [(16, 13, 12, 100, 100, 200, True), (16, 13, 13, 0, 100, 200, False),
                                                          (8, 4, 3, 37, 31, 76, True), (2, 1, 1, 150, 100, 200, True),
                                                          (4, 0, 0, 5, 12, 20, True), (4, 3, 2, 0, 303, 200, False)])
def test_epoch_turn_kernel_vs_stand_in(hip_ops, C, E_in, e, t, rows, S, last_kept):
    """basq_epoch_turn_f64 (classes + columns + fold slot + descriptor in one launch) against the stand-in that
    tests/test_epochs.py holds against explicit per-candidate columns: bit for bit."""
    cpu = CpuStandInOps()
    n = S // 2
    g = torch.Generator().manual_seed(C * 1000 + e * 10 + t)
    reg_blocks = 3 * C
    nb = reg_blocks + e
    R = nb * S + t
    P = torch.rand(C + E_in + 2, rows, S, generator=g, dtype=torch.float64)
    kept = torch.sort(torch.randperm(S - 1, generator=g)[:n - 1 if last_kept else n])[0].to(torch.int32)
    if last_kept:
        kept = torch.cat([kept, torch.tensor([S - 1], dtype=torch.int32)])
    keep_rank = torch.full((S,), -1, dtype=torch.int32)
    keep_rank[kept.long()] = torch.arange(n, dtype=torch.int32)
    kept_full = torch.cat([kept, torch.zeros(S - n, dtype=torch.int32)])
    w_star = torch.rand(S, generator=g, dtype=torch.float64) + 0.5
    tot = torch.rand(S, generator=g, dtype=torch.float64) + 0.5
    info = torch.tensor([n, 0], dtype=torch.int32)
    E_out = (E_in * n + S - 1) // S
    geo_c = cpu.geo_init(2, R, S, reg_blocks * S)
    want = cpu.epoch_turn(P.clone(), C, E_in, E_out, kept_full, keep_rank, w_star, tot, info, geo_c[0], geo_c[1])
    geo_d = hip_ops.geo_init(2, R, S, reg_blocks * S)
    dev = [hip_ops.to_device(x) for x in (P, kept_full, keep_rank, w_star, tot, info)]
    got = hip_ops.epoch_turn(dev[0], C, E_in, E_out, dev[1], dev[2], dev[3], dev[4], dev[5], geo_d[0], geo_d[1])
    torch.cuda.synchronize()
    assert hip_ops.to_host(geo_d[1], "gt").tolist() == geo_c[1].tolist()
    assert torch.equal(got.cpu(), want)
    # a round that did not keep half of the sets: zeros for the columns, the sticky flag in the descriptor
    info_bad = hip_ops.to_device(torch.tensor([n - 1, 0], dtype=torch.int32))
    got = hip_ops.epoch_turn(dev[0], C, E_in, E_out, dev[1], dev[2], dev[3], dev[4], info_bad, geo_d[0], geo_d[1])
    assert not got[C // 2:].any() and int(hip_ops.to_host(geo_d[1], "gt")[3]) == 1


@pytest.mark.parametrize("shard", [None, (0, 700), (611, 903), (2_000, 303), (2_290, 13)])
def test_compaction_of_several_rounds_kernel(hip_ops, shard):
    """basq_reweight_compact_rounds_f64 = basq_reweight_compact_geo_f64 round after round (rows, weights -- the same roundings --
    and ids), launch and buffers sized for upper bounds; a violated round writes nothing.  ``shard``: a rank's slice
    ``[off, off + Rl)`` of the positions (mid-block starts and ends, a slice inside the ragged tail)."""
    S, n, d = 40, 20, 4
    kp = hip_ops.kp(d)
    R = 57 * S + 23
    g = torch.Generator().manual_seed(5)
    off, Rl = shard if shard is not None else (0, R)
    R_max = Rl + 100
    cand = hip_ops.to_device(torch.rand(R_max, kp, generator=g, dtype=torch.float64))
    mu = hip_ops.to_device(torch.rand(R_max, generator=g, dtype=torch.float64))
    wx = hip_ops.to_device(torch.rand(R_max, generator=g, dtype=torch.float64))
    gid = hip_ops.to_device(torch.arange(R_max, dtype=torch.int64) * 7 + 3)
    geo = hip_ops.geo_init(8, R, S, 48 * S, off, Rl)
    outs, cur = [], (cand, mu, gid, wx)
    for r in range(4):
        last = r % 2 == 0
        kept = torch.sort(torch.randperm(S - 1, generator=g)[:n - 1 if last else n])[0].to(torch.int32)
        if last:
            kept = torch.cat([kept, torch.tensor([S - 1], dtype=torch.int32)])
        keep_rank = torch.full((S,), -1, dtype=torch.int32)
        keep_rank[kept.long()] = torch.arange(n, dtype=torch.int32)
        o = dict(keep_rank=hip_ops.to_device(keep_rank), w_star=hip_ops.to_device(torch.rand(S, generator=g, dtype=torch.float64) + 0.5),
                 tot=hip_ops.to_device(torch.rand(S, generator=g, dtype=torch.float64) + 0.5),
                 info=hip_ops.to_device(torch.tensor([n, 0], dtype=torch.int32)))
        hip_ops.round_next(geo[r], o["info"], o["keep_rank"], S, -1, True, geo[r + 1])
        outs.append(o)
        Rn_up = (R >> (r + 1)) + 2 * S
        cur = hip_ops.reweight_compact_geo(*cur, geo[r], geo[r + 1], o["info"], R_max, S, kp, o["keep_rank"], o["w_star"], o["tot"],
                                           Rn_up, n)
    Rn = int(hip_ops.to_host(geo[4], "g4")[7])                 # this rank's share of the survivors
    got = hip_ops.reweight_compact_rounds(cand, mu, gid, wx, geo, outs, R_max, S, kp, Rn + 31, n)
    for a, b in zip(got, cur):
        assert torch.equal(a[:Rn], b[:Rn])
    # three of the four rounds only (a view of the table from row 1 on)
    c1 = hip_ops.reweight_compact_geo(cand, mu, gid, wx, geo[0], geo[1], outs[0]["info"], R_max, S, kp, outs[0]["keep_rank"],
                                      outs[0]["w_star"], outs[0]["tot"], R, n)
    got3 = hip_ops.reweight_compact_rounds(*c1, geo[1:], outs[1:], R, S, kp, Rn + 5, n)
    for a, b in zip(got3, cur):
        assert torch.equal(a[:Rn], b[:Rn])
    # a round that kept another number of sets than expected, a failed elimination, the sticky flag of an earlier round: NOTHING is
    # written (the outputs are poisoned first and must come back untouched; the C entry is called directly to keep the buffers)
    import ctypes as C

    from basq_amd._ops import _ptr

    rows = Rn + 31
    co, mo, go, wo = hip_ops.empty(rows, kp), hip_ops.empty(rows), hip_ops.empty(rows, dtype=torch.int64), hip_ops.empty(rows)

    def call(expect, infos=None, geo_rows=None):
        co.fill_(-7.0), mo.fill_(-7.0), go.fill_(-7), wo.fill_(-7.0)
        arr = lambda key, src: (C.c_void_p * len(outs))(*[o[key].data_ptr() for o in src])      # noqa: E731
        use = outs if infos is None else [dict(o, info=i) for o, i in zip(outs, infos)]
        rc = hip_ops.lib.basq_reweight_compact_rounds_f64(_ptr(cand), _ptr(mu), _ptr(gid), _ptr(wx), _ptr(geo if geo_rows is None else geo_rows),
                                                          len(outs), arr("keep_rank", use), arr("w_star", use), arr("tot", use), arr("info", use),
                                                          R_max, S, kp, rows, expect, _ptr(co), _ptr(mo), _ptr(go), _ptr(wo), hip_ops._stream())
        assert rc == 0
        torch.cuda.synchronize()
        return bool((co == -7.0).all() and (mo == -7.0).all() and (go == -7).all() and (wo == -7.0).all())

    assert not call(n) or Rn == 0                                # the regular call does write (unless this shard has no survivor)
    assert call(n + 1)                                           # another number of kept sets than the host sized for
    failed = [o["info"] for o in outs]
    failed[2] = hip_ops.to_device(torch.tensor([n, 1], dtype=torch.int32))
    assert call(n, infos=failed)                                 # an elimination that stopped early in the third round
    flagged = geo.clone()
    flagged[1, 3] = 1
    assert call(n, geo_rows=flagged)                             # the sticky violation flag in the second round's descriptor
