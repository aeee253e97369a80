"""Epochs without a pairwise evaluation inside (``basq_amd/_epochs.py``): the candidates outside the residue classes carried as
message columns and regrouped with the classes.  Host logic on the CPU stand-in (the kernels' semantics in torch); the HIP
kernels themselves are compared with these stand-ins in ``tests/test_kernels_gpu.py``."""
import pytest
import torch

from basq_amd._engine import EngineTrace, RecombinationEngine
from tests.cases import BY_NAME, build_pool, build_product_kernel, has_golden, load_golden
from tests.cpu_stand_in import CpuStandInOps

# (name, epochs with irregular blocks / tails on the way down)
COLUMN_CASES = ["rbf_1e4", "rbf_2e4_defaults", "rbf_ragged", "rbf_exact_blocks", "matern32_8e3", "matern52_3e4_d32", "rbf_d1",
                # likelihood noise on the block diagonals (the tail block's too), GP posterior and WSABI-L
                "cfg1_posterior_1e4", "wsabil_2e4", "matern52_posterior", "posterior_noise_ragged", "wsabil_noise_ragged"]


def _run(c, columns, trace=None):
    import basq_amd._config as eng

    pts, nys = build_pool(c)
    old = eng.IRR_COLUMNS
    eng.IRR_COLUMNS = columns
    try:
        torch.manual_seed(c["torch_seed"])
        ops = CpuStandInOps()
        idx, w = RecombinationEngine(ops).run(pts, 0, c["N"], nys, c["n"], build_product_kernel(c), trace)
        after = torch.rand(1).item()
    finally:
        eng.IRR_COLUMNS = old
    return idx, w, after, dict(ops.calls)


@pytest.mark.parametrize("name", COLUMN_CASES)
def test_column_epochs_reproduce_the_goldens(name):
    """Indices, weights AND every round's kept sets of the reference-generated golden, on the column path (descriptor-driven,
    traced after the fact) -- and the round-5 path (``IRR_COLUMNS = False``) selects the same batch."""
    if not has_golden(name):
        pytest.skip("fixture not generated")
    c, fx = BY_NAME[name], load_golden(name)
    tr = EngineTrace(host_sync=False)
    idx, w, after, calls = _run(c, True, tr)
    assert idx.tolist() == fx["idx"]
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    assert ((w - gw).abs() / gw).max().item() <= 1e-6
    assert [r["kept"] for r in tr.rounds] == [r["kept"] for r in fx["rounds"]][:len(tr.rounds)]
    idx0, w0, after0, calls0 = _run(c, False)
    assert torch.equal(idx, idx0) and torch.allclose(w, w0, rtol=1e-10, atol=0) and after == after0
    S = 2 * (min(c["n"] - 1, c["m"]) + 1)
    if c["N"] // S >= 8:                                         # at least two classes in round 1: an epoch exists
        assert calls.get("epoch_turn", 0) >= 1 and calls.get("compact_rounds", 0) >= 1
        assert calls0.get("epoch_turn", 0) == 0


def test_column_path_scope():
    """WSABI-M (and the SOBER variant, several ranks) keep the round-5 rounds (``_epochs.eligible``)."""
    for name, uses in (("rbf_2e4_defaults", True), ("cfg1_posterior_1e4", True), ("wsabim_1e4", False), ("wsabil_2e4", True)):
        calls = _run(BY_NAME[name], True)[3]
        assert (calls.get("epoch_turn", 0) > 0) == uses, name


def test_block_capacity_and_turn_capacity_bounds():
    from basq_amd._epochs import block_capacity

    assert block_capacity(1_000_000, 1_000_000, 200, 16) == 8          # 5000 blocks
    assert block_capacity(31_200, 31_499, 200, 16) == 13               # 156 or 157 blocks
    assert block_capacity(3_200, 3_200, 200, 16) == 0
    # an epoch's capacities: e' <= (E n + S - 1) // S, and the true counts stay below them for every outcome of the tail
    S, n = 200, 100
    for e in range(16):
        for t in range(0, S, 37):
            for kept_tail in (0, 1):
                E, ee, tt = e, e, t
                for _ in range(5):
                    E_next = (E * n + S - 1) // S
                    n_irr = ee * n + kept_tail * tt
                    ee, tt = n_irr // S, n_irr % S
                    assert ee <= E_next
                    E = E_next


def test_epoch_turn_stand_in_against_explicit_candidates():
    """``epoch_turn`` (the stand-in the HIP kernel is tested against) vs the thing it replaces: explicit per-candidate columns,
    compacted and re-weighted like candidates, then summed per set.  Two consecutive turns, tail kept / dropped."""
    torch.manual_seed(0)
    ops = CpuStandInOps()
    rows, S, n = 5, 12, 6
    for last_kept in (True, False):
        C = 4
        reg_blocks, e, t = 8, 3, 7                               # 8 regular blocks (multiple of C), 3 behind them, 7 tail points
        nb = reg_blocks + e
        R = nb * S + t
        F = torch.rand(rows, R, dtype=torch.float64)             # one message column per candidate (position order)
        E_in = 3
        P = torch.zeros(C + E_in + 2, rows, S, dtype=torch.float64)
        for b in range(reg_blocks):
            P[b % C] += F[:, b * S:(b + 1) * S]
        for b in range(e):
            P[C + 1 + b] = F[:, (reg_blocks + b) * S:(reg_blocks + b + 1) * S]
        P[C + 1 + E_in][:, :t] = F[:, nb * S:]
        geo = ops.geo_init(4, R, S, reg_blocks * S)
        for turn in range(2):
            kept = torch.sort(torch.randperm(S - 1)[:n - 1 if last_kept else n])[0].to(torch.int32)
            if last_kept:
                kept = torch.cat([kept, torch.tensor([S - 1], dtype=torch.int32)])
            keep_rank = torch.full((S,), -1, dtype=torch.int32)
            keep_rank[kept.long()] = torch.arange(n, dtype=torch.int32)
            w_star = torch.rand(S, dtype=torch.float64) + 0.5
            tot = torch.rand(S, dtype=torch.float64) + 0.5
            info = torch.tensor([n, 0], dtype=torch.int32)
            E_out = (E_in * n + S - 1) // S
            Pn = ops.epoch_turn(P.contiguous(), C, E_in, E_out, kept, keep_rank, w_star, tot, info, geo[turn], geo[turn + 1])
            # the explicit route: every candidate's column through the round
            g = geo[turn]
            R_, n_full = int(g[0]), int(g[1])
            cols = []
            for p in range(R_):
                st = p % S if p < n_full else S - 1
                kr = int(keep_rank[st])
                if kr >= 0:
                    cols.append((F[:, p] * w_star[kr]) / tot[st])
            F = torch.stack(cols, 1)
            Rn = F.shape[1]
            assert Rn == int(geo[turn + 1][0])
            nbn, regn = Rn // S, int(geo[turn + 1][2]) // S
            want = torch.zeros(rows, S, dtype=torch.float64)     # the whole round's message
            for p in range(Rn):
                want[:, p % S if p < nbn * S else S - 1] += F[:, p]
            got = Pn[:C // 2 + 1].sum(0)
            assert torch.allclose(got, want, rtol=1e-13, atol=0)
            assert regn == reg_blocks // 2 ** (turn + 1)
            # the columns themselves
            e_n = nbn - regn
            for b in range(e_n):
                assert torch.equal(Pn[C // 2 + 1 + b], F[:, (regn + b) * S:(regn + b + 1) * S])
            assert torch.equal(Pn[C // 2 + 1 + E_out][:, :Rn - nbn * S], F[:, nbn * S:])
            P, C, E_in = Pn, C // 2, E_out


def test_compaction_of_several_rounds_equals_one_round_at_a_time():
    ops = CpuStandInOps()
    S, n, kp = 10, 5, 3
    R = 20 * S + 7
    g = torch.Generator().manual_seed(3)
    cand = torch.rand(R, kp, generator=g, dtype=torch.float64)
    mu = torch.rand(R, generator=g, dtype=torch.float64)
    gid = torch.arange(R)
    geo = ops.geo_init(5, R, S, 16 * S)
    outs, c1, m1, g1 = [], cand, mu, gid
    for r in range(3):
        kept = torch.sort(torch.randperm(S, generator=g)[:n])[0].to(torch.int32)
        keep_rank = torch.full((S,), -1, dtype=torch.int32)
        keep_rank[kept.long()] = torch.arange(n, dtype=torch.int32)
        o = dict(keep_rank=keep_rank, w_star=torch.rand(S, generator=g, dtype=torch.float64) + 0.5,
                 tot=torch.rand(S, generator=g, dtype=torch.float64) + 0.5, info=torch.tensor([n, 0], dtype=torch.int32))
        ops.round_next(geo[r], o["info"], keep_rank, S, -1, True, geo[r + 1])
        outs.append(o)
        c1, m1, g1, _ = ops.reweight_compact_geo(c1, m1, g1, None, geo[r], geo[r + 1], o["info"], R, S, kp, keep_rank, o["w_star"],
                                                 o["tot"], int(geo[r + 1][0]), n)
    c2, m2, g2, _ = ops.reweight_compact_rounds(cand, mu, gid, None, geo, outs, R, S, kp, int(geo[3][0]), n)
    assert torch.equal(c1, c2) and torch.equal(m1, m2) and torch.equal(g1, g2)
