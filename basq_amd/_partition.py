"""Closed-form bookkeeping of the divide-and-conquer rounds (host side, pure Python).

The reference keeps the survivors as one ascending index list ``idx_story`` and derives
everything from positions in it (``BASQ/_rchq.py:76-78,91,107-130``):

* ``S = 2 (q + 1)`` sets; position ``p`` belongs to set ``p % S`` for ``p < nb * S``
  (``nb = R // S`` full blocks) and to set ``S - 1`` otherwise (the ragged tail);
* after the reduction keeps the sets ``kept`` (ascending), a survivor at block ``b`` whose
  set has rank ``k`` among ``kept`` lands at position ``b * n_keep + k``; tail survivors
  (only if set ``S-1`` was kept) follow at ``nb * n_keep + (p - nb * S)``.

Because the map is closed-form, a rank that holds the contiguous positions
``[off, off + Rl)`` can compute where its survivors go, and how many it keeps, without any
collective: that is what makes the pool shard over GPUs with a single small exchange per
round.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class RoundGeometry:
    R: int          # surviving candidates (all ranks)
    S: int          # number of sets this round
    nb: int         # full blocks: R // S
    n_full: int     # nb * S
    n_tail: int     # R - n_full

    @staticmethod
    def of(R: int, S: int) -> "RoundGeometry":
        nb = R // S
        return RoundGeometry(R, S, nb, nb * S, R - nb * S)


def survivors_before(P: int, geo: RoundGeometry, kept: list) -> int:
    """Number of survivors among global positions ``[0, P)`` when the sets ``kept`` survive."""
    n_keep = len(kept)
    last_kept = n_keep > 0 and kept[-1] == geo.S - 1
    if P <= geo.n_full:
        b, s = divmod(P, geo.S)
        below = sum(1 for k in kept if k < s)
        return b * n_keep + below
    return geo.nb * n_keep + ((P - geo.n_full) if last_kept else 0)


def next_shard(off: int, Rl: int, geo: RoundGeometry, kept: list):
    """-> (new_off, new_Rl) for the rank holding positions ``[off, off + Rl)``."""
    lo = survivors_before(off, geo, kept)
    hi = survivors_before(off + Rl, geo, kept)
    return lo, hi - lo


def initial_shards(N: int, world: int):
    """Contiguous, near-equal shards of the candidate ids ``0..N-1``: [(off, Rl)] per rank."""
    base, extra = divmod(N, world)
    out, off = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((off, n))
        off += n
    return out


def local_blocks(off: int, Rl: int, geo: RoundGeometry) -> int:
    """How many global blocks intersect this rank's block positions."""
    lim = min(off + Rl, geo.n_full)
    if lim <= off:
        return 0
    return (lim + geo.S - 1) // geo.S - off // geo.S


# Launch geometry of the block-sum kernel (must mirror basq_pairwise.hip): a wave covers rows_per_wave(kk) Nystrom rows
# x SETS_PER_WAVE sets, 4 waves per work-group; RESIDENT_WAVES = 256 CUs x 4 SIMDs x 3 waves (register-limited).
SETS_PER_WAVE = 16
RESIDENT_WAVES = 256 * 4 * 3


CHUNK_EFFICIENCY = 0.96          # smallest chunk count reaching this tail efficiency wins (A/B: tools/ab_engine.py)


def rows_per_wave(kk: int) -> int:
    """BASQ_JT_FOR(KK) * 16 in basq_pairwise.hip: 64 rows per wave up to KP = 16, 32 beyond (BASQ_JT_LARGE_FROM = 5)."""
    return 32 if kk >= 5 else 64


def choose_chunks(n_local_blocks: int, m: int, S: int, kk: int = 3, max_chunks: int = 32, min_blocks: int = 4,
                  resident: int = RESIDENT_WAVES, geometry=None) -> int:
    """Number of chunks the block loop is split into.

    All waves of a launch do the same amount of work, so the launch takes ``ceil(waves / resident)``
    "rounds": 4800 waves on 3072 resident slots cost 2 rounds for 1.56 rounds of work (measured: 22 % of
    the kernel time).  Pick the smallest chunk count whose efficiency ``(waves/resident) /
    ceil(waves/resident)`` reaches 96 % (fewer chunks = less partial-sum traffic for the projection), else
    the most efficient one.  ``kk`` = packed row length / 4.  ``geometry`` = (Nystrom rows per work-group, sets per
    wave) of the block-sum form in use (``HipOps.blocksum_geometry``); default: the MFMA form's.
    """
    rows_per_block, sets_per_wave = geometry or (4 * rows_per_wave(kk), SETS_PER_WAVE)
    per_chunk = ((m + rows_per_block - 1) // rows_per_block) * 4 * ((S + sets_per_wave - 1) // sets_per_wave)
    cap = max(1, min(max_chunks, n_local_blocks // min_blocks))
    best, best_eff = 1, -1.0
    for c in range(1, cap + 1):
        rounds = per_chunk * c / resident
        eff = rounds / max(1.0, float(-(-per_chunk * c // resident)))
        if eff >= CHUNK_EFFICIENCY:
            return c
        if eff > best_eff + 1e-9:
            best, best_eff = c, eff
    return best
