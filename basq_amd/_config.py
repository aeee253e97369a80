"""Switches of the recombination engine, in ONE place (tests and ``tools/ab_engine.py`` flip them for A/B runs).

Every switch selects between two implementations of the same arithmetic; none of them changes what is selected (the
golden tests run both sides of the ones that matter).
"""
from __future__ import annotations

# Host LAPACK calls on small matrices (100 x 200 SVD, 1e4 x 99 QR) are slower, not faster, on a many-core
# host with every core in the team (measured: 30 ms per 100x200 gesdd with 128 threads vs ~2 ms with 8).
HOST_LAPACK_THREADS = 8          # tall QR of the fallback path
HOST_SVD_THREADS = 1             # 100x200 / 99x99 SVDs: fastest single-threaded (profiles/r01_host_lapack_threads.txt)

# The range finder's products on basq_skinny_gemm_f64 (False: library GEMMs through torch, for A/B comparisons).
OWN_RANGE_GEMM = True
# CholeskyQR's factor + triangular solve in ONE launch whose solvers start on a column panel as soon as the factor has
# published it (basq_cholqr_f64; same bits as the two separate launches).  False: the two launches of round 2.
FUSED_CHOLQR = True
# The final [q, m] SVD of torch.svd_lowrank only rotates the orthonormal basis Q of the range, and the recombination is invariant
# under such rotations (tests/test_oracle.py::test_selection_invariant_under_basis_rotations): False (default) stops at U = -Q^T --
# one [m, m] x [m, q] product, an LQ reduction, the q x q host SVD and the batch's first host wait less; True: round 3's path.
BASIS_SVD = False
# The GPU range finder may be switched off (tests compare both paths).
GPU_RANGE_FINDER = True
# Per-round null space (:140-143) from the bidiagonalisation's right reflectors on the GPU (basq_nullspace_f64)
# instead of a host LAPACK SVD; False restores the host path (same rows to ~1e-13, see tests).
GPU_NULLSPACE = True

# Multi-rank: every rank reduces the gathered message itself (the kernels sum in a fixed order, so all ranks obtain
# the same survivors bit for bit) instead of rank 0 reducing and broadcasting the result: one collective less per round.
REPLICATED_REDUCTION = True
# Several batches in flight on several ranks (run_many): batch k's reductions run on rank k mod G ONLY and their outcome
# (3 M + 1 doubles) is broadcast, stream-ordered, on a process group of the batch's own -- the chain of single-work-group
# kernels (6.3 ms of a 21-ms batch at the headline size) then divides by G like the wide kernels do.  A single batch at a
# time keeps the replicated form above (nothing to overlap the owner's chain with, one collective less per round).
OWNER_RANK_REDUCTION = True
SHARDED_CHOLQR = True            # multi-rank range finder: Gram product + triangular solve of every CholeskyQR pass on the rank's
                                 # own rows (q x q partial Grams all-gathered, the q x q factor replicated); False: on all m rows
SHARDED_BASIS = True             # multi-rank: split the range finder's Gram products over the ranks (False: rank 0 only)

# Round-1 block sums deferred behind the range finder's launches, to run while the host did the range finder's q x q SVD (rounds
# 2-3: LATE_CHUNKS = 1, LATE_CLASSES = 2).  Round 4: the range finder no longer waits for the host (BASIS_SVD above), and the 14 + 2
# split only costs the 2-class launch its tail efficiency: 19.99 ms per batch with 2 deferred classes, 19.77 with none
# (profiles/r06_q_late_classes_ab.txt).  With BASIS_SVD = True the old values are the better ones.
LATE_CHUNKS = 0                  # chunks of the round-1 block sums deferred behind the range finder (0 = none)
LATE_CLASSES = 0                 # the same in class mode
LATE_CLASSES_PIPELINED = 0       # with several batches in flight another batch's kernels fill that gap: nothing is deferred

# Residue-class block sums: evaluate the pairwise kernel once per EPOCH of log2(C) + 1 rounds.  The chunks of the block
# sums are the residue classes of the block index modulo C; a round that keeps exactly half of the sets sends the
# survivor of (block b, kept rank k) to (block b // 2, set (b % 2) * n + k), so the next round's sums -- again per class,
# modulo C / 2 -- are a gather + rescale of this round's (``basq_regroup_classes_f64``): no candidate is touched.  Only
# the blocks beyond a multiple of C and the ragged tail (< (C + 1) * S points, halving every round) are evaluated directly.
CLASS_SUMS = True
MAX_CLASSES = 16                 # classes at the start of an epoch (power of two): 16 -> the kernel runs in rounds 1, 6, 11
# Inside an epoch the candidates OUTSIDE the residue classes (the < C full blocks behind the regular region + the ragged tail) are
# carried as message columns and regrouped with the classes (basq_epoch_turn_f64): no block sums, projection or compaction between
# two eliminations; their columns are evaluated beside the epoch's first chain on a side stream (basq_amd/_epochs.py; one rank,
# BASQ variant, no noise diagonal).  False: every round evaluates them afresh (round 5's path).
IRR_COLUMNS = True
# Batches in flight evaluate those columns on their own stream (False) or on a side stream per batch in flight (True; A/B:
# tools/bench_many.py --set PIPELINED_SIDE_STREAM=1).  A side stream SHARED by the batches lost (profiles/r08_e_*).
PIPELINED_SIDE_STREAM = False
# Rounds driven by a device-resident descriptor, no host wait per round (any rank count; structured kernels except WSABI-M).
ASYNC_ROUNDS = True

# The range finder's uniforms (8 MB at the headline size) on a copy stream, so that the transfer runs beside the round-1 block
# sums instead of behind them (A/B: tools/ab_engine.py RAND_COPY_STREAM 0 1).
RAND_COPY_STREAM = True

# A synchronous batch waits for the GPU 5-6 times (range finder's SVD, the descriptor table, the last rounds): poll the event
# instead of the runtime's blocking wait (A/B: tools/ab_engine.py SPIN_WAIT 0 1).
SPIN_WAIT = False

PSD_EIG_MAX_M = 4096             # _make_cov_psd: largest Gram whose spectrum is checked (SOBER/_utils.py:122-124)
