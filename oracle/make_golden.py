"""Generate ``tests/golden/*.json`` by running the REFERENCE ITSELF in this container.

Run from the repo root (only where ``/root/reference`` is mounted; the reference
is imported, never copied):

    python -m oracle.make_golden                # all fast cases
    python -m oracle.make_golden --slow         # also the N=1e6 / N=5e5 BASELINE configs (minutes each)
    python -m oracle.make_golden --only NAME... # selected cases

For every case in ``tests/cases.py`` the imported ``BASQ._rchq.recombination``
(``/root/reference/BASQ/_rchq.py:4-25``) is called in float64 with the oracle's
kernel callable and ``torch.manual_seed(case.torch_seed)`` immediately before the
call.  ``Tchernychova_Lyons_CAR`` and ``ker_svd_sparsify`` are wrapped (module
attribute patching, the reference source is untouched) to record per-round
intermediates.  A fixture holds only data: the case parameters, digests of the
regenerated inputs, and the reference's outputs.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

REF_ROOT = "/root/reference"
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _digest(t: torch.Tensor) -> str:
    a = np.ascontiguousarray(t.detach().cpu().numpy().astype("<f8", copy=False))
    return hashlib.sha256(a.tobytes()).hexdigest()


def _import_reference():
    sys.dont_write_bytecode = True           # the mount is read-only
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import BASQ._rchq as ref                 # noqa: N813  (namespace package, torch-only module)
    return ref


def run_case(c, ref):
    from basq_amd.pools import pool_digest
    from tests.cases import build_oracle_kernel, build_pool

    torch.set_default_dtype(torch.float64)   # the reference allocates in the default dtype (_rchq.py:53,79)
    pts_rec, pts_nys = build_pool(c)
    kernel, state = build_oracle_kernel(c)

    rounds = []
    captured = {}
    orig_car = ref.Tchernychova_Lyons_CAR
    orig_basis = ref.ker_svd_sparsify

    def car_spy(X, mu, device, DEBUG=False):
        rec = dict(M=int(X.shape[0]), q=int(X.shape[1]), X_digest=_digest(X), X_abs_sum=float(X.abs().sum()),
                   X_row0=[float(v) for v in X[0, : min(4, X.shape[1])]], mu_in=[float(v) for v in mu])
        out = orig_car(X, mu, device, DEBUG)
        rec["kept"] = [int(v) for v in out[1]]
        rec["w_star"] = [float(v) for v in out[0]]
        rounds.append(rec)
        return out

    def basis_spy(pt, s, kern, device):
        # the randn draw svd_lowrank is about to make (same generator state), for diagnostics only
        st = torch.get_rng_state()
        probe = torch.randn(pt.shape[0], s, dtype=torch.float64)
        torch.set_rng_state(st)
        captured["R_digest"] = _digest(probe)
        S, U = orig_basis(pt, s, kern, device)
        captured["U_shape"] = list(U.shape)
        captured["U_digest"] = _digest(U)
        captured["U_abs_sum"] = float(U.abs().sum())
        captured["U_head"] = [float(v) for v in U[0, :4]]
        captured["S_head"] = [float(v) for v in S[:4]]
        if U.numel() <= 64 * 64:
            captured["U_full"] = [[float(v) for v in row] for row in U]
        return S, U

    ref.Tchernychova_Lyons_CAR = car_spy
    ref.ker_svd_sparsify = basis_spy
    try:
        torch.manual_seed(c["torch_seed"])
        t0 = time.time()
        idx, w = ref.recombination(pts_rec, pts_nys, c["n"], kernel, torch.device("cpu"), init_weights=0)
        dt = time.time() - t0
    finally:
        ref.Tchernychova_Lyons_CAR = orig_car
        ref.ker_svd_sparsify = orig_basis

    # reference-free invariants measured on the reference's own output
    fixture = dict(
        case=c,
        pool_digest=pool_digest(pts_rec),
        obs_digest=None if state is None else pool_digest(state["Xobs"]),
        idx=[int(v) for v in idx],
        w=[float(v) for v in w],
        w_sum_minus_1=float(w.sum() - 1.0),
        n_rounds=len(rounds),
        rounds=rounds,
        basis=captured,
        reference_seconds=dt,
        host=dict(torch=torch.__version__, threads=torch.get_num_threads()),
    )
    return fixture


def slim(fx):
    """Keep fixtures small: per-round mu_in/w_star only for the first and last round of big cases."""
    if len(fx["rounds"]) > 4:
        for r in fx["rounds"][1:-1]:
            r.pop("mu_in", None)
    return fx


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--slow", action="store_true")
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args(argv)
    from tests.cases import CASES

    ref = _import_reference()
    os.makedirs(OUT_DIR, exist_ok=True)
    for c in CASES:
        if args.only and c["name"] not in args.only:
            continue
        if c["slow"] and not (args.slow or args.only):
            continue
        path = os.path.join(OUT_DIR, c["name"] + ".json")
        if os.path.exists(path) and not args.force:
            print("keep", path)
            continue
        fx = slim(run_case(c, ref))
        with open(path, "w") as f:
            json.dump(fx, f, indent=0, separators=(",", ":"))
        print(f"wrote {path}: {len(fx['idx'])} pts, {fx['n_rounds']} rounds, {fx['reference_seconds']:.1f}s", flush=True)


if __name__ == "__main__":
    main()
