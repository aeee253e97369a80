"""Bug hunt beyond the committed fuzz tests: the same differential fuzzes (structured kernels through ``recombination``, the SOBER
variant through ``sober.recombination``) over OTHER case lists.

    python tools/fuzz_more.py --seeds 1 2 3 --count 100

A case outside the bar (indices differ, or weights > 1e-5 relative) is printed together with the reference's own sensitivity (the oracle
against itself with base-kernel values moved by <= 1 ulp): only cases the reference itself reproduces are bugs.
"""
import argparse
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_default_dtype(torch.float64)
import basq_amd                                                    # noqa: E402
from basq_amd import sober                                         # noqa: E402
from oracle.rchq_oracle import recombination_oracle, recombination_sober_oracle   # noqa: E402
from tests.cases import (build_oracle_kernel, build_perturbed_oracle_kernel, build_pool, build_product_kernel,   # noqa: E402
                         observation_gram_condition, structured_fuzz_cases)
from tests.test_sober import _sober_fuzz_cases                     # noqa: E402

DEV = torch.device("cuda", 0)


def dev(ia, wa, ib, wb):
    same = ia.tolist() == ib.tolist()
    return same, (((wa - wb).abs() / wb).max().item() if same and len(wb) else (0.0 if same else float("inf")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--count", type=int, default=100)
    a = ap.parse_args()
    warnings.simplefilter("ignore")
    for seed in a.seeds:
        ok = unstable = bugs = skipped = 0
        for i, c in enumerate(structured_fuzz_cases(seed, a.count)):
            pts, nys = build_pool(c)
            ko, state = build_oracle_kernel(c)
            A = ko(nys, nys)
            ev = torch.linalg.eigvalsh(0.5 * (A + A.T))
            if int((ev > 1e-10 * ev.abs().max()).sum()) < min(c["n"] - 1, c["m"]):
                skipped += 1
                continue
            try:
                torch.manual_seed(c["torch_seed"])
                io, wo = recombination_oracle(pts, nys, c["n"], ko)
            except Exception as e:                                   # noqa: BLE001
                print(f"  structured seed {seed} case {i}: oracle raised {type(e).__name__}")
                skipped += 1
                continue
            try:
                torch.manual_seed(c["torch_seed"])
                ie, we = basq_amd.recombination(pts, nys, c["n"], build_product_kernel(c, state), DEV)
            except Exception as e:                                   # noqa: BLE001
                print(f"  structured seed {seed} case {i}: ENGINE raised {type(e).__name__}: {str(e)[:100]} -- {c}")
                bugs += 1
                continue
            same, rel = dev(ie.cpu(), we.cpu(), io, wo)
            if same and rel <= 1e-5:
                ok += 1
                continue
            moved, ref_rel, explained, n_pat = False, 0.0, False, 0
            for s in range(1, 41):                                  # five patterns as in the committed tests; up to forty before a case
                torch.manual_seed(c["torch_seed"])                   # is called unexplained (seed 8 case 33: 4 of 40 move the indices)
                ip, wp = recombination_oracle(pts, nys, c["n"], build_perturbed_oracle_kernel(c, s))
                sp, rp = dev(ip, wp, io, wo)
                moved = moved or not sp
                ref_rel = max(ref_rel, rp if sp else 0.0)
                n_pat = s
                explained = moved or (same and rel <= 4 * ref_rel)
                if s >= 5 and explained:
                    break
            unstable += explained
            bugs += not explained
            print(f"  structured seed {seed} case {i}: idx equal {same} rel {rel:.2e} | reference vs itself: idx moves {moved} rel {ref_rel:.2e} "
                  f"| cond {observation_gram_condition(c, state):.1e} | {'explained' if explained else 'BUG?'} ({n_pat} patterns) | N={c['N']} d={c['d']} n={c['n']} m={c['m']} {c['kernel']}")
        print(f"structured seed {seed}: ok={ok} unstable(explained)={unstable} unexplained={bugs} skipped={skipped}", flush=True)
        ok = bugs = skipped = 0
        for c, w0 in _sober_fuzz_cases(a.count, seed=seed + 100):
            pts, nys = build_pool(c)
            ko, state = build_oracle_kernel(c)
            A = ko(nys, nys)
            ev = torch.linalg.eigvalsh(0.5 * (A + A.T))
            if int((ev > 1e-8 * ev.abs().max()).sum()) < c["m"] or observation_gram_condition(c, state) > 1e6:
                skipped += 1
                continue
            torch.manual_seed(c["torch_seed"])
            io, wo = recombination_sober_oracle(pts, nys, c["n"], ko, None if w0 is None else w0.clone())
            try:
                torch.manual_seed(c["torch_seed"])
                ie, we = sober.recombination(pts, nys, c["n"], build_product_kernel(c, state), DEV, torch.float64, init_weights=w0)
            except Exception as e:                                   # noqa: BLE001
                print(f"  sober seed {seed} {c['name']}: ENGINE raised {type(e).__name__}: {str(e)[:100]}")
                bugs += 1
                continue
            same, rel = dev(ie.cpu(), we.cpu(), io, wo)
            if same and rel <= 1e-5:
                ok += 1
            else:
                moved, ref_rel = False, 0.0                          # the reference's own sensitivity, as for the structured cases
                for s in range(1, 21):
                    torch.manual_seed(c["torch_seed"])
                    ip, wp = recombination_sober_oracle(pts, nys, c["n"], build_perturbed_oracle_kernel(c, s),
                                                        None if w0 is None else w0.clone())
                    sp, rp = dev(ip, wp, io, wo)
                    moved = moved or not sp
                    ref_rel = max(ref_rel, rp if sp else 0.0)
                explained = moved or (same and rel <= 4 * ref_rel)
                bugs += not explained
                ok += explained
                print(f"  sober seed {seed} {c['name']}: reference vs itself: idx moves {moved} rel {ref_rel:.2e} -> {'explained' if explained else 'BUG?'}")
                print(f"  sober seed {seed} {c['name']}: idx equal {same} rel {rel:.2e} | N={c['N']} d={c['d']} n={c['n']} m={c['m']} weights {c['weights']} {c['kernel']}")
        print(f"sober seed {seed}: ok={ok} outside the bar={bugs} skipped={skipped}", flush=True)


if __name__ == "__main__":
    main()
