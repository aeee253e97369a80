"""Hardware-counter summary of the dominant kernel from the rocprofv3 --pmc passes of ``tools/gpu_jobs.sh pmc``.

    python tools/pmc_summary.py gpurun_out/<tag>  > gpurun_out/<tag>/pmc_summary.json      (-> profiles/r03_pmc.json)

Each pass profiled ``python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline-batch`` with ONE counter group
(never together with a trace domain): ``pmc_fetch`` = FETCH_SIZE, ``pmc_write`` = WRITE_SIZE, ``pmc_pipe`` =
SQ_INSTS_VALU, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE.  The bench runs 1 warm-up + 2 timed + 1 traced
batch = 4 batches; the block-sum launches of the LAST THREE are averaged per batch (the first pays cold caches).

Units and corrections (MI355X micro-architecture guide, HBM / rocprofv3 sections): FETCH_SIZE and WRITE_SIZE are KiB of
L2 <-> fabric requests, Infinity-Cache hits included.  gfx950 tallies the 128-B requests of 16-B-per-lane streaming reads
at 64 B (FETCH_SIZE = half the bytes); this kernel's loads are 8 B per lane (64 lanes x 8 B = 512-B wave requests), an
"uncalibrated" width in the guide's words -- so the file reports the RAW counters and the calibration: the largest launch
(14 of the 16 residue classes of round 1) must fetch its compulsory input -- every packed candidate row and weight once (the
XCD-aware map keeps a candidate slice in ONE L2), the Nystrom rows once -- and the counter reads 0.9-1.0x that figure (0.91
inside bench.py, where part of the rows is still cached from pack_points_kernel; 1.0x cold, profiles/r02_traffic.json): the
half-count does not apply to this access pattern (it would put the counter at 0.5x).  The traffic is two orders below what
8 TB/s would carry in the kernel's duration: this kernel is bound by the fp64 pipe, not by HBM.
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

KERNEL = "blocksum_kernel"


def rows(path):
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    out = []
    for fn in files:
        with open(fn) as f:
            for r in csv.DictReader(f):
                out.append(r)
    return out


def per_dispatch(path):
    """-> list of (dispatch id, kernel name, grid, {counter: value}, duration ns) in dispatch order."""
    acc = {}
    for r in rows(path):
        key = int(r["Dispatch_Id"])
        e = acc.setdefault(key, dict(name=r["Kernel_Name"], grid=int(r["Grid_Size"]), c=defaultdict(float),
                                     ns=int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    return [acc[k] for k in sorted(acc)]


def batches(disp, n_batches=4):
    """Block-sum launches grouped per batch: consecutive launches in dispatch order, ``n_batches`` batches per pass (warm-up + two
    steps + the parity batch).  At the headline size 16 per batch since the end of round 4 (one 16-class launch with the irregular
    blocks of round 1, the fresh evaluations of rounds 6 and 11, the irregular blocks of the rounds in between); 17 before
    (14 + 2 classes in round 1)."""
    bs = [d for d in disp if KERNEL in d["name"]]
    if not bs or len(bs) % n_batches:
        raise SystemExit(f"{len(bs)} block-sum launches: not a multiple of {n_batches} batches")
    per_batch = len(bs) // n_batches
    return [bs[i:i + per_batch] for i in range(0, len(bs), per_batch)]


def main():
    out = sys.argv[1]
    res = {"kernel": "blocksum_kernel<3,0,4> -- all launches of one headline batch (N=1e6, d=10, n=100, m=1e4): 16 residue "
                     "classes in round 1 (one launch since the end of round 4; 14 + 2 before), fresh evaluations in rounds 6 and 11+; round 6: "
                     "the irregular candidates once per EPOCH (their message columns, on the side stream) instead of once per round"}
    res["commit"] = sys.argv[2] if len(sys.argv) > 2 else None      # (the GPU box has no .git: pass `git rev-parse --short HEAD`)
    if res["commit"] is None:
        try:
            res["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        except OSError:
            pass
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from basq_amd._build import source_hash

    # the binary these counters belong to: bench.py replays the file only while the tree's kernel sources hash to the same value
    res["kernel_source_sha256"] = source_hash()
    per = {}
    for tag in ("fetch", "write", "pipe"):
        g = batches(per_dispatch(os.path.join(out, f"pmc_{tag}")))
        per[tag] = g[1:] if len(g) > 1 else g                       # drop the warm-up batch
    def mean_sum(groups, ctr):
        vals = [sum(d["c"].get(ctr, 0.0) for d in grp) for grp in groups]
        return sum(vals) / len(vals) if vals else None

    def mean_big(groups, ctr):
        vals = [max(grp, key=lambda d: d["grid"])["c"].get(ctr, 0.0) for grp in groups]
        return sum(vals) / len(vals) if vals else None

    fetch_kib, write_kib = mean_sum(per["fetch"], "FETCH_SIZE"), mean_sum(per["write"], "WRITE_SIZE")
    if fetch_kib is None or write_kib is None:
        print(json.dumps({"error": "no block-sum dispatches found", "dir": out}))
        return
    res["launches_per_batch"] = len(per["fetch"][0])
    res["fetch_KiB_per_batch"] = fetch_kib
    res["write_KiB_per_batch"] = write_kib
    res["hbm_bytes_per_batch"] = int((fetch_kib + write_kib) * 1024)
    # calibration on the largest launch: the 16 classes of round 1's regular region (4 992 blocks x 200 candidates x 96 B packed
    # rows + mu 8 B; 14 of the 16 -- 873 600 candidates -- while the round was launched as 14 + 2) + the Nystrom rows
    # (10 048 x 96 B), fetched once when the XCD map keeps a candidate slice in ONE L2
    big_fetch = mean_big(per["fetch"], "FETCH_SIZE")
    n_big = 873_600 if res["launches_per_batch"] == 17 else 998_400      # (17 launches per batch: round 1 as 14 + 2 classes, rounds 2-3)
    compulsory = n_big * (96 + 8) + 10_048 * 96
    res["largest_launch"] = {"fetch_KiB": big_fetch, "write_KiB": mean_big(per["write"], "WRITE_SIZE"),
                             "compulsory_fetch_bytes": compulsory,
                             "fetch_over_compulsory": big_fetch * 1024 / compulsory if big_fetch else None,
                             "note": "the residue classes of round 1 in one launch; part of the packed rows is still cached from "
                                     "pack_points_kernel; cold inputs: 1.0x (profiles/r02_traffic.json)"}
    # algorithmic bytes per batch by SURVEY 8d ((8d + 16) per candidate of every evaluated launch + 8 m d per launch) and what
    # the epoch formulation adds on purpose: the class partials [C + 1, m_ext, S] written once (then read once by the projection)
    res["algorithmic_note"] = ("SURVEY 8d input bytes: (8d+16) B per evaluated candidate + 8 m d per launch = ~1.0e8 B per "
                               "batch; the residue-class formulation additionally WRITES its class partials (17 x 16 MB in "
                               "round 1, 17 x 16 MB in round 6): deliberate traffic, 45 GB/s at this kernel's duration")
    if per["pipe"]:
        valu = mean_sum(per["pipe"], "SQ_INSTS_VALU")
        mfma_busy = mean_sum(per["pipe"], "SQ_VALU_MFMA_BUSY_CYCLES")
        gui = mean_sum(per["pipe"], "GRBM_GUI_ACTIVE")
        sq_busy = mean_sum(per["pipe"], "SQ_BUSY_CYCLES")
        res["pipe"] = {"SQ_INSTS_VALU": valu, "SQ_VALU_MFMA_BUSY_CYCLES": mfma_busy, "GRBM_GUI_ACTIVE_sum8xcd": gui,
                       "SQ_BUSY_CYCLES": sq_busy}
        if gui and mfma_busy is not None:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip; an fp64 VALU wave instruction holds the
            # pipe 4 cycles, MFMA_BUSY counts the matrix instruction's busy cycles
            simd_cycles = 1024.0 * gui / 8.0
            res["mfma_util"] = mfma_busy / simd_cycles                       # matrix-instruction busy cycles / SIMD cycles
            res["fp64_pipe_busy"] = (4.0 * valu + mfma_busy) / simd_cycles   # + 4 issue cycles per VALU wave instruction
    # stall / issue counters (one pass per group, `tools/gpu_jobs.sh pmc_stalls`): per-batch sums over the block-sum launches and the
    # ratios the review asked for -- where the fp64 pipe's idle share goes
    stalls = {}
    for d_ in sorted(glob.glob(os.path.join(out, "pmc_stall_*"))):
        if not os.path.isdir(d_):
            continue
        try:
            g = batches(per_dispatch(d_))
        except SystemExit:
            continue
        g = g[1:] if len(g) > 1 else g
        names = sorted({c for grp in g for dsp in grp for c in dsp["c"]})
        for c in names:
            stalls[c] = mean_sum(g, c)
    if stalls:
        wc = stalls.get("SQ_WAVE_CYCLES")
        ratios = {}
        if wc:
            for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                      "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_ANY", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
                if stalls.get(c) is not None:
                    ratios[c + "_per_wave_cycle"] = stalls[c] / wc
        res["stalls"] = {"per_batch": stalls, "ratios": ratios,
                         "note": "SQ_WAVE_CYCLES counts (in quad-cycle units on this part) the cycles waves spent resident; "
                                 "WAIT_INST_ANY: waiting for any instruction issue; WAIT_ANY: waiting on s_waitcnt; ACTIVE_INST_x: "
                                 "cycles an instruction of type x was executing, per wave"}
    res["source"] = ("rocprofv3 --pmc, one counter group per pass (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU "
                     "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) over `python bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --no-roofline-batch`; mean over the batches after the first; KiB x 1024, raw (8-B-per-lane "
                     "loads: no half-count correction applied, see largest_launch)")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
