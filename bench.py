"""bench.py -- recombination batches/sec (N candidates -> n points) at N=1e6, d=10 on 1/2/4/8 MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver launches it as
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` (one rank per GPU, RCCL).
Rank 0 prints ONE JSON line.

* step      = one end-to-end ``recombination(pts_rec[N,d], pts_nys[m,d], n, kernel)`` batch: Nystrom Gram +
              randomised SVD + all divide-and-conquer rounds + reductions -> ``(idx, w)``;
* workload  = BASELINE.json's headline configuration (configs[2]; it fits one GPU): RBF l=2, N=1e6, d=10,
              n=100, m=N/100=1e4, synthetic Gaussian-mixture pool (``basq_amd.pools.gmm_pool``), float64;
* residency = the pool is on the GPU(s) before the clock starts (the reference's boundary hands over torch
              tensors; H2D of the pool is reported separately in DESIGN.md, never in ``value``);
* N > 1     = the SAME batch with the pool sharded over the ranks (strong scaling; one small all-gather +
              broadcast per round);
* roofline  = the dominant kernel (``blocksum_kernel``): algorithmic flops = pairs * (3d + 3)
               over its HIP-event time on the launch stream, against the fp64 vector peak;
* cpu_baseline = the oracle (= the reference's CPU op sequence) on this host's cores, bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_VECTOR_TFLOPS = 78.6      # MI355X nominal: 256 CU x 128 flop/clk x 2.4 GHz (fp64 vector == fp64 matrix peak)
# What the part sustains in a register-only micro-benchmark (tools/microbench.hip, profiles/r02_microbench_fp64_rates.txt,
# 4 waves/SIMD, in-kernel clock 2.32-2.39 GHz): v_fma_f64 only 72.9 TF/s (4.2 cycles per wave instruction),
# v_mfma_f64_16x16x4 only 47.2 TF/s (one issue per 106 cycles), the two interleaved 63.1 TF/s (they do not overlap).
# The block-sum kernel issues BOTH (MFMA distances + VALU exponentials: 83 + 60 of its 143 cycles per 64 pairs), so the mixed
# figure is its practical ceiling; ``frac`` stays against the nominal figure.
MEASURED_FP64_TFLOPS = {"fma_only": 72.9, "mfma_only": 47.2, "mixed_fma_mfma": 63.1}
PEAK_HBM_GBS = 8000.0
POOL_SEEDS = (0, 1, 2, 3, 4)        # seeds 0-4, median

WORKLOAD = dict(N=1_000_000, d=10, n=100, nys_ratio=1e-2, family="rbf", lengthscale=2.0, outputscale=1.0, pool_seed=0)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--N", type=int, default=WORKLOAD["N"])
    ap.add_argument("--d", type=int, default=WORKLOAD["d"])
    ap.add_argument("--n", type=int, default=WORKLOAD["n"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-stride", type=int, default=0, help="run every k-th hot-loop block on the CPU (0 = auto)")
    ap.add_argument("--cpu-threads", type=str, default="", help="comma list of CPU thread counts (default: 8 and all cores)")
    ap.add_argument("--breakdown", action="store_true", help="print a per-phase host timer breakdown to stderr")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists in basq_amd)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # BASQ_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL group, sharded entry, collectives) even with one
    # rank -- the only way to exercise it on a 1-GPU box; never set by the driver.
    force_dist = os.environ.get("BASQ_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        t_init = time.perf_counter()
        dist.init_process_group("nccl", device_id=dev)
        if rank == 0:
            print(f"[bench] RCCL group of {world} up in {time.perf_counter() - t_init:.1f} s", file=sys.stderr)

    import basq_amd
    from basq_amd._partition import initial_shards
    from basq_amd.pools import gmm_pool

    N, d, n = args.N, args.d, args.n
    m = int(N * WORKLOAD["nys_ratio"])
    kern = basq_amd.kernels.StationaryKernel(WORKLOAD["family"], WORKLOAD["lengthscale"], WORKLOAD["outputscale"])
    # pools of seeds 0-4, all resident in HBM before the clock starts (5 x 80 MB); step k uses pool k % 5.
    # Every rank regenerates the same pools (bit-reproducible generator) and keeps its contiguous slice.
    off, Rl = initial_shards(N, world)[rank]
    pools_dev = []
    pool = None
    for sd in POOL_SEEDS:
        p = gmm_pool(N, d, sd)
        if sd == POOL_SEEDS[0]:
            pool = p                                                 # host copy of seed 0: the CPU baseline's input
        pools_dev.append((p[:m].to(dev), p[off:off + Rl].to(dev)))   # PriorSampler: pts_nys = prefix of the pool
        del p

    def one_batch(trace=None, k=0):
        pts_nys, pts_local = pools_dev[k % len(pools_dev)]
        torch.manual_seed(1)                                         # manual_seed(1) before each call
        if world == 1 and not force_dist:
            return basq_amd.recombination(pts_local, pts_nys, n, kern, dev, trace=trace)
        return basq_amd.recombination_sharded(pts_local, off, N, pts_nys, n, kern, dev, trace=trace)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        one_batch(k=k)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        idx, w = one_batch(k=k)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- per-seed latencies (outside the timed region): one synchronised batch per pool seed, median reported ----
    per_seed_ms = []
    for k in range(len(POOL_SEEDS)):
        barrier()
        t1 = time.perf_counter()
        one_batch(k=k)
        barrier()
        per_seed_ms.append(1e3 * (time.perf_counter() - t1))

    # ---- kernel-level roofline: one extra traced batch (HIP events on the launch stream, no host syncs) ----
    tr = basq_amd.EngineTrace(time_kernels=True, host_sync=False)
    one_batch(tr)
    torch.cuda.synchronize()
    k_ms = sum(a.elapsed_time(b) for a, b, _ in tr.kernel_events)
    k_pairs = sum(info["pairs"] for _, _, info in tr.kernel_events)
    k_launches = len(tr.kernel_events)
    # what the REFERENCE's loop evaluates for the same batch (m * sum_r R_r + m^2): the residue-class block sums
    # evaluate about half of it (rounds inside an epoch regroup the previous sums instead)
    ref_pairs = float(m) * sum(r["R"] for r in tr.rounds) + float(m) * m
    flops = k_pairs * (3 * d + 3)
    bytes_alg = sum(info["R"] * (8 * d + 16) + 8 * info["m"] * d for _, _, info in tr.kernel_events)
    achieved_tf = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0

    # HBM bytes of the largest block-sum launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs, KiB units; 8-B-per-lane loads, so the gfx950 16-B half-count does not apply)
    traffic, traffic_src, pipe_busy = None, None, None
    pmc = os.path.join(ROOT, "profiles", "r02_traffic.json")
    if os.path.exists(pmc) and N == WORKLOAD["N"] and d == WORKLOAD["d"] and world == 1:
        with open(pmc) as f:
            rec = json.load(f)
        traffic, traffic_src = rec["bytes_per_launch"], rec["source"]
        try:                                                         # (4 INSTS_VALU + MFMA_BUSY) / SIMD-cycles, same passes
            pipe_busy = float(rec["pipe"]["fp64_pipe_busy"].rsplit("=", 1)[1])
        except (KeyError, ValueError, IndexError):
            pipe_busy = None

    if args.breakdown:
        tb = basq_amd.EngineTrace(time_kernels=False, host_sync=True)
        one_batch(tb)
        if rank == 0:
            print("phase breakdown (s):", {k: round(v, 4) for k, v in tb.timers.items()}, file=sys.stderr)
            print("rounds:", [(r["R"], r["S"], len(r["kept"])) for r in tb.rounds], file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.cpu_baseline import baseline_matrix
        from oracle.kernels_oracle import StationaryOracle

        # SURVEY §8d matrix: {float64, float32} x {8 threads, all host cores}; ~5 s of CPU work per cell at N=1e6
        stride = args.cpu_stride or max(1, N // 25_000)
        ncpu = os.cpu_count() or 1
        threads = [int(t) for t in args.cpu_threads.split(",") if t] or sorted({min(8, ncpu), ncpu})
        cells = baseline_matrix(pool, pool[:m], n, StationaryOracle(WORKLOAD["family"], WORKLOAD["lengthscale"],
                                                                    WORKLOAD["outputscale"]), stride, seed=1,
                                thread_counts=threads)
        f64 = [c for c in cells if c["dtype"] == "float64"]
        f32 = [c for c in cells if c["dtype"] == "float32"]
        best = min(f64, key=lambda c: c["seconds_per_batch"])        # parity-mode arithmetic (same as the GPU path)
        best32 = min(f32, key=lambda c: c["seconds_per_batch"])
        cpu = dict(
            value=1.0 / best["seconds_per_batch"], unit="batches/s", cores=best["threads"], kind="port",
            dtype="f64", host_cores=ncpu,
            value_f32=1.0 / best32["seconds_per_batch"], cores_f32=best32["threads"],
            matrix=[dict(dtype=c["dtype"], threads=c["threads"], seconds_per_batch=round(c["seconds_per_batch"], 2),
                         measured_seconds=round(c["measured_seconds"], 2), abbreviated=bool(c.get("abbreviated", False)))
                    for c in cells],
            sample=(f"oracle (reference op sequence, torch CPU) on pool seed 0: Gram+svd_lowrank, all "
                    f"{best['n_rounds']} rounds' projection/SVD/elimination in full; hot loop every {stride}th block "
                    f"({best['kernel_calls_run']}/{best['kernel_calls_total']} kernel calls), loop time scaled; value = "
                    f"fastest float64 cell ({best['threads']} threads: {best['measured_seconds']:.1f}s measured -> "
                    f"{best['seconds_per_batch']:.1f}s/batch); un-sampled anchor: profiles/r02_cpu_full_batch.txt"),
        )

    if rank == 0:
        value = args.steps / dt
        out = {
            "metric": "recombination batches/sec (N candidates -> n points) at N=1e6 d=10",
            "value": value,
            "unit": "batches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "median_ms_per_seed": sorted(per_seed_ms)[len(per_seed_ms) // 2],
            "ms_per_seed": [round(v, 3) for v in per_seed_ms],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"RBF kernel l=2, N={N} candidates, d={d}, n={n} recombination, m={m} Nystrom points, "
                                   f"float64, pool sharded over {world} GPU(s); pool seeds {list(POOL_SEEDS)} cycled over steps",
                       "N": N, "d": d, "n": n, "m": m, "kernel": "rbf", "parallelism": f"pool-sharded x{world}"},
            "roofline": {
                # contract vocabulary is hbm|mfma: this kernel is bound by the fp64 PIPE, which f64 MFMA and fp64 VALU share
                # on gfx950 (roughly half of its busy cycles are MFMA, half the VALU exp epilogue) -- neither HBM nor a
                # GEMM-shaped MFMA bound; "bound_detail" says so.
                "bound": "mfma", "bound_detail": "fp64 pipe: on gfx950 the f64 matrix instructions (either form) and the fp64 "
                                                   "VALU share ONE pipe of 16 lanes x 1 op per cycle per SIMD "
                                                   "(profiles/r02_l_microbench_mfma_f64_4x4x4.txt: v_mfma_f64_4x4x4 runs at "
                                                   "that rate, 75.7 TF/s; 16x16x4 at 47.4; interleaved with v_fma_f64 the times "
                                                   "add).  Per pair: KP lane-FMAs for the exponent argument on the matrix "
                                                   "instruction + ~13 VALU lane-ops for the table exponential; not HBM",
                "kernel": "blocksum_kernel (fused pairwise-kernel block sums, BASQ/_rchq.py:79-99; MFMA distances + VALU exp)",
                "achieved": achieved_tf, "peak": PEAK_FP64_VECTOR_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / PEAK_FP64_VECTOR_TFLOPS,
                "peak_measured": MEASURED_FP64_TFLOPS["mixed_fma_mfma"],
                "frac_of_measured_peak": achieved_tf / MEASURED_FP64_TFLOPS["mixed_fma_mfma"],
                "peak_measured_source": "profiles/r02_microbench_fp64_rates.txt (fma-only 72.9, mfma-only 47.2, both "
                                        "interleaved 63.1 TF/s: the figure that applies to this kernel's instruction mix)",
                "traffic": traffic,
                "traffic_source": ("committed PMC pass, not measured in this run: " + traffic_src) if traffic_src else None,
                "fp64_pipe_busy_pmc": pipe_busy,
                "fp64_pipe_busy_source": "committed PMC pass (same file), not measured in this run" if pipe_busy else None,
                "launches_per_batch": k_launches, "kernel_ms_per_batch": k_ms, "pairs_per_batch": k_pairs,
                "reference_pairs_per_batch": ref_pairs,
                "whole_batch_TFLOPs_by_reference_count": ref_pairs * (3 * d + 3) / (dt / args.steps) / 1e12,
                "flops_per_pair": 3 * d + 3,
                # executed lane operations per pair (KP = 4 ceil((d + 2) / 4) on the matrix instruction + 13.06 VALU, counted
                # in the kernel's ISA) and the share of the kernel time the fp64 pipe needs for them at the nominal 2.4 GHz
                "fp64_pipe_lane_ops_per_pair": 4 * ((d + 2 + 3) // 4) + 13.06,
                "fp64_pipe_time_frac_nominal_clock": ((4 * ((d + 2 + 3) // 4) + 13.06) * k_pairs / (1024 * 16 * 2.4e9))
                / (k_ms * 1e-3) if k_ms > 0 else None,
                "hbm_algorithmic_GBs": bytes_alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0,
                "hbm_peak_GBs": PEAK_HBM_GBS,
                "note": "compute-bound on the fp64 pipe (SURVEY 8d): 33 flop per pair by SURVEY's count (3d + 3); the kernel "
                        "executes ~25 fp64 lane operations per pair on the one fp64 pipe; launch durations by HIP events on "
                        "the launch stream over one traced batch, rank 0; traffic / pipe occupancy: committed PMC passes",
            },
            "cpu_baseline": cpu,
            "result_digest": {"n_selected": int(idx.numel()), "w_sum": float(w.sum().item())},
        }
        line = json.dumps(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner to the C-level stdout buffer; push it out first so that the JSON line is the LAST
        # line of rank 0's output
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
