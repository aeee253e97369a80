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
* N > 1     = the SAME batch with the pool sharded over the ranks (strong scaling; one small all-gather per round,
              stream-ordered: no host wait per round on any rank count);
* value     = steps one after the other (each call returns before the next starts: the reference's synchronous call);
              ``value_concurrent2`` (``concurrent`` object) = max(K, 12) of the same steps with TWO independent batches in flight
              (``basq_amd.recombination_many``: the reference's own pair, selection + quadrature, ``BASQ/_basq.py:82-88,
              104-106``), results bit-identical to the sequential runs; ``value_concurrent3`` / ``value_concurrent4`` on one GPU, ``value_concurrent4`` /
              ``value_concurrent8`` on multi-GPU lines (owner-rank reductions: batch k's chain on rank k mod N), with an ``rccl``
              object naming the process group the line was measured on;
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

# (before the first HIP call: hardware queues for the batches in flight of `value_concurrent*` -- see basq_amd/__init__.py; the
#  sequential `value` is unaffected)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch                                                     # noqa: E402

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
CONFIG_STEPS = 3                    # timed batches per entry of `configs` (after two warm-up batches)
WATCHDOG_EXIT_CODE = 3              # the runs with batches in flight did not finish on several ranks (the line is printed first)
TOTAL_LIMIT_EXIT_CODE = 5           # several ranks: the whole run exceeded BASQ_BENCH_TOTAL_LIMIT_S (no line)

WORKLOAD = dict(N=1_000_000, d=10, n=100, nys_ratio=1e-2, family="rbf", lengthscale=2.0, outputscale=1.0, pool_seed=0)


def roofline_self_check(k_ms, chain_ms, class_ms, per_seed_ms, steady_ms_per_launch, clock_in_situ_mhz):
    """The checks a roofline line must pass before it is printed (review r04: 17 ms of block sums inside an 18.6-ms batch went out
    unnoticed) -> dict with the figures, one boolean per check and ``ok``.

    (1) ``fits_in_batch``: everything the traced batch timed runs on ONE stream, one after the other -- block sums + the chains of
        null space + elimination must fit into a synchronised batch (the median over the pool seeds, 2 % slack);
    (2) ``class_launch_plausible``: the round-1 class launch cannot take more than 1.25 x its back-to-back time scaled by the clock
        it was given (2.03 GHz when no sample exists: the in-situ clock of every run so far; the ramp after a chain costs ~18 %)."""
    out = {}
    med = sorted(per_seed_ms)[len(per_seed_ms) // 2] if per_seed_ms else None
    if med is not None and k_ms > 0:
        out["blocksum_plus_chain_ms"] = round(k_ms + chain_ms, 3)
        out["median_ms_per_seed"] = round(med, 3)
        out["fits_in_batch"] = bool(k_ms + chain_ms <= 1.02 * med)
    if steady_ms_per_launch is not None and class_ms:
        clock = clock_in_situ_mhz or 2030.0
        bound = 1.25 * steady_ms_per_launch * (2400.0 / clock)
        out["class_launch_ms"] = round(max(class_ms), 3)
        out["class_launch_bound_ms"] = round(bound, 3)
        out["class_launch_plausible"] = bool(max(class_ms) <= bound)
    out["ok"] = all(v for v in out.values() if isinstance(v, bool))
    return out


def newest_counters(profiles_dir, tree_hash=None):
    """-> (record, file name, stale) of the newest committed counter summary ``profiles/r*_pmc.json`` (names sort by round and
    letter); ``stale`` = the record's ``kernel_source_sha256`` is absent or differs from the tree's (``tree_hash``: for tests)."""
    import glob

    files = sorted(glob.glob(os.path.join(profiles_dir, "r[0-9][0-9]_*pmc.json")) + glob.glob(os.path.join(profiles_dir, "r[0-9][0-9]_pmc.json")),
                   key=os.path.basename)
    if not files:
        return None, None, None
    with open(files[-1]) as f:
        rec = json.load(f)
    if tree_hash is None:
        from basq_amd._build import source_hash

        tree_hash = source_hash()
    return rec, os.path.basename(files[-1]), rec.get("kernel_source_sha256") != tree_hash


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--N", type=int, default=WORKLOAD["N"])
    ap.add_argument("--d", type=int, default=WORKLOAD["d"])
    ap.add_argument("--n", type=int, default=WORKLOAD["n"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-stride", type=int, default=0, help="run every k-th hot-loop block on the CPU (0 = auto)")
    ap.add_argument("--cpu-threads", type=str, default="", help="comma list of CPU thread counts (default: 8 and all cores)")
    ap.add_argument("--breakdown", action="store_true", help="print a per-phase host timer breakdown to stderr")
    ap.add_argument("--no-roofline-batch", action="store_true", help="skip the traced batch, the per-seed latencies and the "
                    "pipelined runs (counter passes: tools/gpu_jobs.sh pmc)")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the runs with several batches in flight")
    ap.add_argument("--no-configs", action="store_true", help="skip the other one-GPU BASELINE configurations (`configs`)")
    ap.add_argument("--no-h2d", action="store_true", help="skip `value_incl_h2d` (the pool handed over from host memory)")
    ap.add_argument("--plain", action="store_true", help="for kernel statistics under rocprofv3: the timed loop, the per-seed "
                    "batches and the traced batch only -- no batches in flight, no back-to-back repetition of the round-1 "
                    "launch, no clock sampler (every block-sum launch in the profile then belongs to a batch)")
    return ap.parse_args()


def self_launch(n_ranks: int) -> int:
    """``python bench.py --gpus N`` WITHOUT a launcher (no ``RANK`` in the environment): start the N ranks here, as children of a
    parent that never touches the GPU -- ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <same arguments>``, the command the driver would have used -- relay what they print, and make rank 0's
    JSON line the LAST line of this process's standard output.  -> the launcher's exit code (non-zero if any rank failed or a
    watchdog fired).  Nothing here may call into HIP: a parent that had initialised the GPU could not start children safely."""
    import socket
    import subprocess

    have = torch.cuda.device_count()                             # (counting devices does not initialise the runtime)
    if have < n_ranks:
        raise SystemExit(f"bench.py --gpus {n_ranks}: this node shows {have} GPU(s)")
    with socket.socket() as sk:                                  # a free rendezvous port on the loop-back interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] no launcher in the environment: starting {n_ranks} rank(s) through torch.distributed.run (port {port})",
          file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line_json = None
    for ln in proc.stdout:                                       # the ranks' stderr goes straight through
        if ln.startswith('{"metric"'):
            line_json = ln.rstrip("\n")                          # held back: printed LAST
        else:
            sys.stdout.write(ln)
            sys.stdout.flush()
    rc = proc.wait()
    if line_json is not None:
        print(line_json, flush=True)
    elif rc == 0:
        rc = 4                                                   # the ranks left without a line: never a success
    return rc


def main():
    args = parse()
    force_dist_env = os.environ.get("BASQ_BENCH_FORCE_DIST") == "1"
    if "RANK" not in os.environ and (args.gpus > 1 or force_dist_env):
        # no launcher: be the launcher (before anything touches the GPU)
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                                        # an external launcher decides the rank count
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists in basq_amd)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # BASQ_BENCH_FORCE_DIST=1: take the multi-rank code path (launcher, RCCL group, sharded entry, collectives) even with
    # one rank -- the only way to exercise it on a 1-GPU box (tests/test_bench_dist_gpu.py); never set by the driver.
    force_dist = force_dist_env and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        t_init = time.perf_counter()
        dist.init_process_group("nccl", device_id=dev)
        if rank == 0:
            print(f"[bench] RCCL group of {world} up in {time.perf_counter() - t_init:.1f} s", file=sys.stderr)

    if dist is not None and world > 1:
        # several ranks: a bound on the WHOLE run -- a rank that is stuck in a collective (first contact of the sharded path with a real
        # multi-GPU node) must not hold the node until the driver's own limit; no line is printed (there is nothing measured to report),
        # the exit code is a distinct non-zero one, and every rank runs the same timer
        import threading

        def total_limit():
            print(f"[bench] rank {rank}: the run did not finish within BASQ_BENCH_TOTAL_LIMIT_S; leaving with code {TOTAL_LIMIT_EXIT_CODE}",
                  file=sys.stderr, flush=True)
            os._exit(TOTAL_LIMIT_EXIT_CODE)

        whole = threading.Timer(float(os.environ.get("BASQ_BENCH_TOTAL_LIMIT_S", "1500")), total_limit)
        whole.daemon = True
        whole.start()

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
    pools_pinned = []                                                # (one rank only: the pools in page-locked host memory, for
    pool = None                                                      #  `value_incl_h2d`)
    want_h2d = world == 1 and not (args.no_roofline_batch or args.plain or args.no_h2d)
    for sd in POOL_SEEDS:
        p = gmm_pool(N, d, sd)
        if sd == POOL_SEEDS[0]:
            pool = p                                                 # host copy of seed 0: the CPU baseline's input
        pools_dev.append((p[:m].to(dev), p[off:off + Rl].to(dev)))   # PriorSampler: pts_nys = prefix of the pool
        if want_h2d:
            pools_pinned.append(p.pin_memory())
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

    # The ONE full garbage collection every Python process owes after its imports and set-up, paid here instead of inside the
    # timed loop: the imports leave the collector's older generations primed, the first young-generation pass of the loop
    # cascades into a full collection, and that walks every object torch has created -- 35-44 ms, the "48.6-ms step among twenty
    # of 18.6 ms" of round 4's driver run (tools/stall_probe.py, profiles/r07_a_stall_probe_*.txt).  Later passes are
    # young-generation only (0.05-0.2 ms every 5-20 batches) and stay inside the timed region.  It is paid after the FIRST
    # warm-up batch (which creates what the set-up still owes: streams, pinned buffers, the kernels' first launches), so that the
    # remaining warm-up batches run between those 40 ms of an idle GPU and the timed loop: the first timed step no longer opens
    # on a shader clock that has dropped (it was ~1 ms slower than the rest).
    import gc

    for k in range(args.warmup):
        one_batch(k=k)
        if k == 0:
            gc.collect()
    if args.warmup == 0:
        gc.collect()
    barrier()
    t0 = time.perf_counter()
    step_marks = []
    for k in range(args.steps):
        idx, w = one_batch(k=k)
        step_marks.append(time.perf_counter())                  # (host clock when call k returned: no extra synchronisation)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- per-seed latencies (outside the timed region): one synchronised batch per pool seed, median reported ----
    per_seed_ms = []
    parity = None
    golden_names = ["cfg3_rbf_1e6"] + [f"cfg3_rbf_1e6_pool{sd}" for sd in POOL_SEEDS[1:]]
    check_golden = (N, d, n) == (WORKLOAD["N"], WORKLOAD["d"], WORKLOAD["n"])
    for k in range(0 if args.no_roofline_batch else len(POOL_SEEDS)):
        barrier()
        t1 = time.perf_counter()
        i_k, w_k = one_batch(k=k)
        barrier()
        per_seed_ms.append(1e3 * (time.perf_counter() - t1))
        # every pool the timed loop cycles through has a golden produced by running the reference itself (tests/golden/,
        # fixtures only: nothing of the reference is read here): the batches `value` is measured on ARE the reference's
        gpath = os.path.join(ROOT, "tests", "golden", golden_names[k] + ".json")
        if check_golden and os.path.exists(gpath):
            with open(gpath) as f:
                fx = json.load(f)
            same = i_k.cpu().tolist() == fx["idx"]
            gw = torch.tensor(fx["w"], dtype=torch.float64)
            rel = float(((w_k.cpu() - gw).abs() / gw).max().item()) if same else float("nan")
            parity = parity or dict(pools_checked=0, indices_identical=True, max_rel_weight_error=0.0,
                                    source="tests/golden/cfg3_rbf_1e6*.json (reference-generated, oracle/make_golden.py)")
            parity["pools_checked"] += 1
            parity["indices_identical"] = bool(parity["indices_identical"] and same)
            parity["max_rel_weight_error"] = max(parity["max_rel_weight_error"], rel) if same else float("nan")

    # ---- kernel-level roofline: traced batches (HIP events on the launch stream, no host syncs: the batch stays on the code
    #      path the timed steps take -- descriptor-driven rounds -- and reports its launches after the fact).  The roofline
    #      numbers come from a batch WITHOUT the clock sampler; the shader clock is sampled in a batch of its own, and its
    #      samples are only kept if the sampled launches took what the unsampled ones took (round 4's driver run: the sampler's
    #      stream landed on the hardware queue of the launch stream, the 8-ms sampler ran IN FRONT of the block sums instead of
    #      beside them, and its time was booked as the kernel's: frac 0.255 instead of 0.488) ----
    from basq_amd._ops import HipOps

    def traced_batch(sampler=None):
        t = basq_amd.EngineTrace(time_kernels=not args.no_roofline_batch, sample_clock=sampler, host_sync=False)
        one_batch(t)
        torch.cuda.synchronize()
        return t

    def launch_figures(t):
        """-> block-sum ms, pairs, launches, ms of the chains (null space + elimination), ms of each class launch."""
        ms = sum(a.elapsed_time(b) for a, b, _ in t.kernel_events)
        pairs = sum(info["pairs"] for _, _, info in t.kernel_events)
        chain = sum(a.elapsed_time(b) for a, b in t.chain_events)
        cls_ms = [a.elapsed_time(b) for a, b, info in t.kernel_events if info.get("chunks", 0) >= 2 and info.get("class_mod", 0) > 0]
        return ms, pairs, len(t.kernel_events), chain, cls_ms

    tr = traced_batch()
    k_ms, k_pairs, k_launches, chain_ms, class_ms = launch_figures(tr)

    # the shader clock the class launches ran at: one wave on a second stream samples it every 250 us while the launch runs.
    # A full fp64 load that follows the previous batch's chain of single-work-group reductions opens at ~2.05 GHz and gains
    # only ~20 MHz per ms (tools/clock_probe.hip, profiles/r04_l_shader_clock_after_idle.txt)
    clock_ops = HipOps(dev, stream=torch.cuda.Stream(device=dev))
    clk, clk_w, clock_note = [], 0.0, None
    if not (args.plain or args.no_roofline_batch or world > 1 or force_dist):
        tr_clk = traced_batch(clock_ops)
        sampled_ms = launch_figures(tr_clk)[4]
        # the sampler must have run BESIDE the launches: each sampled class launch within 15 % of its unsampled twin
        beside = len(sampled_ms) == len(class_ms) and all(b <= 1.15 * a + 0.05 for a, b in zip(class_ms, sampled_ms))
        if beside:
            for a, b, info in tr_clk.kernel_events:
                if "clock_mhz" in info:
                    w_ms = a.elapsed_time(b)
                    series = info.pop("clock_mhz").tolist()[: max(1, min(32, int(w_ms / 0.25)))]  # the samples inside the launch
                    mean = sum(series) / len(series)
                    clk.append(dict(classes=info["chunks"], ms=round(w_ms, 3), mhz_mean=round(mean), mhz_first=round(series[0]),
                                    mhz_last=round(series[-1])))
                    clk_w += mean * w_ms
        else:
            clock_note = ("clock samples DISCARDED: the sampler's stream was serialised with the launch stream on this box "
                          f"(class launches {[round(v, 3) for v in class_ms]} ms unsampled, {[round(v, 3) for v in sampled_ms]} ms "
                          "with the sampler); the roofline figures are from the unsampled batch and unaffected")
    clk_ms = sum(c["ms"] for c in clk)
    clock_in_situ = clk_w / clk_ms if clk_ms > 0 else None              # MHz, time-weighted over the class launches
    # what the REFERENCE's loop evaluates for the same batch (m * sum_r R_r + m^2): the residue-class block sums
    # evaluate about half of it (rounds inside an epoch regroup the previous sums instead)
    ref_pairs = float(m) * sum(r["R"] for r in tr.rounds) + float(m) * m
    bytes_alg = sum(info["R"] * (8 * d + 16) + 8 * info["m"] * d for _, _, info in tr.kernel_events)

    # ---- the same kernel in STEADY STATE: the round-1 class launch repeated back to back.  Inside a batch that launch opens on
    #      a chip that clocked down during the previous batch's chain of single-work-group reductions (tools/idle_probe.py:
    #      6.35 ms back to back, 7.5-7.8 ms after a chain or >= 10 ms of idle time); `achieved` above is the in-situ figure.
    steady = None
    if not (args.no_roofline_batch or args.plain) and world == 1 and not force_dist:
        from basq_amd._partition import RoundGeometry

        ops = HipOps(dev)
        pts_nys, pts_local = pools_dev[0]
        spec = kern.spec(d)
        cen = ops.col_mean(pts_nys)
        pa, pb = ops.pack(spec, pts_nys, cen, 0, pad_rows_to=64), ops.pack(spec, pts_local, cen, 1)
        mu0, _ = ops.init_state(N, 0, N)
        S2 = 2 * n
        geo0 = RoundGeometry.of(N, S2)
        C0 = 16
        Rr = (geo0.nb // C0) * C0 * S2
        if Rr > 0:
            run = lambda: ops.blocksum(spec, pa, m, pb, mu0, None, Rr, 0, geo0.n_full, S2, C0, class_mod=C0)   # noqa: E731
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 4
            # one more launch with the clock sampled beside it (kept only if the sampler did run BESIDE the launch)
            ev = ops.record_event()
            clock_ops.wait_event(ev)
            clk_s = clock_ops.shader_clock_mhz(max(1, int(ms / 0.25)), 250)
            run()
            e2 = ops.record_event()
            torch.cuda.synchronize()
            clk_steady = float(clk_s.mean().item()) if ev.elapsed_time(e2) <= 1.15 * ms + 0.05 else None
            tf = float(Rr) * m * (3 * d + 3) / (ms * 1e-3) / 1e12
            steady = dict(achieved=tf, frac=tf / PEAK_FP64_VECTOR_TFLOPS, ms_per_launch=ms, pairs_per_launch=float(Rr) * m,
                          shader_clock_MHz=clk_steady,
                          note="the 16-class round-1 launch repeated back to back (chip at its sustained clock); inside a batch "
                               "the same launch starts on a chip that clocked down during the preceding chain of "
                               "single-work-group reductions: profiles/r04_h_block_sums_after_idle_or_chain.txt, "
                               "profiles/r04_l_shader_clock_after_idle.txt")
        del pa, pb, mu0

    # ---- self-check of the roofline figures before they are printed (review r04: 17 ms of block sums inside an 18.6-ms batch
    #      went out unnoticed).  (1) everything the traced batch timed runs on ONE stream, one after the other: block sums + the
    #      chains of null space + elimination must fit into a synchronised batch; (2) the round-1 class launch cannot take more
    #      than 1.25 x its back-to-back time scaled by the clock it was given (2.4 GHz nominal when no sample exists: the ramp
    #      after a chain costs ~18 %).  A violated check retakes the traced batch once; a second violation marks the line.
    def roofline_checks(k_ms_, chain_ms_, class_ms_):
        return roofline_self_check(k_ms_, chain_ms_, class_ms_, per_seed_ms, steady["ms_per_launch"] if steady else None,
                                   clock_in_situ)

    check = roofline_checks(k_ms, chain_ms, class_ms) if not args.no_roofline_batch else {"ok": True}
    roofline_suspect = False
    retake = not check["ok"]
    if dist is not None:
        # a traced batch is a collective operation (every rank takes part in its exchanges): the decision to retake it is
        # rank 0's, broadcast -- ranks deciding on their own timings could disagree and leave each other in a collective
        flag = torch.tensor([1 if retake else 0], dtype=torch.int32, device=dev)
        dist.broadcast(flag, src=0)
        retake = bool(int(flag.item()))
    if retake:
        first = dict(check, kernel_ms_per_batch=k_ms, chain_ms_per_batch=chain_ms)
        tr = traced_batch()
        k_ms, k_pairs, k_launches, chain_ms, class_ms = launch_figures(tr)
        check = roofline_checks(k_ms, chain_ms, class_ms)
        check["first_attempt"] = first
        roofline_suspect = not check["ok"]
    flops = k_pairs * (3 * d + 3)
    achieved_tf = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0

    # Hardware counters of the block-sum launches of one batch: rocprofv3 --pmc passes of THIS bench command (one counter
    # group per pass, tools/gpu_jobs.sh pmc -> tools/pmc_summary.py), committed with the commit they were taken on.
    # FETCH_SIZE / WRITE_SIZE in KiB; the kernel's loads are 8 B per lane (not the 16-B-per-lane streams whose gfx950
    # half-count the micro-architecture guide describes): calibrated in the file against the launch's compulsory bytes.
    # Counters cannot be collected from inside the run, so they are REPLAYED from the file -- and bound to the binary: every file
    # carries the hash of the kernel sources + compiler flags it was taken on (`kernel_source_sha256`, basq_amd._build.source_hash);
    # if the tree's hash differs, `counters_stale` is true and traffic / mfma_util / fp64_pipe_busy_pmc are null.
    pmc_rec, pmc_name, counters_stale = None, None, None
    if N == WORKLOAD["N"] and d == WORKLOAD["d"] and world == 1:
        pmc_rec, pmc_name, counters_stale = newest_counters(os.path.join(ROOT, "profiles"))
    pmc_live = None if counters_stale else pmc_rec

    # ---- the same steps with the pool handed over from HOST memory (outside `value`): the reference's boundary starts from host
    #      tensors (``BASQ/_sampler.py:31-34``: ``prior.sample`` on the CPU); page-locked here, one copy per batch on the launch stream
    h2d = None
    if pools_pinned:
        def one_batch_h2d(k):
            pts = pools_pinned[k % len(pools_pinned)].to(dev, non_blocking=True)
            torch.manual_seed(1)
            return basq_amd.recombination(pts, pts[:m], n, kern, dev)

        for k in range(2):
            one_batch_h2d(k)
        barrier()
        t1 = time.perf_counter()
        for k in range(args.steps):
            i_h, w_h = one_batch_h2d(k)
        barrier()
        dth = time.perf_counter() - t1
        i_d, w_d = one_batch(k=args.steps - 1)
        h2d = dict(value=args.steps / dth, unit="batches/s", steps=args.steps, ms_per_step=1e3 * dth / args.steps,
                   bytes_per_batch=int(N * d * 8), bit_identical_to_resident=bool(torch.equal(i_h, i_d) and torch.equal(w_h, w_d)),
                   note="pool [N, d] float64 copied from page-locked host memory inside every step; never part of `value`")
        del pools_pinned[:]

    # ---- the other BASELINE configurations that fit one GPU (configs[1], [3], [4] and tutorial 03's WSABI-M at config 5's size),
    #      outside `value`: a few batches each, checked against the reference-generated golden of the same case ----
    configs_out = None
    if world == 1 and not force_dist and not (args.no_configs or args.no_roofline_batch or args.plain):
        from basq_amd.pools import kernel_for_case

        configs_out = []
        for name, label in (("cfg2_rbf_1e5", "configs[1]"), ("cfg4_matern52_1e6_d32", "configs[3] on ONE GPU"),
                            ("cfg5_wsabil_5e5", "configs[4]"), ("cfg5m_wsabim_5e5", "configs[4]'s size with WSABI-M")):
            gpath = os.path.join(ROOT, "tests", "golden", name + ".json")
            if not os.path.exists(gpath):
                continue
            with open(gpath) as f:
                fx = json.load(f)
            c = fx["case"]
            pts_c = gmm_pool(c["N"], c["d"], c["pool_seed"]).to(dev)
            nys_c = pts_c[: c["m"]]
            kern_c = kernel_for_case(c)

            def run_c():
                torch.manual_seed(c["torch_seed"])
                return basq_amd.recombination(pts_c, nys_c, c["n"], kern_c, dev)

            run_c()
            run_c()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(CONFIG_STEPS):
                i_c, w_c = run_c()
            torch.cuda.synchronize()
            ms_c = 1e3 * (time.perf_counter() - t1) / CONFIG_STEPS
            same = i_c.cpu().tolist() == fx["idx"]
            gw = torch.tensor(fx["w"], dtype=torch.float64)
            configs_out.append(dict(name=name, baseline_config=label, N=c["N"], d=c["d"], n=c["n"], m=c["m"],
                                    kernel=c["kernel"]["family"] + ("" if c["kernel"]["posterior"] is None else "+posterior")
                                    + ("" if c["kernel"]["warp"] == "none" else "+" + c["kernel"]["warp"]),
                                    steps=CONFIG_STEPS, ms_per_batch=round(ms_c, 3), batches_per_s=round(1e3 / ms_c, 2),
                                    indices_identical=bool(same),
                                    max_rel_weight_error=float(((w_c.cpu() - gw).abs() / gw).max().item()) if same else None))
            del pts_c, nys_c, kern_c
        torch.cuda.empty_cache()

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

        # SURVEY §8d matrix: {float64, float32} x {8 threads, all host cores}; ~3.5 s of CPU work per cell at N=1e6 (~15 s in all)
        stride = args.cpu_stride or max(1, N // 70_000)         # every 14th block at N = 1e6: ~3.5 s of CPU work per cell
        ncpu = os.cpu_count() or 1
        # 8 threads (comparable with SURVEY's probe) and 32 (the largest team that still helps: beyond it the 200-column
        # blocks of the hot loop thrash -- profiles/r02_cpu_full_batch.txt); never more than the host has
        threads = [int(t) for t in args.cpu_threads.split(",") if t] or sorted({min(8, ncpu), min(32, ncpu)})
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

    rccl_info = None
    if dist is not None:
        import basq_amd._config as bcfg

        # proof that the N ranks sat on N DISTINCT GPUs: every rank's device identity, gathered
        prop = torch.cuda.get_device_properties(dev)
        ident = {"rank": rank, "local_rank": local_rank, "name": prop.name,
                 "uuid": str(getattr(prop, "uuid", "")) or None,
                 "pci": "%04x:%02x:%02x" % (getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_bus_id", 0),
                                            getattr(prop, "pci_device_id", 0)),
                 "hip_visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}
        idents = [None] * dist.get_world_size()
        dist.all_gather_object(idents, ident)
        rccl_info = {"world": dist.get_world_size(), "backend": dist.get_backend(), "devices": idents,
                     "distinct_devices": len({(i["uuid"], i["pci"]) for i in idents}),
                     "sequential_batch": "reduction replicated on every rank (one all-gather of the (q+1) x 2n message per round)",
                     "batches_in_flight": ("owner-rank reductions: batch k's null space + elimination on rank k mod world, outcome "
                                           "broadcast (3*2n+1 doubles) on the batch's own process group"
                                           if bcfg.OWNER_RANK_REDUCTION else "reduction replicated on every rank"),
                     "process_groups_for_batches_in_flight": "one per batch in flight (TorchDistComm.for_slot)"}

    def build_line(concurrent, note=None):
        value = args.steps / dt
        out = {
            "metric": "recombination batches/sec (N candidates -> n points) at N=1e6 d=10",
            "value": value,
            "unit": "batches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "ms_each_step": [round(1e3 * (b - a), 3) for a, b in zip([t0] + step_marks[:-1], step_marks)],
            "median_ms_per_seed": sorted(per_seed_ms)[len(per_seed_ms) // 2] if per_seed_ms else None,
            "ms_per_seed": [round(v, 3) for v in per_seed_ms],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"RBF kernel l=2, N={N} candidates, d={d}, n={n} recombination, m={m} Nystrom points, "
                                   f"float64, pool sharded over {world} GPU(s); synthetic 8-component mixture pool with unit-variance "
                                   f"Irwin-Hall(12) components (sum of 12 uniforms - 6: support +-6, bit-portable; basq_amd/pools.py), "
                                   f"pool seeds {list(POOL_SEEDS)} cycled over steps",
                       "N": N, "d": d, "n": n, "m": m, "kernel": "rbf", "parallelism": f"pool-sharded x{world}"},
            "value_concurrent2": concurrent[2]["value"] if 2 in concurrent else None,
            "value_concurrent3": concurrent[3]["value"] if 3 in concurrent else None,
            "value_concurrent4": concurrent[4]["value"] if 4 in concurrent else None,
            "value_concurrent8": concurrent[8]["value"] if 8 in concurrent else None,
            "concurrent": [concurrent[k] for k in sorted(concurrent)] or None,
            # proof of the N-rank run for the driver's SCALE record: the process group this line was measured on
            "rccl": rccl_info,
            "roofline": {
                # The contract's vocabulary is hbm | mfma.  On gfx950 the f64 matrix instructions and the fp64 VALU share ONE
                # pipe per SIMD (16 lanes x 1 op per cycle; profiles/r02_l_microbench_mfma_f64_4x4x4.txt), and this kernel
                # keeps it busy with both -- KP lane-FMAs per pair for the exponent argument on the matrix instruction, ~13
                # VALU lane-ops for the table exponential -- so "mfma" stands for that pipe; HBM is three orders below.
                "bound": "mfma",
                "kernel": "blocksum_kernel (fused pairwise-kernel block sums, BASQ/_rchq.py:79-99; MFMA distances + VALU exp)",
                "achieved": achieved_tf, "peak": PEAK_FP64_VECTOR_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / PEAK_FP64_VECTOR_TFLOPS,
                "peak_measured": MEASURED_FP64_TFLOPS["mixed_fma_mfma"],
                "frac_of_measured_peak": achieved_tf / MEASURED_FP64_TFLOPS["mixed_fma_mfma"],
                "peak_measured_source": "profiles/r02_microbench_fp64_rates.txt (fma-only 72.9, mfma-only 47.2, both "
                                        "interleaved 63.1 TF/s: the figure that applies to this kernel's instruction mix)",
                # counters (rocprofv3 --pmc passes over this command, per BATCH = all block-sum launches of one batch)
                "traffic": pmc_live["hbm_bytes_per_batch"] if pmc_live else None,
                "hbm_GBs_counter": (pmc_live["hbm_bytes_per_batch"] / (k_ms * 1e-3) / 1e9) if (pmc_live and k_ms > 0) else None,
                "hbm_frac": (pmc_live["hbm_bytes_per_batch"] / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if (pmc_live and k_ms > 0) else None,
                "mfma_util": pmc_live.get("mfma_util") if pmc_live else None,
                "fp64_pipe_busy_pmc": pmc_live.get("fp64_pipe_busy") if pmc_live else None,
                "stall_counters": pmc_live.get("stalls") if pmc_live else None,
                "counters_stale": counters_stale,
                "counters_source": (f"profiles/{pmc_name} (commit {pmc_rec.get('commit')}, kernel sources "
                                    f"{str(pmc_rec.get('kernel_source_sha256'))[:12]}, {pmc_rec.get('source')}); bytes per "
                                    "batch from the counters, time from this run's HIP events"
                                    + ("; STALE: taken on other kernel sources than this tree's -- figures withheld" if counters_stale else ""))
                if pmc_rec else None,
                "steady_state": steady,
                # the clock the class launches (most of kernel_ms_per_batch) actually had, and the fraction against the fp64
                # peak AT that clock: what the kernel leaves on the table, as opposed to what the power manager withholds
                "shader_clock_MHz_in_situ": clock_in_situ, "shader_clock_samples": clk or None, "shader_clock_note": clock_note,
                "frac_at_in_situ_clock": (achieved_tf / (PEAK_FP64_VECTOR_TFLOPS * clock_in_situ / 2400.0)) if clock_in_situ else None,
                "launches_per_batch": k_launches, "kernel_ms_per_batch": k_ms, "pairs_per_batch": k_pairs,
                # of which the kernel values of the message columns (basq_amd/_epochs.py), evaluated on the SIDE stream beside the first
                # chain of every epoch; their launches are timed by HIP events on THAT stream and are part of `launches_per_batch`,
                # `pairs_per_batch` and `kernel_ms_per_batch` (0.4 % of the pairs at the headline size, ~0.2 ms of short launches)
                "side_stream_pairs_per_batch": getattr(tr, "side_pairs", 0.0),
                # the chains of single-work-group kernels (finalize + null space + elimination per round) of the same traced batch
                "chain_ms_per_batch": chain_ms, "chain_rounds": len(tr.chain_events),
                "self_check": check, "roofline_suspect": roofline_suspect,
                "reference_pairs_per_batch": ref_pairs,
                "whole_batch_TFLOPs_by_reference_count": ref_pairs * (3 * d + 3) / (dt / args.steps) / 1e12,
                "flops_per_pair": 3 * d + 3,
                # executed lane operations per pair (KP = 4 ceil((d + 2) / 4) on the matrix instruction + 12.06 VALU, counted
                # in the kernel's ISA: 13.06 before the quadratic exponential; the counters see 13.4 VALU instructions per 64
                # pairs incl. prologues) and the share of the kernel time the fp64 pipe needs for them at the nominal 2.4 GHz
                "fp64_pipe_lane_ops_per_pair": 4 * ((d + 2 + 3) // 4) + 12.06,
                "fp64_pipe_time_frac_nominal_clock": ((4 * ((d + 2 + 3) // 4) + 12.06) * k_pairs / (1024 * 16 * 2.4e9))
                / (k_ms * 1e-3) if k_ms > 0 else None,
                "hbm_algorithmic_GBs": bytes_alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0,
                "hbm_peak_GBs": PEAK_HBM_GBS,
                "note": "33 flop per pair by SURVEY's count (3d + 3); launch durations by HIP events on the launch stream over "
                        "one traced batch that takes the timed code path (descriptor-driven rounds), rank 0, NO clock sampler in "
                        "that batch (the clock is sampled in a second traced batch and kept only if its launches took the same time)",
            },
            "cpu_baseline": cpu,
            "value_incl_h2d": h2d["value"] if h2d else None,
            "incl_h2d": h2d,
            "configs": configs_out,
            "result_digest": {"n_selected": int(idx.numel()), "w_sum": float(w.sum().item())},
            "parity_vs_golden": parity,
        }
        out["concurrent_timed_out"] = bool(note)                 # machine-readable twin of `concurrent_note`
        if note:
            out["concurrent_note"] = note
        return json.dumps(out)

    def emit(line):
        # RCCL writes a version banner to the C-level stdout buffer; push it out first so that the JSON line is the LAST
        # line of rank 0's output
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(line, flush=True)

    # ---- several batches in flight: LAST, and on several ranks under a watchdog.  `value` and everything above come from the
    #      sequential path (replicated reductions, one process group).  The runs with batches in flight use owner-rank reductions
    #      on one process group per batch in flight -- gloo-tested on 2-8 ranks, never run on multi-GPU hardware -- so a rank that
    #      does not get through them within the limit prints the line it already has (rank 0) and leaves; the driver always gets
    #      its JSON line.
    watchdog = None
    concurrent = {}
    if dist is not None and world > 1:
        import threading

        def give_up():
            if rank == 0:
                emit(build_line(dict(concurrent), note="the runs with batches in flight did not ALL finish within the watchdog's limit "
                                                       "on this machine (those that did are reported): value and roofline are from "
                                                       "the sequential path, completed before"))
            # The line is out (flushed) and carries `concurrent_timed_out`; the process leaves with a NON-ZERO code of its own
            # (3): a hang of the owner-rank path must not read as a successful run.  Every rank runs this timer, so no peer is
            # left waiting in a collective for a rank that has gone; nothing is restarted or re-executed.
            try:
                sys.stdout.flush()
                sys.stderr.flush()
            finally:
                os._exit(WATCHDOG_EXIT_CODE)

        watchdog = threading.Timer(float(os.environ.get("BASQ_BENCH_CONCURRENT_LIMIT_S", "240")), give_up)
        watchdog.daemon = True
        watchdog.start()
    # ---- several batches in flight (outside the timed region of `value`): the same K steps through recombination_many ----
    if not (args.no_concurrent or args.no_roofline_batch or args.plain):
        # several ranks: batch k's reductions live on rank k mod G (owner-rank mode, basq_amd/_config.py), so the chains only
        # spread over all G GPUs with at least G batches in flight
        for k_fl in ([2, 3, 4] if world == 1 and not force_dist else sorted({2, 4, max(4, min(world, 8))})):
            n_c = max(args.steps, 12, 3 * k_fl)                      # enough steps for the pipeline's fill and drain not to dominate
            calls, seeds = [], [1] * n_c
            for k in range(n_c):
                pts_nys, pts_local = pools_dev[k % len(pools_dev)]
                calls.append((pts_local, pts_nys, n, kern) if (world == 1 and not force_dist)
                             else (pts_local, off, N, pts_nys, n, kern))

            lat = []

            def many():
                del lat[:]
                if world == 1 and not force_dist:
                    return basq_amd.recombination_many(calls, dev, in_flight=k_fl, seeds=seeds, timings=lat)
                return basq_amd.recombination_many_sharded(calls, dev, in_flight=k_fl, seeds=seeds, timings=lat)

            many()                                                   # warm-up: the slots' streams, buffers, workspaces
            barrier()
            t1 = time.perf_counter()
            res_c = many()
            barrier()
            dtc = time.perf_counter() - t1
            if dist is not None:
                tt = torch.tensor([dtc], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtc = float(tt.item())
            # every pipelined result against the sequential run of the same step (same pool, same seed): bit for bit
            same = True
            for k in range(min(n_c, len(pools_dev))):
                i1, w1 = one_batch(k=k)
                same = same and torch.equal(i1, res_c[k][0]) and torch.equal(w1, res_c[k][1])
            lat_ms = sorted(1e3 * (b - a) for a, b in lat)           # first launch -> result, per batch (rank 0's host clock)
            concurrent[k_fl] = dict(in_flight=k_fl, value=n_c / dtc, unit="batches/s", steps=n_c,
                                    ms_per_step=1e3 * dtc / n_c, latency_ms_median=lat_ms[len(lat_ms) // 2],
                                    latency_ms_max=lat_ms[-1], bit_identical_to_sequential=bool(same))

    if watchdog is not None:
        watchdog.cancel()
    line = build_line(concurrent) if rank == 0 else None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(line)


if __name__ == "__main__":
    main()
