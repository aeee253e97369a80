"""Where does a slow step of bench.py's timed loop come from?  (review r04: one 48.6-ms step among twenty of 18.6 ms)

    python tools/stall_probe.py [--steps 60] [--warmup 5] [--gc default|freeze|off]

The timed loop of bench.py (same pools, same seeds, same call), with per-step host timers and, per step: Python garbage collections
(generation, duration: ``gc.callbacks``), the caching allocator's device allocations / frees (``torch.cuda.memory_stats`` deltas), new
pinned host buffers (``HipOps._pinned`` misses), and the time of the Gaussian draw.  Steps slower than 1.1 x the median are printed
with what happened inside them.
"""
import argparse
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basq_amd                                                    # noqa: E402
from basq_amd import _ops                                          # noqa: E402
from basq_amd.pools import gmm_pool                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gc", default="default", choices=["default", "collect", "freeze", "off"],
                    help="after the warm-up: nothing | one full collection (what bench.py does) | collect + freeze | collect + disable")
    ap.add_argument("--N", type=int, default=1_000_000)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    N, d, n = a.N, 10, 100
    m = N // 100
    kern = basq_amd.kernels.StationaryKernel("rbf", 2.0, 1.0)
    pools = []
    for sd in range(5):
        p = gmm_pool(N, d, sd)
        pools.append((p[:m].to(dev), p.to(dev)))
        del p

    events = []                                                  # (t, kind, detail) on the host clock
    t_gc = [0.0]

    def on_gc(phase, info):
        if phase == "start":
            t_gc[0] = time.perf_counter()
        else:
            events.append((t_gc[0], "gc", f"gen{info['generation']} {1e3 * (time.perf_counter() - t_gc[0]):.2f} ms collected {info['collected']}"))

    gc.callbacks.append(on_gc)
    orig_pinned = _ops.HipOps._pinned

    def pinned(self, shape, dtype, tag):
        cache = self.__dict__.setdefault("_pin_cache", {})
        miss = (tag, tuple(shape), dtype) not in cache
        t0 = time.perf_counter()
        buf = orig_pinned(self, shape, dtype, tag)
        if miss and time.perf_counter() - t0 > 1e-4:             # (torch's host allocator caches pinned blocks: a miss costs ~10 us)
            events.append((t0, "pinned", f"new {tag} {tuple(shape)} {1e3 * (time.perf_counter() - t0):.2f} ms"))
        return buf

    _ops.HipOps._pinned = pinned
    orig_rand = torch.rand

    def one(k):
        nys, pts = pools[k % 5]
        torch.manual_seed(1)
        return basq_amd.recombination(pts, nys, n, kern, dev)

    for k in range(a.warmup):
        one(k)
    torch.cuda.synchronize()
    if a.gc == "collect":
        gc.collect()
    elif a.gc == "freeze":
        gc.collect()
        gc.freeze()
    elif a.gc == "off":
        gc.collect()
        gc.disable()
    keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "allocation.all.allocated", "segment.all.allocated",
            "reserved_bytes.all.current")
    marks, stats = [], []
    t0 = time.perf_counter()
    s_prev = torch.cuda.memory_stats(dev)
    for k in range(a.steps):
        one(k)
        marks.append(time.perf_counter())
        s = torch.cuda.memory_stats(dev)
        stats.append({key: s.get(key, 0) - s_prev.get(key, 0) for key in keys})
        s_prev = s
    torch.cuda.synchronize()
    ms = [1e3 * (b - a_) for a_, b in zip([t0] + marks[:-1], marks)]
    med = sorted(ms)[len(ms) // 2]
    print(f"gc mode {a.gc}: {a.steps} steps, median {med:.2f} ms, mean {sum(ms) / len(ms):.2f} ms, max {max(ms):.2f} ms; "
          f"max/median {max(ms) / med:.3f}; gc counts {gc.get_count()} thresholds {gc.get_threshold()}; "
          f"reserved {torch.cuda.memory_reserved(dev) / 2**30:.2f} GiB")
    print("all steps (ms):", [round(v, 1) for v in ms])
    starts = [t0] + marks[:-1]
    for k, v in enumerate(ms):
        inside = [(kind, det) for (t, kind, det) in events if starts[k] <= t < marks[k]]
        nz = {key: val for key, val in stats[k].items() if val and key in ("num_device_alloc", "num_device_free", "num_alloc_retries", "segment.all.allocated",
                                                                           "reserved_bytes.all.current")}
        if v > 1.1 * med or inside or nz:
            print(f"  step {k:3d} (pool {k % 5}): {v:7.2f} ms {'SLOW' if v > 1.1 * med else '    '} | {inside} | allocator {nz}")


if __name__ == "__main__":
    main()
