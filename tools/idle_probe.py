"""How long does the round-1 block-sum launch take when the chip was idle -- or busy on ONE compute unit -- just before?

    python tools/idle_probe.py

The 14-class launch of a headline batch back to back, after 2 / 10 / 50 / 200 ms without work, after the candidate packing,
and right after 14 x (null space + elimination), the chain that ends every batch.  Finding (profiles/r04_h_*): 6.35 ms back to
back, 7.5-7.8 ms after >= 10 ms of idle time OR after the 6-ms chain: the chip clocks down while one CU works and the block sums
that follow pay ~1.2 ms for the ramp.  Batches in flight keep it loaded (part of what recombination_many gains).
The last block looks at the shape of the effect with eight short launches (two classes, 1.2 ms each) after a chain: those
are NOT slower than back to back -- the penalty belongs to the long, chip-filling launch, not to the first millisecond after
the chain (no counter for the shader clock is readable from here; the mechanism is not pinned down).
"""
import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from basq_amd._ops import HipOps
from basq_amd._partition import RoundGeometry
from basq_amd.kernels import StationaryKernel
from basq_amd.pools import gmm_pool
ops = HipOps("cuda:0"); R, m, d, n = 1_000_000, 10_000, 10, 100; S = 2 * n
spec = StationaryKernel("rbf", 2.0).spec(d)
pts = ops.to_device(gmm_pool(R, d, 0)); nys = pts[:m].contiguous(); c = ops.col_mean(nys)
A = ops.pack(spec, nys, c, 0, pad_rows_to=64); B = ops.pack(spec, pts, c, 1); mu, _ = ops.init_state(R, 0, R)
geo = RoundGeometry.of(R, S); Rr = (geo.nb // 16) * 16 * S
def run(nch, cm=16):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.blocksum(spec, A, m, B, mu, None, Rr, 0, geo.n_full, S, nch, class_mod=cm); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
run(16); run(16)
print("back to back 16 classes:", [round(run(16), 3) for _ in range(4)])
print("back to back 14 classes:", [round(run(14), 3) for _ in range(4)])
for idle in (0.002, 0.01, 0.05, 0.2):
    ts = []
    for _ in range(4):
        time.sleep(idle); ts.append(round(run(14), 3))
    print(f"after {idle*1e3:.0f} ms idle, 14 classes:", ts)
# preceded by a pack (as in the engine)
ts = []
for _ in range(4):
    time.sleep(0.02); B2 = ops.pack(spec, pts, c, 1); ts.append(round(run(14), 3))
print("after 20 ms idle + pack, 14 classes:", ts)
# preceded by a CHAIN: 14 x (null space + elimination) = ~6 ms with ONE compute unit busy, as at the end of every batch
g = torch.Generator().manual_seed(0)
X = torch.randn(100, 200, generator=g, dtype=torch.float64); X[0] = 1.0
Xd = ops.to_device(X); mu0 = ops.to_device(torch.rand(200, generator=g, dtype=torch.float64) + 0.1)
def chain(k=14):
    for _ in range(k):
        P = ops.nullspace(Xd, 100, 200); ops.car_eliminate(P, mu0.clone(), 200, 100)
ts = []
for _ in range(4):
    torch.cuda.synchronize(); chain(); ts.append(round(run(14), 3))
print("right after a 6-ms chain (one CU busy), 14 classes:", ts)
ts = []
for _ in range(4):
    torch.cuda.synchronize(); run(16); chain(); ts.append(round(run(14), 3))
print("block sums, chain, block sums (as consecutive batches):", ts)
# a filler on a second stream during the chain: a throw-away block-sum launch (the chip stays loaded)
side = torch.cuda.Stream()
ts = []
for _ in range(4):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ops2 = HipOps("cuda:0", stream=side); ops2.blocksum(spec, A, m, B, mu, None, Rr, 0, geo.n_full, S, 12, class_mod=16)
    chain()
    torch.cuda.current_stream().wait_stream(side)
    ts.append(round(run(14), 3))
print("chain with a filler launch beside it, then 14 classes:", ts)
# the shape of the ramp: after a chain, eight launches of two classes each (~0.95 ms of work per launch at full speed)
def run_events(nch, k):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
    evs[0].record()
    for i in range(k):
        ops.blocksum(spec, A, m, B, mu, None, Rr, 0, geo.n_full, S, nch, class_mod=16)
        evs[i + 1].record()
    torch.cuda.synchronize()
    return [round(evs[i].elapsed_time(evs[i + 1]), 3) for i in range(k)]
torch.cuda.synchronize(); run_events(2, 8)
print("back to back, 8 x 2 classes:", run_events(2, 8))
for _ in range(3):
    torch.cuda.synchronize(); chain()
    print("after a chain,  8 x 2 classes:", run_events(2, 8))
