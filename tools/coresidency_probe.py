"""Do one round's reductions (null space + elimination) START while another stream's round-1 block sums fill the chip?

    [BASQ_HIP_LIB=<variant>] python tools/coresidency_probe.py

Stream A: the 14-class block-sum launch of a headline batch (~6-7 ms).  Stream B, 1 ms later: null space + elimination of a
100 x 200 round.  Reported: when B's two kernels finished relative to A's start and end.  With the single-work-group kernels
(1024 threads x 128 VGPRs; 160 KB of LDS) B waits for A's grid to drain; a form whose work-groups fit into the hole ONE
block-sum work-group leaves (4 waves, <= 204 VGPRs) could run beside it.
"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from basq_amd._ops import HipOps
from basq_amd._partition import RoundGeometry
from basq_amd.kernels import StationaryKernel
from basq_amd.pools import gmm_pool

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
oa, ob = HipOps("cuda:0", stream=sa), HipOps("cuda:0", stream=sb)
R, m, d, n = 1_000_000, 10_000, 10, 100
S = 2 * n
spec = StationaryKernel("rbf", 2.0).spec(d)
with torch.cuda.stream(sa):
    pts = oa.to_device(gmm_pool(R, d, 0)); nys = pts[:m].contiguous(); c = oa.col_mean(nys)
    A = oa.pack(spec, nys, c, 0, pad_rows_to=64); B = oa.pack(spec, pts, c, 1); mu, _ = oa.init_state(R, 0, R)
geo = RoundGeometry.of(R, S); Rr = (geo.nb // 16) * 16 * S
g = torch.Generator().manual_seed(0)
X = torch.randn(100, 200, generator=g, dtype=torch.float64); X[0] = 1.0
with torch.cuda.stream(sb):
    Xd = ob.to_device(X); mu0 = ob.to_device(torch.rand(200, generator=g, dtype=torch.float64) + 0.1)
torch.cuda.synchronize()


def wide():
    oa.blocksum(spec, A, m, B, mu, None, Rr, 0, geo.n_full, S, 14, class_mod=16)


def chain(k=1):
    for _ in range(k):
        P = ob.nullspace(Xd, 100, 200); ob.car_eliminate(P, mu0.clone(), 200, 100)


with torch.cuda.stream(sa):
    wide()
with torch.cuda.stream(sb):
    chain()
torch.cuda.synchronize()
for k in (1, 6):
    for _ in range(3):
        ev = {key: torch.cuda.Event(enable_timing=True) for key in ("a0", "a1", "b0", "b1")}
        with torch.cuda.stream(sa):
            ev["a0"].record(sa); wide(); ev["a1"].record(sa)
        time.sleep(0.001)
        with torch.cuda.stream(sb):
            ev["b0"].record(sb); chain(k); ev["b1"].record(sb)
        torch.cuda.synchronize()
        print(f"{k} round(s) of reductions enqueued 1 ms into a 14-class launch: block sums {ev['a0'].elapsed_time(ev['a1']):6.2f} ms; "
              f"reductions finish {ev['a0'].elapsed_time(ev['b1']):6.2f} ms after the launch began "
              f"({ev['b0'].elapsed_time(ev['b1']):6.2f} ms after they were enqueued)")
# alone, for reference
for k in (1, 6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sb):
        e0.record(sb); chain(k); e1.record(sb)
    torch.cuda.synchronize()
    print(f"{k} round(s) of reductions alone: {e0.elapsed_time(e1):6.2f} ms")
