import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import basq_amd
from basq_amd import sober
from basq_amd._basis import make_cov_psd
from oracle.make_golden_sober import TUTORIAL_CASES, tutorial_inputs
from tests.cases import build_product_kernel
dev = torch.device("cuda", 0)
c = TUTORIAL_CASES[1]
pts, nys = tutorial_inputs(c)
pts, nys = pts.to(dev), nys.to(dev)
kern = build_product_kernel(c)
for variant in ("sober", "basq"):
    for rep in range(2):
        tr = basq_amd.EngineTrace(host_sync=True)
        torch.manual_seed(1)
        if variant == "sober":
            sober.recombination(pts, nys, c["n"], kern, dev, torch.float64, trace=tr)
        else:
            basq_amd.recombination(pts, nys, c["n"], kern, dev, trace=tr)
    print(variant, {k: round(v * 1e3, 2) for k, v in tr.timers.items()})
A = kern(nys, nys)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("make_cov_psd", t(lambda: make_cov_psd(A.clone())))
As = torch.sqrt(A * A.T)
print("cholesky_ex+item", t(lambda: torch.linalg.cholesky_ex(As).info.item()))
print("eigvalsh gpu", t(lambda: torch.linalg.eigvalsh(As)))
Ah = As.cpu()
print("eigvalsh cpu", t(lambda: torch.linalg.eigvalsh(Ah)))
print("d2h", t(lambda: As.cpu()))
