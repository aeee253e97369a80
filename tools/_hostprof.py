import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basq_amd
from basq_amd.pools import gmm_pool
N, d, n = 1_000_000, 10, 100
pts = gmm_pool(N, d, seed=21).to("cuda:0"); nys = pts[: N // 100].contiguous()
kern = basq_amd.kernels.StationaryKernel("rbf", 2.0)
dev = torch.device("cuda:0")
for _ in range(2):
    torch.manual_seed(3); basq_amd.recombination(pts, nys, n, kern, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    torch.manual_seed(3); basq_amd.recombination(pts, nys, n, kern, dev)
torch.cuda.synchronize()
print("ms/batch", (time.perf_counter() - t0) / 3 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    torch.manual_seed(3); basq_amd.recombination(pts, nys, n, kern, dev)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
