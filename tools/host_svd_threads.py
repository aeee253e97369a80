"""How many host threads should the small LAPACK calls use?  (100x200 full SVD, 99x99 SVD, 1e4x99 QR)"""
import time
import torch
torch.manual_seed(0)
A = torch.randn(100, 200, dtype=torch.float64)
B = torch.randn(99, 99, dtype=torch.float64)
C = torch.randn(10000, 99, dtype=torch.float64)
print("host threads available:", torch.get_num_threads())
for t in (1, 2, 4, 8, 16, 32):
    torch.set_num_threads(t)
    out = []
    for M, f in ((A, lambda x: torch.linalg.svd(x)), (B, lambda x: torch.linalg.svd(x)), (C, lambda x: torch.linalg.qr(x))):
        f(M)
        t0 = time.perf_counter()
        for _ in range(10):
            f(M)
        out.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"threads={t:3d}  svd100x200 {out[0]:7.2f} ms   svd99x99 {out[1]:7.2f} ms   qr1e4x99 {out[2]:7.2f} ms")
