"""Ad-hoc timing of the LDS-DMA GEMM (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, K, N = 32000, 512, 2048
X = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
dG = torch.randn(M, N, generator=g).to(dev)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
pl = hipops.split_planes(W); plt = hipops.split_planes(W, transpose=True)
C = torch.empty(M, N, device=dev); dX = torch.empty(M, K, device=dev)
t1 = timeit(lambda: hipops.gemm_x3w(X, pl, C, M, N, K))
t2 = timeit(lambda: hipops.gemm_x3w(dG, plt, dX, M, K, N))
fl = 2.0 * M * N * K / 1e9
t3 = timeit(lambda: hipops.gemm(X, W, C, M, N, K, transB=True, precision=1))
t4 = timeit(lambda: hipops.gemm(dG, W, dX, M, K, N, precision=1))
print(f"register-staged bf16x3: xproj NT {t3*1e3:.0f} us  dX NN {t4*1e3:.0f} us", flush=True)
print(f"LDS-DMA x3w: xproj NT {t1*1e3:.0f} us ({fl/t1:.0f} TF)  dX NN {t2*1e3:.0f} us ({fl/t2:.0f} TF)", flush=True)
