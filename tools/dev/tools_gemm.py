"""Ad-hoc GEMM timing on the path's shapes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, K, N = 32000, 512, 2048
X = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
dG = torch.randn(M, N, generator=g).to(dev)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for prec in (0, 1):
    C = torch.empty(M, N, device=dev)
    t1 = timeit(lambda: hipops.gemm(X, W, C, M, N, K, transB=True, precision=prec))
    dX = torch.empty(M, K, device=dev)
    t2 = timeit(lambda: hipops.gemm(dG, W, dX, M, K, N, precision=prec))
    dW = torch.empty(N, K, device=dev)
    t3 = timeit(lambda: hipops.gemm(dG, X, dW, N, K, M, transA=True, lda=N, splitk=8, precision=prec))
    fl = 2.0 * M * N * K / 1e9
    print(f"precision {prec}: xproj NT {t1*1e3:.0f} us ({fl/t1:.0f} TF)  dX NN {t2*1e3:.0f} us ({fl/t2:.0f} TF)  dW TN {t3*1e3:.0f} us ({fl/t3:.0f} TF)", flush=True)
pl = hipops.split_planes(W); plt = hipops.split_planes(W, transpose=True)   # (N,K) and (K,N)->... planes of W^T
C = torch.empty(M, N, device=dev); dX = torch.empty(M, K, device=dev)
t1 = timeit(lambda: hipops.gemm_x3w(X, pl, C, M, N, K))
t2 = timeit(lambda: hipops.gemm_x3w(dG, plt, dX, M, K, N))
fl = 2.0 * M * N * K / 1e9
print(f"x3w (LDS-DMA): xproj NT {t1*1e3:.0f} us ({fl/t1:.0f} TF)  dX NN {t2*1e3:.0f} us ({fl/t2:.0f} TF)", flush=True)
ref = (X.double() @ W.double().t())
print(f"x3w max rel err vs fp64: {float((C.double()-ref).abs().max()/ref.abs().max()):.2e}")
refx = dG.double() @ W.double()
print(f"x3w dX max rel err vs fp64: {float((dX.double()-refx).abs().max()/refx.abs().max()):.2e}")
for prec in (0, 1):
    C = torch.empty(M, N, device=dev)
    hipops.gemm(X, W, C, M, N, K, transB=True, precision=prec)
    print(f"precision {prec} max rel err vs fp64: {float((C.double()-ref).abs().max()/ref.abs().max()):.2e}")
