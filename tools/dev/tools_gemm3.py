"""x3w NT / NN products and the TN weight-gradient product on the path's shapes: time only (development aid; the tile and
diagnostic variants are chosen by PGASR_X3W_TILE / PGASR_X3W_DIAG / PGASR_TN_TILE in the environment)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, K, N = 32000, 512, 2048
reps = int(os.environ.get("REPS", "5"))
X = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
dG = torch.randn(M, N, generator=g).to(dev)
def timeit(fn, n=reps):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
pl = hipops.split_planes(W); plt = hipops.split_planes(W, transpose=True)
C = torch.empty(M, N, device=dev); dX = torch.empty(M, K, device=dev); dW = torch.empty(N, K, device=dev)
which = os.environ.get("WHICH", "nt,nn,tn").split(",")
fl = 2.0 * M * N * K / 1e9
out = []
if "nt" in which:
    t = timeit(lambda: hipops.gemm_x3w(X, pl, C, M, N, K)); out.append(f"xproj NT {t*1e3:.0f} us ({fl/t:.0f} TF)")
if "nn" in which:
    t = timeit(lambda: hipops.gemm_x3w(dG, plt, dX, M, K, N)); out.append(f"dX NN {t*1e3:.0f} us ({fl/t:.0f} TF)")
if "tn" in which:
    for sk in (8, 16):
        t = timeit(lambda: hipops.gemm(dG, X, dW, N, K, M, transA=True, lda=N, splitk=sk, precision=1)); out.append(f"dW TN splitk {sk}: {t*1e3:.0f} us ({fl/t:.0f} TF)")
print("  ".join(out), flush=True)
if os.environ.get("CHECK"):
    ref = dG.double().t() @ X.double()
    print(f"dW max rel err vs fp64: {float((dW.double()-ref).abs().max()/ref.abs().max()):.2e}")
