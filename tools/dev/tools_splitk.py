"""Split-K sweep for the two weight-gradient GEMM shapes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
T, B, G, I, H = 1000, 32, 2048, 512, 256
dg = torch.randn(T, B, G, generator=g).to(dev); x = torch.randn(T, B, I, generator=g).to(dev); out = torch.randn(T, B, 2 * H, generator=g).to(dev)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for sk in (2, 4, 8, 16, 32):
    dwih = torch.empty(G, I, device=dev)
    t1 = timeit(lambda: hipops.gemm(dg, x, dwih, M=G, N=I, K=T * B, transA=True, lda=G, splitk=sk))
    dwhh = torch.zeros(2, 4 * H, H, device=dev)
    K = (T - 1) * B
    t2 = timeit(lambda: hipops.gemm(dg, out, dwhh, M=4 * H, N=H, K=K, transA=True, lda=G, ldb=2 * H, ldc=H, a_off=B * G, b_off=0,
                                    strideA=4 * H - B * G, strideB=B * 2 * H + H, strideC=4 * H * H, batch=2, splitk=sk))
    print(f"splitk {sk:2d}: dW_ih {t1:.0f} us   dW_hh {t2:.0f} us", flush=True)
