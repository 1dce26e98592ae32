"""The CTC head at the headline shape (development aid): the fused head + log-softmax kernel against the general fp32 GEMM + log-softmax pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
M, K, V = 32000, 512, 29
x = torch.randn(M, K, device=dev); W = torch.randn(V, K, device=dev) * 0.1; b = torch.randn(V, device=dev)
y = torch.empty(M, V, device=dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, reps=20):
    ts = []
    for _ in range(reps):
        flush.zero_()                       # x out of the caches, as after a sweep that wrote it long ago
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def old():
    hipops.gemm(x, W, y, M=M, N=V, K=K, transB=True, bias=b, precision=0)
    hipops.log_softmax_rows(y.view(1, M, V))


print(f"general fp32 GEMM + log_softmax pass: {timeit(old):.1f} us")
us = timeit(lambda: hipops.head_logsoftmax(x, W, b))
print(f"fused head kernel:                    {us:.1f} us = {M * K * 4 / us / 1e6:.2f} TB/s of x")
