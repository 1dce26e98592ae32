"""The input layer's weight gradient (dW = dpre^T xnorm: 32 batches of 512 x 80 x 1000, instance norm applied on load) at split-K 1 / 2 / 4:
the last dependent GEMM of the step's tail (development aid, round 5)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
B, F, T, N = 32, 80, 1000, 512
g = torch.Generator().manual_seed(0)
x = torch.randn(B, F, T, generator=g).to(dev)
dpre = torch.randn(T, B, N, generator=g).to(dev)
mean, rstd = hipops.instnorm_stats(x, 1e-5)
ref = None
for sk in (1, 2, 4, 8, 1):
    dW = torch.zeros(N, F, device=dev)
    def run():
        hipops.gemm(dpre, x, dW, M=N, N=F, K=T, transA=True, transB=True, lda=B * N, ldb=T, ldc=F, strideA=N, strideB=F * T, strideC=0,
                    batch=B, sum_batches=True, norm_operand=2, shift=mean, scale=rstd, splitk=sk)
    run(); torch.cuda.synchronize()
    if ref is None:
        ref = dW.clone()
    err = float((dW - ref).abs().max() / ref.abs().max())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"splitk": sk, "us": round(e0.elapsed_time(e1) / 20 * 1e3, 1), "max_rel_diff_vs_splitk1": err}), flush=True)
