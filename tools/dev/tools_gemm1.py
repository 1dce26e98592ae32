"""One bf16x3 xproj GEMM shape repeated (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, K, N = 32000, 512, 2048
X = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
C = torch.empty(M, N, device=dev)
for _ in range(3):
    hipops.gemm(X, W, C, M, N, K, transB=True, precision=1)
torch.cuda.synchronize()
