"""Time the CTC lattice + gradient at the headline shape (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
T, B, V, L = 1000, 32, 29, 100
g = torch.Generator().manual_seed(0)
lp = torch.log_softmax(torch.randn(T, B, V, generator=g), 2).to(dev)
tg = torch.randint(1, V, (B, L), generator=g, dtype=torch.int32).to(dev)
il = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), L, dtype=torch.int32, device=dev)
for need in (False, True):
    hipops.ctc_loss_grad(lp, tg, il, tl, need_grad=need); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hipops.ctc_loss_grad(lp, tg, il, tl, need_grad=need)
    e1.record(); torch.cuda.synchronize()
    print(f"ctc {'lattice+grad' if need else 'lattice only'}: {e0.elapsed_time(e1) * 100:.1f} us")
