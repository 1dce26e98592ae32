"""Time the CTC lattice + gradient kernels at the headline shape (development aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from policy_gradient_asr_amd import hipops
DEV = "cuda:0"
T, B, V, L = 1000, 32, 29, 100
g = torch.Generator().manual_seed(0)
lp = torch.log_softmax(torch.randn(T, B, V, generator=g), 2).to(DEV)
tg = torch.randint(1, V, (B, L), generator=g, dtype=torch.int32).to(DEV)
il = torch.full((B,), T, dtype=torch.int32, device=DEV); tl = torch.full((B,), L, dtype=torch.int32, device=DEV)
for _ in range(3):
    nll, h = hipops.ctc_lattice(lp, tg, il, tl)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    nll, h = hipops.ctc_lattice(lp, tg, il, tl)
e1.record(); torch.cuda.synchronize()
print(f"ctc lattice: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us  ({e0.elapsed_time(e1) / 10 / T * 1e3:.3f} us/frame), nll[0] = {float(nll[0]):.4f}")
