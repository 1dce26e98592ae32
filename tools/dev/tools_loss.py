"""Timing of the loss-phase kernels at the headline shape (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
T, B, V, L = 1000, 32, 29, 100
lp = torch.log_softmax(torch.randn(T, B, V, generator=g), 2).to(dev)
tg = torch.randint(1, V, (B, L), generator=g, dtype=torch.int32).to(dev)
il = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), L, dtype=torch.int32, device=dev)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"ctc loss+grad: {timeit(lambda: hipops.ctc_loss_grad(lp, tg, il, tl)):.0f} us")
gr, sm = hipops.frame_argmax_sample(lp, seed=1)
paths = torch.stack((gr, sm)).contiguous()
tok, tlen = hipops.ctc_collapse(paths, il)
print("hyp lengths:", tlen.float().mean(dim=1).tolist())
ref2 = torch.cat((tg, tg)).contiguous(); rl2 = torch.cat((tl, tl)).contiguous()
print(f"edit distance (64 pairs): {timeit(lambda: hipops.edit_distance(ref2, rl2, tok.view(2*B, T), tlen.view(2*B).contiguous())):.0f} us")
print(f"beam16 decode (32 utts): {timeit(lambda: hipops.ctc_beam_search(lp, il, beam=16), 2):.0f} us")
print(f"beam5 decode (32 utts): {timeit(lambda: hipops.ctc_beam_search(lp, il, beam=5), 2):.0f} us")
