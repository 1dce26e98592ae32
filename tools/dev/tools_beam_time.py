"""Time the prefix beam search kernels at the headline shape (development aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from policy_gradient_asr_amd import hipops
DEV = "cuda:0"
T, B, V = 1000, 32, 29
g = torch.Generator().manual_seed(0)
for name, scale in (("peaked", 2.0), ("flat(untrained)", 0.05)):
    lp = torch.log_softmax(torch.randn(T, B, V, generator=g) * scale, 2).to(DEV)
    for beam in (16, 5):
        for generic in (False, True):
            for _ in range(2):
                hipops.ctc_beam_search(lp, None, beam=beam, generic=generic)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                hipops.ctc_beam_search(lp, None, beam=beam, generic=generic)
            e1.record(); torch.cuda.synchronize()
            print(f"{name:16s} beam {beam:2d} {'generic' if generic else 'small  '}: {e0.elapsed_time(e1)/5:.3f} ms  ({e0.elapsed_time(e1)/5/T*1e3:.2f} us/frame)", flush=True)
