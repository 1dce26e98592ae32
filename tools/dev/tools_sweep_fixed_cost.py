"""Fixed cost of a sweep launch (development aid): stand-alone forward / backward sweeps of T = 1 .. 64 frames at B = 32, events around
the hipops call (prepare kernel + sweep kernel).  duration(T) = fixed + T * per-step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
B, H = 32, 256
G, I = 8 * H, 2 * H
g = torch.Generator().manual_seed(0)
params = []
for d in range(2):
    params += [torch.randn(4 * H, I, generator=g) * 0.05, (torch.rand(4 * H, H, generator=g) * 2 - 1) / 16, torch.zeros(4 * H), torch.zeros(4 * H)]
params = [p.to(dev) for p in params]
wih, bias, pf, pb = hipops.lstm_pack(params, I)
for T in (1, 2, 8, 32, 64, 256):
    x = torch.randn(T, B, I, generator=g).to(dev)
    ln = torch.full((B,), T, dtype=torch.int32, device=dev)
    gates0 = torch.empty(T, B, G, device=dev)
    hipops.gemm(x, wih, gates0, M=T * B, N=G, K=I, transB=True, bias=bias)
    out = torch.empty(T, B, I, device=dev); cbuf = torch.empty(T, B, I, device=dev)
    dy = torch.randn(T, B, I, generator=g).to(dev) * 1e-2
    res = {}
    for name in ("fwd", "bwd"):
        best = 1e9
        for rep in range(8):
            gt = gates0.clone()
            if name == "bwd":
                hipops.lstm_layer_fwd(gt, out, cbuf, pf, ln, T, B)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            if name == "fwd":
                hipops.lstm_layer_fwd(gt, out, cbuf, pf, ln, T, B)
            else:
                hipops.lstm_layer_bwd(gt, out, cbuf, dy, pb, ln, T, B)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        res[name] = best
    print(f"T = {T:4d}: forward {res['fwd']:7.1f} us   backward {res['bwd']:7.1f} us", flush=True)
