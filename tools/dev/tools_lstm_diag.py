"""Diagnostic: forward-sweep time with parts of the per-step memory traffic disabled (results invalid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
T, B = 1000, 32
g = torch.Generator().manual_seed(0)
params = []
for d in range(2):
    params += [torch.randn(1024, 512, generator=g) * 0.05, (torch.rand(1024, 256, generator=g) * 2 - 1) / 16, torch.zeros(1024), torch.zeros(1024)]
params = [p.to(dev) for p in params]
wih, bias, pf, pb = hipops.lstm_pack(params, 512)
gates0 = torch.randn(T, B, 2048, generator=g).to(dev)
out = torch.empty(T, B, 512, device=dev); cbuf = torch.empty(T, B, 512, device=dev)
dout = torch.randn(T, B, 512, generator=g).to(dev) * 1e-3
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
def run(flags, bwd=False):
    hipops.LSTM_FLAGS = flags
    gates = gates0.clone()
    hipops.lstm_layer_fwd(gates, out, cbuf, pf, lengths, T, B); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    gs = [gates0.clone() for _ in range(3)]
    e0.record()
    for gt in gs:
        if bwd: hipops.lstm_layer_bwd(gt, out, cbuf, dout, pb, lengths, T, B)
        else: hipops.lstm_layer_fwd(gt, out, cbuf, pf, lengths, T, B)
    e1.record(); torch.cuda.synchronize()
    hipops.LSTM_FLAGS = 0
    return e0.elapsed_time(e1) / 3
for name, fl in (("normal", 0), ("no stores", 1 << 8), ("no loads", 2 << 8), ("no stores+no loads", 3 << 8), ("write-through", 1),
                 ("first poll taken (no dependency)", 4 << 8), ("no exchange loads", 8 << 8), ("no exchange loads, no publish", 24 << 8),
                 ("compute chain only (no I/O at all)", 27 << 8)):
    print(f"fwd {name:20s}: {run(fl):.3f} ms   bwd: {run(fl, True):.3f} ms", flush=True)
