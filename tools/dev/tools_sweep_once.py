"""Stand-alone forward + backward sweep at the headline shape under PGASR_LSTM_FLAGS (development aid; PMC target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
T, B = 1000, int(os.environ.get("SWEEP_B", "32"))
g = torch.Generator().manual_seed(0)
params = []
for d in range(2):
    params += [torch.randn(1024, 512, generator=g) * 0.05, (torch.rand(1024, 256, generator=g) * 2 - 1) / 16, torch.zeros(1024), torch.zeros(1024)]
params = [p.to(dev) for p in params]
hipops.set_precision(os.environ.get("PREC", "bf16x3"))      # "f32": three planes / six products
wih, bias, pf, pb = hipops.lstm_pack(params, 512)
gates0 = torch.randn(T, B, 2048, generator=g).to(dev)
out = torch.empty(T, B, 512, device=dev); cbuf = torch.empty(T, B, 512, device=dev)
dout = torch.randn(T, B, 512, generator=g).to(dev) * 1e-3
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
res = []
for bwd in (False, True):
    gs = [gates0.clone() for _ in range(4)]
    hipops.lstm_layer_fwd(gs[0], out, cbuf, pf, lengths, T, B); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for gt in gs[1:]:
        if bwd: hipops.lstm_layer_bwd(gt, out, cbuf, dout, pb, lengths, T, B)
        else: hipops.lstm_layer_fwd(gt, out, cbuf, pf, lengths, T, B)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 3)
hipops.lstm_assert_no_timeouts()
print(f"{hipops.get_precision()} flags {hipops.LSTM_FLAGS:#x}: fwd {res[0]:.3f} ms   bwd {res[1]:.3f} ms", flush=True)
