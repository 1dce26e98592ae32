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
gates = torch.randn(T, B, 2048, generator=g).to(dev)
out = torch.empty(T, B, 512, device=dev); cbuf = torch.empty(T, B, 512, device=dev)
dout = (torch.randn(T, B, 512, generator=g) * 1e-3).to(dev)
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
hipops.lstm_layer_fwd(gates, out, cbuf, pf, lengths, T, B)
ws = hipops.lstm_layer_bwd(gates.clone(), out, cbuf, dout, pb, lengths, T, B)
torch.cuda.synchronize()
print("failed poll attempts per step (member 5, wave 0), per cluster:", (ws[160:176].view(torch.int32).float() / T).tolist())
