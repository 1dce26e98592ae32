"""Diagnostic: per-phase cycle breakdown of one LSTM forward sweep (needs `make stamps`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PGASR_HIP_LIB"] = os.path.join(ROOT, "policy_gradient_asr_amd", "libpgasr_hip_stamps.so")
sys.path.insert(0, ROOT)
import torch
from policy_gradient_asr_amd import hipops, _lib
dev = torch.device("cuda:0")
T, B = 1000, 32
lib = _lib.load()
g = torch.Generator().manual_seed(0)
params = []
for d in range(2):
    params += [torch.randn(1024, 512, generator=g) * 0.05, (torch.rand(1024, 256, generator=g) * 2 - 1) / 16,
               torch.zeros(1024), torch.zeros(1024)]
params = [p.to(dev) for p in params]
wih, bias, pf, pb = hipops.lstm_pack(params, 512)
gates = torch.randn(T, B, 2048, generator=g).to(dev)
out = torch.empty(T, B, 512, device=dev); cbuf = torch.empty(T, B, 512, device=dev)
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
for it in range(3):
    ws = hipops.lstm_layer_fwd(gates.clone(), out, cbuf, pf, lengths, T, B)
torch.cuda.synchronize()
nb = lib.pgasr_lstm_workspace_bytes(T, B, 0)
hello = ws[256:256 + 4 * 16 * 4].view(torch.int32).view(4, 16).cpu()
print("XCC id per cluster member (rows = clusters):")
print((hello & 0xF).tolist())
st = ws[nb - 4096 * 64: nb].view(torch.int64).view(4096, 8)[:T].cpu().double()
d = st[5:T - 5]
segs = [(0, 1, "poll: h_{t-1} words valid"), (1, 2, "MFMA + partial write"), (2, 3, "LDS barrier"), (3, 6, "sum + cell + publish")]
tot = (d[1:, 0] - d[:-1, 0]).mean()
print(f"cycles per step (100 MHz wall clock x?): {tot:.0f}")
for i, j, nme in segs:
    x = d[:, j] - d[:, i]
    print(f"  {nme:28s} {x.mean():8.0f}  (min {x.min():.0f}, p50 {x.median():.0f}, max {x.max():.0f})")
x = d[1:, 0] - d[:-1, 6]
print(f"  {'loop back (loader/progress)':28s} {x.mean():8.0f}  (min {x.min():.0f}, p50 {x.median():.0f}, max {x.max():.0f})")
