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
hipops.set_precision(os.environ.get("PREC", "bf16x3"))      # "f32": the three-plane / six-product sweep (NP = 3)
wih, bias, pf, pb = hipops.lstm_pack(params, 512)
gates = torch.randn(T, B, 2048, generator=g).to(dev)
out = torch.empty(T, B, 512, device=dev); cbuf = torch.empty(T, B, 512, device=dev)
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
hipops.LSTM_FLAGS = int(os.environ.get("PGASR_LSTM_FLAGS", "0"), 0)
for it in range(3):
    ws = hipops.lstm_layer_fwd(gates.clone(), out, cbuf, pf, lengths, T, B)
torch.cuda.synchronize()
nb = lib.pgasr_lstm_workspace_bytes(T, B, 0)
hello = ws[256:256 + 4 * 64 * 4].view(torch.int32).view(4, 64)[:, :16].cpu()     # 64 words per cluster, the first 16 are the members'
print("XCC id per cluster member (rows = clusters):")
print((hello & 0xF).tolist())
hw = (hello >> 16) & 0xFFFF
print("CU id (HW_ID[11:8]) / SH / SE per member:")
for r in range(4):
    print([f"se{(int(v) >> 13) & 7}.sh{(int(v) >> 12) & 1}.cu{(int(v) >> 8) & 15}" for v in hw[r].tolist()])
    ids = [int(v) >> 8 for v in hw[r].tolist()]
    print("   distinct CUs:", len(set(ids)), "of", len(ids))
raw = ws[nb - 4096 * 64: nb].view(torch.int64)[:10].cpu().double()
print(f"in-kernel clock: {float(raw[8]) / float(raw[9]) * 100:.0f} MHz  (s_memtime ticks / s_memrealtime ticks x 100 MHz over the whole sweep)")
st = raw[:8] / T
names = {0: "loop top (loader section, progress word)", 1: "poll: h_{t-1} words valid", 2: "MFMA + partial write", 3: "LDS barrier",
         4: "LDS reads + 4-way sum", 5: "cell math + publish", 6: "result staging (LDS writes)"}
print(f"cycles per step: {float(st.sum()):.0f}  (flags {hipops.LSTM_FLAGS:#x})")
for k, nme in names.items():
    print(f"  {nme:42s} {float(st[k]):8.0f}")
