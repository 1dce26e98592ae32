"""Round 4 (diagnostic build, `make diag`): per-segment cycles of one x6c tile -- workgroup 0, every wave -- at the input-gradient
shape (M=32000, N=512, K=2048: one tile per CU, 128 steps).  PGASR_X6_VAR selects the structure, PGASR_X6_DIAG switches parts off."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PGASR_HIP_LIB"] = os.path.join(ROOT, "policy_gradient_asr_amd", "libpgasr_hip_diag.so")
sys.path.insert(0, ROOT)
import torch
from policy_gradient_asr_amd import hipops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, N, K = 32000, int(os.environ.get("N", "512")), int(os.environ.get("K", "2048"))
A = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
C = torch.empty(M, N, device=dev)
p3 = hipops.split_planes(W, planes=3)
for _ in range(3):
    hipops.gemm_x3w(A, p3, C, M, N, K)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 128)()
lib.pgasr_diag_x6_stamps.restype = ctypes.c_int
assert lib.pgasr_diag_x6_stamps(ctypes.byref(buf)) == 0
steps = K // 16
names = {0: "loop top", 1: "multiply (VAR0: first half; ping-pong: whole)", 2: "mid barrier", 3: "wait A regs (vmcnt)", 4: "convert", 5: "issue W DMA x3",
         6: "issue A loads x2", 7: "(to end-of-step waits)", 8: "vmcnt(12)+lgkmcnt(0)", 9: "barrier", 10: "multiply second half (VAR0)", 11: "after loop"}
print(f"VAR={os.environ.get('PGASR_X6_VAR', '0')} DIAG={os.environ.get('PGASR_X6_DIAG', '0')}  N={N} K={K}  steps={steps}")
for w in (0, 1, 4, 5):
    row = [buf[w * 16 + i] for i in range(16)]
    clk = row[12] / max(row[13], 1) * 100
    tot = sum(row[:11]) / steps
    print(f" wave {w}: {tot:7.0f} cycles/step at {clk:.0f} MHz :: " + "  ".join(f"[{i}] {row[i] / steps:.0f}" for i in range(11) if row[i]))
print("  segments: " + "; ".join(f"[{k}] {v}" for k, v in names.items()))
