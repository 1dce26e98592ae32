"""Diagnostic (needs `make stamps`): cycle breakdown of a BACKWARD sweep -- alone on the chip, and the last one of a train step
(layer 0's: fed by the input gradient of layer 1, streamed, with the weight-gradient products beside it).  PREC=f32|bf16x3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PGASR_HIP_LIB"] = os.path.join(ROOT, "policy_gradient_asr_amd", "libpgasr_hip_stamps.so")
sys.path.insert(0, ROOT)
import torch
import bench
from policy_gradient_asr_amd import hipops, _lib
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "f32")
hipops.set_precision(prec)
lib = _lib.load()
T, B = bench.T, bench.B_PER_GPU
NAMES = {0: "loop top + pre-poll cell work", 1: "poll (partial sums valid)", 2: "cell gradient + plane stores", 3: "LDS barrier",
         5: "fragment reads + first 2 tiles' MFMAs", 6: "their tag + stores", 7: "last 2 tiles' MFMAs", 4: "their tag + stores"}


def report(tag, ws):
    nb = lib.pgasr_lstm_workspace_bytes(T, B, 1)
    raw = ws[nb - 4096 * 64: nb].view(torch.int64)[:10].cpu().double()
    st = raw[:8] / T
    print(f"{tag}: in-kernel clock {float(raw[8]) / float(raw[9]) * 100:.0f} MHz, {float(raw[9]) / 100 / T:.3f} us per step, {float(st.sum()):.0f} cycles per step")
    print("   " + "   ".join(f"{n}: {float(st[k]):.0f}" for k, n in NAMES.items()), flush=True)


# ---- alone
g = torch.Generator().manual_seed(0)
params = []
for d in range(2):
    params += [torch.randn(1024, 512, generator=g) * 0.05, (torch.rand(1024, 256, generator=g) * 2 - 1) / 16, torch.zeros(1024), torch.zeros(1024)]
params = [p.to(dev) for p in params]
wih, bias, pf, pb = hipops.lstm_pack(params, 512)
x = torch.randn(T, B, 512, generator=g).to(dev); dy = (torch.randn(T, B, 512, generator=g) * 1e-2).to(dev)
ln = torch.full((B,), T, dtype=torch.int32, device=dev)
gates0 = torch.empty(T, B, 2048, device=dev)
hipops.gemm(x, wih, gates0, M=T * B, N=2048, K=512, transB=True, bias=bias)
out = torch.empty(T, B, 512, device=dev); cbuf = torch.empty(T, B, 512, device=dev)
hipops.lstm_layer_fwd(gates0, out, cbuf, pf, ln, T, B)
for it in range(3):
    ws = hipops.lstm_layer_bwd(gates0.clone(), out, cbuf, dy, pb, ln, T, B)
torch.cuda.synchronize()
report(f"{prec} backward sweep ALONE", ws)
for it in range(3):
    words = torch.zeros(64, dtype=torch.int32, device=dev)
    ws = hipops.lstm_layer_bwd(gates0.clone(), out, cbuf, dy, pb, ln, T, B, slab=words)
torch.cuda.synchronize()
report(f"{prec} backward sweep alone, STREAMED (nobody listening)", ws)

# ---- in the step
torch.manual_seed(0)
model = Seq2Seq(bench.V, n_feats=bench.F); model.apply(weights); model = model.to(dev).train()
trainer = PolicyGradientTrainer(model, lr=5e-4, lam=1.0, seed=1234, precision=prec)
xb, t, fm, tm = [v.to(dev) for v in bench.synth_batch(0)]
for it in range(8):
    trainer.step(xb, t, fm, tm)
torch.cuda.synchronize()
hipops.lstm_assert_no_timeouts()
for key, w_ in hipops._ws_cache.items():
    if key[0] == "lstm_bwd":
        report(f"{prec} LAST backward sweep of a train step (layer 0: fed + streamed)", w_)
