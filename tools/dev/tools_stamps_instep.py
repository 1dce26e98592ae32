"""Diagnostic: cycle breakdown and in-kernel clock of the LAST forward sweep of a train step (needs `make stamps`),
to compare with tools_stamps.py (the same sweep alone on the chip)."""
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
torch.manual_seed(0)
model = Seq2Seq(bench.V, n_feats=bench.F); model.apply(weights); model = model.to(dev).train()
trainer = PolicyGradientTrainer(model, lr=5e-4, lam=1.0, seed=1234)
x, t, fm, tm = [v.to(dev) for v in bench.synth_batch(0)]
for it in range(8):
    trainer.step(x, t, fm, tm)
torch.cuda.synchronize()
hipops.lstm_assert_no_timeouts()
lib = _lib.load()
T, B = bench.T, bench.B_PER_GPU
nb = lib.pgasr_lstm_workspace_bytes(T, B, 0)
for key, ws in hipops._ws_cache.items():
    if key[0] != "lstm_fwd":
        continue
    raw = ws[nb - 4096 * 64: nb].view(torch.int64)[:10].cpu().double()
    print(f"in-step forward sweep: in-kernel clock {float(raw[8]) / float(raw[9]) * 100:.0f} MHz, {float(raw[9]) / 100 / T:.3f} us per step")
    st = raw[:8] / T
    names = {0: "loop top", 1: "poll", 2: "MFMA + partial write", 3: "LDS barrier", 4: "LDS reads + sum", 5: "cell + publish", 6: "staging"}
    print(f"cycles per step: {float(st.sum()):.0f}  " + "  ".join(f"{n} {float(st[k]):.0f}" for k, n in names.items()))
