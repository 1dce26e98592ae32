"""Soak: many train steps, then every LSTM workspace's error word must still be 0 and the loss finite (development aid).
PREC=f32|bf16x3, REWARD_MODE=utterance|per_step, argv[1] = steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev); m.train()
tr = PolicyGradientTrainer(m, seed=1, precision=os.environ.get("PREC") or None, reward_mode=os.environ.get("REWARD_MODE", "utterance"))
batch = [v.to(dev) for v in synth_batch(1)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
t0 = time.perf_counter()
losses = []
for i in range(n):
    losses.append(tr.step(*batch))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
bad = 0
for key, ws in hipops._ws_cache.items():
    if key[0].startswith("lstm"):
        bad += int(ws[:4].view(torch.int32).item() != 0)
ls = torch.stack(losses).float().cpu()
print(f"peak device memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB; "
      f"{n} steps in {dt:.2f} s = {dt / n * 1e3:.2f} ms/step; lstm error words set: {bad}; losses finite: {bool(torch.isfinite(ls).all())}; first/last loss {float(ls[0]):.3f} / {float(ls[-1]):.3f}")
