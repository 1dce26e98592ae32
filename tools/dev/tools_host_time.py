"""Host-side enqueue time per train step vs GPU time (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev); m.train()
tr = PolicyGradientTrainer(m, seed=1)
batch = [v.to(dev) for v in synth_batch(1)]
for _ in range(3): tr.step(*batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): tr.step(*batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/20:.2f} ms/step; total {1e3*(t2-t0)/20:.2f} ms/step")
