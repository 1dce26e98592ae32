"""Ad-hoc timing of the path's pieces on the GPU (development aid, not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from policy_gradient_asr_amd import hipops
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev).eval()
tr = PolicyGradientTrainer(m, seed=1)
batch = synth_batch(dev, 1)
for i in range(2):
    t0 = time.perf_counter(); tr.step(*batch); torch.cuda.synchronize()
    print(f"step {i}: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
hipops.profile_reset(True)
t0 = time.perf_counter()
for i in range(3):
    tr.step(*batch)
torch.cuda.synchronize()
print(f"3 steps: {(time.perf_counter()-t0)*1e3/3:.1f} ms/step", flush=True)
for k, (ms, n) in hipops.profile_collect().items():
    print(f"  {k}: {ms/3:.2f} ms/step over {n/3:.0f} launches", flush=True)
