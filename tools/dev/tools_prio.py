"""A/B: the train step issued on a high-priority stream (side streams stay at normal priority) (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F
print("priority range", torch.cuda.Stream.priority_range())
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev); m.train()
tr = PolicyGradientTrainer(m, seed=1)
batch = [v.to(dev) for v in synth_batch(1)]
hi = torch.cuda.Stream(priority=-1)
def run(stream, n):
    if stream is None:
        for _ in range(n): tr.step(*batch)
    else:
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for _ in range(n): tr.step(*batch)
        torch.cuda.current_stream().wait_stream(stream)
for rnd in range(3):
    for name, st in (("default", None), ("high-priority main", hi)):
        run(st, 3); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(st, 10); torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter()-t0)*100:.3f} ms/step", flush=True)
