"""GPU time from the start of a train step to the launch of its first forward sweep, and from the end of the last
backward sweep to the end of the step (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev); m.train()
tr = PolicyGradientTrainer(m, seed=1)
batch = [v.to(dev) for v in synth_batch(1)]
marks = []
orig_f, orig_b = hipops.lstm_layer_fwd, hipops.lstm_layer_bwd
def fwd(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("f", e)); return orig_f(*a, **k)
def bwd(*a, **k):
    r = orig_b(*a, **k); e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("b", e)); return r
hipops.lstm_layer_fwd, hipops.lstm_layer_bwd = fwd, bwd
for _ in range(5): tr.step(*batch)
torch.cuda.synchronize()
rows = []
for _ in range(20):
    del marks[:]
    s = torch.cuda.Event(enable_timing=True); s.record()
    tr.step(*batch)
    e = torch.cuda.Event(enable_timing=True); e.record()
    rows.append((s, list(marks), e))
torch.cuda.synchronize()
import statistics as st
front = [r[0].elapsed_time(r[1][0][1]) for r in rows]
f = [[r[1][i][1].elapsed_time(r[1][i + 1][1]) for i in range(2)] for r in rows]
loss = [r[1][2][1].elapsed_time(r[1][3][1]) for r in rows]      # third fwd launch -> end of first bwd sweep
bw = [[r[1][i][1].elapsed_time(r[1][i + 1][1]) for i in (3, 4)] for r in rows]
tail = [r[1][5][1].elapsed_time(r[2]) for r in rows]
tot = [r[0].elapsed_time(r[2]) for r in rows]
print(f"step {st.mean(tot):.3f} ms: front {st.mean(front):.3f} | fwd launch-to-launch {st.mean(x[0] for x in f):.3f} {st.mean(x[1] for x in f):.3f} | "
      f"fwd3 launch -> bwd3 end {st.mean(loss):.3f} | bwd end-to-end {st.mean(x[0] for x in bw):.3f} {st.mean(x[1] for x in bw):.3f} | tail {st.mean(tail):.3f}")
