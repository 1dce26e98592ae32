"""Phases and instrumented-kernel times of the headline train step in one precision mode (development aid).
PREC=f32|bf16x3 (default f32), STEPS (default 40).  Prints ms per step, the five phases, the six sweeps and the
event-timed GEMM launches (two extra steps)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F

prec = os.environ.get("PREC", "f32")
steps = int(os.environ.get("STEPS", "40"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev).train()
tr = PolicyGradientTrainer(m, lr=5e-4, lam=1.0, seed=1234, precision=prec)
batch = [v.to(dev) for v in synth_batch(100)]
for i in range(5):
    tr.step(*batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    tr.step(*batch)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
hipops.profile_reset(True, only=("lstm_",))
marks = []
for i in range(8):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    tr.step(*batch)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    marks.append((e0, e1))
ph = hipops.profile_phases(marks)
hipops.profile_reset(True)
for i in range(2):
    tr.step(*batch)
prof = {k: (v[0] / 2, v[1] / 2) for k, v in hipops.profile_collect().items()}
hipops.profile_reset(False)
hipops.lstm_assert_no_timeouts()
print(json.dumps({"precision": prec, "ms_per_step": ms, "phases": ph, "instrumented_ms_per_step": prof}), flush=True)
