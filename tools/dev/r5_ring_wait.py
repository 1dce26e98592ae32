"""What does the loader waves' wait for the staging ring cost, and are the helpers late?  (development aid, round 5; needs the diagnostic
library: PGASR_HIP_LIB=.../libpgasr_hip_diag.so, built with -DPGASR_LSTM_DIAG.)  Runs headline train steps, then reads the counters the last
forward sweep (layer 3) and the last backward sweep (layer 1: fed + streamed) left in their workspaces' start-up words:
  loader of member 5 / member 0:  waits whose first poll found the slot not ready, retries, cycles in ring_wait, steps
  helper 0:                       cycles waiting for its turn (back-pressure), for the feed's tiles, moving a step's rows; steps; lead when ready"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F

prec = os.environ.get("PREC", "f32")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev).train()
tr = PolicyGradientTrainer(m, lr=5e-4, lam=1.0, seed=1234, precision=prec)
batch = [v.to(dev) for v in synth_batch(100)]
T, B = 1000, 32
for i in range(6):
    tr.step(*batch)
torch.cuda.synchronize()
for name, backward in (("last forward sweep (fed)", False), ("last backward sweep (fed + streamed)", True)):
    ws = hipops._lstm_ws(T, B, backward, dev)
    hello = ws[256:256 + 4 * 64 * 4].view(torch.int32).cpu().view(4, 64)
    for cl in range(4):
        w = hello[cl].tolist()
        l5, l0, hp = w[48:52], w[52:56], w[56:61]
        def loader(x):
            return {"late_waits": x[0], "retries": x[1], "cycles_per_step_in_ring_wait": round(16 * x[2] / max(x[3], 1)), "steps": x[3]}
        n = max(hp[3], 1)
        print(json.dumps({"sweep": name, "cluster": cl, "loader_member5": loader(l5), "loader_member0": loader(l0),
                          "helper0": {"cycles_per_iteration": {"back_pressure": round(16 * hp[0] / n), "fed_wait": round(16 * hp[1] / n), "move": round(16 * hp[2] / n)},
                                      "iterations": hp[3], "mean_lead_when_ready_steps": round(hp[4] / n, 2)}}), flush=True)
