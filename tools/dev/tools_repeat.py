"""Race hunt: the same train step (parameters frozen, CTC objective only, no dropout) must give the SAME gradient bits
every time it is repeated -- feed-ahead counters, K-quarter slabs, exchange words and side-stream GEMMs included
(development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for mode in ("eval", "train"):
    torch.manual_seed(0)
    m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev)
    m = m.eval() if mode == "eval" else m.train()
    tr = PolicyGradientTrainer(m, lr=0.0, lam=0.0, seed=1)
    batch = [v.to(dev) for v in synth_batch(1)]
    ref, bad = None, 0
    for i in range(n):
        if mode == "train":
            m.encoder._drop_calls = 0          # same dropout masks every step
        tr.step(*batch)
        g = tr.gflat.clone()
        if ref is None:
            ref = g
        elif not torch.equal(g, ref):
            bad += 1
    hipops.lstm_assert_no_timeouts()
    print(f"{mode}: {n} repeats, {bad} differing gradients, |g| = {float(ref.abs().sum()):.4f}", flush=True)
