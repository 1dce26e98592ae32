"""A/B a functional.grad_overlap switch on the train step (development aid): python tools_ab.py split_tail"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd.functional import grad_overlap
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F
name = sys.argv[1]
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev); m.train()
tr = PolicyGradientTrainer(m, seed=1)
batch = [v.to(dev) for v in synth_batch(1)]
for rnd in range(3):
    for val in (False, True):
        setattr(grad_overlap, name, val)
        for _ in range(2): tr.step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): tr.step(*batch)
        torch.cuda.synchronize()
        print(f"{name}={val}: {(time.perf_counter()-t0)*100:.3f} ms/step", flush=True)
