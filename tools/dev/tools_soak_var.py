"""Soak with ragged batches (BASELINE config 5: lengths U[T/2, T]) and odd shapes: many train steps, then no sweep may
have timed out and every loss must be finite (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
dev = torch.device("cuda:0")
V, F = 29, 80
def batch(B, T, L, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g)
    lens[0] = T
    x = torch.randn(B, F, T, generator=g)
    fmask = (torch.arange(T)[None, :] < lens[:, None]).float()
    x = x * fmask[:, None, :]
    tl = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
    targets = torch.randint(1, V, (B, L), generator=g)
    tmask = (torch.arange(L)[None, :] < tl[:, None]).long()
    targets = targets * tmask
    return tuple(t.to(dev) for t in (x, targets, fmask, tmask))
for (B, T, L, n) in ((32, 1000, 100, 150), (24, 777, 60, 150), (16, 333, 30, 150), (5, 97, 8, 150), (40, 200, 20, 100)):
    torch.manual_seed(0)
    m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev); m.train()
    tr = PolicyGradientTrainer(m, seed=1)
    t0 = time.perf_counter(); losses = []
    for i in range(n):
        losses.append(tr.step(*batch(B, T, L, i % 7)))
    hipops.lstm_assert_no_timeouts()
    ls = torch.stack(losses).float().cpu()
    print(f"B={B} T={T}: {n} steps {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step, finite={bool(torch.isfinite(ls).all())}, fed_ok={hipops.lstm_fed_ok(T, B)}", flush=True)
