"""Ad-hoc timing of the path's pieces on the GPU (development aid, not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev); m.train(os.environ.get("PGASR_TT_TRAIN","0")=="1")
tr = PolicyGradientTrainer(m, seed=1)
batch = [v.to(dev) for v in synth_batch(1)]
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

# ---- precision of one BLSTM layer vs fp64 torch-CPU (T=300,B=16) ----
import numpy as np
from policy_gradient_asr_amd import functional as Fh
g = torch.Generator().manual_seed(0)
Tn, Bn = 300, 16
lstm = torch.nn.LSTM(512, 256, 1, bidirectional=True).double()
xx = torch.randn(Tn, Bn, 512, generator=g).double(); dy = torch.randn(Tn, Bn, 512, generator=g).double() * 1e-4
xr = xx.clone().requires_grad_(True)
o, _ = lstm(xr); o.backward(dy)
names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
ps = [getattr(lstm, n_).detach().float().to(dev).requires_grad_(True) for n_ in names]
xg = xx.float().to(dev).requires_grad_(True)
y = Fh.blstm_layer(xg, torch.full((Bn,), Tn, dtype=torch.int32, device=dev), ps)
y.backward(dy.float().to(dev)); torch.cuda.synchronize()
re = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
print(f"precision: out {re(y.detach(), o.detach()):.2e}  dx {re(xg.grad, xr.grad):.2e}  " +
      "  ".join(f"{n_[7:]} {re(p_.grad, getattr(lstm, n_).grad):.2e}" for n_, p_ in zip(names, ps)), flush=True)

# ---- A/B: weight-gradient overlap on/off (interleaved) ----
import time
for rnd in range(2):
    for ov in (False, True):
        tr.overlap_weight_grads = ov
        tr.step(*batch); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            tr.step(*batch)
        torch.cuda.synchronize()
        print(f"overlap={ov}: {(time.perf_counter()-t0)*1e3/5:.2f} ms/step", flush=True)

# ---- A/B: confine side-stream GEMMs to the XCDs the sweep does not use ----
from policy_gradient_asr_amd.functional import grad_overlap
tr.overlap_weight_grads = True
for rnd in range(2):
    for cf in (False, True):
        grad_overlap.confine = cf
        tr.step(*batch); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            tr.step(*batch)
        torch.cuda.synchronize()
        print(f"confine={cf}: {(time.perf_counter()-t0)*1e3/5:.2f} ms/step", flush=True)
