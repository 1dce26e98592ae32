"""Where do the sweeps lose time INSIDE the step?  (development aid, round 5)
The headline f32 train step with the sweeps' diagnostic switches (results invalid, timing meaningful):
  flags 0x000  the product
  flags 0x100  storer waves issue no bulk stores (gates / c / h / dgates rows never leave the CU)
  flags 0x200  loader waves issue no LDS-DMA (the staged rows are never read from the ring)
  flags 0x300  both
Prints ms per step, the phases and the six sweeps for each variant.  PREC=f32|bf16x3, STEPS (default 30), NOWG=1: no weight-gradient
products (their workgroups hold the CUs the next layer's feed wants at the start of its sweep: how much is that?)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F

prec = os.environ.get("PREC", "f32")
steps = int(os.environ.get("STEPS", "30"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev).train()
tr = PolicyGradientTrainer(m, lr=0.0, lam=1.0, seed=1234, precision=prec)     # lr 0: garbage gradients of the diagnostic variants change nothing
batch = [v.to(dev) for v in synth_batch(100)]
hipops.adam_step = lambda *a, **k: None       # the diagnostic variants produce garbage gradients: keep the parameters (all variants alike)
if os.environ.get("NOWG"):                    # no weight-gradient products at all (results invalid): what do the fed sweeps cost without their competition?
    hipops.lstm_wgrads = lambda *a, **k: None
for flags in [int(f, 0) for f in os.environ.get("FLAGS", "0,0x100,0x200,0x300,0").split(",")]:
    hipops.LSTM_FLAGS = flags
    for i in range(4):
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
    hipops.profile_reset(False)
    for w in hipops.lstm_error_word_tensors(dev):
        w.zero_()
    print(json.dumps({"flags": hex(flags), "ms_per_step": round(ms, 3),
                      "phases": {k: ([round(x, 3) for x in v] if isinstance(v, list) else round(v, 3)) for k, v in (ph or {}).items()}}), flush=True)
hipops.LSTM_FLAGS = 0
