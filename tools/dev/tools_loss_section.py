"""Diagnostic: event-timed timeline of every hipops launch between the last forward sweep and the first backward sweep of one
train step (no profiler attached: rocprof's own per-launch cost inflates exactly these gaps)."""
import os, sys, types, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from policy_gradient_asr_amd import hipops
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Seq2Seq(bench.V, n_feats=bench.F); model.apply(weights); model = model.to(dev).train()
trainer = PolicyGradientTrainer(model, lr=5e-4, lam=1.0, seed=1234)
trainer.overlap_weight_grads = os.environ.get("X_NO_OVERLAP") != "1"
x, t, fm, tm = [v.to(dev) for v in bench.synth_batch(0)]
for it in range(6):
    trainer.step(x, t, fm, tm)
torch.cuda.synchronize()
log = []
on = [False]
skip = {"profile_reset", "profile_collect", "profile_phases", "lstm_assert_no_timeouts", "lstm_error_words", "lstm_busy_ptr", "gemm_x3w_ok", "lstm_fed_ok"}
for name, fn in list(vars(hipops).items()):
    if isinstance(fn, types.FunctionType) and not name.startswith("_") and name not in skip:
        def wrap(fn=fn, name=name):
            def inner(*a, **k):
                if not on[0]:
                    return fn(*a, **k)
                s = torch.cuda.current_stream()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                h0 = time.perf_counter()
                e0.record(s)
                r = fn(*a, **k)
                e1.record(s)
                log.append((name, s.cuda_stream & 0xFFFF, e0, e1, h0))
                if name.startswith("lstm_layer"):
                    e2 = torch.cuda.Event(enable_timing=True); e2.record(s)
                    log.append(("(second marker right behind the sweep)", s.cuda_stream & 0xFFFF, e1, e2, h0))
                return r
            return inner
        setattr(hipops, name, wrap())
_ftm = model.encoder.forward_time_major
def ftm(*a, **k):
    r = _ftm(*a, **k)
    if on[0]:
        s_ = torch.cuda.current_stream()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter(); e0.record(s_)
        r[1].new_zeros(1).add_(1)          # a tiny kernel on the main stream
        e1.record(s_)
        log.append(("(marker) encoder done + tiny kernel", s_.cuda_stream & 0xFFFF, e0, e1, h0))
    return r
model.encoder.forward_time_major = ftm
for rep in range(3):
    del log[:]
    on[0] = True
    g0 = torch.cuda.Event(enable_timing=True); g0.record(); hs = time.perf_counter()
    trainer.step(x, t, fm, tm)
    on[0] = False
    torch.cuda.synchronize()
fw = [i for i, l in enumerate(log) if l[0].startswith("lstm_layer_fwd")]
bw = [i for i, l in enumerate(log) if l[0].startswith("lstm_layer_bwd")]
ref = log[fw[-1]][3]
print(f"loss section (events; main-stream time from the end of the last forward sweep to the start of the first backward sweep: "
      f"{ref.elapsed_time(log[bw[0]][2]) * 1e3:.0f} us)")
print("   GPU start -> end (us after the last forward sweep)        stream  call          [host call time / GPU start, us after step start]")
for name, sid, e0, e1, h0 in (log if os.environ.get('WHOLE_STEP') else log[fw[-1]:bw[0] + 1]):
    print(f"  {ref.elapsed_time(e0) * 1e3:8.1f} -> {ref.elapsed_time(e1) * 1e3:8.1f}  ({e0.elapsed_time(e1) * 1e3:6.1f})  s{sid:04x}  {name:24s} [{(h0 - hs) * 1e6:8.0f} / {g0.elapsed_time(e0) * 1e3:8.0f}]")
