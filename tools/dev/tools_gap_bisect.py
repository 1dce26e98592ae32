"""Diagnostic: the dead time between the last forward sweep and the head GEMM, in progressively smaller programs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from policy_gradient_asr_amd import hipops, functional as Fh
from policy_gradient_asr_amd.model import Seq2Seq, weights
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Seq2Seq(bench.V, n_feats=bench.F); model.apply(weights); model = model.to(dev).train()
x, t, fm, tm = [v.to(dev) for v in bench.synth_batch(0)]
marks = []
_fwd = hipops.lstm_layer_fwd
def fwd(*a, **k):
    r = _fwd(*a, **k)
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("sweep_end", e))
    return r
hipops.lstm_layer_fwd = fwd
_gemm = hipops.gemm
def gemm(A, B, C, **k):
    if k.get("N") == bench.V:
        e = torch.cuda.Event(enable_timing=True); e.record()
        r = _gemm(A, B, C, **k)
        e2 = torch.cuda.Event(enable_timing=True); e2.record(); marks.append(("head", e, e2))
        return r
    return _gemm(A, B, C, **k)
hipops.gemm = gemm
def report(tag):
    torch.cuda.synchronize()
    ends = [m for m in marks if m[0] == "sweep_end"]; heads = [m for m in marks if m[0] == "head"]
    per = len(ends) // len(heads)
    gaps = [ends[per * (i + 1) - 1][1].elapsed_time(heads[i][2]) * 1e3 for i in range(len(heads))]
    print(f"{tag}: last forward sweep end -> head GEMM end (35 us of kernel): " + " ".join(f"{g:.0f}" for g in gaps[-6:]), flush=True)
    del marks[:]
mode = os.environ.get("MODE", "")
if mode == "nosides":          # every "side" stream is the calling stream itself: one queue in the whole program
    Fh.grad_overlap.side_stream = classmethod(lambda cls: torch.cuda.current_stream())
    Fh.grad_overlap.second_side_stream = classmethod(lambda cls: torch.cuda.current_stream())
print("MODE", mode, "FEED_AHEAD", Fh.FEED_AHEAD)
with torch.no_grad():
    for i in range(8): model.logits(x, fm)
report("forward only, no_grad, train mode, host running ahead")
for i in range(8):
    z, _ = model.logits(x, fm)
report("forward only, with autograd graph")
