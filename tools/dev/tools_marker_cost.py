"""Diagnostic: what an event record / a satisfied event wait costs the stream it is issued on (behind a 500-us sleeper, so
that the host is ahead and the packets are processed back to back)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
tiny = torch.zeros(1, device=dev); zero_words = torch.zeros(8, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
with torch.cuda.stream(side): tiny.add_(1)
torch.cuda.synchronize()
N = 40
def run(between):
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        hipops.stream_gate(zero_words.data_ptr(), timeout_us=500)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(N):
            tiny.add_(1)
            between()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / N)
    return best
old = torch.cuda.Event(); old.record(); torch.cuda.synchronize()
def nothing(): pass
def rec(): torch.cuda.Event().record()
def rec_timing(): torch.cuda.Event(enable_timing=True).record()
def wait_done(): torch.cuda.current_stream().wait_event(old)
def side_waits_main(): side.wait_stream(torch.cuda.current_stream())
def rec2(): rec(); rec()
def wait_other():
    # a REAL barrier packet: the event is recorded on the other stream now (the host is ahead: it is not complete yet when
    # the wait is enqueued) but long complete when the main stream gets there
    with torch.cuda.stream(side):
        ev = torch.cuda.Event(); ev.record()
    torch.cuda.current_stream().wait_event(ev)
base = run(nothing)
print(f"tiny kernel alone: {base:.2f} us per iteration")
for name, fn in (("+ event record", rec), ("+ 2 event records", rec2), ("+ timing event record", rec_timing), ("+ wait on a completed event", wait_done),
                 ("+ side.wait_stream(main)", side_waits_main), ("+ wait on another stream's (earlier) event", wait_other)):
    print(f"{name:32s}: +{run(fn) - base:.2f} us", flush=True)
