"""Diagnostic: dispatch latency on the main stream while ONE other stream sits on an unsatisfied event wait (a barrier packet at
the head of its hardware queue) -- which streams of a pool interfere, and with how many blocked at once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
T, B = 1000, 32
W = torch.randn(29, 512, device=dev) * 0.05; bvec = torch.zeros(29, device=dev)
x = torch.randn(T * B, 512, device=dev); y = torch.empty(T * B, 29, device=dev)
tiny = torch.zeros(1, device=dev)
A2 = torch.randn(4096, 4096, device=dev); C2 = torch.empty(4096, 4096, device=dev)
def head(): hipops.gemm(x, W, y, M=T * B, N=29, K=512, transB=True, bias=bvec)
def chain(n=6):
    for _ in range(n): head()
zero_words = torch.zeros(8, dtype=torch.int32, device=dev)
def long_kernel():
    if os.environ.get("PRED", "gemms") == "sleeper":
        hipops.stream_gate(zero_words.data_ptr(), timeout_us=1000)       # ONE kernel (one sleeping wave) of 1 ms
    else:
        for _ in range(4): hipops.gemm(A2, A2, C2, M=4096, N=4096, K=4096)
NPOOL = int(os.environ.get("NPOOL", "12"))
pool = [torch.cuda.Stream() for _ in range(NPOOL)]
for sd in pool:                      # make every stream real (its hardware queue is created on first use)
    with torch.cuda.stream(sd): tiny.add_(1)
torch.cuda.synchronize()
SERVE = int(os.environ.get("SERVE", "-1"))      # -1: the chain runs on the default stream, else on pool[SERVE]
def measure(blocked):
    res = []
    serving = torch.cuda.current_stream() if SERVE < 0 else pool[SERVE]
    for rep in range(7):
        torch.cuda.synchronize()
        with torch.cuda.stream(serving):
            ea = torch.cuda.Event(enable_timing=True); eb = torch.cuda.Event(enable_timing=True)
            long_kernel()                      # ~1 ms during which the host enqueues everything below
            ea.record(); chain(); eb.record()
            later = torch.cuda.Event(); later.record()
        for i in blocked:
            if i == SERVE: continue
            pool[i].wait_event(later)
            with torch.cuda.stream(pool[i]): tiny.add_(1)
        torch.cuda.synchronize()
        res.append(ea.elapsed_time(eb) * 1e3)
    res.sort()
    return res[3], res[0], res[-1]
print("predecessor:", os.environ.get("PRED", "gemms"), " serving stream:", "default" if SERVE < 0 else f"pool[{SERVE}]")
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}; pool of {NPOOL} streams; 6 dependent 35-us kernels on the default stream")
print("nothing blocked: %.1f us (min %.1f max %.1f)" % measure([]))
for i in range(NPOOL):
    print(f"stream {i:2d} blocked: %.1f us (min %.1f max %.1f)" % measure([i]), flush=True)
print("all blocked: %.1f us (min %.1f max %.1f)" % measure(list(range(NPOOL))))
