"""Diagnostic: a loop shaped like the forward pass -- side stream: [wait for the previous iteration's end][a few small kernels]
[event]; main: [wait for that event][ONE long kernel][a 250-workgroup GEMM] -- with the host running ahead.  Reported: end of
the long kernel -> end of the GEMM (35 us of kernel), per choice of side stream and kind of long kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
T, B = 1000, 32
g = torch.Generator().manual_seed(0)
params = []
for d in range(2):
    params += [torch.randn(1024, 512, generator=g) * 0.05, (torch.rand(1024, 256, generator=g) * 2 - 1) / 16, torch.zeros(1024), torch.zeros(1024)]
params = [p.to(dev) for p in params]
wih, bias, pf, pb = hipops.lstm_pack(params, 512)
gates0 = torch.randn(T, B, 2048, generator=g).to(dev)
out = torch.empty(T, B, 512, device=dev); cbuf = torch.empty(T, B, 512, device=dev)
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
W = torch.randn(29, 512, device=dev) * 0.05; bvec = torch.zeros(29, device=dev)
y = torch.empty(T * B, 29, device=dev)
tiny = torch.zeros(1, device=dev); small = torch.zeros(1 << 16, device=dev)
zero_words = torch.zeros(8, dtype=torch.int32, device=dev)
def head(): hipops.gemm(out.view(-1, 512), W, y, M=T * B, N=29, K=512, transB=True, bias=bvec)
def long_sweep(): hipops.lstm_layer_fwd(gates0, out, cbuf, pf, lengths, T, B)
def long_sleeper(): hipops.stream_gate(zero_words.data_ptr(), timeout_us=1000)
pool = [torch.cuda.Stream() for _ in range(int(os.environ.get("NPOOL", "8")))]
for sd in pool:
    with torch.cuda.stream(sd): tiny.add_(1)
torch.cuda.synchronize()
main = torch.cuda.current_stream()
def loop(side, long_kernel, side_kernels=9, iters=8):
    gaps = []
    for it in range(iters):
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(side_kernels): small.add_(1)
            main.wait_stream(side)
        long_kernel()
        ea = torch.cuda.Event(enable_timing=True); ea.record()
        head()
        eb = torch.cuda.Event(enable_timing=True); eb.record()
        gaps.append((ea, eb))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 for a, b in gaps]
Ax = torch.randn(T * B, 512, device=dev); planes = hipops.split_planes(torch.randn(2048, 512, device=dev) * 0.05)
def loop3(side, long_kernel, iters=8, front=True):
    """three long kernels per iteration, each behind its own event of the side stream (the per-layer weight packs)"""
    gaps = []
    for it in range(iters):
        evs = []
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for l in range(3):
                    for _ in range(3): small.add_(1)
                    e = torch.cuda.Event(); e.record(); evs.append(e)
        if front:
            for _ in range(4): small.add_(1)          # "front end" kernels on the main stream
        for l in range(3):
            if evs: main.wait_event(evs[l])
            if os.environ.get("WITH_X3W") == "1":
                hipops.gemm_x3w(Ax, planes, gates0.view(T * B, 2048), T * B, 2048, 512)
            long_kernel()
        ea = torch.cuda.Event(enable_timing=True); ea.record()
        head()
        eb = torch.cuda.Event(enable_timing=True); eb.record()
        gaps.append((ea, eb))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 for a, b in gaps]
if os.environ.get("LOOP3") == "1":
    for lname, lk in (("sleeper", long_sleeper), ("forward sweep", long_sweep)):
        print(f"3 x {lname:14s} no side stream  :", " ".join(f"{v:5.0f}" for v in loop3(None, lk)))
        for i, sd in enumerate(pool[:5]):
            print(f"3 x {lname:14s} side = stream {i:2d}:", " ".join(f"{v:5.0f}" for v in loop3(sd, lk)), flush=True)
    sys.exit(0)
for lname, lk in (("sleeper", long_sleeper), ("forward sweep", long_sweep)):
    print(f"{lname:14s} no side stream  :", " ".join(f"{v:5.0f}" for v in loop(None, lk)))
    for i, sd in enumerate(pool):
        print(f"{lname:14s} side = stream {i:2d}:", " ".join(f"{v:5.0f}" for v in loop(sd, lk)), flush=True)
