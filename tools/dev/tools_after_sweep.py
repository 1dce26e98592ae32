"""Diagnostic: how long after a stand-alone forward sweep does the next kernel on the same stream finish, by kind of kernel."""
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
big = torch.empty(16 << 20, device=dev); tiny = torch.zeros(1, device=dev)
A2 = torch.randn(4096, 4096, device=dev); B2 = torch.randn(4096, 4096, device=dev); C2 = torch.empty(4096, 4096, device=dev)
kinds = {
    "tiny add_": lambda: tiny.add_(1),
    "64 MB fill": lambda: big.zero_(),
    "head GEMM (250 workgroups)": lambda: hipops.gemm(out.view(-1, 512), W, y, M=T * B, N=29, K=512, transB=True, bias=bvec),
    "log_softmax rows": lambda: hipops.log_softmax_rows(y.view(T, B, 29)),
}
def before_sweep():
    gs = gates0.clone()
    hipops.lstm_layer_fwd(gs, out, cbuf, pf, lengths, T, B)
def before_gemm():
    for _ in range(8): hipops.gemm(A2, B2, C2, M=4096, N=4096, K=4096)
Ax = torch.randn(T * B, 512, device=dev); planes = hipops.split_planes(torch.randn(2048, 512, device=dev) * 0.05); Cx = torch.empty(T * B, 2048, device=dev)
def before_x3w_sweep():
    hipops.gemm_x3w(Ax, planes, Cx, T * B, 2048, 512)
    before_sweep()
def before_x3w():
    hipops.gemm_x3w(Ax, planes, Cx, T * B, 2048, 512)
for bname, before in (("forward sweep", before_sweep), ("x3w, then sweep", before_x3w_sweep), ("x3w GEMM", before_x3w), ("8 x 4096^3 GEMM", before_gemm)):
    for name, fn in kinds.items():
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
        alone = e0.elapsed_time(e1) / 3 * 1e3
        out_line = f"after {bname:16s}: {name:28s} back-to-back {alone:7.1f} us; end of predecessor -> end of kernel"
        for nwait in (0, 5):
            res = []
            sides = [torch.cuda.Stream() for _ in range(nwait)]
            for rep in range(5):
                ea = torch.cuda.Event(enable_timing=True); eb = torch.cuda.Event(enable_timing=True)
                before()
                ea.record()
                fn()
                eb.record()
                later = torch.cuda.Event(); later.record()
                for sd in sides:                       # other queues blocked on a barrier that waits for this stream, as in a train step
                    sd.wait_event(later)
                    with torch.cuda.stream(sd):
                        tiny.add_(1)
                torch.cuda.synchronize()
                res.append(ea.elapsed_time(eb) * 1e3)
            out_line += f"  [{nwait} waiting queues] {sorted(res)[2]:7.1f} us"
        print(out_line, flush=True)
