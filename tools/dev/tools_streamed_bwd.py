"""Streamed backward sweep at the headline shape (development aid): what the slab publication costs the sweep and how far behind
the sweep's end the gated weight-gradient launch finishes.

  plain sweep | streamed sweep, nobody listening | streamed sweep + gated dW beside it (sweep end, dW end) | sweep then un-gated dW
REPS (default 5), T (default 1000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops, streams
dev = torch.device("cuda:0")
T, B, H = int(os.environ.get("T", "1000")), 32, 256
G, I = 8 * H, 2 * H
reps = int(os.environ.get("REPS", "5"))
g = torch.Generator().manual_seed(0)
params = []
for d in range(2):
    params += [torch.randn(4 * H, I, generator=g) * 0.05, (torch.rand(4 * H, H, generator=g) * 2 - 1) / 16, torch.zeros(4 * H), torch.zeros(4 * H)]
params = [p.to(dev) for p in params]
hipops.set_precision(os.environ.get("PREC", "bf16x3"))      # "f32": three-plane sweep, six-product weight gradients
wih, bias, pf, pb = hipops.lstm_pack(params, I)
x = torch.randn(T, B, I, generator=g).to(dev)
dy = (torch.randn(T, B, I, generator=g) * 1e-2).to(dev)
ln = torch.full((B,), T, dtype=torch.int32, device=dev)
gates0 = torch.empty(T, B, G, device=dev)
hipops.gemm(x, wih, gates0, M=T * B, N=G, K=I, transB=True, bias=bias)
out = torch.empty(T, B, I, device=dev); cbuf = torch.empty(T, B, I, device=dev)
hipops.lstm_layer_fwd(gates0, out, cbuf, pf, ln, T, B)
side = streams.side_stream("tool_streamed")
modes = os.environ.get("MODES", "plain,streamed_alone,streamed+dW,plain+dW_beside_unrelated,sequential").split(",")
assert not any("dW" in m for m in modes) or hipops.streams_concurrent(side)      # (a counter-collecting profiler serialises kernels: sweep-only modes there)
dwih = torch.empty(G, I, device=dev); dwhh = torch.empty(2, 4 * H, H, device=dev)
busy = hipops.lstm_busy_ptr(T, B, True, dev)


def run(mode):
    dg = gates0.clone()
    words = torch.zeros(64, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    if mode == "plain":
        hipops.lstm_layer_bwd(dg, out, cbuf, dy, pb, ln, T, B); e1.record(); e2 = e1
    elif mode == "streamed_alone":
        hipops.lstm_layer_bwd(dg, out, cbuf, dy, pb, ln, T, B, slab=words); e1.record(); e2 = e1
    elif mode == "streamed+dW":
        ws = hipops.lstm_layer_bwd(dg, out, cbuf, dy, pb, ln, T, B, slab=words); e1.record()
        with torch.cuda.stream(side):
            side.wait_event(e0)
            hipops.stream_gate(busy, need=4, timeout_us=5000, running=words)
            hipops.lstm_wgrads(dg, x, out, T, B, I, dwih, dwhh, busy_ptr=busy, slab=words, err_ws=ws)
            e2.record()
    elif mode == "plain+dW_beside_unrelated":
        # the interference alone: the same GEMM work beside the sweep on ANOTHER (complete) dgates tensor
        hipops.lstm_layer_bwd(dg, out, cbuf, dy, pb, ln, T, B); e1.record()
        with torch.cuda.stream(side):
            side.wait_event(e0)
            hipops.stream_gate(busy)
            hipops.lstm_wgrads(other, x, out, T, B, I, dwih, dwhh, busy_ptr=busy)
            e2.record()
    else:   # sequential
        hipops.lstm_layer_bwd(dg, out, cbuf, dy, pb, ln, T, B); e1.record()
        hipops.lstm_wgrads(dg, x, out, T, B, I, dwih, dwhh)
        e2.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), e0.elapsed_time(e2)


other = gates0.clone()
for mode in modes:
    run(mode)
    r = [run(mode) for _ in range(reps)]
    print(f"{mode:28s} sweep end {min(a for a, _ in r):.3f} ms (median {sorted(a for a, _ in r)[len(r) // 2]:.3f})   all done {min(b for _, b in r):.3f} ms (median {sorted(b for _, b in r)[len(r) // 2]:.3f})", flush=True)
hipops.lstm_assert_no_timeouts()
