"""Stand-alone FED sweeps at the headline shape, producer first (PMC target, development aid).

The train step launches a fed sweep BEFORE the GEMM that produces its rows and lets the two run side by side; a counter-collecting
profiler serialises kernels, so that order cannot be profiled.  Here the producer (pgasr_gemm_x3w_feed_f32, the same kernel and
decomposition) runs FIRST, to completion, and the fed sweep (pgasr_lstm_layer_fwd_fed / _bwd_fed: the kernels of the timed step,
helpers polling the tile counters, agent-scope loads of the fed rows, dropout mask applied by the backward helpers; the backward
sweep also streamed: slab publications and L2 write-backs by its flusher workgroups) then finds every tile counted complete.  What the counters see per sweep launch is the fed path's traffic without the wait."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
T, B, H = 1000, 32, 256
G, I = 8 * H, 2 * H
reps = int(os.environ.get("REPS", "3"))
g = torch.Generator().manual_seed(0)
params = []
for d in range(2):
    params += [torch.randn(4 * H, I, generator=g) * 0.05, (torch.rand(4 * H, H, generator=g) * 2 - 1) / 16, torch.zeros(4 * H), torch.zeros(4 * H)]
params = [p.to(dev) for p in params]
hipops.set_precision(os.environ.get("PREC", "bf16x3"))      # "f32": three planes / six products (sweeps and feeds)
NPL = hipops.LSTM_PLANES
wih, bias, pf, pb = hipops.lstm_pack(params, I)
planes = hipops.split_planes(wih)
planes_t = hipops.split_planes(wih, transpose=True)
x = torch.randn(T, B, I, generator=g).to(dev)
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
words = 2 * ((T * B + 255) // 256)
out = torch.empty(T, B, I, device=dev); cbuf = torch.empty(T, B, I, device=dev)
for r in range(reps):
    # forward: projection first, then the sweep fed by it
    gates = torch.empty(T, B, G, device=dev)
    done = torch.zeros(words, dtype=torch.int32, device=dev)
    hipops.gemm_x3w_feed(x, planes, gates, T * B, G, I, bias, 0, done)
    hipops.lstm_layer_fwd(gates, out, cbuf, pf, lengths, T, B, fed=done, fed_need=hipops.x3w_feed_col_tiles(G, NPL))
    # backward: the input gradient of a layer above (dgates_above x W_ih) first, then the sweep fed by it (mask applied by its helpers)
    dg_above = torch.randn(T, B, G, generator=torch.Generator().manual_seed(r)).to(dev) * 1e-3
    dout = torch.empty(T, B, I, device=dev)
    done2 = torch.zeros(words, dtype=torch.int32, device=dev)
    hipops.gemm_x3w_feed(dg_above, planes_t, dout, T * B, I, G, None, 0, done2, order=1)
    # .. and STREAMED, as in the timed step: the flusher workgroups write the XCD's L2 back once per time slab (nobody listens here)
    slab = torch.zeros(64, dtype=torch.int32, device=dev)
    hipops.lstm_layer_bwd(gates, out, cbuf, dout, pb, lengths, T, B, want_dbias=True, fed=done2, fed_need=hipops.x3w_feed_col_tiles(I, NPL), drop=(0.3, 0x5EED, 3), slab=slab)
torch.cuda.synchronize()
hipops.lstm_assert_no_timeouts()
print("fed sweeps done", flush=True)
