"""Round 4 (verdict item 7): does the backward sweep's write-back waste (WRITE_SIZE 2.6x its algorithmic bytes: the helpers' L2 staging
ring written back) cost time when the HBM write path is NOT idle?  The stand-alone backward sweep at the headline shape, alone and
beside a device-to-device streaming copy (pgasr_stream_copy, WG workgroups that only take otherwise idle CUs -- on the four XCDs the
sweep leaves free AND on the idle CUs of its own XCDs) that runs for the whole sweep.  PREC=bf16x3|f32, WGS="0 64 128 220"."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops, streams
dev = torch.device("cuda:0")
T, B = 1000, 32
g = torch.Generator().manual_seed(0)
params = []
for d in range(2):
    params += [torch.randn(1024, 512, generator=g) * 0.05, (torch.rand(1024, 256, generator=g) * 2 - 1) / 16, torch.zeros(1024), torch.zeros(1024)]
params = [p.to(dev) for p in params]
hipops.set_precision(os.environ.get("PREC", "bf16x3"))
wih, bias, pf, pb = hipops.lstm_pack(params, 512)
gates0 = torch.randn(T, B, 2048, generator=g).to(dev)
out = torch.empty(T, B, 512, device=dev); cbuf = torch.empty(T, B, 512, device=dev)
dout = torch.randn(T, B, 512, generator=g).to(dev) * 1e-3
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
hipops.lstm_layer_fwd(gates0.clone(), out, cbuf, pf, lengths, T, B)
side = streams.side_stream("stream_copy_test")
nbytes = 1 << 30
src = torch.empty(nbytes, dtype=torch.uint8, device=dev); dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
src.random_(0, 255)
main = torch.cuda.current_stream()
for wgs in [int(v) for v in os.environ.get("WGS", "0 64 128 220").split()]:
    res = []
    for rep in range(4):
        gt = gates0.clone()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
        start = torch.cuda.Event(); start.record()
        if wgs:
            with torch.cuda.stream(side):
                side.wait_event(start)
                c0.record()
                for _ in range(4):          # ~4 GB moved: longer than the sweep
                    hipops.stream_copy(dst, src, workgroups=wgs)
                c1.record()
        e0.record()
        hipops.lstm_layer_bwd(gt, out, cbuf, dout, pb, lengths, T, B)
        e1.record()
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1), c0.elapsed_time(c1) if wgs else 0.0))
    sw = sorted(r[0] for r in res[1:])[1]
    cp = sorted(r[1] for r in res[1:])[1]
    rate = (4 * 2 * nbytes / (cp * 1e-3) / 1e12) if wgs else 0.0
    print(f"{hipops.get_precision()} backward sweep: {sw:.3f} ms beside a stream copy of {wgs} workgroups ({rate:.2f} TB/s of read + write traffic over {cp:.2f} ms)", flush=True)
hipops.lstm_assert_no_timeouts()
