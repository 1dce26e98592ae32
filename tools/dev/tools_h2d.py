"""How fast is the batch's H2D copy, and is it a DMA or a kernel?  (development aid)"""
import os, sys, time, torch
dev = torch.device("cuda:0")
x = torch.randn(32, 80, 1000).pin_memory()
d = torch.empty_like(x, device=dev)
s = torch.cuda.Stream()
for mode in ("pinned", "pageable"):
    src = x if mode == "pinned" else x.clone()
    for _ in range(3):
        d.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(20):
        d.copy_(src, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(mode, f"{ms:.3f} ms per 10.24 MB copy = {x.numel()*4/ms/1e6:.1f} GB/s; host {1e3*(time.perf_counter()-t0)/20:.3f} ms", flush=True)
