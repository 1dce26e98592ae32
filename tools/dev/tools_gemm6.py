"""Round 4: the path's big products in the six-product (fp32-faithful) arithmetic of gemm_x6.hip against the bf16x3 kernels and
the exact fp32 MFMA kernel, same process (development aid).  Shapes: input projection (32000 x 2048 x 512), input gradient
(32000 x 512 x 2048), dW_ih (2048 x 512 x 32000, split-K 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
QUICK = os.environ.get("QUICK", "") == "1"      # only the six-product kernels (A/B of variants, PMC passes)

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

M = 32000
for N, K in ((2048, 512), (512, 2048)):
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    C = torch.empty(M, N, device=dev)
    p2 = hipops.split_planes(W, planes=2); p3 = hipops.split_planes(W, planes=3, packed=False); p3p = hipops.split_planes(W, planes=3, packed=True)
    ref = (A[:512].double() @ W.double().t()).cpu()
    res = {}
    e2 = e0 = float("nan")
    if not QUICK:
        res["x3c (bf16x3)"] = timeit(lambda: hipops.gemm_x3w(A, p2, C, M, N, K)); e2 = float((C[:512].double().cpu() - ref).abs().max() / ref.abs().max())
    # the product path packs W (hipops.X6_PACKED); the row-major planes are timed beside it outside QUICK (= profiled) runs only
    res["x6c (six products)"] = timeit(lambda: hipops.gemm_x3w(A, p3p, C, M, N, K)); e3 = float((C[:512].double().cpu() - ref).abs().max() / ref.abs().max())
    if not QUICK:
        res["x6c row-major W planes"] = timeit(lambda: hipops.gemm_x3w(A, p3, C, M, N, K)); e3p = float((C[:512].double().cpu() - ref).abs().max() / ref.abs().max()); assert e3p == e3
    if not QUICK:
        res["fp32 MFMA"] = timeit(lambda: hipops.gemm(A, W, C, M, N, K, transB=True, precision=0), reps=5); e0 = float((C[:512].double().cpu() - ref).abs().max() / ref.abs().max())
    gf = 2.0 * M * N * K / 1e9
    for k, us in res.items():
        print(f"C=A W^T M={M} N={N} K={K}: {k:22s} {us:7.1f} us  {gf / us * 1e3:7.1f} TF fp32-equivalent", flush=True)
    print(f"   max-norm error vs fp64: bf16x3 {e2:.2e}  six-product {e3:.2e}  fp32 MFMA {e0:.2e}", flush=True)
    us6 = res["x6c (six products)"]
    print(f"   six-product issued bf16: {6 * gf / us6:.3f} PF = {6 * gf / us6 / 2.5:.1%} of 2.5 PF", flush=True)

Mw, Nw, Kw = 2048, 512, 32000
A = torch.randn(Kw, Mw, generator=g).to(dev); B = torch.randn(Kw, Nw, generator=g).to(dev)
C = torch.empty(Mw, Nw, device=dev)
ref = (A.double().t()[:256] @ B.double()).cpu()
for prec, name in (((2, "t6 (six products)"),) if QUICK else ((1, "t256 (bf16x3)"), (2, "t6 (six products)"), (0, "fp32 MFMA"))):
    us = timeit(lambda: hipops.gemm(A, B, C, Mw, Nw, Kw, transA=True, splitk=16, precision=prec), reps=10 if prec else 3)
    err = float((C[:256].double().cpu() - ref).abs().max() / ref.abs().max())
    print(f"dW = dY^T X {Mw} x {Nw} x {Kw} split-K 16: {name:20s} {us:7.1f} us  err {err:.2e}", flush=True)
