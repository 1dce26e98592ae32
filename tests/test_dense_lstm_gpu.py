"""GEMM / affine / LSTM / head parity against torch-CPU (the oracle's arithmetic), through the
C ABI.  Tolerance 1e-3 relative on outputs and gradients (north_star)."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lib_load():
    from policy_gradient_asr_amd import _lib
    return _lib.load()


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("tA,tB,M,N,K", [
    (0, 0, 128, 128, 16), (0, 1, 300, 200, 77), (1, 0, 129, 65, 40), (1, 1, 64, 257, 128),
    (0, 1, 1000, 29, 512), (0, 0, 500, 512, 29), (1, 0, 29, 512, 3000),
    (1, 0, 260, 132, 100), (1, 0, 1024, 256, 4000),     # TN through the transposing-read LDS image (vector loads)
])
@pytest.mark.parametrize("precision,tol", [(0, 1e-5), (1, 3e-5)])
def test_gemm_layouts(tA, tB, M, N, K, precision, tol):
    from policy_gradient_asr_amd import hipops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((N, K) if tB else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    want = (A.t() if tA else A).double() @ (B.t() if tB else B).double() + bias.double()
    C = torch.empty(M, N, device=DEV)
    hipops.gemm(A.to(DEV), B.to(DEV), C, M, N, K, transA=bool(tA), transB=bool(tB), bias=bias.to(DEV), precision=precision)
    assert rel_err(C.cpu(), want) < tol
    # split-K + accumulate + leaky epilogue
    C2 = torch.ones(M, N, device=DEV)
    hipops.gemm(A.to(DEV), B.to(DEV), C2, M, N, K, transA=bool(tA), transB=bool(tB), splitk=3, bias=bias.to(DEV),
                act=1, slope=0.01, accumulate=True, precision=precision)
    want2 = torch.nn.functional.leaky_relu(want, 0.01) + 1.0
    assert rel_err(C2.cpu(), want2) < tol


@pytest.mark.parametrize("M,N,K,with_bias,with_dact", [
    (300, 128, 32, True, False),        # one k-tile, ragged M inside one row tile
    (1000, 256, 96, True, True),        # 3 k-tiles (one per stage), ragged last row tile, both epilogues
    (513, 384, 160, False, True),       # 5 k-tiles: the 3-stage ring wraps
    (4096, 512, 2048, False, False),    # dX shape of the path (K = 8H)
    # N % 256 == 0: the 256 x 256 tile (one 16-deep k-step per stage, four stages, k-steps in pairs)
    (300, 256, 32, True, True),         # two k-steps: fewer than the ring's four stages
    (777, 768, 64, True, False),        # four k-steps = one turn of the ring; ragged last row tile; three column tiles
    (2000, 2048, 512, False, True),     # the input projection's K and N
    (1031, 512, 352, True, True),       # 22 k-steps: the ring wraps five times, odd number of pairs
])
@pytest.mark.parametrize("tile", ["128", "256", "c"])
def test_gemm_x3w_lds_dma(M, N, K, with_bias, with_dact, tile, monkeypatch):
    """pgasr_gemm_x3w_f32 (LDS-DMA tiles, pre-split weight planes) against fp64, in each of its three tile structures
    (PGASR_X3W_TILE, read by the library at every call; N % 256 != 0 always takes the 256 x 128 kernel)."""
    from policy_gradient_asr_amd import hipops
    monkeypatch.setenv("PGASR_X3W_TILE", tile)
    monkeypatch.setattr(hipops, "LSTM_PLANES", 2)       # the two-plane (bf16x3) kernels are what this test is about; the library default is "f32"
    monkeypatch.setattr(hipops, "GEMM_PRECISION", 1)
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1
    bias = torch.randn(N, generator=g) if with_bias else None
    y = torch.randn(M, N, generator=g) if with_dact else None
    want = A.double() @ W.double().t()
    if with_bias:
        want = want + bias.double()
    if with_dact:
        want = want * torch.where(y > 0, 1.0, 0.01).double()
    planes = hipops.split_planes(W.to(DEV))
    hi = (planes[0].cpu().to(torch.int32) << 16).view(torch.float32)
    lo = (planes[1].cpu().to(torch.int32) << 16).view(torch.float32)
    assert rel_err(hi.double() + lo.double(), W) < 2e-5          # x = hi + lo to ~2^-17
    C = torch.full((M, N), float("nan"), device=DEV)
    hipops.gemm_x3w(A.to(DEV), planes, C, M, N, K, bias=None if bias is None else bias.to(DEV),
                    dact_y=None if y is None else y.to(DEV), slope=0.01)
    assert rel_err(C.cpu(), want) < 3e-5
    # transposed planes: W given as (K, N)
    planes_t = hipops.split_planes(W.t().contiguous().to(DEV), transpose=True)
    assert torch.equal(planes_t[0], planes[0]) and torch.equal(planes_t[1], planes[1])
    assert hipops.gemm_x3w_ok(M, N, K) and not hipops.gemm_x3w_ok(M, N + 1, K) and not hipops.gemm_x3w_ok(M, N, K + 8)


@pytest.mark.parametrize("M,N,K,order,eighths,quarters,halves", [
    (8192, 512, 2048, 1, 0, 16, 16),     # the input-gradient feed of the step as shipped: 16 groups in quarters, 16 in halves (every tile split here)
    (8192, 512, 2048, 1, 2, 3, 4),       # eighths in front (A/B knob), then quarters, halves, whole tiles
    (8192, 512, 2048, 0, 0, 0, 0),       # whole tiles only
    (4000, 512, 2048, 1, 1, 1, 100),     # ragged last row tile; more halves asked for than tiles exist
    (4096, 2048, 512, 0, 0, 2, 0),       # the forward projection: K = 512, two groups in quarters (PGASR_X6_FWD_SPLIT_GROUPS)
    (4096, 2048, 512, 0, 0, 0, 0),
])
def test_gemm_x6w_feed_graded_head(M, N, K, order, eighths, quarters, halves, monkeypatch):
    """pgasr_gemm_x6w_feed_f32 without a consumer (no XCD mask): the graded head of round 5 -- tile groups in K-eighths, -quarters and -halves
    in front of the whole tiles, every tile the fixed-order sum of its parts (16-byte accesses to the parked accumulator sets) -- against fp64
    at fp32-GEMM accuracy, every row tile counted exactly once per direction half, and the same bits on a second run."""
    from policy_gradient_asr_amd import hipops
    monkeypatch.setenv("PGASR_X6_SPLIT8_GROUPS", str(eighths))
    monkeypatch.setenv("PGASR_X6_SPLIT_GROUPS", str(quarters))
    monkeypatch.setenv("PGASR_X6_SPLIT2_GROUPS", str(halves))
    monkeypatch.setenv("PGASR_X6_FWD_SPLIT_GROUPS", str(quarters))
    g = torch.Generator().manual_seed(M + N + K + eighths)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = torch.randn(N, K, generator=g) * 0.1
    bias = torch.randn(N, generator=g).to(DEV)
    want = A.double().cpu() @ W.double().t() + bias.double().cpu()
    pack = hipops.split_planes(W.to(DEV), planes=3, packed=True)
    mt = (M + 255) // 256
    outs = []
    for _ in range(2):
        C = torch.full((M, N), float("nan"), device=DEV)
        done = torch.zeros(2 * mt, dtype=torch.int32, device=DEV)
        hipops.gemm_x3w_feed(A, pack, C, M, N, K, bias, 0, done, order=order)
        torch.cuda.synchronize()
        assert done.cpu().tolist() == [hipops.x3w_feed_col_tiles(N, 3)] * (2 * mt)
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0].cpu(), want) < 2e-6
    # the same feed in two launches (pgasr_gemm_x6w_feed_phase_f32: the K-split head one item per workgroup, then the rest): the same bits
    items = int(_lib_load().pgasr_gemm_x6w_feed_head_items(M, N, K))
    assert 0 <= items <= 256 and (items > 0) == (eighths + quarters > 0)
    if items > 0:
        C = torch.full((M, N), float("nan"), device=DEV)
        done = torch.zeros(2 * mt, dtype=torch.int32, device=DEV)
        ws = hipops.gemm_x3w_feed(A, pack, C, M, N, K, bias, 0, done, order=order, phase=1)
        hipops.gemm_x3w_feed(A, pack, C, M, N, K, bias, 0, done, order=order, phase=2, ws=ws)
        torch.cuda.synchronize()
        assert done.cpu().tolist() == [hipops.x3w_feed_col_tiles(N, 3)] * (2 * mt)
        assert torch.equal(C, outs[0])
        # .. and with the queue words in a block the CALLER zeroed (no memset in either launch: they may go onto two streams)
        C2 = torch.full((M, N), float("nan"), device=DEV)
        done = torch.zeros(2 * mt, dtype=torch.int32, device=DEV)
        ctrl = torch.zeros(256, dtype=torch.int32, device=DEV)
        ws = hipops.gemm_x3w_feed(A, pack, C2, M, N, K, bias, 0, done, order=order, phase=1, ctrl=ctrl)
        hipops.gemm_x3w_feed(A, pack, C2, M, N, K, bias, 0, done, order=order, phase=2, ws=ws, ctrl=ctrl)
        torch.cuda.synchronize()
        assert torch.equal(C2, outs[0]) and int(ctrl[0]) >= mt * (N // 256)
    else:
        with pytest.raises(Exception):
            hipops.gemm_x3w_feed(A, pack, torch.empty(M, N, device=DEV), M, N, K, bias, 0, torch.zeros(2 * mt, dtype=torch.int32, device=DEV), order=order, phase=1)


@pytest.mark.parametrize("M,N,K,with_bias,with_dact", [
    (300, 256, 64, True, True),         # four 16-deep steps: the prologue alone fills the ring; ragged M inside one row tile
    (777, 768, 80, True, False),        # five steps (3-unrolled loop ends mid-turn); ragged last row tile; three column tiles
    (513, 256, 112, False, True),       # seven steps
    (2000, 2048, 512, False, True),     # the input projection's K and N
    (4096, 512, 2048, False, False),    # dX shape of the path (K = 8H): 128 steps
    (1031, 512, 352, True, True),       # 22 steps
])
def test_gemm_x6w_six_product(M, N, K, with_bias, with_dact):
    """pgasr_gemm_x6w_f32 (gemm_x6.hip: three bf16 planes per operand, six MFMA products, 16-deep steps) against fp64 at
    fp32-GEMM accuracy -- the arithmetic of the "f32" precision mode -- and the exactness of the 3-plane split."""
    from policy_gradient_asr_amd import hipops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1
    bias = torch.randn(N, generator=g) if with_bias else None
    y = torch.randn(M, N, generator=g) if with_dact else None
    want = A.double() @ W.double().t()
    if with_bias:
        want = want + bias.double()
    if with_dact:
        want = want * torch.where(y > 0, 1.0, 0.01).double()
    planes = hipops.split_planes(W.to(DEV), planes=3, packed=False)
    assert len(planes) == 3
    parts = [(p_.cpu().to(torch.int32) << 16).view(torch.float32).double() for p_ in planes]
    assert float(((parts[0] + parts[1] + parts[2]) - W.double()).abs().max() / W.abs().max()) < 2.0 ** -23     # x = hi + mid + lo
    two = hipops.split_planes(W.to(DEV), planes=2)
    assert torch.equal(two[0], planes[0])                       # the hi plane is the same rounding in both splits
    C = torch.full((M, N), float("nan"), device=DEV)
    hipops.gemm_x3w(A.to(DEV), planes, C, M, N, K, bias=None if bias is None else bias.to(DEV),
                    dact_y=None if y is None else y.to(DEV), slope=0.01)
    e6 = rel_err(C.cpu(), want)
    # the PACKED operand (the three planes tile by tile in the kernel's LDS-image order): the same bytes by another route, so the same bits
    pack = hipops.split_planes(W.to(DEV), planes=3, packed=True)
    assert isinstance(pack, hipops.X6Pack) and len(pack) == 3 and pack.pack.numel() == N * K * 6
    img = pack.pack.view(torch.int16).view(N // 256, K // 16, 3, 256, 2, 8).cpu()          # [tile][step][plane][row][chunk position][8]
    r = torch.arange(256)
    for pl in range(3):
        dense = planes[pl].cpu().view(N // 256, 256, K // 16, 2, 8).permute(0, 2, 1, 3, 4)    # [tile][step][row][k-chunk][8]
        for cp in range(2):
            assert torch.equal(img[:, :, pl, :, cp, :], dense[:, :, r, cp ^ ((r >> 3) & 1), :])
    Cp = torch.full((M, N), float("nan"), device=DEV)
    hipops.gemm_x3w(A.to(DEV), pack, Cp, M, N, K, bias=None if bias is None else bias.to(DEV), dact_y=None if y is None else y.to(DEV), slope=0.01)
    assert torch.equal(Cp, C)
    pack_t = hipops.split_planes(W.t().contiguous().to(DEV), transpose=True, planes=3, packed=True)
    assert torch.equal(pack_t.pack, pack.pack)
    with pytest.raises(Exception):
        hipops.gemm_x3w(A.to(DEV)[:, :K - 16].contiguous(), pack, Cp, M, N, K - 16)        # a pack knows the product it was made for
    # the fp32 MFMA kernel on the same product: the six-product result must be of the same quality
    C0 = torch.empty(M, N, device=DEV)
    hipops.gemm(A.to(DEV), W.to(DEV), C0, M, N, K, transB=True, bias=None if bias is None else bias.to(DEV), precision=0)
    if with_dact:
        C0 = C0 * torch.where(y.to(DEV) > 0, 1.0, 0.01)
    e0 = rel_err(C0.cpu(), want)
    print(f"[x6w] M={M} N={N} K={K}: six-product {e6:.2e}  fp32 MFMA {e0:.2e}")
    assert e6 < 2e-6 and e6 < 4 * e0 + 2e-7
    planes_t = hipops.split_planes(W.t().contiguous().to(DEV), transpose=True, planes=3, packed=False)
    assert all(torch.equal(a, b) for a, b in zip(planes_t, planes))
    assert hipops.gemm_x3w_ok(M, N, K, planes=3) and not hipops.gemm_x3w_ok(M, N + 128, K, planes=3) and not hipops.gemm_x3w_ok(M, N, K + 8, planes=3)


@pytest.mark.parametrize("M,K,V", [(32000, 512, 29), (77, 64, 7), (33, 256, 32), (1, 128, 1), (4000, 1024, 29)])
def test_head_logsoftmax_kernel_vs_fp64(M, K, V):
    """pgasr_head_logsoftmax (A4: Linear + log_softmax in one exact-fp32 kernel): logits and log-probs against fp64, ragged row
    counts (the tail block), the limits V = 32 / K = 1024, and what it refuses."""
    from policy_gradient_asr_amd import hipops, _lib
    g = torch.Generator().manual_seed(M + K + V)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(V, K, generator=g) * 0.1
    b = torch.randn(V, generator=g)
    want = x.double() @ W.double().t() + b.double()
    want_lp = torch.log_softmax(want, dim=1)
    z, lp = hipops.head_logsoftmax(x.to(DEV), W.to(DEV), b.to(DEV))
    assert rel_err(z.cpu(), want) < 2e-6
    assert float((lp.cpu().double() - want_lp).abs().max()) < 2e-5 * max(1.0, float(want_lp.abs().max()))
    z2, lp2 = hipops.head_logsoftmax(x.to(DEV), W.to(DEV), b.to(DEV), want_logits=False)
    assert z2 is None and torch.equal(lp2, lp)
    assert torch.equal(hipops.log_softmax_rows(z.view(1, M, V)).view(M, V), lp) or float((hipops.log_softmax_rows(z.view(1, M, V)).view(M, V) - lp).abs().max()) < 1e-5
    assert not hipops.head_logsoftmax_ok(K + 8, V) and not hipops.head_logsoftmax_ok(K, 33)
    with pytest.raises(_lib.PgasrError):
        hipops.head_logsoftmax(x.to(DEV), torch.randn(33, K).to(DEV), torch.zeros(33).to(DEV))


@pytest.mark.parametrize("busy_mask", [0x00, 0x0F, 0xA5, 0xFE, 0xFF])
def test_gemm_queue_mode_is_placement_independent(busy_mask):
    """Queue mode (pgasr_gemm_f32 xcc_busy != NULL): whichever XCDs are declared busy -- none, half, all but
    one, or all (then the unmasked second launch does everything) -- the result is bit-identical to the plain
    launch, for a weight-gradient shaped TN product with batch-sum slabs and accumulation."""
    from policy_gradient_asr_amd import hipops
    g = torch.Generator().manual_seed(7)
    M, N, K = 1024, 512, 3000
    A = torch.randn(K, M, generator=g).to(DEV)
    B = torch.randn(K, N, generator=g).to(DEV)
    base = torch.randn(M, N, generator=g).to(DEV)
    want = base.clone()
    hipops.gemm(A, B, want, M, N, K, transA=True, splitk=4, accumulate=True, precision=1)
    busy = torch.tensor([(busy_mask >> i) & 1 for i in range(8)], dtype=torch.int32, device=DEV)
    got = base.clone()
    hipops.gemm(A, B, got, M, N, K, transA=True, splitk=4, accumulate=True, precision=1, xcc_busy=busy.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = base.double().cpu() + A.double().cpu().t() @ B.double().cpu()
    assert rel_err(got.cpu(), ref) < 1e-5


@pytest.mark.parametrize("M,N,K,splitk,batch", [
    (256, 256, 64, 2, 1),            # one tile, one pair of k-steps per slab (fewer than the ring's four stages)
    (1024, 512, 3200, 4, 1),         # dW_ih-like; slabs of 800 -> kper 800 = 25 x 32
    (512, 256, 2016, 5, 2),          # two batches (dW_hh's two directions), a shorter last slab, 63 pairs
    (2048, 512, 8000, 8, 1),         # eight row tiles x two column tiles x eight slabs = 128 items
])
@pytest.mark.parametrize("precision,tol", [(1, 1e-5), (2, 1.5e-6)])
def test_gemm_tn_256_tile_weight_gradient_shapes(M, N, K, splitk, batch, precision, tol, monkeypatch):
    """TN products with both operands k-major (dW = dY^T X, model.py:39-44 backward) on the 256 x 256 LDS-DMA tile of
    gemm_dma.hip (taken by pgasr_gemm_f32 for M, N % 256 == 0, K % 32 == 0, split-K slabs of multiples of 32): against
    fp64, with the reduce epilogue (accumulate into C), with batch strides, plain and in queue mode under every kind of
    XCD mask -- bit-identical whichever workgroup computes which item.  precision 2: the same shapes on the six-product kernel
    of gemm_x6.hip (three planes per operand, 16-deep steps) at fp32-GEMM accuracy."""
    from policy_gradient_asr_amd import hipops
    monkeypatch.setenv("PGASR_TN_TILE", "256")          # opt-in kernel (read by the library at every call)
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(batch, K, M, generator=g).to(DEV)
    B = torch.randn(batch, K, N, generator=g).to(DEV)
    base = torch.randn(batch, M, N, generator=g).to(DEV)
    ref = base.double().cpu() + torch.einsum("bkm,bkn->bmn", A.double().cpu(), B.double().cpu())
    outs = []
    for mask in (None, 0x00, 0x0F, 0xFE, 0xFF):
        busy = None if mask is None else torch.tensor([(mask >> i) & 1 for i in range(8)], dtype=torch.int32, device=DEV)
        got = base.clone()
        hipops.gemm(A, B, got, M, N, K, transA=True, splitk=splitk, accumulate=True, precision=precision, batch=batch,
                    strideA=K * M, strideB=K * N, strideC=M * N, xcc_busy=None if busy is None else busy.data_ptr())
        torch.cuda.synchronize()
        outs.append(got)
    assert rel_err(outs[0].cpu(), ref) < tol
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # sum_batches: the two batches' slabs summed into ONE output
    if batch > 1:
        one = torch.zeros(M, N, device=DEV)
        hipops.gemm(A, B, one, M, N, K, transA=True, splitk=splitk, precision=precision, batch=batch, sum_batches=True,
                    strideA=K * M, strideB=K * N, strideC=0)
        assert rel_err(one.cpu(), (ref - base.double().cpu()).sum(0)) < tol
    if precision == 2:          # shapes the TN kernel does not take have no six-product form: the call says so
        with pytest.raises(Exception):
            hipops.gemm(A, B, torch.empty(M, N, device=DEV), M, N + 4, K, transA=True, splitk=splitk, precision=2)


def test_instnorm_affine_fwd_bwd():
    from policy_gradient_asr_amd import functional as Fh
    B, F, T, N = 3, 80, 50, 512
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, F, T, generator=g) * 3 + 1.5
    W = torch.randn(N, F, generator=g) * 0.1; b = torch.randn(N, generator=g) * 0.1
    Wc = W.clone().requires_grad_(True); bc = b.clone().requires_grad_(True)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.linear(model_ref.instance_norm(x).transpose(1, 2), Wc, bc))
    dy = torch.randn(T, B, N, generator=g)
    ref.transpose(0, 1).backward(dy)
    Wg = W.to(DEV).requires_grad_(True); bg = b.to(DEV).requires_grad_(True)
    y = Fh.InstNormAffineFn.apply(x.to(DEV), Wg, bg)
    y.backward(dy.to(DEV))
    assert rel_err(y.detach().cpu(), ref.detach().transpose(0, 1)) < 1e-5
    assert rel_err(Wg.grad.cpu(), Wc.grad) < 1e-4
    assert rel_err(bg.grad.cpu(), bc.grad) < 1e-4


@pytest.mark.parametrize("B,F,T", [(3, 80, 50), (2, 7, 33), (4, 80, 1000), (1, 120, 4099)])
def test_instnorm_stats_vector_and_scalar_paths(B, F, T):
    """mean / rstd over all F*T values of an utterance (model.py:37,48): 16-byte-load path (F*T % 4 == 0), the
    scalar path, and a count that leaves a remainder after the four-deep unrolled loop."""
    from policy_gradient_asr_amd import hipops
    g = torch.Generator().manual_seed(B * F + T)
    x = torch.randn(B, F, T, generator=g) * 2.5 - 0.7
    mean, rstd = hipops.instnorm_stats(x.to(DEV), 1e-5)
    xd = x.double().reshape(B, -1)
    want_mean = xd.mean(dim=1)
    want_rstd = 1.0 / torch.sqrt(xd.var(dim=1, unbiased=False) + 1e-5)
    assert rel_err(mean.cpu(), want_mean) < 1e-6
    assert rel_err(rstd.cpu(), want_rstd) < 1e-6


def _lstm_case(T, B, lens, seed, in_dim=512):
    g = torch.Generator().manual_seed(seed)
    lstm = torch.nn.LSTM(in_dim, 256, 1, bidirectional=True)
    x = torch.randn(T, B, in_dim, generator=g)
    dy = torch.randn(T, B, 512, generator=g)
    lengths = torch.tensor(lens, dtype=torch.int64)
    for b, n in enumerate(lens):
        dy[n:, b] = 0  # padded outputs carry no gradient in the packed reference either
    return lstm, x, dy, lengths


@pytest.mark.parametrize("T,B,lens", [
    (5, 2, [5, 5]), (40, 4, [40, 33, 1, 17]), (64, 16, [64] * 16), (30, 19, list(range(30, 11, -1))),
    (200, 32, [200] * 32),
    (1, 5, [1] * 5),                                # a single frame: no recurrent exchange at all
    (3, 17, [3] * 9 + [1] * 8),                     # one utterance over a 16-utterance group: a second, nearly empty cluster pair
    (7, 48, [7] * 20 + [4] * 28),                   # three groups = six clusters
    (5, 80, [5] * 50 + [2] * 30),                   # ten clusters: two share an XCD, no room for helper workgroups
    (6, 128, [6] * 70 + [3] * 40 + [1] * 18),       # the compiled-in limit B = 128: sixteen clusters, two per XCD
])
def test_blstm_layer_vs_torch_cpu(T, B, lens):
    from policy_gradient_asr_amd import functional as Fh
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    lstm, x, dy, lengths = _lstm_case(T, B, lens, seed=T + B)
    xr = x.clone().requires_grad_(True)
    pk = pack_padded_sequence(xr, lengths, enforce_sorted=False)
    out, _ = lstm(pk)
    out, _ = pad_packed_sequence(out, total_length=T)
    out.backward(dy)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
             "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    params = [getattr(lstm, n).detach().to(DEV).requires_grad_(True) for n in names]
    xg = x.to(DEV).requires_grad_(True)
    y = Fh.blstm_layer(xg, lengths.to(torch.int32).to(DEV), params)
    y.backward(dy.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(y.detach().cpu(), out.detach()) < 1e-4
    for b, n in enumerate(lens):
        assert torch.all(y[n:, b] == 0)
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-3
    for n, p in zip(names, params):
        assert rel_err(p.grad.cpu(), getattr(lstm, n).grad) < 1e-3, n


LSTM_NAMES = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
              "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]


@pytest.mark.parametrize("T,B,lens", [
    (40, 4, [40, 33, 1, 17]), (30, 19, list(range(30, 11, -1))), (200, 32, [200] * 32),
    (5, 80, [5] * 50 + [2] * 30),                   # ten clusters: no helper workgroups
    (1000, 16, [1000] * 8 + list(range(993, 500, -64))),     # the headline's chain length
    (1000, 32, [1000] * 32),                                 # the headline's shape: the B = 32 fed forward sweep (four clusters)
])
def test_blstm_layer_f32_mode_vs_torch_cpu(T, B, lens):
    """precision mode "f32" -- the reference's arithmetic (nn.LSTM in torch fp32, model.py:39-44): exact fp32 MFMA for
    the hoisted products, 3-plane / 6-product sweeps.  Outputs and every gradient within 1e-5 (max norm) of torch-CPU
    run in fp64 on the same fp32 parameters, i.e. at the level of torch's own fp32 rounding; the opt-in bf16x3 mode is
    measured beside it on the same case and must be the less exact of the two."""
    from policy_gradient_asr_amd import functional as Fh, hipops
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    lstm, x, dy, lengths = _lstm_case(T, B, lens, seed=T + B + 1)
    l64 = torch.nn.LSTM(512, 256, 1, bidirectional=True).double()
    l64.load_state_dict({k: v.double() for k, v in lstm.state_dict().items()})
    xr = x.double().requires_grad_(True)
    if T >= 500:
        # the packed semantics without torch's PackedSequence path (minutes per case at T = 1000 in fp64): every utterance reversed
        # within its own length, oracle/model_ref.py -- pinned against the PackedSequence path by tests/test_oracle_cpu.py
        from oracle import model_ref
        out = model_ref.blstm_layer_packed_equivalent(xr.transpose(0, 1), lengths, [getattr(l64, n) for n in LSTM_NAMES]).transpose(0, 1)
    else:
        out, _ = l64(pack_padded_sequence(xr, lengths, enforce_sorted=False))
        out, _ = pad_packed_sequence(out, total_length=T)
    out.backward(dy.double())
    errs = {}
    for mode in ("f32", "bf16x3"):
        with hipops.precision(mode):
            params = [getattr(lstm, n).detach().to(DEV).requires_grad_(True) for n in LSTM_NAMES]
            xg = x.to(DEV).requires_grad_(True)
            y = Fh.blstm_layer(xg, lengths.to(torch.int32).to(DEV), params)
            y.backward(dy.to(DEV))
            torch.cuda.synchronize()
        hipops.lstm_assert_no_timeouts()
        e = {"out": rel_err(y.detach().cpu(), out.detach()), "dx": rel_err(xg.grad.cpu(), xr.grad)}
        for n, p in zip(LSTM_NAMES, params):
            e[n] = rel_err(p.grad.cpu(), getattr(l64, n).grad)
        errs[mode] = e
        for b, n in enumerate(lens):
            assert torch.all(y[n:, b] == 0)
    worst = {m: max(e.values()) for m, e in errs.items()}
    print(f"[precision] T={T} B={B}: worst rel err vs fp64  f32 mode {worst['f32']:.2e} (out {errs['f32']['out']:.2e})   "
          f"bf16x3 mode {worst['bf16x3']:.2e} (out {errs['bf16x3']['out']:.2e})")
    assert worst["f32"] < 1e-5, errs["f32"]
    assert worst["bf16x3"] < 1e-3 and errs["f32"]["out"] < errs["bf16x3"]["out"]


def test_blstm_write_through_protocol_matches_default():
    """flags bit 0 forces the cross-XCD (write-through) hand-off; results must be identical to the
    default (placement-selected) protocol bit for bit."""
    from policy_gradient_asr_amd import functional as Fh, hipops
    T, B, lens = 50, 20, [50] * 10 + list(range(40, 30, -1))
    lstm, x, dy, lengths = _lstm_case(T, B, lens, seed=5)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
             "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    res = []
    for flags in (0, 1):
        hipops.LSTM_FLAGS = flags
        try:
            params = [getattr(lstm, n).detach().to(DEV).requires_grad_(True) for n in names]
            xg = x.to(DEV).requires_grad_(True)
            y = Fh.blstm_layer(xg, lengths.to(torch.int32).to(DEV), params)
            y.backward(dy.to(DEV))
            torch.cuda.synchronize()
            res.append([y.detach().cpu(), xg.grad.cpu()] + [p.grad.cpu() for p in params])
        finally:
            hipops.LSTM_FLAGS = 0
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("T,B,lens", [
    (200, 32, [200] * 32),                          # 8 time steps per 256-row tile: the headline geometry
    (61, 20, [61] * 7 + list(range(60, 47, -1))),   # steps straddle row tiles, ragged last tile, second group of 4
    (1, 3, [1] * 3), (9, 16, [9] * 16),
])
@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_blstm_fed_by_concurrent_projection_matches_sequential(T, B, lens, mode):
    """The forward sweep fed by a projection GEMM that runs beside it (functional.FEED_AHEAD: sweep launched first,
    helper workgroups wait per row tile) gives bit-identical outputs, saved activations and gradients to the
    projection-then-sweep order; B > 32 falls back to the sequential order by itself.  mode "f32": the same with the
    six-product feed kernel (gemm_x6.hip) and the 3-plane sweeps."""
    from policy_gradient_asr_amd import functional as Fh, hipops
    assert hipops.lstm_fed_ok(T, B) and not hipops.lstm_fed_ok(T, 33)
    if not hipops.streams_concurrent(Fh.grad_overlap.second_side_stream()):
        pytest.skip("kernels of different streams are serialised here (profiler / launch-blocking): feed-ahead is off by itself")
    lstm, x, dy, lengths = _lstm_case(T, B, lens, seed=7 + T)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
             "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    res = []
    prev = Fh.FEED_AHEAD
    try:
        for feed in (False, True, True):
            Fh.FEED_AHEAD = feed
            params = [getattr(lstm, n).detach().to(DEV).requires_grad_(True) for n in names]
            xg = x.to(DEV).requires_grad_(True)
            with hipops.precision(mode):
                y = Fh.blstm_layer(xg, lengths.to(torch.int32).to(DEV), params)
                y.backward(dy.to(DEV))
            torch.cuda.synchronize()
            hipops.lstm_check_error(hipops._lstm_ws(T, B, False, xg.device), B, False)
            res.append([y.detach().cpu(), xg.grad.cpu()] + [p.grad.cpu() for p in params])
    finally:
        Fh.FEED_AHEAD = prev
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("T,B,lens,flags", [
    (200, 32, [200] * 32, 0),                               # the headline geometry
    (1000, 32, [1000] * 20 + list(range(999, 987, -1)), 0),   # the headline length: 11 slabs
    (130, 32, [130] * 9 + list(range(129, 106, -1)), 0),    # very ragged
    (20, 32, [20] * 32, 0),                                 # a single slab
    (2, 32, [2] * 20 + [1] * 12, 0),                        # the shortest sweep the products take: dW_hh sums ONE frame pair
    (997, 32, [997] * 5 + list(range(996, 969, -1)), 0),    # a length that no slab size divides (configs[4]'s ragged batches)
    (200, 32, [200] * 32, 1),                               # write-through protocol: the storer waves release their own stores
])
@pytest.mark.parametrize("mode,tol", [("bf16x3", 2e-5), ("f32", 2e-6)])
def test_streamed_backward_sweep_feeds_its_own_weight_gradients(T, B, lens, flags, mode, tol):
    """pgasr_lstm_layer_bwd_streamed + pgasr_lstm_wgrads_streamed: the weight-gradient products of a layer run BESIDE the backward
    sweep that is still writing their dgates operand (flusher workgroup: L2 write-back per time slab, slab_done words; gated TN
    kernel: waits per slab, agent-scope loads).  Bit-identical to the sequential order (sweep, then the same launch un-gated), and
    equal to the fp64 product of the same dgates to bf16x3 accuracy -- mode "f32": to fp32-GEMM accuracy, on the six-product
    gated kernel (gemm_x6.hip) behind a 3-plane sweep."""
    from policy_gradient_asr_amd import functional as Fh, hipops, streams
    dev = torch.device(DEV)
    with hipops.precision(mode):
        _streamed_case(T, B, lens, flags, tol, dev)


@pytest.mark.parametrize("T,B,lens", [
    (200, 16, [200] * 16),                                  # one cluster per direction: half of the headline's batch per GPU
    (1000, 16, [1000] * 9 + list(range(999, 992, -1))),     # the headline length
    (131, 16, [131] * 3 + list(range(130, 117, -1))),       # ragged, a length no slab size divides
])
def test_streamed_backward_sweep_with_sixteen_utterances(T, B, lens):
    """B = 16 (round 5): the time slabs of the weight-gradient sums are whole 16-row steps of the six-product TN kernel, so the
    "f32" mode streams them beside the sweep at B % 16 == 0 (the bf16x3 kernel steps 32 rows and keeps B % 32 == 0).  Same checks
    as above: bit-identical to the sequential order, fp32-GEMM accuracy against fp64."""
    from policy_gradient_asr_amd import hipops
    with hipops.precision("f32"):
        _streamed_case(T, B, lens, 0, 2e-6, torch.device(DEV))


def _streamed_case(T, B, lens, flags, tol, dev):
    from policy_gradient_asr_amd import functional as Fh, hipops, streams
    side = streams.side_stream("test_streamed")
    if not hipops.streams_concurrent(side):
        pytest.skip("kernels of different streams are serialised here (profiler / launch-blocking)")
    assert hipops.lstm_wgrads_ok(T, B, 512) and not hipops.lstm_wgrads_ok(T, 20, 512)
    assert hipops.lstm_wgrads_ok(T, 16, 512, 3) and not hipops.lstm_wgrads_ok(T, 16, 512, 2)      # 16-row steps: the six-product kernel only
    edges = hipops.lstm_wgrad_slabs(T)
    assert edges[0] == 0 and edges[-1] == T and all(a < b for a, b in zip(edges, edges[1:]))
    if T == 1000:
        assert edges == [0, 16, 40, 72, 112, 160, 224, 304, 408, 536, 664, 792, 920, 1000]
    lstm, x, dy, lengths = _lstm_case(T, B, lens, seed=11 + T)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
             "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    params = [getattr(lstm, n).detach().to(dev).contiguous() for n in names]
    wih, bias, pf, pb = hipops.lstm_pack(params, 512)
    xd, dyd, ln = x.to(dev), dy.to(dev), lengths.to(torch.int32).to(dev)
    G, H = 2048, 256
    gates0 = torch.empty(T, B, G, device=dev)
    hipops.gemm(xd, wih, gates0, M=T * B, N=G, K=512, transB=True, bias=bias)
    out = torch.empty(T, B, 2 * H, device=dev); cbuf = torch.empty(T, B, 2 * H, device=dev)
    hipops.lstm_layer_fwd(gates0, out, cbuf, pf, ln, T, B)           # gates0 := saved activations
    prev_flags = hipops.LSTM_FLAGS
    hipops.LSTM_FLAGS = flags
    try:
        # sequential order
        dg_ref = gates0.clone()
        hipops.lstm_layer_bwd(dg_ref, out, cbuf, dyd, pb, ln, T, B)
        dwih_ref = torch.empty(G, 512, device=dev); dwhh_ref = torch.empty(2, 4 * H, H, device=dev)
        hipops.lstm_wgrads(dg_ref, xd, out, T, B, 512, dwih_ref, dwhh_ref)
        d64, x64, o64 = dg_ref.double().cpu(), xd.double().cpu(), out.double().cpu()
        want_ih = d64.reshape(T * B, G).t() @ x64.reshape(T * B, 512)
        assert rel_err(dwih_ref.cpu(), want_ih) < tol
        want_hh0 = d64[1:, :, :4 * H].reshape(-1, 4 * H).t() @ o64[:-1, :, :H].reshape(-1, H)
        want_hh1 = d64[:-1, :, 4 * H:].reshape(-1, 4 * H).t() @ o64[1:, :, H:].reshape(-1, H)
        assert rel_err(dwhh_ref[0].cpu(), want_hh0) < tol and rel_err(dwhh_ref[1].cpu(), want_hh1) < tol
        for rep in range(3):
            dg = gates0.clone()
            words = torch.zeros(64, dtype=torch.int32, device=dev)
            dwih = torch.full((G, 512), float("nan"), device=dev); dwhh = torch.full((2, 4 * H, H), float("nan"), device=dev)
            report = torch.zeros(2, dtype=torch.int32, device=dev)       # zeroed BEFORE the event the side stream waits for
            torch.cuda.synchronize()
            before = torch.cuda.Event(); before.record()
            ws = hipops.lstm_layer_bwd(dg, out, cbuf, dyd, pb, ln, T, B, slab=words)
            busy = hipops.lstm_busy_ptr(T, B, True, dev)
            if rep == 2:
                torch.cuda.synchronize()        # a consumer that comes LATE: the sweep is over, its busy counters are back to zero --
                                                # the gate must open on the publications, not sit out its (here 100 ms, the entry point's maximum) time-out
            with torch.cuda.stream(side):
                side.wait_event(before)
                hipops.stream_gate(busy, need=2 * ((B + 15) // 16), timeout_us=100000 if rep == 2 else 5000, running=words, report=report)
                hipops.lstm_wgrads(dg, xd, out, T, B, 512, dwih, dwhh, busy_ptr=busy, slab=words, err_ws=ws)
            torch.cuda.synchronize()
            how, held_us = report.tolist()
            # the gate's own word, not a host clock: it never left by time-out, and the late consumer's left on the publication
            assert how in (hipops.GATE_OPENED_ON_BUSY, hipops.GATE_OPENED_ON_PUBLICATION), (rep, how, held_us)
            if rep == 2:
                assert how == hipops.GATE_OPENED_ON_PUBLICATION and held_us < 1000, (how, held_us)
            hipops.lstm_check_error(ws, B, True)
            nc = 2 * ((B + 15) // 16)
            assert words[:nc].tolist() == [len(edges) - 1] * nc
            assert torch.equal(dg, dg_ref)
            assert torch.equal(dwih, dwih_ref), float((dwih - dwih_ref).abs().max())
            assert torch.equal(dwhh, dwhh_ref), float((dwhh - dwhh_ref).abs().max())
    finally:
        hipops.LSTM_FLAGS = prev_flags


def test_sweep_error_word_is_sticky_and_checked():
    """A sweep that gives up on a bounded wait sets the first word of its workspace; no later launch clears it and
    hipops.lstm_assert_no_timeouts() (called by bench.py and model.train) raises.  The flag is forged here."""
    from policy_gradient_asr_amd import functional as Fh, hipops, _lib
    T, B, lens = 6, 4, [6, 6, 5, 3]
    lstm, x, dy, lengths = _lstm_case(T, B, lens, seed=3)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
             "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    params = [getattr(lstm, n).detach().to(DEV) for n in names]
    run = lambda: Fh.blstm_layer(x.to(DEV), lengths.to(torch.int32).to(DEV), params)
    run()
    hipops.lstm_assert_no_timeouts()
    ws = hipops._lstm_ws(T, B, False, torch.device(DEV))
    ws[:4].view(torch.int32).fill_(1)
    try:
        run()                                        # a later launch must not wipe the flag
        with pytest.raises(_lib.PgasrError):
            hipops.lstm_assert_no_timeouts()
    finally:
        ws[:4].zero_()
    hipops.lstm_assert_no_timeouts()


def test_encoder_matches_reference_golden(golden_dir):
    """Encoder on the MI355X vs the reference model.Encoder outputs (tests/golden)."""
    from policy_gradient_asr_amd.model import Encoder
    z = np.load(os.path.join(golden_dir, "encoder_cases.npz"))
    p = model_ref.init_params(n_feats=120, vocab=29, seed=0)
    enc = Encoder()
    enc.load_state_dict({k: v for k, v in p.items() if not k.startswith("head.")}, strict=True)
    assert list(enc.state_dict().keys()) == list(z["names"])
    enc = enc.to(DEV).eval()
    for cid in range(3):
        x = torch.from_numpy(z[f"x{cid}"]).to(DEV); mask = torch.from_numpy(z[f"mask{cid}"]).to(DEV)
        with torch.no_grad():
            y = enc(x, mask)
        assert y.shape == z[f"y{cid}"].shape
        assert rel_err(y.cpu(), z[f"y{cid}"]) < 1e-3
        lens = mask.sum(1).int().tolist()
        for b, n in enumerate(lens):
            assert torch.all(y[b, n:] == 0)


def test_blstm_fused_output_dropout_equals_separate_dropout_pass():
    """nn.LSTM's inter-layer dropout (model.py:42) written by the sweep's storer waves: the returned tensor is bit for bit
    pgasr_dropout(output) with the same (p, seed, offset), and so are the gradients of the two formulations."""
    from policy_gradient_asr_amd import functional as Fh, hipops
    T, B = 37, 19
    lens = [37] * 10 + [20] * 5 + [1] * 4
    lstm, x, dy, lengths = _lstm_case(T, B, lens, seed=5)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
             "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    cfg = (0.3, 0x5EED, 7)
    res = []
    for fused in (False, True):
        params = [getattr(lstm, n).detach().to(DEV).requires_grad_(True) for n in names]
        xg = x.to(DEV).requires_grad_(True)
        if fused:
            y = Fh.blstm_layer(xg, lengths.to(torch.int32).to(DEV), params, out_dropout=cfg)
        else:
            y = Fh.DropoutFn.apply(Fh.blstm_layer(xg, lengths.to(torch.int32).to(DEV), params), *cfg)
        y.backward(dy.to(DEV))
        torch.cuda.synchronize()
        res.append([y.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in params])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    keep = res[0][0] != 0
    assert 0.6 < float(keep[:1].float().mean()) < 0.8           # p = 0.3 on the full-length frames


def test_seq2seq_logprobs_and_grads_vs_oracle():
    from policy_gradient_asr_amd.model import Seq2Seq
    B, F, T, V = 4, 80, 60, 29
    p = model_ref.init_params(n_feats=F, vocab=V, seed=3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, F, T, generator=g)
    lens = [60, 41, 60, 25]
    mask = torch.zeros(B, T)
    for b, n in enumerate(lens):
        mask[b, :n] = 1; x[b, :, n:] = 0
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    enc = model_ref.encoder_forward_torch(pr, x, mask)
    lp_ref = model_ref.head_forward_torch(pr, enc)
    w = torch.randn(T, B, V, generator=g) * mask.t()[:, :, None]
    (lp_ref * w).sum().backward()
    m = Seq2Seq(V, n_feats=F)
    sd = {("encoder." + k if not k.startswith("head.") else k): v for k, v in p.items()}
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    lp = m(x.to(DEV), None, mask.to(DEV), DEV)
    (lp * w.to(DEV)).sum().backward()
    assert rel_err(lp.detach().cpu(), lp_ref.detach()) < 1e-3
    for k, v in m.named_parameters():
        rk = k[len("encoder."):] if k.startswith("encoder.") else k
        assert rel_err(v.grad.cpu(), pr[rk].grad) < 1e-3, k       # north_star tolerance


def test_stream_gate_opens_on_busy_word_or_timeout():
    """pgasr_stream_gate: a hint kernel -- returns at once when a busy word is set, after the time-out otherwise."""
    import time
    from policy_gradient_asr_amd import hipops
    words = torch.zeros(8, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); hipops.stream_gate(words.data_ptr(), 8, timeout_us=20000); torch.cuda.synchronize()
    waited = time.perf_counter() - t0
    assert 0.015 < waited < 0.5                       # ~20 ms time-out, never a hang
    words[5] = 1
    torch.cuda.synchronize()
    t0 = time.perf_counter(); hipops.stream_gate(words.data_ptr(), 8, timeout_us=20000); torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.01
    with pytest.raises(Exception):
        hipops.stream_gate(words.data_ptr(), 8, timeout_us=10 ** 7)       # beyond the documented 100 ms cap
