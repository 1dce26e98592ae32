"""torch.autograd glue over the HIP kernels.  Activations are time-major (T,B,C).
No function here has a CPU path: tensors must be on the MI355X."""
import os

import torch

from . import hipops
from . import streams

HID = hipops.HID
FEED_AHEAD = os.environ.get("PGASR_FEED_AHEAD", "1") != "0"   # input projections run beside the forward sweeps they feed
JOIN_FEED = os.environ.get("PGASR_JOIN_FEED", "0") == "1"   # A/B: main stream joins the feed stream after every fed forward sweep
FEED_BWD = os.environ.get("PGASR_FEED_BWD", "1") != "0"       # .. and the upper layers' input-gradient GEMMs beside the backward sweeps
STREAM_DW = os.environ.get("PGASR_STREAM_DW", "1") != "0"     # a layer's weight-gradient GEMMs run beside ITS OWN backward sweep (streamed sweep)
LEAKY_SLOPE = 0.01   # F.leaky_relu default, model.py:50


def _pick_splitk(M, N, K, target_wgs=512, batch=1, beside_sweep=False):
    """Split-K factor of a weight-gradient product.  Shapes the 256 x 256 TN kernel takes (gemm_c256.hip: M, N % 256 == 0,
    K % 32 == 0, one workgroup per CU) get one work item per available CU: 128 beside a sweep (it leaves four XCDs
    free), 256 in the tail -- measured round 3 on dW_ih (2048 x 512 x 32000): 128 items 345 us on half the chip (the
    128 x 128 kernel: ~510 us there), 256 items 224 us on the whole chip (276 us).  Other shapes: 128 x 128 tiles, 2 workgroups
    per CU (measured on dW_ih: 4 -> 508 us, 8 -> 326 us).  A function of the layer's shapes only, never of the overlap
    mode, so that every mode gives the same bits."""
    if M % 256 == 0 and N % 256 == 0 and K % 32 == 0 and K >= 64:
        tiles = (M // 256) * (N // 256) * batch
        sk = max(1, min((128 if beside_sweep else 256) // tiles, K // 256))
        while sk > 1 and (sk - 1) * (-(-K // sk // 32) * 32) >= K:     # no empty slab once slabs are rounded up to 32
            sk -= 1
        return sk
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    sk = max(1, min(64, target_wgs // max(tiles, 1)))
    while sk > 1 and K // sk < 256:
        sk //= 2
    return max(sk, 1)


class grad_overlap:
    """Opt-in (used by train_step): run the weight-gradient GEMMs of a BLSTM layer on a side stream,
    concurrently with the next layer's backward sweep (which only needs dX), and accumulate them
    straight into the parameters' pre-allocated .grad buffers.  ``finish()`` makes the current
    stream wait for the side stream; call it after loss.backward()."""
    enabled = False
    confine = os.environ.get("PGASR_CONFINE", "1") != "0"     # keep side-stream GEMMs off the XCDs of the concurrent LSTM sweep
    confine_feed = os.environ.get("PGASR_CONFINE_FEED", "1") != "0"   # .. and the GEMMs that FEED a sweep (A/B switch)
    _sides = {}        # one side stream and one pending list PER main stream (micro-batches run on their own streams)
    _pendings = {}     # (ready event, closure) of the layer above, issued right AFTER the next sweep is launched

    @classmethod
    def _key(cls):
        return torch.cuda.current_stream().cuda_stream

    _main_of = {}      # side stream handle -> the main stream it serves

    @classmethod
    def side_stream(cls):
        k = cls._key()
        if k not in cls._sides:
            cls._sides[k] = streams.side_stream("weight_gradients")
            cls._main_of[cls._sides[k].cuda_stream] = k
        return cls._sides[k]

    upper_grads_hook = None   # callable(swept_event): called by the FIRST layer's backward once every gradient above it is issued
    tail_dx_first = os.environ.get("PGASR_TAIL_DX_FIRST", "1") != "0"
    split_tail = True  # no sweep follows the first layer's backward: run its dW_hh beside its dW_ih
    _sides2 = {}

    @classmethod
    def second_side_stream(cls):
        k = cls._key()
        if k not in cls._sides2:
            cls._sides2[k] = streams.side_stream("feed")
        return cls._sides2[k]

    _sides3 = {}
    _streamed_unjoined = {}   # main stream -> True while the streamed-weight-gradient stream carries work of this step

    @classmethod
    def third_side_stream(cls):
        """The stream of the weight-gradient launches that consume a backward sweep WHILE it runs (BLSTMLayerFn.backward)."""
        k = cls._key()
        if k not in cls._sides3:
            cls._sides3[k] = streams.side_stream("streamed_weight_gradients")
        return cls._sides3[k]

    _recent = {}       # the last event recorded on the main stream in the running backward pass (per main stream)
    _feed_unjoined = {}   # main stream -> True while the feed stream carries work of this step that nothing has joined yet

    @classmethod
    def note_event(cls, ev):
        cls._recent[cls._key()] = ev

    @classmethod
    def take_event(cls):
        return cls._recent.pop(cls._key(), None)

    @classmethod
    def pending(cls):
        return cls._pendings.setdefault(cls._key(), [])

    # input gradients that are produced LATE (BLSTMLayerFn.backward): data_ptr of the placeholder tensor -> record
    _deferred = {}

    @classmethod
    def flush(cls, busy_ptr=0, launched_after=None, first=None):
        """Issue the deferred weight-gradient GEMMs on the side stream.  Called right after a sweep kernel
        has been launched: a kernel with a large grid enqueued BEFORE the sweep holds up the dispatch of
        everything behind it, on any stream (rocprof: the sweep's 5 us memset waited 616 us for the GEMM
        in front of it), so the order of enqueueing is sweep first, GEMMs second."""
        pending = cls.pending()
        if not pending and first is None:
            return
        side = cls.side_stream()
        prev = hipops.GEMM_XCC_BUSY_PTR
        hipops.GEMM_XCC_BUSY_PTR = busy_ptr if cls.confine else 0
        try:
            with torch.cuda.stream(side):
                for ready, fn in pending:
                    side.wait_event(ready)
                if launched_after is not None and busy_ptr:
                    # enqueue order says nothing about dispatch order when the host runs ahead: wait for the point
                    # on the main stream just before the sweep, then for the sweep's clusters to register
                    side.wait_event(launched_after)
                    hipops.stream_gate(busy_ptr)
                if first is not None:
                    first()          # the GEMM that feeds the sweep just launched goes ahead of the weight gradients
                for ready, fn in pending:
                    fn()
        finally:
            hipops.GEMM_XCC_BUSY_PTR = prev
            del pending[:]

    @classmethod
    def finish(cls):
        if cls._deferred:
            cls._deferred.clear()
            raise RuntimeError("a deferred input gradient was never consumed by a BLSTM layer (grad_overlap/FEED_AHEAD)")
        cls.flush()
        cls._recent.pop(cls._key(), None)
        if cls._key() in cls._sides:
            torch.cuda.current_stream().wait_stream(cls.side_stream())
        if cls._streamed_unjoined.pop(cls._key(), False) and cls._key() in cls._sides3:
            torch.cuda.current_stream().wait_stream(cls._sides3[cls._key()])
        if cls._feed_unjoined.pop(cls._key(), False) and cls._key() in cls._sides2:
            # ONE join of the feed stream per step, and only when the tail of backward has not made it already (the first
            # layer's dW_hh runs on that stream and is joined through the weight-gradient stream): a fed forward sweep is
            # not followed by a join of its own, which is safe only while the sweep really waits for every tile -- after a
            # sweep time-out (survivable since round 2) or with overlap_weight_grads off the feed GEMMs could still be
            # touching buffers that the next step's allocations reuse
            torch.cuda.current_stream().wait_stream(cls.second_side_stream())


# The input layer's weight gradient is B batches of a (512 x F x T) product: at B = 32 that is 128 work items of 128 x 128 on 256 CUs, each
# a thousand frames deep, and it is the last dependent GEMM of the step's tail.  Split-K 2 fills the chip: 166 -> 99 us at the headline shape
# (4: 98, 8: 133; tools/dev/r5_affine_dw.py, round 5).  One value for every order of the step, so they keep giving the same bits.
AFFINE_DW_SPLITK = 2       # for T >= 256 frames


class InstNormAffineFn(torch.autograd.Function):
    """model.py:48-50: InstanceNorm2d over the (F,T) plane -> Linear(F,512) -> leaky_relu.
    x (B,F,T) -> y (T,B,N).  The normalisation is applied while the GEMM loads its tile."""

    @staticmethod
    def forward(ctx, x, weight, bias, consumer_applies_dact=False):
        """consumer_applies_dact: the op consuming y multiplies its input gradient by
        leaky'(y) itself (BLSTMLayerFn's dact_y epilogue), so backward must not repeat it."""
        B, F, T = x.shape
        N = weight.shape[0]
        x = x.contiguous(); weight = weight.contiguous(); bias = bias.contiguous()
        ctx.consumer_applies_dact = bool(consumer_applies_dact)
        mean, rstd = hipops.instnorm_stats(x, 1e-5)
        y = torch.empty(T, B, N, dtype=torch.float32, device=x.device)
        # exact fp32 MFMA (precision 0), not the bf16x3 split: leaky_relu' is a step function of this product's SIGN, and a
        # pre-activation that is zero to 1e-5 relative (the split's error) flips it -- measured at the headline shape:
        # ~160 of 16 M pre-activations changed side against the fp32 CPU path, each one scaling a summand of the
        # input layer's weight/bias gradient by 100 (1e-2 max-norm error on those two tensors under a REINFORCE
        # gradient, 6e-4 under CTC alone).  K = F is tiny: the exact product costs ~10 us more per step.
        hipops.gemm(x, weight, y, M=T, N=N, K=F, transA=True, transB=True, lda=T, ldb=F, ldc=B * N,
                    strideA=F * T, strideB=0, strideC=N, batch=B, bias=bias, act=1, slope=LEAKY_SLOPE,
                    norm_operand=1, shift=mean, scale=rstd, precision=0)
        ctx.save_for_backward(x, weight, mean, rstd, y)
        ctx.bias_ref, ctx.weight_ref = bias, weight
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd, y = ctx.saved_tensors
        B, F, T = x.shape
        N = weight.shape[0]
        # d(pre-activation) = dy * leaky'(y)   (sign(y) == sign(pre))
        if ctx.consumer_applies_dact:
            dpre = dy.contiguous()
        else:
            dpre = torch.where(y > 0, dy, dy * LEAKY_SLOPE).contiguous()
        bg = getattr(ctx.bias_ref, "grad", None)
        if grad_overlap.enabled and bg is not None and bg.is_contiguous() and bg.dtype == torch.float32:
            # the tail of the step is ONE dependent chain by now: the bias gradient (two column-sum kernels, ~50 us) leaves it --
            # accumulated straight into bias.grad on the side stream, beside the weight-gradient GEMM (joined by grad_overlap.finish)
            main = torch.cuda.current_stream()
            side = grad_overlap.side_stream()
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                hipops.colsum(dpre, T * B, N, N, bg, accumulate=True)
            streams.hold(dpre, side)
            wg = getattr(ctx.weight_ref, "grad", None)
            direct = wg is not None and wg.is_contiguous() and wg.dtype == torch.float32 and tuple(wg.shape) == (N, F)
            dW = wg if direct else torch.empty(N, F, dtype=torch.float32, device=x.device)
            # .. and the weight gradient is summed into weight.grad by the GEMM's own slab reduction (one launch less on the chain)
            hipops.gemm(dpre, x, dW, M=N, N=F, K=T, transA=True, transB=True, lda=B * N, ldb=T, ldc=F,
                        strideA=N, strideB=F * T, strideC=0, batch=B, sum_batches=True, norm_operand=2,
                        shift=mean, scale=rstd, accumulate=direct, splitk=AFFINE_DW_SPLITK if T >= 256 else 1)
            return (None, None if direct else dW, None, None)
        return (None, *InstNormAffineFn.param_grads(x, mean, rstd, dpre, N), None)

    @staticmethod
    def param_grads(x, mean, rstd, dpre, N):
        B, F, T = x.shape
        dW = torch.empty(N, F, dtype=torch.float32, device=x.device)
        hipops.gemm(dpre, x, dW, M=N, N=F, K=T, transA=True, transB=True, lda=B * N, ldb=T, ldc=F,
                    strideA=N, strideB=F * T, strideC=0, batch=B, sum_batches=True, norm_operand=2,
                    shift=mean, scale=rstd, splitk=AFFINE_DW_SPLITK if T >= 256 else 1)
        db = torch.empty(N, dtype=torch.float32, device=x.device)
        hipops.colsum(dpre, T * B, N, N, db)
        return dW, db


class DropoutFn(torch.autograd.Function):
    """Inverted dropout whose mask is a pure function of (seed, offset, index): backward re-applies
    the same kernel call to the gradient (no stored mask)."""

    @staticmethod
    def forward(ctx, x, p, seed, offset, *opt):
        """fuse_leaky_backward: x is the OUTPUT of a leaky_relu whose producer leaves the leaky' factor to its
        consumer (InstNormAffineFn(consumer_applies_dact=True)); backward then applies mask and leaky'(x) in one pass."""
        ctx.cfg = (float(p), int(seed), int(offset))
        x = x.contiguous()
        ctx.fused = bool(opt[0]) if opt else False          # opt = (fuse_leaky_backward,)
        ctx.nopt = len(opt)
        if ctx.fused:
            ctx.save_for_backward(x)
        return hipops.dropout(x, p, seed, offset)

    @staticmethod
    def backward(ctx, dy):
        p, seed, offset = ctx.cfg
        rec = grad_overlap._deferred.get(dy.data_ptr())
        if rec is not None and not ctx.fused and rec["drop"] is None and dy.shape == rec["dx"].shape:
            rec["drop"] = (p, seed, offset)      # the sweep that consumes dy applies this mask while it stages the rows
            return (dy, None, None, None) + (None,) * ctx.nopt
        dact = ctx.saved_tensors[0] if ctx.fused else None
        return (hipops.dropout(dy.contiguous(), p, seed, offset, dact_y=dact, slope=LEAKY_SLOPE), None, None, None) + (None,) * ctx.nopt


class LinearFn(torch.autograd.Function):
    """y (rows,N) = x (rows,K) W^T + b   (the CTC head)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1])
        rows, K = x2.shape
        N = weight.shape[0]
        weight = weight.contiguous()
        y = torch.empty(rows, N, dtype=torch.float32, device=x.device)
        hipops.gemm(x2, weight, y, M=rows, N=N, K=K, transB=True, bias=bias.contiguous())
        ctx.save_for_backward(x2, weight)
        ctx.shp = shp
        ctx.param_refs = (weight, bias)
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        rows, K = x2.shape
        N = weight.shape[0]
        dy2 = dy.contiguous().view(rows, N)
        dx = torch.empty(rows, K, dtype=torch.float32, device=dy.device)
        hipops.gemm(dy2, weight, dx, M=rows, N=K, K=N)
        wg, bg = (p.grad for p in ctx.param_refs)
        if grad_overlap.enabled and wg is not None and bg is not None and wg.is_contiguous() and bg.is_contiguous():
            # only dx is on the chain: the weight gradient joins the side-stream work that runs beside the next sweep
            done = torch.cuda.Event()
            done.record()
            grad_overlap.note_event(done)

            def weight_grads():
                hipops.gemm(dy2, x2, wg, M=N, N=K, K=rows, transA=True, lda=N, splitk=_pick_splitk(N, K, rows), accumulate=True)
                hipops.colsum(dy2, rows, N, N, bg, accumulate=True)
            grad_overlap.pending().append((done, weight_grads))
            side = grad_overlap.side_stream()
            streams.hold(dy2, side); streams.hold(x2, side)
            return dx.view(ctx.shp), None, None
        dW = torch.empty(N, K, dtype=torch.float32, device=dy.device)
        hipops.gemm(dy2, x2, dW, M=N, N=K, K=rows, transA=True, lda=N, splitk=_pick_splitk(N, K, rows))
        db = torch.empty(N, dtype=torch.float32, device=dy.device)
        hipops.colsum(dy2, rows, N, N, db)
        return dx.view(ctx.shp), dW, db


class HeadFn(torch.autograd.Function):
    """(logits, log_probs) = the CTC head and its log-softmax from ONE kernel (``pgasr_head_logsoftmax``: exact fp32, one pass over x).
    ``log_probs`` is handed to the fused loss as a by-product (not differentiable: the loss's gradient arrives through ``logits``,
    d/dlogits of a function of log_softmax(logits)); the backward is LinearFn's."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1])
        weight = weight.contiguous()
        logits, lp = hipops.head_logsoftmax(x2, weight, bias.contiguous())
        ctx.save_for_backward(x2, weight)
        ctx.shp = shp
        ctx.param_refs = (weight, bias)
        lp = lp.view(*shp[:-1], -1)
        ctx.mark_non_differentiable(lp)
        return logits.view(*shp[:-1], -1), lp

    @staticmethod
    def backward(ctx, dy, _dlp=None):
        return LinearFn.backward(ctx, dy)


class LogSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        lp = hipops.log_softmax_rows(logits.contiguous())
        ctx.save_for_backward(lp)
        return lp

    @staticmethod
    def backward(ctx, g):
        (lp,) = ctx.saved_tensors
        return g - torch.exp(lp) * g.sum(dim=-1, keepdim=True)


class BLSTMLayerFn(torch.autograd.Function):
    """One bidirectional LSTM layer (model.py:39-44) on a time-major input (T,B,I) with
    per-utterance lengths (packed-sequence semantics of model.py:52-55)."""

    @staticmethod
    def forward(ctx, x, lengths, dact_y, sweep_follows, prepacked, out_dropout, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        """out_dropout = (p, seed, offset) or None: the layer RETURNS dropout(out) -- nn.LSTM's inter-layer dropout
        (model.py:42), written by the sweep's storer waves next to out -- instead of out; backward applies the same mask to
        the incoming gradient (or hands it to the helpers of its fed sweep)."""
        T, B, I = x.shape
        x = x.contiguous()
        if prepacked is None:
            prepacked = prepack_blstm((w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r), I)
        elif prepacked.ready is not None:
            torch.cuda.current_stream().wait_event(prepacked.ready)
        wih_perm, bias_perm, pack_f, pack_b = prepacked.wih_perm, prepacked.bias_perm, prepacked.pack_f, prepacked.pack_b
        ctx.planes_t = prepacked.planes_t
        npl = len(prepacked.planes) if prepacked.planes is not None else hipops.LSTM_PLANES   # bf16 planes per fp32 operand: the arithmetic of this layer's big products
        ctx.nplanes = npl
        ctx.fed_bwd = prepacked.fed_bwd       # zeroed counters for this layer's deferred input-gradient GEMM (or None)
        prepacked.fed_bwd = None
        ctx.slab_words = prepacked.slab       # zeroed publication words for this layer's streamed backward sweep (or None)
        prepacked.slab = None
        G = 2 * 4 * HID
        gates = torch.empty(T, B, G, dtype=torch.float32, device=x.device)
        out = torch.empty(T, B, 2 * HID, dtype=torch.float32, device=x.device)
        cbuf = torch.empty(T, B, 2 * HID, dtype=torch.float32, device=x.device)
        ctx.out_dropout = tuple(out_dropout) if out_dropout is not None else None
        out_drop = torch.empty_like(out) if out_dropout is not None else None
        dkw = {"out_drop": out_drop, "drop": ctx.out_dropout}
        x3w = prepacked.planes is not None and hipops.gemm_x3w_ok(T * B, G, I, planes=npl)   # LDS-DMA kernel, pre-split weight planes
        if (x3w and FEED_AHEAD and hipops.lstm_fed_ok(T, B) and hipops.x3w_feed_col_tiles(G, npl) > 0
                and hipops.streams_concurrent(grad_overlap.second_side_stream())):
            # the projection leaves the critical path: the sweep is launched FIRST and its helper workgroups wait for
            # the row tiles a GEMM on the side stream produces, in consumption order, on the XCDs the sweep leaves free
            main = torch.cuda.current_stream()
            side = grad_overlap.second_side_stream()     # not the stream the weight packs of the later layers are queued on
            need_words = 2 * ((T * B + 255) // 256)
            done = prepacked.fed_fwd if (prepacked.fed_fwd is not None and prepacked.fed_fwd.numel() >= need_words) else None
            prepacked.fed_fwd = None          # single use
            if done is None:
                done = torch.zeros(need_words, dtype=torch.int32, device=x.device)
            # HEAD: the first tile groups of the projection are computed on THIS stream in front of the sweep (one work item
            # per workgroup, whole chip, ~30 us), so that the rows of the sweep's first steps are in memory when it starts;
            # without it a fed sweep sits ~50-70 us on its first rows (stream gate + launch + first tiles).  The rest of
            # the queue follows on the feed stream as before.
            head = hipops.x3w_feed_head_items(G, I, planes=npl) > 0
            ws_feed = hipops.gemm_x3w_feed(x, prepacked.planes, gates, T * B, G, I, bias_perm, 0, done, phase=1) if head else None
            zeroed = torch.cuda.Event()
            zeroed.record()
            hipops.lstm_layer_fwd(gates, out, cbuf, pack_f, lengths, T, B, fed=done, fed_need=hipops.x3w_feed_col_tiles(G, npl), **dkw)
            side.wait_event(zeroed)
            busy = hipops.lstm_busy_ptr(T, B, False, x.device)
            with torch.cuda.stream(side):
                hipops.stream_gate(busy)
                hipops.gemm_x3w_feed(x, prepacked.planes, gates, T * B, G, I, bias_perm, busy if grad_overlap.confine_feed else 0, done,
                                     phase=2 if head else 0, ws=ws_feed)
            if ws_feed is not None:
                streams.hold(ws_feed, side)
            for t_ in (x, gates, done, bias_perm) + tuple(prepacked.planes):
                streams.hold(t_, side)
            grad_overlap._feed_unjoined[main.cuda_stream] = True
            # no join: the sweep cannot end before every row tile it waited for is complete, so the stream that ran the sweep
            # is already behind the feed's stores; what is left of the GEMM then (workgroups finding the tile counter
            # exhausted) touches no operand.  The next fed sweep's GEMM is ordered on the feed stream itself.
            if JOIN_FEED:
                main.wait_stream(side)
        else:
            if x3w and npl == 3 and hipops.x3w_feed_col_tiles(G, npl) > 0 and T * B * G * 4 < 2 ** 31:
                # the sequential order of a six-product projection: the SAME kernel with the same decomposition as the fed order (its
                # first tile groups are fixed-order sums of K-quarters, gemm_x6.hip), no XCD mask, before the sweep -- both orders give
                # the same bits (like the input-gradient product in backward)
                hipops.gemm_x3w_feed(x, prepacked.planes, gates, T * B, G, I, bias_perm, 0,
                                     torch.zeros(2 * ((T * B + 255) // 256), dtype=torch.int32, device=x.device))
            elif x3w:
                hipops.gemm_x3w(x, prepacked.planes, gates, T * B, G, I, bias=bias_perm)
            else:
                hipops.gemm(x, wih_perm, gates, M=T * B, N=G, K=I, transB=True, bias=bias_perm)
            hipops.lstm_layer_fwd(gates, out, cbuf, pack_f, lengths, T, B, **dkw)
        ctx.save_for_backward(x, lengths, gates, out, cbuf, wih_perm, pack_b, dact_y if dact_y is not None else x.new_empty(0))
        ctx.has_dact = dact_y is not None
        ctx.sweep_follows = bool(sweep_follows)   # backward: another layer's sweep runs right after this one
        ctx.param_refs = (w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r)
        return out if out_drop is None else out_drop

    @staticmethod
    def backward(ctx, dout):
        x, lengths, gates, out, cbuf, wih_perm, pack_b, dact_y = ctx.saved_tensors
        T, B, I = x.shape
        G = 2 * 4 * HID
        dev = x.device
        npl = ctx.nplanes
        dout = dout.contiguous()
        rec = grad_overlap._deferred.pop(dout.data_ptr(), None) if grad_overlap.enabled else None
        if ctx.out_dropout is not None:
            # forward returned dropout(out): the gradient w.r.t. out is the same mask applied to dout
            if rec is not None and rec["drop"] is None and dout.shape == rec["dx"].shape:
                rec["drop"] = ctx.out_dropout      # dout does not exist yet: the fed sweep's helpers mask the rows they stage
            else:
                dout = hipops.dropout(dout, *ctx.out_dropout)
        if grad_overlap.enabled:
            # "the main stream is about to launch the sweep": the last event recorded on it in this backward pass (the head's
            # `done`, the `swept` of the layer above) is late enough -- the side stream's gate kernel waits for the sweep's
            # clusters anyway -- and every event record costs the stream ~5 us (tools/dev/tools_marker_cost.py)
            before = grad_overlap.take_event()
            if before is None:
                before = torch.cuda.Event()
                before.record()
        # STREAMED sweep: this layer's own weight-gradient products run beside it on a third stream and consume its dgates
        # rows slab by slab (the sweep publishes, pgasr_lstm_layer_bwd_streamed) -- they used to wait for the sweep's end and
        # run beside the NEXT layer's sweep, which left the first layer's in the tail of the step
        targets = [p.grad for p in ctx.param_refs]
        stream_own = (grad_overlap.enabled and STREAM_DW and all(t is not None and t.is_contiguous() for t in targets)
                      and hipops.lstm_wgrads_ok(T, B, I, npl) and hipops.lstm_fed_ok(T, B)
                      and hipops.streams_concurrent(grad_overlap.third_side_stream()))
        slab = None
        if stream_own:
            slab = ctx.slab_words if (ctx.slab_words is not None and ctx.slab_words.numel() >= 64) else None
            ctx.slab_words = None             # single use
            if slab is None:
                slab = torch.zeros(64, dtype=torch.int32, device=dev)
        if rec is not None:
            # dout does not exist yet: the layer above left its input-gradient GEMM (and the dropout between the layers)
            # to us.  Its first tile groups are computed here, in front of the sweep; then the sweep goes, and the rest of
            # the GEMM follows on the side stream and feeds it row tile by row tile.
            if grad_overlap.enabled and rec["head"]():
                before = torch.cuda.Event()       # the side stream's part must come behind the head (one queue)
                before.record()
            sweep_ws, dbias_part = hipops.lstm_layer_bwd(gates, out, cbuf, dout, pack_b, lengths, T, B, want_dbias=True,
                                                         fed=rec["done"], fed_need=rec["need"], drop=rec["drop"], slab=slab)
        else:
            sweep_ws, dbias_part = hipops.lstm_layer_bwd(gates, out, cbuf, dout, pack_b, lengths, T, B, want_dbias=True, slab=slab)   # gates := dgates
        dg = gates
        if grad_overlap.enabled:
            # the layer above left its weight-gradient GEMMs for now: they run beside THIS sweep, whose clusters
            # register the XCDs they occupy in busy counters that the queue-mode GEMM workgroups consult
            busy = hipops.lstm_busy_ptr(T, B, True, dev)
            grad_overlap.flush(busy, launched_after=before, first=(lambda: rec["launch"](busy)) if rec is not None else None)
            if rec is not None:
                streams.hold(dout, grad_overlap.side_stream())
            fed_done = None
            if stream_own and grad_overlap._key() in grad_overlap._sides:
                # the point of the side stream behind everything queued there for THIS sweep's time (the GEMM that feeds it, the
                # head's weight gradients) and in FRONT of what the hook below may queue (N > 1: an all-reduce that waits for
                # this sweep's end -- the streamed products must not wait for that)
                fed_done = torch.cuda.Event()
                with torch.cuda.stream(grad_overlap.side_stream()):
                    fed_done.record()
            swept = torch.cuda.Event()
            swept.record()
            grad_overlap.note_event(swept)
            if not ctx.sweep_follows and grad_overlap.upper_grads_hook is not None:
                # first layer: the head's and the upper layers' weight gradients are all on the side stream now
                # (N > 1: the trainer starts their all-reduce there, behind this sweep and under the tail GEMMs)
                grad_overlap.upper_grads_hook(swept)
        dx = None
        defer = (ctx.needs_input_grad[0] and grad_overlap.enabled and FEED_AHEAD and FEED_BWD and ctx.sweep_follows and not ctx.has_dact
                 and ctx.planes_t is not None and hipops.gemm_x3w_ok(T * B, I, G, planes=npl) and I % 256 == 0 and I == 2 * HID
                 and hipops.x3w_feed_col_tiles(I, npl) > 0
                 and hipops.lstm_fed_ok(T, B) and hipops.streams_concurrent(grad_overlap.side_stream()))
        if defer:
            # another layer's backward sweep consumes dx (through at most a dropout): leave the product to that layer's
            # backward, which launches its sweep FIRST and lets this GEMM feed it from the side stream
            dx = torch.empty(T, B, I, dtype=torch.float32, device=dev)
            need_words = 2 * ((T * B + 255) // 256)
            done = ctx.fed_bwd if (ctx.fed_bwd is not None and ctx.fed_bwd.numel() >= need_words) else None
            ctx.fed_bwd = None                # single use: a second backward through this node zeroes its own
            if done is None:
                done = torch.zeros(need_words, dtype=torch.int32, device=dev)
            planes_t = ctx.planes_t

            feed_ws = []

            def head(dg=dg, dx=dx, done=done, planes_t=planes_t):
                """On the consuming sweep's stream, in front of its launch: the first tile groups (see the forward path)."""
                if hipops.x3w_feed_head_items(I, G, planes=npl) > 0:
                    feed_ws.append(hipops.gemm_x3w_feed(dg, planes_t, dx, T * B, I, G, None, 0, done, order=1, phase=1))
                    return True
                return False

            def launch(busy_ptr, dg=dg, dx=dx, done=done, planes_t=planes_t):
                hipops.gemm_x3w_feed(dg, planes_t, dx, T * B, I, G, None, busy_ptr if grad_overlap.confine_feed else 0, done, order=1,
                                     phase=2 if feed_ws else 0, ws=feed_ws[0] if feed_ws else None)
                side_ = torch.cuda.current_stream()
                for t_ in (dg, dx, done) + tuple(planes_t) + tuple(feed_ws):
                    streams.hold(t_, side_)
            grad_overlap._deferred[dx.data_ptr()] = {"dx": dx, "done": done, "need": hipops.x3w_feed_col_tiles(I, npl), "drop": None,
                                                     "launch": launch, "head": head}
        elif ctx.needs_input_grad[0]:
            dx = torch.empty(T, B, I, dtype=torch.float32, device=dev)
            if (ctx.sweep_follows and not ctx.has_dact and ctx.planes_t is not None and hipops.gemm_x3w_ok(T * B, I, G, planes=npl)
                    and I == 2 * HID and hipops.x3w_feed_col_tiles(I, npl) > 0 and T * B * I * 4 < 2 ** 31):
                # the product that feed-ahead would have run beside the next sweep, in the sequential order: the SAME kernel
                # with the same decomposition (its first tiles are fixed-order sums of K-quarters), no XCD mask, before the
                # sweep -- so that both orders give the same bits
                done = torch.zeros(2 * ((T * B + 255) // 256), dtype=torch.int32, device=dev)
                hipops.gemm_x3w_feed(dg, ctx.planes_t, dx, T * B, I, G, None, 0, done, order=1)
            elif ctx.planes_t is not None and hipops.gemm_x3w_ok(T * B, I, G, planes=npl):
                hipops.gemm_x3w(dg, ctx.planes_t, dx, T * B, I, G,
                                dact_y=dact_y if ctx.has_dact else None, slope=LEAKY_SLOPE)
            else:
                hipops.gemm(dg, wih_perm, dx, M=T * B, N=I, K=G, dact_y=dact_y if ctx.has_dact else None,
                            slope=LEAKY_SLOPE)
        def weight_grads(accumulate_into=None, hh_stream=None):
            """hh_stream: run the dW_hh product on that stream, beside dW_ih (used where no sweep follows: the tail of
            the step is then two GEMM chains wide instead of one long one)."""
            cur = torch.cuda.current_stream()
            if hipops.lstm_wgrads_ok(T, B, I, npl):
                # both products in one launch, summed over the time slabs that the streamed order uses: the same bits
                dwih = torch.empty(G, I, dtype=torch.float32, device=dev)
                dwhh = torch.empty(2, 4 * HID, HID, dtype=torch.float32, device=dev)
                hipops.lstm_wgrads(dg, x, out, T, B, I, dwih, dwhh, busy_ptr=hipops.GEMM_XCC_BUSY_PTR, planes=npl)
                dbias = torch.empty(G, dtype=torch.float32, device=dev)
                hipops.colsum(dbias_part, dbias_part.shape[0], G, G, dbias)
                if accumulate_into is not None:
                    hipops.lstm_unpack_grads(dwih, dbias, dwhh, I, accumulate_into, accumulate=True)
                    return None
                gl_ = [torch.empty(4 * HID, I, device=dev), torch.empty(4 * HID, HID, device=dev),
                       torch.empty(4 * HID, device=dev), torch.empty(4 * HID, device=dev),
                       torch.empty(4 * HID, I, device=dev), torch.empty(4 * HID, HID, device=dev),
                       torch.empty(4 * HID, device=dev), torch.empty(4 * HID, device=dev)]
                hipops.lstm_unpack_grads(dwih, dbias, dwhh, I, gl_)
                return gl_

            def hh():
                dwhh_ = torch.zeros(2, 4 * HID, HID, dtype=torch.float32, device=dev)
                if T > 1:
                    # dW_hh[d] = sum_t dgates_t[d]^T h_{prev(t)}[d];  prev = t-1 (fwd) / t+1 (rev); one launch, 2 batches
                    K = (T - 1) * B
                    hipops.gemm(dg, out, dwhh_, M=4 * HID, N=HID, K=K, transA=True, lda=G, ldb=2 * HID, ldc=HID,
                                a_off=B * G, b_off=0, strideA=4 * HID - B * G, strideB=B * 2 * HID + HID,
                                strideC=4 * HID * HID, batch=2, splitk=_pick_splitk(4 * HID, HID, K, 256, batch=2, beside_sweep=ctx.sweep_follows))   # x2 batches
                return dwhh_
            if hh_stream is not None:
                hh_stream.wait_stream(cur)
                with torch.cuda.stream(hh_stream):
                    dwhh = hh()
                    for t_ in (dg, out):
                        streams.hold(t_, hh_stream)
                streams.hold(dwhh, cur)
            dwih = torch.empty(G, I, dtype=torch.float32, device=dev)
            hipops.gemm(dg, x, dwih, M=G, N=I, K=T * B, transA=True, lda=G,
                        splitk=_pick_splitk(G, I, T * B, 512 if ctx.sweep_follows else 256, beside_sweep=ctx.sweep_follows))    # by layer, not by mode, so that overlap on/off give the same bits
            dbias = torch.empty(G, dtype=torch.float32, device=dev)
            hipops.colsum(dbias_part, dbias_part.shape[0], G, G, dbias)      # the sweep summed dgates over t per group
            if hh_stream is not None:
                cur.wait_stream(hh_stream)
                if hh_stream is grad_overlap._sides2.get(grad_overlap._main_of.get(cur.cuda_stream)):
                    grad_overlap._feed_unjoined.pop(grad_overlap._main_of[cur.cuda_stream], None)   # joined through this stream
            else:
                dwhh = hh()
            if accumulate_into is not None:
                hipops.lstm_unpack_grads(dwih, dbias, dwhh, I, accumulate_into, accumulate=True)
                return None
            gl = [torch.empty(4 * HID, I, device=dev), torch.empty(4 * HID, HID, device=dev),
                  torch.empty(4 * HID, device=dev), torch.empty(4 * HID, device=dev),
                  torch.empty(4 * HID, I, device=dev), torch.empty(4 * HID, HID, device=dev),
                  torch.empty(4 * HID, device=dev), torch.empty(4 * HID, device=dev)]
            hipops.lstm_unpack_grads(dwih, dbias, dwhh, I, gl)
            return gl

        if stream_own:
            s3 = grad_overlap.third_side_stream()
            # Everything queued on the side stream for THIS sweep's time goes first -- above all the GEMM that feeds the sweep:
            # workgroups that wait for the sweep must not hold the CUs its producer needs (the sweep would wait for rows, the
            # products for the sweep: a circle until the time-outs); and the head's weight gradients beside the top layer's
            # sweep, which otherwise sit behind the waiting workgroups until the sweep ends and then delay the next feed
            with torch.cuda.stream(s3):
                s3.wait_event(before)            # the main stream just before the sweep; then ALL of the sweep's clusters must have registered
                if fed_done is not None:
                    s3.wait_event(fed_done)
                hipops.stream_gate(busy, need=2 * ((B + 15) // 16), timeout_us=5000, running=slab)
                dwih = torch.empty(G, I, dtype=torch.float32, device=dev)
                dwhh = torch.empty(2, 4 * HID, HID, dtype=torch.float32, device=dev)
                hipops.lstm_wgrads(dg, x, out, T, B, I, dwih, dwhh, busy_ptr=busy if grad_overlap.confine else 0,
                                   slab=slab, err_ws=sweep_ws, planes=npl)
                s3.wait_event(swept)             # the bias partial sums are written at the sweep's very end
                dbias = torch.empty(G, dtype=torch.float32, device=dev)
                hipops.colsum(dbias_part, dbias_part.shape[0], G, G, dbias)
                hipops.lstm_unpack_grads(dwih, dbias, dwhh, I, targets, accumulate=True)
            for t_ in (dg, x, out, dbias_part, slab):
                streams.hold(t_, s3)
            grad_overlap._streamed_unjoined[grad_overlap._key()] = True
            return (dx, None, None, None, None, None) + (None,) * 8
        if grad_overlap.enabled and all(t is not None and t.is_contiguous() for t in targets):
            side = grad_overlap.side_stream()
            tail = not ctx.sweep_follows
            hh_stream = grad_overlap.second_side_stream() if (tail and grad_overlap.split_tail) else None
            ready = swept
            if tail and dx is not None and grad_overlap.tail_dx_first:
                # no sweep is left to hide behind and the input gradient heads the only dependent chain of the tail
                # (dropout/leaky', affine gradients, Adam): it gets the machine to itself, the weight-gradient GEMMs
                # start when it is done and run beside that chain
                ready = torch.cuda.Event()
                ready.record()
            grad_overlap.pending().append((ready, lambda: weight_grads(accumulate_into=targets, hh_stream=hh_stream)))
            if tail:
                grad_overlap.flush()       # nothing left to hide behind: go now
            for t_ in (dg, x, out, dbias_part):
                streams.hold(t_, side)
            return (dx, None, None, None, None, None) + (None,) * 8
        gl = weight_grads()
        return (dx, None, None, None, None, None, *gl)


class PackedBLSTM:
    """What a sweep needs of one layer's parameters, in kernel layouts (made once per step)."""
    __slots__ = ("wih_perm", "bias_perm", "pack_f", "pack_b", "planes", "planes_t", "ready", "fed_fwd", "fed_bwd", "slab")


def prepack_blstm(params, in_dim):
    pk = PackedBLSTM()
    params = [p.contiguous() for p in params]
    pk.wih_perm, pk.bias_perm, pk.pack_f, pk.pack_b = hipops.lstm_pack(params, in_dim)
    G = 2 * 4 * HID
    # both GEMM roles of W_ih: N=G,K=in and N=in,K=G, pre-split into the bf16 planes of the precision mode: (hi, lo) for the
    # bf16x3 kernels, (hi, mid, lo) for the six-product kernels of the "f32" mode (gemm_x6.hip: 256-wide tiles, 16-deep steps)
    npl = hipops.LSTM_PLANES
    ok = in_dim % 128 == 0 if npl == 2 else (in_dim % 256 == 0 and in_dim >= 64)
    pk.planes = hipops.split_planes(pk.wih_perm, planes=npl) if ok else None                      # (G, in): forward projection
    pk.planes_t = hipops.split_planes(pk.wih_perm, transpose=True, planes=npl) if ok else None    # (in, G): input gradient
    pk.ready = None
    pk.fed_fwd = pk.fed_bwd = pk.slab = None      # pre-zeroed tile counters / publication words of this step (prepack_blstm_layers)
    return pk


def prepack_blstm_layers(layer_params, in_dims, rows=0):
    """All layers' packs on the side stream (they depend on the parameters only); each carries the event the
    consuming stream waits for.  rows = T*B > 0: the step's feed-ahead tile counters are zeroed here as well (one fill
    for all layers, off the main stream) instead of one fill in front of every fed sweep."""
    main = torch.cuda.current_stream()
    side = grad_overlap.side_stream()
    side.wait_stream(main)          # the parameters may still be being written (previous step's Adam)
    out = []
    with torch.cuda.stream(side):
        nl = len(in_dims)
        words = 2 * ((rows + 255) // 256)
        counters = torch.zeros(3 * nl, max(words, 64), dtype=torch.int32, device=layer_params[0][0].device) if rows > 0 else None
        if counters is not None:
            streams.hold(counters, main)
        for li, (params, in_dim) in enumerate(zip(layer_params, in_dims)):
            pk = prepack_blstm(params, in_dim)
            if counters is not None:
                pk.fed_fwd, pk.fed_bwd, pk.slab = counters[2 * li], counters[2 * li + 1], counters[2 * nl + li]
            for t in (pk.wih_perm, pk.bias_perm, pk.pack_f, pk.pack_b) + tuple(pk.planes or ()) + tuple(pk.planes_t or ()):
                streams.hold(t, main)
            out.append(pk)
        # ONE event for all layers (the packs take ~0.1 ms, the front end that runs meanwhile longer): a cross-stream wait
        # costs the waiting stream ~10 us even when it is long satisfied (tools/dev/tools_marker_cost.py)
        ready = torch.cuda.Event()
        ready.record()
        out[0].ready = ready
    return out


def blstm_layer(x, lengths, params, dact_y=None, sweep_follows=False, prepacked=None, out_dropout=None):
    """sweep_follows: in the backward pass the sweep of the layer BELOW runs right after this layer's
    (True for every layer but the first) -- lets the overlapped weight-gradient GEMMs stay off its XCDs.
    prepacked: a PackedBLSTM made from ``params`` (else packed here).
    out_dropout = (p, seed, offset): return dropout(output) (the inter-layer dropout fused into the sweep)."""
    return BLSTMLayerFn.apply(x, lengths, dact_y, sweep_follows, prepacked, out_dropout, *params)
