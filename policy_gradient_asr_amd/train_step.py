"""The train-step body of model.py:225-239 for the CTC + policy-gradient objective, one process
per GPU, gradients all-reduced over RCCL (replaces nn.DataParallel, model.py:201).

Parameters and gradients live in two flat fp32 buffers (nn.Parameters are views), so the
data-parallel exchange is ONE all-reduce of the whole gradient (19.15 MB at F=80,V=29) and the
optimizer is one Adam over the flat buffer.  Utterances are independent through forward, CTC,
decode, reward and REINFORCE gradient, so the gradient all-reduce is the only collective; the loss
is normalised by the GLOBAL batch so 1-GPU and N-GPU gradients agree to fp32 rounding.
"""
import os
import threading

import torch
import torch.distributed as dist

# Leading fp32 words of both flat buffers (256 bytes: the parameters keep their alignment).  Word 0 of the GRADIENT
# buffer carries the step's error flag through the gradient all-reduce: every rank writes 1.0 there when one of its
# sweeps timed out, the SUM is > 0 on every rank, and the guarded Adam of every rank reads that word -- all replicas skip
# the update or none does (the rank-local guard let replicas diverge, and let the other ranks apply a reduced gradient
# that contained the failed rank's garbage).
FLAG_PAD = 64
_step_lock = threading.Lock()     # the host layer keeps per-process state (grad_overlap, held tensors): one step at a time


def flatten_parameters(model, pad=0):
    """Re-point every parameter (and its .grad) at a slice of one flat buffer (``pad`` leading words stay free)."""
    params = [p for p in model.parameters()]
    dev = params[0].device
    total = pad + sum(p.numel() for p in params)
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    gflat = torch.zeros(total, dtype=torch.float32, device=dev)
    off = pad
    with torch.no_grad():
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.view(-1))
            p.data = flat[off:off + n].view_as(p.data)
            p.grad = gflat[off:off + n].view_as(p.data)
            off += n
    return flat, gflat


def shard_slice(global_batch, rank, world):
    """Contiguous B/N utterances per rank (global_batch must divide evenly)."""
    if global_batch % world:
        raise ValueError("global batch must be a multiple of the world size")
    per = global_batch // world
    return slice(rank * per, (rank + 1) * per)


def balance_by_frames(lengths, world):
    """Variable-T batches (BASELINE config 5): assign utterances to ranks so that every rank gets
    the same COUNT and nearly the same total number of frames (ranks then reach the all-reduce
    together).  Greedy longest-first into the lightest non-full rank.  Returns a list of index
    lists, one per rank."""
    n = len(lengths)
    if n % world:
        raise ValueError("batch must be a multiple of the world size")
    per = n // world
    order = sorted(range(n), key=lambda i: -int(lengths[i]))
    loads = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min((r for r in range(world) if len(out[r]) < per), key=lambda r: loads[r])
        out[r].append(i)
        loads[r] += int(lengths[i])
    return out


class DataParallelStep:
    """zero_grad -> forward_loss -> backward -> all-reduce(sum) -> Adam, on flat buffers.
    Subclasses provide ``forward_loss(batch, global_batch) -> scalar loss`` already divided by the
    GLOBAL batch size, so the summed gradient is the global-batch gradient."""

    def __init__(self, model, lr=5e-4, world_size=1, process_group=None, precision=None):
        """precision: "f32" (the library default: the reference's arithmetic, torch fp32 -- every big product as six bf16 MFMA terms
        of three-plane operands, exact to 2^-24: three-plane sweeps, six-product W_ih projections / input and weight gradients,
        gemm_x6.hip; the small products on the exact fp32 MFMA) or "bf16x3" (opt-in: 3-term products of two-plane operands, within
        the 1e-3 bar, ~20 % faster) -- hipops.PRECISION_MODES; None = whatever mode is set when a step runs."""
        if precision is not None:
            from . import hipops
            if precision not in hipops.PRECISION_MODES:
                raise ValueError(f"precision must be one of {sorted(hipops.PRECISION_MODES)} or None")
        self.precision = precision
        self.model = model
        self.world = world_size
        self.pg = process_group
        self.flat, self.gflat = flatten_parameters(model, FLAG_PAD)
        self.flat_param = torch.nn.Parameter(self.flat)
        self.flat_param.grad = self.gflat
        # Adam(lr=5e-4) is the reference's commented choice (model.py:207)
        self.lr = lr
        if self.flat.is_cuda:
            self.exp_avg = torch.zeros_like(self.flat)
            self.exp_avg_sq = torch.zeros_like(self.flat)
            self.applied = torch.zeros(2, dtype=torch.int32, device=self.flat.device)   # updates really applied (pgasr_adam_step)
            self.opt = None
        else:   # CPU is only the gloo plumbing test: torch's Adam
            self.opt = torch.optim.Adam([self.flat_param], lr=lr)
            self.applied_cpu = 0
        self.nstep = 0                      # CALLS of step() (seeds the sampler / dropout offsets); see applied_steps()
        self.collective = self.world > 1    # tests set this on a 1-rank group to exercise the plumbing
        self._early = None                  # (split, work) of an all-reduce issued during backward
        if self.world > 1:
            dist.broadcast(self.flat, src=0, group=self.pg)   # identical replicas

    def param_offset(self, name):
        """Offset in the flat buffers of the parameter called ``name`` (model.named_parameters() order)."""
        off = FLAG_PAD
        for n, p in self.model.named_parameters():
            if n == name:
                return off
            off += p.numel()
        raise KeyError(name)

    def reduce_upper(self, split):
        """Start the all-reduce of gflat[split:] now (its gradients are complete on the CURRENT stream) and leave
        gflat[:split] to ``reduce_rest``: two buckets, the first one hidden under what is left of backward.
        Every rank must call it at the same point of its step (collectives are matched by order)."""
        if not self.collective or self._early is not None:
            return
        # The blocking form on purpose: issued under the weight-gradient side stream it makes only THAT stream wait for the
        # collective (the caller's stream joins the side stream before Adam anyway).  With async_op=True the whole step ran
        # 9.0 -> 12.6 ms on a 1-rank RCCL group -- every phase slower, forward sweeps of the next step included, host enqueue
        # time unchanged (bench.py, PGASR_BENCH_SOLO_COLLECTIVE=1); a dummy kernel in the same place and the blocking
        # form both cost nothing.
        dist.all_reduce(self.gflat[split:], op=dist.ReduceOp.SUM, group=self.pg)
        self._early = (split, None)

    def reduce_rest(self):
        if not self.collective:
            return
        if self._early is None:
            dist.all_reduce(self.gflat, op=dist.ReduceOp.SUM, group=self.pg)
            return
        split, work = self._early
        self._early = None
        if work is not None:
            work.wait()                 # the current stream waits for the first bucket
        if split > 0:
            dist.all_reduce(self.gflat[:split], op=dist.ReduceOp.SUM, group=self.pg)

    def forward_loss(self, batch, global_batch):  # pragma: no cover - abstract
        raise NotImplementedError

    def write_local_error_flag(self):
        """Word 0 of the gradient buffer := 1.0 iff THIS rank's step produced invalid gradients (a persistent sweep gave up on a bounded
        wait; the words are sticky), else 0.0 -- one small launch on the current stream (``pgasr_error_flag``).  Returns False where
        there is nothing to report (CPU plumbing, no sweep workspace yet)."""
        if not self.flat.is_cuda:
            return False
        from . import _lib, hipops
        words = hipops.lstm_error_word_tensors(self.flat.device)
        if not words:
            return False
        lib = _lib.load()
        _lib.check(lib.pgasr_error_flag(words[0].data_ptr(), words[1].data_ptr() if len(words) > 1 else None, self.gflat.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "pgasr_error_flag")
        return True

    def applied_steps(self):
        """Number of Adam updates really applied (synchronises): ``nstep`` minus the updates the guard skipped."""
        if self.opt is not None:
            return self.applied_cpu
        return int(self.applied[self.nstep & 1].item())

    def set_applied_steps(self, n):
        """Checkpoint resume: the bias correction continues from ``n`` applied updates."""
        if self.opt is not None:
            self.applied_cpu = int(n)
        else:
            self.applied.fill_(int(n))

    def backward(self, loss):
        loss.backward()

    def step(self, *batch):
        return self._locked(self._step, batch)

    def compute_gradients(self, *batch):
        """The step WITHOUT its exchange and update: zero_grad -> forward_loss -> backward in the step's own orders (fed sweeps,
        streamed weight gradients, side streams), gradients left in ``gflat`` (the parameters' ``.grad`` views), every side stream
        joined.  Returns the detached local loss.  What the full-size parity tests compare with the oracle; also the hook for a
        caller that accumulates micro-batches itself."""
        return self._locked(self._gradients, batch).detach()

    def _locked(self, fn, batch):
        from . import streams
        # every side stream of the step is joined into the calling stream by the time backward() returns, so tensors that
        # cross streams are kept alive until the next step begins instead of being handed to record_stream (streams.hold)
        if not _step_lock.acquire(blocking=False):
            raise RuntimeError("policy_gradient_asr_amd: one train step at a time per process (the host layer's overlap / "
                               "stream state is process-global); a second trainer may only step between the steps of the first")
        try:
            with streams.managed_step():
                if self.precision is not None and self.flat.is_cuda:
                    from . import hipops
                    with hipops.precision(self.precision):
                        return fn(*batch)
                return fn(*batch)
        finally:
            _step_lock.release()

    def _gradients(self, *batch):
        local_b = batch[0].shape[0]
        self.gflat.zero_()
        loss = self.forward_loss(batch, local_b * self.world)
        self.backward(loss)
        return loss

    def _step(self, *batch):
        loss = self._gradients(*batch)
        if self.collective:
            # the error flag travels with the last gradient bucket (word 0, see FLAG_PAD): SUM > 0 on every rank iff any
            # rank's gradients are invalid
            self.write_local_error_flag()
        self.reduce_rest()
        self.nstep += 1
        if self.opt is None:
            from . import hipops
            # a sweep that timed out in this step (or an earlier one: the words are sticky) left invalid gradients:
            # the update is skipped on the device -- on every rank, since the guard is the reduced flag -- and does not
            # count for the bias correction; the host raises at its next check (hipops.lstm_assert_no_timeouts)
            guards = [self.gflat.data_ptr()] if self.collective else hipops.lstm_error_words(self.flat.device)
            hipops.adam_step(self.flat, self.gflat, self.exp_avg, self.exp_avg_sq, self.nstep, lr=self.lr,
                             guards=guards, applied=self.applied)
        elif self.collective and float(self.gflat[0]) > 0:
            pass                                # CPU plumbing (gloo tests): same rule, checked on the host
        else:
            self.opt.step()
            self.applied_cpu += 1
        return loss.detach()


class PolicyGradientTrainer(DataParallelStep):
    """step(x, targets, fmask, tmask): x (B,F,T) fp32; targets (B,L) int (pad 0); fmask (B,T);
    tmask (B,L) -- the collate_custom batch (data.py:107-116) after model.py:227-230's squeeze.
    Returns the detached local loss (no host sync).
    ONE trainer steps at a time per process: the overlap / stream / held-tensor state of the host layer (functional.grad_overlap,
    streams) is process-global and guarded by a step lock -- a second trainer may step between the steps of the first, a concurrent
    ``step()`` raises RuntimeError.
    Data parallel: ranks are expected to hold CONTIGUOUS shards of the global batch (``shard_slice``): the sampler addresses its draws by
    global utterance index ``rank * local_B + b``, which makes N ranks sample exactly what one process holding the whole batch samples.
    With ``balance_by_frames`` shards (variable lengths, configs[4]) the draws are still distinct and the training valid, but that
    identity with the single-process run does not hold."""

    def __init__(self, model, lr=5e-4, lam=1.0, seed=0, blank=0, world_size=1, process_group=None, rank=0,
                 reward_decoder="greedy", beam_size=16, precision=None, reward_mode="utterance"):
        """reward_mode: "utterance" (default) -- one reward R = -ED / |y| per utterance, the sum of the reference's per-step rewards
        (policy_grad.py:10-15) up to a constant the baseline removes; "per_step" -- the per-step rewards themselves, as rewards-to-go
        per frame against the greedy path's reward-to-go at the same frame (loss.PGCTCLossFn; greedy baseline only).
        reward_decoder: which hypothesis the self-critical baseline reward comes from -- "greedy" (best path) or
        "beam": the reference's own reward definition (policy_grad.py:6-8: prefix beam search -> collapse_fn ->
        edit distance), decoded on the device with ``beam_size`` (BASELINE config 5: 16; the reference passes 5)."""
        super().__init__(model, lr=lr, world_size=world_size, process_group=process_group, precision=precision)
        if reward_decoder not in ("greedy", "beam"):
            raise ValueError("reward_decoder must be 'greedy' or 'beam'")
        if reward_mode not in ("utterance", "per_step"):
            raise ValueError("reward_mode must be 'utterance' or 'per_step'")
        if reward_mode == "per_step" and reward_decoder != "greedy":
            raise ValueError("per-step rewards need the frame-aligned greedy baseline (reward_decoder='greedy')")
        self.reward_decoder, self.beam_size, self.reward_mode = reward_decoder, int(beam_size), reward_mode
        self.lam = lam
        # ONE sampling seed for all ranks: a rank addresses its draws by GLOBAL utterance index (contiguous shards: rank *
        # local batch), so N ranks sample exactly the paths of one process holding the whole batch -- the N-rank REINFORCE
        # gradient equals the single-process one, like the CTC part (tests/test_dp_rccl_gpu.py, lambda = 1)
        self.seed = seed
        self.rank = rank
        if hasattr(model, "encoder"):
            model.encoder.dropout_seed = 0x5EED + 104729 * rank
        self.blank = blank
        self._one = None
        self.last_stats = None
        self.overlap_weight_grads = True
        # A batch whose size the fast orders do not take (feed-ahead: B <= 32; streamed weight gradients: B % 16 == 0 in "f32", B % 32
        # == 0 in "bf16x3") is padded with EMPTY utterances (no frames, no target) up to the next such size: the last, ragged batch of an
        # epoch (model.py:221-222 leaves it as it comes) then runs the same orders as every other batch.  An empty utterance adds nothing
        # to the loss or to any gradient (CTC of nothing against nothing is 0, its rewards are 0, the sweeps skip it), the loss keeps
        # its normalisation by the REAL batch, the sampler its addressing; a 16-utterance group costs a sweep the same full or not.
        self.pad_ragged_batches = True
        # N > 1: the gradients of the head and of BLSTM layers 1, 2 (2/3 of the bytes) are all-reduced under the tail
        # of backward (the first layer's three GEMMs and the affine gradients), the rest after it
        self.early_reduce = os.environ.get("PGASR_EARLY_REDUCE", "1") != "0"
        try:
            self.upper_split = self.param_offset("encoder.blstm.weight_ih_l1")
        except KeyError:
            self.upper_split = None

    MAX_LOCAL_BATCH = 128      # pgasr_lstm_layer_fwd/bwd: at most 16 clusters of 16 utterances are co-resident
    MAX_VOCAB = 64             # CTC lattice / frame kernels: one wave per (t, b) row

    def _check_limits(self, x, targets):
        """The kernels' compiled-in limits, stated where the caller can read them (otherwise the first symptom is a
        PGASR_ERR_UNSUPPORTED from deep inside the step)."""
        if x.dim() != 3 or targets.dim() != 2:
            raise ValueError("step(x, targets, fmask, tmask): x (B,F,T), targets (B,L)")
        if x.shape[0] > self.MAX_LOCAL_BATCH:
            raise ValueError(f"local batch {x.shape[0]} > {self.MAX_LOCAL_BATCH}: the persistent LSTM sweeps keep at most 16 "
                             "clusters of 16 utterances resident on the chip; use more ranks or smaller batches")
        vocab = getattr(getattr(self.model, "head", None), "out_features", None)
        if vocab is not None and vocab > self.MAX_VOCAB:
            raise ValueError(f"alphabet of {vocab} symbols > {self.MAX_VOCAB}: the CTC / sampling kernels hold one frame's scores in one wave")
        if self.reward_decoder == "beam" and self.beam_size > 128:
            raise ValueError("beam_size > 128 is not supported by pgasr_ctc_beam_search")

    def staging_stream(self):
        """The stream on which the NEXT batch is to be staged into HBM once ``step()`` has returned (model.py:227-230's
        ``.to(device)``, taken off the critical path): the loss section's side stream.  Work queued there now runs beside
        this step's backward pass, is ordered after everything of the step before it (so the buffers of step k-1 are
        free) and needs no stream of its own -- an extra stream whose first packet waits for an event shares a hardware
        queue with one of the step's streams and holds up the GEMMs queued behind it (measured: 9.7 -> 13.8 ms)."""
        from .loss import PGCTCLossFn
        main = torch.cuda.current_stream()
        side = PGCTCLossFn._lattice_streams.get(main.cuda_stream)
        if side is None:
            from . import streams
            side = streams.side_stream("loss_section")
            PGCTCLossFn._lattice_streams[main.cuda_stream] = side
        return side

    def _upper_grads_issued(self, swept):
        """Called from the first BLSTM layer's backward once its sweep has been launched: every gradient of
        gflat[upper_split:] has been issued on the side stream by then.  The collective is ordered after them AND
        after the sweep (``swept``): a collective kernel never runs beside a sweep's latency chain."""
        from .functional import grad_overlap
        side = grad_overlap.side_stream()
        s3 = grad_overlap._sides3.get(grad_overlap._key()) if grad_overlap._streamed_unjoined.get(grad_overlap._key()) else None
        with torch.cuda.stream(side):
            side.wait_event(swept)
            if s3 is not None:
                side.wait_stream(s3)      # streamed sweeps: the upper layers' weight gradients were issued THERE (the first layer's are not yet)
            self.reduce_upper(self.upper_split)

    def backward(self, loss):
        """Weight-gradient GEMMs run on a side stream under the next layer's backward sweep."""
        from .functional import grad_overlap
        from .loss import PGCTCLossFn
        grad_overlap.enabled = self.overlap_weight_grads
        early = self.collective and self.early_reduce and self.overlap_weight_grads and self.upper_split is not None
        grad_overlap.upper_grads_hook = self._upper_grads_issued if early else None
        if self._one is None or self._one.device != loss.device:
            self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
        PGCTCLossFn.unit_seed_ptr = self._one.data_ptr()   # the seed gradient below IS 1: no fill, no 3.7 MB multiply on the chain
        try:
            loss.backward(gradient=self._one)
        except BaseException:
            grad_overlap._deferred.clear()      # do not let finish() mask the error with its own complaint
            raise
        finally:
            PGCTCLossFn.unit_seed_ptr = None
            grad_overlap.enabled = False
            grad_overlap.upper_grads_hook = None
            grad_overlap.finish()

    def _padded(self, x, targets, fmask, tmask):
        """The batch with empty utterances appended up to the next size the fast orders take (see ``pad_ragged_batches``)."""
        from . import hipops
        B = x.shape[0]
        q = 16 if hipops.LSTM_PLANES == 3 else 32
        Bp = -(-B // q) * q
        if not (self.pad_ragged_batches and x.is_cuda and Bp != B and Bp <= 32 and targets.dim() == 2 and fmask.dim() == 2):
            return x, targets, fmask, tmask
        n = Bp - B
        grow = lambda t_: torch.cat((t_, t_.new_zeros((n,) + tuple(t_.shape[1:]))), dim=0)
        return grow(x), grow(targets), grow(fmask), grow(tmask)

    def forward_loss(self, batch, global_batch):
        from .loss import pg_ctc_loss
        x, targets, fmask, tmask = batch
        from . import hipops
        self._check_limits(x, targets)
        real_b = x.shape[0]
        x, targets, fmask, tmask = self._padded(x, targets, fmask, tmask)
        padded = x.shape[0] != real_b
        if (fmask.dtype == torch.float32 and tmask.dtype == torch.int64 and targets.dtype == torch.int64 and targets.dim() == 2
                and targets.shape[1] > 0 and fmask.is_contiguous() and tmask.is_contiguous() and targets.is_contiguous()):
            in_len, tg_len, tg = hipops.batch_prep(fmask, tmask, targets)       # the collate_custom dtypes: one launch
        else:
            in_len = None
            tg_len = tmask.sum(dim=1).to(torch.int32).contiguous()
            tg = targets.to(torch.int32).contiguous()
        logits, in_len = self.model.logits(x, fmask, in_len)
        loss, nll, R_s, R_g = pg_ctc_loss(logits, in_len, tg, tg_len, lam=self.lam, seed=self.seed,
                                          offset=self.nstep + 1, global_batch=global_batch, blank=self.blank,
                                          beam=self.beam_size if self.reward_decoder == "beam" else 0,
                                          sample_base=self.rank * real_b if (self.world > 1 or padded) else -1,
                                          per_step=self.reward_mode == "per_step")
        self.last_stats = (nll[:real_b], R_s[:real_b], R_g[:real_b]) if padded else (nll, R_s, R_g)
        return loss
