"""Losses of the path.

``customNLLLoss`` keeps the reference's name, constructor and call signature (loss.py:5-17),
including its quirk that a falsy ignore_index (None or 0) ignores nothing.  ``PGCTCLossFn`` is
the spec'd objective the reference lacks: CTC + lambda * REINFORCE with a
self-critical (greedy) baseline, computed by the HIP kernels in one fused gradient pass.
"""
import torch
import torch.nn as nn

from . import hipops
from . import streams


class customNLLLoss(nn.Module):
    """sum_i mean_b( -inp[i, b, target[b, i]] ) for inp (L,B,V) log-probs, target (B,L)."""

    def __init__(self, ignore_index=None):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, inp, target):
        L, B, V = inp.shape
        tgt = target.t().unsqueeze(-1).long()                 # (L,B,1)
        picked = -inp.gather(2, tgt).squeeze(-1)              # (L,B)
        if self.ignore_index:                                 # loss.py:9: falsy -> ignore nothing
            keep = (target.t() != self.ignore_index).to(inp.dtype)
            return ((picked * keep).sum(dim=1) / keep.sum(dim=1)).sum()
        return picked.mean(dim=1).sum()


class PGCTCLossFn(torch.autograd.Function):
    """loss = (1/Bg) sum_b [ nll_b / max(L_b,1)  -  lam * (R_s,b - R_g,b) * sum_{t<T_b} log p(pi_t,b) ]

    pi ~ softmax(logits) per frame (Philox, seed-addressable), R = -ED(y, collapse(path)) / max(L,1)
    for the sampled (R_s) path and for the baseline hypothesis (R_g): the greedy best path (beam = 0), or -- the
    reference's own reward definition, policy_grad.py:6-8 -- the prefix-beam-search hypothesis of width ``beam``
    after collapse_fn.  Bg = global batch (all ranks).  The reward is the utterance-level R = -ED / |y|: the
    reference's per-step r_t (policy_grad.py:10-15) telescope to |y| - ED(y, yhat), i.e. to the
    same R up to the constant |y| that the baseline subtracts; the per-t values themselves are available from
    policy_grad.rewards_all_t, and the gradient uses their sum (one coefficient per utterance).
    ``per_step = True`` (opt-in) puts the per-step rewards themselves into the gradient, as rewards-to-go: the coefficient of
    frame t becomes lam/Bg (G_s(t) - G_g(t)) / max(L,1) with G(t) = ED(y, yhat[:c(t)]) - ED(y, yhat) the sum of the rewards of the
    characters that start at frames >= t (c(t) = characters started before t), for the sampled path and -- the baseline -- for the
    greedy path at the same frame (``pgasr_pg_step_coefs``); frame 0 carries the utterance coefficient.  Greedy baseline only (a
    beam hypothesis has no frame alignment).
    Returns (loss, stats) where stats = (nll (B), R_s (B), R_g (B)) detached."""

    _lattice_streams = {}      # one side stream per calling stream
    # The trainer seeds loss.backward() with ITS OWN tensor of value 1 and registers that tensor's address here: only when
    # the incoming gradient IS that tensor (nothing between this function's output and the seed scaled it) is the
    # 3.7 MB multiply skipped.  Any other g -- a subclass that scales the loss, gradient accumulation with 1/k, a second
    # trainer -- takes grad * g.
    unit_seed_ptr = None
    unit_hits = 0              # how often the shortcut was taken (tests)
    @staticmethod
    def forward(ctx, logits, in_len, targets, tg_len, lam, seed, offset, global_batch, blank, beam=0, sample_base=-1, per_step=False,
                log_probs=None):
        T, B, V = logits.shape
        if per_step and beam > 0:
            raise ValueError("per-step rewards need the frame-aligned greedy baseline (beam = 0)")
        dev = logits.device
        # log-probs the head kernel already produced for exactly this tensor (model.Seq2Seq.logits), else one pass over the logits
        lp = log_probs
        if lp is None or lp.shape != logits.shape or not lp.is_contiguous():
            lp = hipops.log_softmax_rows(logits.contiguous())
        # The alpha/beta lattice (96 workgroups, a serial chain of T frames, ~0.27 ms at T=1000) is the long pole of this
        # section and stays on the CALLING stream; sampling, collapse, (beam search,) edit distance and the reward
        # arithmetic (~0.13 ms with the greedy baseline) run beside it on a side stream and are joined before the gradient
        # pass.  (Round 1 had it the other way round: two cross-stream hops then sat on the critical chain.)
        main = torch.cuda.current_stream()
        side = PGCTCLossFn._lattice_streams.setdefault(main.cuda_stream, None) or streams.side_stream("loss_section")
        PGCTCLossFn._lattice_streams[main.cuda_stream] = side
        # sample_base >= 0: this shard's first utterance in the GLOBAL batch -- the draws are then addressed globally
        lay = {"batch_stride": int(global_batch), "batch_offset": int(sample_base)} if sample_base >= 0 else {}
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if beam > 0:
                # baseline hypothesis = prefix beam search + collapse_fn (policy_grad.py:6-8), rows [0] of the pair buffers
                _, sample = hipops.frame_argmax_sample(lp, seed=seed, offset=offset, want_greedy=False, **lay)
                tokens = torch.zeros(2, B, T, dtype=torch.int32, device=dev)
                tok_len = torch.empty(2, B, dtype=torch.int32, device=dev)
                hipops.ctc_beam_search(lp, in_len, beam=beam, blank=blank, collapse=True, out=(tokens[0], tok_len[0]))
                hipops.ctc_collapse(sample[None], in_len, blank=blank, out=(tokens[1:], tok_len[1:]))
            else:
                greedy, sample = hipops.frame_argmax_sample(lp, seed=seed, offset=offset, **lay)
                paths = torch.stack((greedy, sample), dim=0)                          # (2,T,B)
                tokens, tok_len = hipops.ctc_collapse(paths, in_len, blank=blank)     # (2,B,T), (2,B)
            if per_step:
                dist, prefix = hipops.edit_distance(targets.repeat(2, 1), tg_len.repeat(2), tokens.view(2 * B, T), tok_len.view(2 * B),
                                                    want_prefix=True)
            else:
                dist = hipops.edit_distance(targets.repeat(2, 1), tg_len.repeat(2), tokens.view(2 * B, T), tok_len.view(2 * B))
            R_g, R_s, coef, utt_scale = hipops.pg_rewards(dist, tg_len, lam, 1.0 / float(global_batch))
            if per_step:
                coef = hipops.pg_step_coefs(paths, in_len, prefix, tok_len.view(2 * B), tg_len, lam, 1.0 / float(global_batch), blank=blank)
        nll, lattice = hipops.ctc_lattice(lp, targets, in_len, tg_len, blank=blank)
        main.wait_stream(side)
        for t_ in (sample, R_g, R_s, coef, utt_scale):
            streams.hold(t_, main)
        grad = hipops.ctc_grad_from_lattice(lp, in_len, tg_len, lattice, utt_scale=utt_scale, pg_coef=coef, pg_path=sample)
        loss = hipops.pg_loss_value(lp, sample, in_len, nll, utt_scale, coef).sum()
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(nll, R_s, R_g)
        ctx.set_materialize_grads(False)        # no zero-filled gradients for the three statistics
        return loss, nll, R_s, R_g

    @staticmethod
    def backward(ctx, g, *unused):
        (grad,) = ctx.saved_tensors
        if PGCTCLossFn.unit_seed_ptr is not None and g.data_ptr() == PGCTCLossFn.unit_seed_ptr and g.numel() == 1:
            PGCTCLossFn.unit_hits += 1
            return (grad,) + (None,) * 12
        return (grad * g,) + (None,) * 12


def pg_ctc_loss(logits, in_len, targets, tg_len, lam=1.0, seed=0, offset=0, global_batch=None, blank=0, beam=0, sample_base=-1,
                per_step=False, log_probs=None):
    """beam > 0: the baseline reward comes from the prefix-beam-search hypothesis of that width (see PGCTCLossFn).
    sample_base >= 0 (data parallel): index of this shard's first utterance in the global batch; the sampled paths are
    then those of the single-process global batch with the same seed.
    log_probs: log_softmax(logits) if the caller already has it (the head kernel's by-product, ``logits.log_probs`` of
    Seq2Seq.logits -- picked up from that attribute when not given)."""
    B = logits.shape[1]
    if log_probs is None:
        # the by-product is valid only for the tensor as the head kernel wrote it: any in-place edit since bumps _version
        log_probs = getattr(logits, "log_probs", None)
        if log_probs is not None and getattr(logits, "log_probs_version", None) != logits._version:
            log_probs = None
    return PGCTCLossFn.apply(logits, in_len, targets, tg_len, float(lam), int(seed), int(offset),
                             int(global_batch or B), int(blank), int(beam), int(sample_base), bool(per_step), log_probs)


class CTCLoss(nn.Module):
    """CTC-only objective ('mean' reduction: per-utterance nll / max(L,1), mean over the batch)."""

    def __init__(self, blank=0):
        super().__init__()
        self.blank = blank

    def forward(self, logits, in_len, targets, tg_len, global_batch=None):
        loss, _, _, _ = pg_ctc_loss(logits, in_len, targets, tg_len, lam=0.0, global_batch=global_batch,
                                    blank=self.blank)
        return loss
