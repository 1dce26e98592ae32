"""CTC loss and gradient, numpy float64 (TEST INFRASTRUCTURE ONLY).

The reference has no CTC call site (SURVEY.md header fact 3, §8a row A5); the
conventions it fixes are blank = index 0 (CTCdecoder.py:41) and index 0 = '<pad>'
for targets (model.py:195, data.py:99).  The arithmetic follows Graves et al. 2006
(alpha/beta over the blank-extended label sequence l' of S = 2L+1 states) and is
checked against torch.nn.functional.ctc_loss in tests/test_oracle_cpu.py.

Gradient convention: d(sum_b nll_b)/d(logits[t,b,v]) where log-probs are
log_softmax(logits) -- i.e. softmax - occupancy/P, the only level at which torch's
CTC backward is meaningful (SURVEY.md §8c last row).
"""
import numpy as np

NEG_INF = -np.inf


def log_softmax(x, axis=-1):
    x = np.asarray(x, dtype=np.float64)
    m = np.max(x, axis=axis, keepdims=True)
    s = x - m
    return s - np.log(np.sum(np.exp(s), axis=axis, keepdims=True))


def _lse2(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), NEG_INF, r)


def _lse3(a, b, c):
    m = np.maximum(np.maximum(a, b), c)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m) + np.exp(c - m))
    return np.where(np.isneginf(m), NEG_INF, r)


def ctc_alpha_beta(lp, target, blank=0):
    """lp: (T,V) log-probs of ONE utterance (already cut to its length),
    target: (L,) ints.  Returns (alpha, beta, nll); beta includes the emission
    at t, so alpha[t,s]+beta[t,s]-lp[t,l'_s] is the log occupancy numerator."""
    T, V = lp.shape
    L = len(target)
    S = 2 * L + 1
    ext = np.full(S, blank, dtype=np.int64)
    ext[1::2] = target
    # skip transition s-2 -> s allowed when l'_s != blank and l'_s != l'_{s-2}
    can_skip = np.zeros(S, dtype=bool)
    can_skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])

    alpha = np.full((T, S), NEG_INF)
    alpha[0, 0] = lp[0, blank]
    if S > 1:
        alpha[0, 1] = lp[0, ext[1]]
    for t in range(1, T):
        a = alpha[t - 1]
        a1 = np.concatenate(([NEG_INF], a))[:S]
        a2 = np.concatenate(([NEG_INF, NEG_INF], a))[:S]
        a2 = np.where(can_skip, a2, NEG_INF)
        alpha[t] = _lse3(a, a1, a2) + lp[t, ext]

    beta = np.full((T, S), NEG_INF)
    beta[T - 1, S - 1] = lp[T - 1, blank]
    if S > 1:
        beta[T - 1, S - 2] = lp[T - 1, ext[S - 2]]
    # skip out of s into s+2 allowed when l'_{s+2} != blank and != l'_s
    skip_fwd = np.zeros(S, dtype=bool)
    skip_fwd[:-2] = can_skip[2:]
    for t in range(T - 2, -1, -1):
        b = beta[t + 1]
        b1 = np.concatenate((b, [NEG_INF]))[1:]
        b2 = np.concatenate((b, [NEG_INF, NEG_INF]))[2:]
        b2 = np.where(skip_fwd, b2, NEG_INF)
        beta[t] = _lse3(b, b1, b2) + lp[t, ext]

    if S > 1:
        ll = _lse2(alpha[T - 1, S - 1], alpha[T - 1, S - 2])
    else:
        ll = alpha[T - 1, S - 1]
    return alpha, beta, -float(ll), ext


def ctc_loss_and_grad(logits, targets, input_lengths, target_lengths, blank=0):
    """logits (T,B,V) any float; targets (B,Lmax) int (pad ignored);
    returns nll (B,) float64, grad (T,B,V) float64 = d sum_b nll_b / d logits,
    zero for frames t >= input_lengths[b] (same as torch's ctc_loss backward)."""
    logits = np.asarray(logits, dtype=np.float64)
    T, B, V = logits.shape
    lp_all = log_softmax(logits, axis=2)
    nll = np.zeros(B)
    grad = np.zeros((T, B, V))
    for b in range(B):
        Tb = int(input_lengths[b])
        Lb = int(target_lengths[b])
        lp = lp_all[:Tb, b]
        tgt = np.asarray(targets[b][:Lb], dtype=np.int64)
        alpha, beta, nll_b, ext = ctc_alpha_beta(lp, tgt, blank)
        nll[b] = nll_b
        if not np.isfinite(nll_b):
            # infeasible alignment: loss inf; torch gives nan/inf grads here.  We
            # define grad = 0 for such utterances (documented in DESIGN.md).
            continue
        ab = alpha + beta  # (Tb,S)
        occ = np.full((Tb, V), NEG_INF)
        for s in range(len(ext)):
            occ[:, ext[s]] = _lse2(occ[:, ext[s]], ab[:, s])
        with np.errstate(over="ignore"):
            grad[:Tb, b] = np.exp(lp) - np.exp(occ + nll_b - lp)
    return nll, grad


def ctc_reduce_mean(nll, target_lengths):
    """torch 'mean' reduction: per-utterance nll / max(L,1), then batch mean."""
    tl = np.maximum(np.asarray(target_lengths, dtype=np.float64), 1.0)
    return float(np.mean(nll / tl))
