"""Decoders, edit distance, reward, sampler and REINFORCE gradient on the CPU
(TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Restates, in this repo's own code:
  * prefix beam search        -- CTCdecoder.py:41-116 (+ logsumexp :31-39)
  * collapse_fn               -- CTCdecoder.py:119-131
  * edit_dist / evaluate      -- metrics.py:4-31
  * reward (evident intent)   -- policy_grad.py:4-16 (raises as written)
and defines the pieces the reference lacks: greedy best-path
decode, a counter-based (Philox4x32-10) per-frame sampler, the REINFORCE gradient.
"""
import math

import numpy as np

NEG_INF = -float("inf")


# --------------------------------------------------------------------------- #
# prefix beam search (CTCdecoder.py:41-116)
# --------------------------------------------------------------------------- #
def _lse(*xs):
    """Stable log-sum-exp of python floats with the all -inf guard
    (CTCdecoder.py:31-39)."""
    top = max(xs)
    if top == NEG_INF:
        return NEG_INF
    acc = 0.0
    for x in xs:
        acc += math.exp(x - top)
    return top + math.log(acc)


def prefix_beam_search(probs, beam_size=100, blank=0):
    """probs: (T,V) *probabilities*; returns (best prefix tuple, -log score).

    Ordering rules that make the result reproducible:
    candidates are created symbol-major / parent-rank-minor, a prefix's slot is
    created on first touch, and the per-frame sort is stable and descending in
    lse(p_blank, p_nonblank) so ties keep first-touch order."""
    T, V = probs.shape
    with np.errstate(divide="ignore"):
        logp = np.log(probs)
    beam = [((), 0.0, NEG_INF)]
    for t in range(T):
        table = {}  # prefix -> [p_b, p_nb]; dict keeps first-touch order

        def slot(key):
            e = table.get(key)
            if e is None:
                e = [NEG_INF, NEG_INF]
                table[key] = e
            return e

        for s in range(V):
            p = logp[t, s]
            for prefix, p_b, p_nb in beam:
                if s == blank:
                    e = slot(prefix)
                    e[0] = _lse(e[0], p_b + p, p_nb + p)
                    continue
                last = prefix[-1] if prefix else None
                e = slot(prefix + (s,))
                if s != last:
                    e[1] = _lse(e[1], p_b + p, p_nb + p)
                else:
                    # repeated symbol: only paths that ended in blank extend
                    e[1] = _lse(e[1], p_b + p)
                    # ... and the non-blank mass stays on the unchanged prefix
                    e2 = slot(prefix)
                    e2[1] = _lse(e2[1], p_nb + p)
        ranked = sorted(table.items(), key=lambda kv: _lse(kv[1][0], kv[1][1]),
                        reverse=True)[:beam_size]
        beam = [(k, v[0], v[1]) for k, v in ranked]
    best = beam[0]
    return best[0], -_lse(best[1], best[2])


def collapse_fn(preds):
    """Drop adjacent duplicate characters of an already-decoded string
    (CTCdecoder.py:119-131): 'aabbcc'->'abc', ''->''."""
    out = []
    for ch in preds:
        if not out or ch != out[-1]:
            out.append(ch)
    return "".join(out)


# --------------------------------------------------------------------------- #
# greedy best-path decode (SURVEY §8a A9 -- not in the reference)
# --------------------------------------------------------------------------- #
def greedy_decode(scores, lengths=None, blank=0):
    """scores (T,B,V) logits or log-probs -> list of B int lists.
    argmax per frame (first max wins) -> collapse repeats -> drop blank."""
    scores = np.asarray(scores)
    T, B, V = scores.shape
    best = np.argmax(scores, axis=2)  # first max wins
    out = []
    for b in range(B):
        Tb = T if lengths is None else int(lengths[b])
        seq, prev = [], -1
        for t in range(Tb):
            k = int(best[t, b])
            if k != prev and k != blank:
                seq.append(k)
            prev = k
        out.append(seq)
    return out


def collapse_path(path, blank=0):
    """CTC collapse of one sampled/argmax frame path (1-D ints)."""
    seq, prev = [], -1
    for k in path:
        k = int(k)
        if k != prev and k != blank:
            seq.append(k)
        prev = k
    return seq


# --------------------------------------------------------------------------- #
# edit distance / CER / WER (metrics.py:4-31)
# --------------------------------------------------------------------------- #
def edit_dist(s1, s2):
    """Levenshtein distance (sub=ins=del=1) between reference s1 and prediction
    s2; returns the TUPLE (distance, len(s1)) like metrics.py:21."""
    n = len(s1)
    row = list(range(n + 1))
    for i in range(1, len(s2) + 1):
        diag = row[0]
        row[0] = i
        for j in range(1, n + 1):
            up = row[j]
            if s2[i - 1] == s1[j - 1]:
                row[j] = diag
            else:
                row[j] = 1 + min(row[j - 1], diag, up)
            diag = up
    return int(row[n]), n


def evaluate(s1, s2):
    """(CER, WER) as metrics.py:23-31; divides by zero on an empty reference
    exactly like the reference does."""
    d, n = edit_dist(s1, s2)
    cer = d / n
    d, n = edit_dist(s1.split(" "), s2.split(" "))
    wer = d / n
    return cer, wer


# --------------------------------------------------------------------------- #
# reward (policy_grad.py:4-16, evident intent: element [0] of edit_dist's tuple)
# --------------------------------------------------------------------------- #
def reward_from_string(true_y, pred_y, t):
    """r_t for an already decoded+collapsed string pred_y (policy_grad.py:10-16)."""
    if t > 1:
        return -(edit_dist(true_y, pred_y[:t + 1])[0] - edit_dist(true_y, pred_y[:t])[0])
    if t == 1:
        return -(edit_dist(true_y, pred_y[:t + 1])[0] - len(true_y))
    raise ValueError("reward is undefined for t <= 0 (policy_grad.py:10-16)")


def reward(true_y, probs, t, ind2char, beam_size=5):
    """Full restatement of policy_grad.reward with the tuple defect fixed."""
    seq, _ = prefix_beam_search(probs, beam_size=beam_size)
    s = collapse_fn("".join(ind2char[i] for i in seq))
    return reward_from_string(true_y, s, t)


# --------------------------------------------------------------------------- #
# Philox4x32-10 sampler (SURVEY §8a A12 -- spec, not in the reference)
# --------------------------------------------------------------------------- #
_PH_M0 = np.uint64(0xD2511F53)
_PH_M1 = np.uint64(0xCD9E8D57)
_PH_W0 = 0x9E3779B9
_PH_W1 = 0xBB67AE85
_M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Counters: uint32 arrays; key: python ints."""
    c0 = c0.astype(np.uint64); c1 = c1.astype(np.uint64)
    c2 = c2.astype(np.uint64); c3 = c3.astype(np.uint64)
    k0 &= 0xFFFFFFFF; k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = _PH_M0 * c0
        p1 = _PH_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _M32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _M32
        n0 = (hi1 ^ c1 ^ np.uint64(k0)) & _M32
        n1 = lo1
        n2 = (hi0 ^ c3 ^ np.uint64(k1)) & _M32
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _PH_W0) & 0xFFFFFFFF
        k1 = (k1 + _PH_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def sampler_uniforms(T, B, seed, offset=0):
    """u[t,b] in [0,1): 24 high bits of Philox word 0 with counter
    (t*B+b, offset, 0, 0) and key (seed_lo, seed_hi).  Exactly representable in
    fp32, so the HIP sampler and this function see the same u bit for bit."""
    idx = np.arange(T * B, dtype=np.uint64)
    c0 = (idx & _M32).astype(np.uint32)
    c1 = np.full(T * B, offset & 0xFFFFFFFF, dtype=np.uint32)
    z = np.zeros(T * B, dtype=np.uint32)
    x0, _, _, _ = philox4x32_10(c0, c1, z, z, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return ((x0 >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)).reshape(T, B)


def sample_paths(logits, seed, offset=0):
    """Inverse-CDF sample of softmax(logits[t,b,:]) per frame, float64.
    Returns (paths (T,B) int64, cdf (T,B,V) float64, u (T,B))."""
    logits = np.asarray(logits, dtype=np.float64)
    T, B, V = logits.shape
    m = logits.max(axis=2, keepdims=True)
    e = np.exp(logits - m)
    cdf = np.cumsum(e, axis=2) / e.sum(axis=2, keepdims=True)
    u = sampler_uniforms(T, B, seed, offset)
    paths = (cdf <= u[..., None]).sum(axis=2)
    paths = np.minimum(paths, V - 1)
    return paths.astype(np.int64), cdf, u


def step_rewards(true_y, pred):
    """The reference's per-step rewards as it evidently means them (policy_grad.py:10-15 with element [0] of edit_dist's tuple):
    r_1 = |y| - ED(y, pred[:2]),  r_t = ED(y, pred[:t]) - ED(y, pred[:t+1]) for t > 1; returned for t = 1 .. len(pred)."""
    out = []
    for t in range(1, len(pred) + 1):
        if t > 1:
            out.append(-(edit_dist(true_y, pred[:t + 1])[0] - edit_dist(true_y, pred[:t])[0]))
        else:
            out.append(-(edit_dist(true_y, pred[:t + 1])[0] - len(true_y)))
    return out


def reward_to_go(path, true_y, blank=0):
    """Per-frame reward-to-go of one frame path (1-D ints, its valid frames only): character j of the collapsed path earns
    rho_j = ED(y, yhat[:j-1]) - ED(y, yhat[:j]) at the frame where it starts; G[t] = sum of rho_j over the characters that start at
    frames >= t.  (In the reference's r_t: r_1 = rho_1 + rho_2, r_t = rho_{t+1}.)  Summed character by character, no telescoping."""
    starts, seq, prev = [], [], -1
    for t, k in enumerate(path):
        k = int(k)
        if k != prev and k != blank:
            starts.append(t); seq.append(k)
        prev = k
    # d[i] = ED(y, seq[:i]) row by row (the recurrence of edit_dist above, keeping every row's last entry)
    n = len(true_y)
    row = list(range(n + 1))
    d = [n]
    for i in range(1, len(seq) + 1):
        diag, row[0] = row[0], i
        for jj in range(1, n + 1):
            up = row[jj]
            row[jj] = diag if seq[i - 1] == true_y[jj - 1] else 1 + min(row[jj - 1], diag, up)
            diag = up
        d.append(row[n])
    rho = [d[j - 1] - d[j] for j in range(1, len(seq) + 1)]
    per_frame = np.zeros(len(path), dtype=np.float64)
    for r, st in zip(rho, starts):
        per_frame[st] += r
    G = per_frame[::-1].cumsum()[::-1].copy()                 # G[t] = sum of the rewards earned at frames >= t
    return G, seq, rho


def reinforce_grad(logits, paths, coef, lengths):
    """d/dlogits of  sum_b coef[b] * ( - sum_{t<len_b} log softmax(logits)[t,b,path] )
    = coef[b] * (softmax - onehot(path)), zero for t >= len_b  (SURVEY §8a A12,
    coef[b] = lambda * (R_b - baseline_b) * scale)."""
    logits = np.asarray(logits, dtype=np.float64)
    T, B, V = logits.shape
    m = logits.max(axis=2, keepdims=True)
    e = np.exp(logits - m)
    sm = e / e.sum(axis=2, keepdims=True)
    onehot = np.zeros_like(sm)
    tt, bb = np.meshgrid(np.arange(T), np.arange(B), indexing="ij")
    onehot[tt, bb, np.asarray(paths)] = 1.0
    coef = np.asarray(coef, dtype=np.float64)
    g = (sm - onehot) * (coef[None, :, None] if coef.ndim == 1 else coef[:, :, None])      # (B,) per utterance or (T,B) per frame
    mask = (np.arange(T)[:, None] < np.asarray(lengths)[None, :])
    return g * mask[..., None]
