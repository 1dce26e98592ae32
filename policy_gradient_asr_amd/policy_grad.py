"""Drop-in for the reference's policy_grad.py.

``reward(true_y, pred_y, t, ind2char, ctc_decoder)`` keeps the reference's signature
(policy_grad.py:4).  As written the reference raises for every t (it subtracts the tuples
edit_dist returns, policy_grad.py:11-15; t = 0 leaves r_t unbound) -- SURVEY §8a A11.  This
implementation realises the evident intent: use element [0] of the tuple,
    t > 1 : r_t = ED(y, yhat[:t])   - ED(y, yhat[:t+1])
    t == 1: r_1 = |y|               - ED(y, yhat[:2])
and raises ValueError for t <= 0.  The decode (beam 5) and the edit distances run on the device;
``rewards_all_t`` returns every r_t of an utterance from ONE decode and ONE prefix-distance launch
instead of a full beam decode per t (the reference's dominant cost, SURVEY §3 stack 3).
"""
import torch

from . import hipops
from .CTCdecoder import collapse_fn, _device


def _decode_collapsed(pred_y, ind2char, ctc_decoder):
    seq, _score = ctc_decoder.decode(pred_y, beam_size=5)
    return collapse_fn("".join(ind2char[ind] for ind in seq))


def _prefix_distances(true_y, pred_s):
    """ED(true_y, pred_s[:i]) for i = 0..len(pred_s) from one kernel launch."""
    dev = _device()
    table = {}
    enc = lambda s: [table.setdefault(ch, len(table) + 1) for ch in s]
    r, h = enc(true_y), enc(pred_s)
    ref = torch.tensor([r + [0] * (1 - min(len(r), 1))], dtype=torch.int32, device=dev)
    hyp = torch.tensor([h + [0] * (1 - min(len(h), 1))], dtype=torch.int32, device=dev)
    rl = torch.tensor([len(r)], dtype=torch.int32, device=dev); hl = torch.tensor([len(h)], dtype=torch.int32, device=dev)
    _, prefix = hipops.edit_distance(ref, rl, hyp, hl, want_prefix=True)
    return prefix[0, :len(h) + 1].tolist()


def reward(true_y, pred_y, t, ind2char, ctc_decoder):
    if t <= 0:
        raise ValueError("reward is undefined for t <= 0 (policy_grad.py:10-16)")
    pred_s = _decode_collapsed(pred_y, ind2char, ctc_decoder)
    pd = _prefix_distances(true_y, pred_s)
    n = len(pred_s)
    d = lambda i: pd[min(i, n)]          # slices past the end return the whole string
    if t > 1:
        return -(d(t + 1) - d(t))
    return -(d(t + 1) - len(true_y))


def rewards_all_t(true_y, pred_y, ind2char, ctc_decoder):
    """[r_1, ..., r_{len(yhat)}] from one decode; sum telescopes to |y| - ED(y, yhat)."""
    pred_s = _decode_collapsed(pred_y, ind2char, ctc_decoder)
    pd = _prefix_distances(true_y, pred_s)
    n = len(pred_s)
    d = lambda i: pd[min(i, n)]
    out = []
    for t in range(1, n + 1):
        out.append(-(d(t + 1) - d(t)) if t > 1 else -(d(t + 1) - len(true_y)))
    return out, pred_s
