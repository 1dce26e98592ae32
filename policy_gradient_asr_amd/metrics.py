"""Drop-in for the reference's metrics.py (edit_dist, evaluate, save_predictions) on the HIP
Levenshtein kernel (csrc/editdist.hip)."""
import os

import torch

from . import hipops
from .CTCdecoder import _device


def _encode(seqs):
    """Map arbitrary hashable symbols (chars or word strings) to dense int32 ids >= 1."""
    table = {}
    out = []
    for s in seqs:
        ids = []
        for tok in s:
            if tok not in table:
                table[tok] = len(table) + 1
            ids.append(table[tok])
        out.append(ids)
    return out


def edit_dist_batch(refs, hyps, device=None):
    """Levenshtein distances of many (reference, hypothesis) pairs in one launch.
    refs/hyps: lists of sequences (str or list of tokens).  Returns a list of ints."""
    dev = _device(device)
    n = len(refs)
    enc = [_encode([r, h]) for r, h in zip(refs, hyps)]
    R = max(max((len(e[0]) for e in enc), default=0), 1)
    Hy = max(max((len(e[1]) for e in enc), default=0), 1)
    ref = torch.zeros(n, R, dtype=torch.int32); hyp = torch.zeros(n, Hy, dtype=torch.int32)
    rl = torch.zeros(n, dtype=torch.int32); hl = torch.zeros(n, dtype=torch.int32)
    for i, (r, h) in enumerate(enc):
        ref[i, :len(r)] = torch.tensor(r, dtype=torch.int32); hyp[i, :len(h)] = torch.tensor(h, dtype=torch.int32)
        rl[i], hl[i] = len(r), len(h)
    d = hipops.edit_distance(ref.to(dev), rl.to(dev), hyp.to(dev), hl.to(dev))
    return [int(x) for x in d.tolist()]


def edit_dist(s1, s2):
    """(distance, len(s1)) between reference s1 and prediction s2 (metrics.py:4-21); str for CER,
    list of words for WER."""
    return edit_dist_batch([s1], [s2])[0], len(s1)


def evaluate(s1, s2):
    """(CER, WER) (metrics.py:23-31).  Like the reference, divides by zero on an empty reference."""
    w1, w2 = s1.split(" "), s2.split(" ")
    d = edit_dist_batch([s1, w1], [s2, w2])
    cer = d[0] / len(s1)
    wer = d[1] / len(w1)
    return cer, wer


def save_predictions(target, predicted, model_path):
    """predicted.txt with one 'target|prediction' line per utterance (metrics.py:33-37)."""
    path = os.path.join(model_path, "predicted.txt")
    with open(path, "w") as fo:
        for i in range(len(target)):
            fo.write(target[i] + "|" + predicted[i] + "\n")
