"""Acoustic model + losses on the CPU (TEST INFRASTRUCTURE ONLY).

Restates model.py:34-56 (Encoder), model.py:19-25 (weights), loss.py:5-17
(customNLLLoss) and the spec'd CTC head (nn.Linear(512,V) +
log_softmax, blank=0) with torch-CPU / numpy ops.  Two forms of the BLSTM:

  * ``encoder_forward_torch``: torch.nn.functional + torch's CPU LSTM over a
    packed sequence, the same dispatch the reference reaches (model.py:52-55).
    This is what bench.py's cpu_baseline leg times.
  * ``blstm_numpy``: explicit float64 time loop (gate order i,f,g,o; two biases;
    reverse direction starting at each utterance's own last frame; zeros past the
    length) -- an independent check of the packed-sequence semantics.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

H = 256          # model.py:40
D_IN = 512       # model.py:38-39
N_LAYERS = 3     # model.py:41


def param_names(n_layers=N_LAYERS):
    """state_dict names of the reference Encoder."""
    names = ["input_layer.weight", "input_layer.bias"]
    for l in range(n_layers):
        for sfx in ("", "_reverse"):
            for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                names.append(f"blstm.{k}_l{l}{sfx}")
    return names


def init_params(n_feats=120, vocab=29, seed=0, dtype=torch.float32):
    """Xavier-normal / bias 0.1 on Linear (model.py:19-25); torch-default
    U(-1/sqrt(H), 1/sqrt(H)) on the LSTM; deterministic from ``seed``."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    std = (2.0 / (n_feats + D_IN)) ** 0.5
    p["input_layer.weight"] = torch.randn(D_IN, n_feats, generator=g, dtype=dtype) * std
    p["input_layer.bias"] = torch.full((D_IN,), 0.1, dtype=dtype)
    k = 1.0 / (H ** 0.5)
    for l in range(N_LAYERS):
        for sfx in ("", "_reverse"):
            p[f"blstm.weight_ih_l{l}{sfx}"] = (torch.rand(4 * H, D_IN, generator=g, dtype=dtype) * 2 - 1) * k
            p[f"blstm.weight_hh_l{l}{sfx}"] = (torch.rand(4 * H, H, generator=g, dtype=dtype) * 2 - 1) * k
            p[f"blstm.bias_ih_l{l}{sfx}"] = (torch.rand(4 * H, generator=g, dtype=dtype) * 2 - 1) * k
            p[f"blstm.bias_hh_l{l}{sfx}"] = (torch.rand(4 * H, generator=g, dtype=dtype) * 2 - 1) * k
    std = (2.0 / (2 * H + vocab)) ** 0.5
    p["head.weight"] = torch.randn(vocab, 2 * H, generator=g, dtype=dtype) * std
    p["head.bias"] = torch.full((vocab,), 0.1, dtype=dtype)
    return p


def instance_norm(x, eps=1e-5):
    """x (B,F,T): one mean / biased variance per utterance over all F*T values,
    padding included (model.py:37,48)."""
    mu = x.mean(dim=(1, 2), keepdim=True)
    var = x.var(dim=(1, 2), unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


def lstm_flat_weights(p):
    flat = []
    for l in range(N_LAYERS):
        for sfx in ("", "_reverse"):
            for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                flat.append(p[f"blstm.{k}_l{l}{sfx}"])
    return flat


def reverse_within_length(x, lengths):
    """x (B,T,C), lengths (B,): frame t < len of utterance b <- frame len-1-t; frames past the length stay where they are."""
    B, T = x.shape[0], x.shape[1]
    t = torch.arange(T)[None, :]
    L = lengths.to(torch.int64)[:, None]
    idx = torch.where(t < L, L - 1 - t, t)
    return torch.gather(x, 1, idx[:, :, None].expand_as(x))


def blstm_layer_packed_equivalent(x, lengths, w):
    """One bidirectional LSTM layer with the PACKED-sequence semantics of model.py:52-55 (the reverse direction starts at each
    utterance's own last valid frame; outputs past the length are exactly 0) WITHOUT torch's packed-sequence path, which on the
    CPU takes minutes at T = 1000: the forward direction is an ordinary LSTM over the padded batch (padding comes after the valid
    frames and cannot reach them), the reverse direction the same LSTM over each utterance reversed within its own length; outputs
    are masked and the reverse ones flipped back.  The same arithmetic per valid frame as nn.LSTM(bidirectional) on a
    PackedSequence -- checked against it (outputs and gradients) and against the reference's own Encoder outputs in
    tests/test_oracle_cpu.py.  x (B,T,I); w: the eight tensors weight_ih, weight_hh, bias_ih, bias_hh (forward), then reverse."""
    B, T, I = x.shape
    Hd = w[1].shape[1]
    mask = (torch.arange(T)[None, :] < lengths.to(torch.int64)[:, None])[:, :, None].to(x.dtype)
    outs = []
    for d in (0, 1):
        inp = x if d == 0 else reverse_within_length(x, lengths)
        cell = torch.nn.LSTM(I, Hd, 1, batch_first=True).to(x.dtype)
        names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")
        o, _ = torch.func.functional_call(cell, dict(zip(names, w[4 * d:4 * d + 4])), (inp,))
        o = o * mask
        outs.append(o if d == 0 else reverse_within_length(o, lengths))
    return torch.cat(outs, dim=2)


def encoder_forward_torch(p, x, mask, packed=True, leaky_side=None, return_pre=False, fast_packed=False):
    """p: dict of tensors; x (B,F,T); mask (B,T) 1/0 -> (B,T,512).
    Eval mode (no dropout).  model.py:47-56.
    fast_packed (with packed): the packed semantics through ``blstm_layer_packed_equivalent`` (full-size ragged batches).
    leaky_side (B,T,512) bool: which side of leaky_relu (model.py:50) each pre-activation is put on, instead of its own
    sign -- a DISCRETE choice, like an arg-max: a full-size parity test that shares the discrete choices of the device
    path (tests/test_train_step_gpu.py) passes the device's sides after counting how many differ from the oracle's own.
    return_pre: also return the pre-activations (B,T,512)."""
    B, Fdim, T = x.shape
    h = instance_norm(x).transpose(1, 2)                      # (B,T,F)
    pre = F.linear(h, p["input_layer.weight"], p["input_layer.bias"])
    h = F.leaky_relu(pre) if leaky_side is None else torch.where(leaky_side, pre, 0.01 * pre)
    lengths = mask.sum(dim=1).to(torch.int64).cpu()
    lstm = torch.nn.LSTM(D_IN, H, N_LAYERS, bidirectional=True, batch_first=True).to(x.dtype)
    lstm.eval()
    # run the module with OUR tensors as its parameters so gradients flow back to ``p``
    sd = {k[len("blstm."):]: v for k, v in p.items() if k.startswith("blstm.")}
    if packed and fast_packed:
        out = h
        for l in range(N_LAYERS):
            out = blstm_layer_packed_equivalent(out, lengths, [sd[f"{k}_l{l}{sfx}"] for sfx in ("", "_reverse")
                                                               for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")])
    elif packed:
        pk = pack_padded_sequence(h, lengths, enforce_sorted=False, batch_first=True)
        out, _ = torch.func.functional_call(lstm, sd, (pk,))
        out, _ = pad_packed_sequence(out, total_length=T, batch_first=True)
    else:
        out, _ = torch.func.functional_call(lstm, sd, (h,))
    return (out, pre) if return_pre else out


def head_forward_torch(p, enc_out):
    """(B,T,512) -> (T,B,V) log-probs."""
    logits = F.linear(enc_out, p["head.weight"], p["head.bias"])
    return F.log_softmax(logits, dim=2).transpose(0, 1)


def head_logits_torch(p, enc_out):
    return F.linear(enc_out, p["head.weight"], p["head.bias"]).transpose(0, 1)


def custom_nll(inp, target, ignore_index=None):
    """loss.py:5-17: sum over the first dim of ``inp`` (L,B,V) of a batch-mean
    NLL against target[:, i]; a falsy ignore_index (None or 0) ignores nothing."""
    total = 0.0
    L = inp.shape[0]
    for i in range(L):
        tgt = target[:, i]
        picked = -inp[i, torch.arange(inp.shape[1]), tgt]
        if ignore_index:
            keep = tgt != ignore_index
            total = total + picked[keep].sum() / keep.sum()
        else:
            total = total + picked.mean()
    return total


# --------------------------------------------------------------------------- #
# independent float64 BLSTM (explicit loops)
# --------------------------------------------------------------------------- #
def _sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def lstm_dir_numpy(x, lengths, w_ih, w_hh, b_ih, b_hh, reverse):
    """x (B,T,I) float64 -> (B,T,H); zeros past each length; the reverse
    direction starts from a zero state at t = len_b-1."""
    B, T, _ = x.shape
    Hh = w_hh.shape[1]
    out = np.zeros((B, T, Hh))
    for b in range(B):
        h = np.zeros(Hh)
        c = np.zeros(Hh)
        steps = range(int(lengths[b]) - 1, -1, -1) if reverse else range(int(lengths[b]))
        for t in steps:
            g = w_ih @ x[b, t] + b_ih + w_hh @ h + b_hh
            i = _sigmoid(g[0:Hh]); f = _sigmoid(g[Hh:2 * Hh])
            gg = np.tanh(g[2 * Hh:3 * Hh]); o = _sigmoid(g[3 * Hh:4 * Hh])
            c = f * c + i * gg
            h = o * np.tanh(c)
            out[b, t] = h
    return out


def blstm_numpy(p, x, lengths, n_layers=N_LAYERS):
    """p: dict of numpy float64 arrays keyed like param_names(); x (B,T,512)."""
    cur = x
    for l in range(n_layers):
        outs = []
        for sfx, rev in (("", False), ("_reverse", True)):
            outs.append(lstm_dir_numpy(cur, lengths,
                                       p[f"blstm.weight_ih_l{l}{sfx}"], p[f"blstm.weight_hh_l{l}{sfx}"],
                                       p[f"blstm.bias_ih_l{l}{sfx}"], p[f"blstm.bias_hh_l{l}{sfx}"], rev))
        cur = np.concatenate(outs, axis=2)
    return cur
