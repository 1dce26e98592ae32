"""The reference's attention decoder on the CPU (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Restates model.py:58-94 (``Attention.forward``) and model.py:99-117 (``Decoder.forward``) AS EXECUTED, quirks included, in
closed form (numpy float64) instead of the reference's per-frame Python loops.

``Attention.forward(dec_t (B,H), enc_out (B,T,H))`` (model.py:62-92), per utterance b and encoder frame i:
    temp1[r,k] = exp(d[r] * e_i[k])                   an OUTER product of the decoder state with the frame (bmm of (H,1) x (1,H),
                                                      model.py:69) -- not a dot-product score: there is one "score" per (r, k) pair
    temp2[r]   = sum_k temp1[r,k]                     (model.py:70)
    a[r,k]     = temp1[r,k] / temp2[k]                model.py:73 divides the (H,H) matrix by the (H,) vector: numpy/torch broadcasting
                                                      lines the vector up with the LAST axis, so entry [r,k] is divided by ROW k's sum,
                                                      not row r's -- the rows of `a` do not sum to one (defect recorded in DESIGN.md)
    c_t[r,k]  += a[r,k] * e_i[k]                      (model.py:83,89: the frame broadcast over r)
and finally c_t = sum over r (model.py:91):
    c[b,k] = sum_i e[b,i,k] * S1[b,i,k] / S2[b,i,k],   S1 = sum_r exp(d[b,r] e[b,i,k]),   S2 = sum_k' exp(d[b,k] e[b,i,k'])
H_dec must equal H_enc (the bmm output is square), every encoder frame counts (no mask), exp is taken without a max shift
(fp32 overflow gives inf/inf = nan in the reference; this restatement shifts by the exact maxima, which changes nothing while the
reference's own exponentials are finite).

``Decoder.forward(target_inputs (B,L), encoder_outputs)`` (model.py:109-116): embed -> one-layer LSTM(128 -> H) -> per decoder
step t the context above -> cat(dec_out[:,t], c_t) -> a list of L tensors (B,2H) that the reference stacks only to PRINT its
shape; the function returns None.  ``decoder_preds`` returns that stacked (L,B,2H) tensor.
"""
import numpy as np


def attention_ctx(dec_t, enc_out):
    """dec_t (B,H), enc_out (B,T,H) -> c_t (B,H) float64: model.py:62-92 as executed."""
    d = np.asarray(dec_t, dtype=np.float64)
    e = np.asarray(enc_out, dtype=np.float64)
    B, T, H = e.shape
    if d.shape != (B, H):
        raise ValueError("Attention.forward needs dec_t (B,H) with the encoder's H (model.py:69: the bmm output is (H,H))")
    c = np.zeros((B, H))
    for b in range(B):
        for i in range(T):
            x = np.outer(d[b], e[b, i])                  # [r,k] = d_r e_k  (model.py:69 before the exp)
            m1 = x.max(axis=0)                           # per k over r
            s1 = np.exp(x - m1[None, :]).sum(axis=0)     # S1[k] / exp(m1[k])
            m2 = x.max(axis=1)                           # per row over k'
            s2 = np.exp(x - m2[:, None]).sum(axis=1)     # S2[row] / exp(m2[row]); used at index k (the broadcasting quirk)
            c[b] += e[b, i] * np.exp(m1 - m2) * s1 / s2
    return c


def lstm_layer_unidirectional(x, w_ih, w_hh, b_ih, b_hh):
    """x (B,L,I) -> h (B,L,H): torch's single-layer batch_first LSTM (gate order i,f,g,o; two biases; zero initial state), fp64."""
    x = np.asarray(x, dtype=np.float64)
    B, L, _ = x.shape
    H = w_hh.shape[1]
    w_ih, w_hh, b = np.asarray(w_ih, np.float64), np.asarray(w_hh, np.float64), np.asarray(b_ih, np.float64) + np.asarray(b_hh, np.float64)
    h = np.zeros((B, H)); c = np.zeros((B, H))
    out = np.zeros((B, L, H))
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    for t in range(L):
        g = x[:, t] @ w_ih.T + h @ w_hh.T + b
        i, f, gg, o = sig(g[:, :H]), sig(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), sig(g[:, 3 * H:])
        c = f * c + i * gg
        h = o * np.tanh(c)
        out[:, t] = h
    return out


def decoder_preds(params, target_inputs, encoder_outputs):
    """params: embed_layer.weight (V,128), lstm.weight_ih_l0 (4H,128), lstm.weight_hh_l0 (4H,H), lstm.bias_ih_l0, lstm.bias_hh_l0.
    -> (L,B,2H): what model.py:111-116 builds (and only prints the shape of)."""
    emb = np.asarray(params["embed_layer.weight"], np.float64)[np.asarray(target_inputs)]          # (B,L,128)
    dec_out = lstm_layer_unidirectional(emb, params["lstm.weight_ih_l0"], params["lstm.weight_hh_l0"],
                                        params["lstm.bias_ih_l0"], params["lstm.bias_hh_l0"])
    L = dec_out.shape[1]
    return np.stack([np.concatenate([dec_out[:, t], attention_ctx(dec_out[:, t], encoder_outputs)], axis=1) for t in range(L)])
