"""Tensor-level wrappers over the C ABI (one function per entry point of
include/pgasr_hip.h).  torch is used for device memory and the current stream only.
Every wrapper requires CUDA(HIP) tensors and raises otherwise -- no CPU fallback."""
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t, dtype, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.PgasrError(f"{name} must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise _lib.PgasrError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise _lib.PgasrError(f"{name} must be contiguous")
    return t


def _p(t):
    return 0 if t is None else t.data_ptr()


_ws_cache = {}


def _workspace(nbytes, device, tag):
    key = (tag, device, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def ctc_loss_grad(log_probs, targets, input_lengths, target_lengths, blank=0, utt_scale=None,
                  pg_coef=None, pg_path=None, need_grad=True):
    """log_probs (T,B,V) fp32; targets (B,Lmax) int32; lengths int32.
    Returns (nll (B,), grad_logits (T,B,V) or None)."""
    lib = _lib.load()
    _req(log_probs, torch.float32, "log_probs")
    T, B, V = log_probs.shape
    _req(targets, torch.int32, "targets"); _req(input_lengths, torch.int32, "input_lengths")
    _req(target_lengths, torch.int32, "target_lengths")
    _req(utt_scale, torch.float32, "utt_scale"); _req(pg_coef, torch.float32, "pg_coef")
    _req(pg_path, torch.int32, "pg_path")
    Lmax = targets.shape[1] if targets.dim() == 2 else 0
    if Lmax == 0:
        targets = torch.zeros(B, 1, dtype=torch.int32, device=log_probs.device)
    nbytes = lib.pgasr_ctc_workspace_bytes(T, B, V, Lmax)
    ws = _workspace(nbytes, log_probs.device, "ctc")
    nll = torch.empty(B, dtype=torch.float32, device=log_probs.device)
    grad = torch.empty_like(log_probs) if need_grad else None
    st = lib.pgasr_ctc_loss_grad(_p(log_probs), _p(targets), _p(input_lengths), _p(target_lengths),
                                 T, B, V, Lmax, blank, _p(utt_scale), _p(pg_coef), _p(pg_path),
                                 _p(nll), _p(grad), _p(ws), ws.numel(), _stream())
    _lib.check(st, "pgasr_ctc_loss_grad")
    return nll, grad


def frame_argmax_sample(scores, seed=0, offset=0, want_greedy=True, want_sample=True):
    lib = _lib.load()
    _req(scores, torch.float32, "scores")
    T, B, V = scores.shape
    g = torch.empty(T, B, dtype=torch.int32, device=scores.device) if want_greedy else None
    s = torch.empty(T, B, dtype=torch.int32, device=scores.device) if want_sample else None
    st = lib.pgasr_frame_argmax_sample(_p(scores), T, B, V, int(seed) & (2 ** 64 - 1), int(offset) & 0xFFFFFFFF,
                                       _p(g), _p(s), _stream())
    _lib.check(st, "pgasr_frame_argmax_sample")
    return g, s


def ctc_collapse(paths, lengths, blank=0):
    """paths (P,T,B) int32 -> tokens (P,B,T) int32, token_lengths (P,B) int32."""
    lib = _lib.load()
    _req(paths, torch.int32, "paths"); _req(lengths, torch.int32, "lengths")
    P, T, B = paths.shape
    tokens = torch.zeros(P, B, T, dtype=torch.int32, device=paths.device)
    tl = torch.empty(P, B, dtype=torch.int32, device=paths.device)
    st = lib.pgasr_ctc_collapse(_p(paths), _p(lengths), P, T, B, blank, _p(tokens), _p(tl), _stream())
    _lib.check(st, "pgasr_ctc_collapse")
    return tokens, tl


def edit_distance(ref, ref_len, hyp, hyp_len, want_prefix=False):
    """ref (N,R) int32, hyp (N,Hy) int32 -> dist (N,) int32 [, prefix (N,Hy+1) int32]."""
    lib = _lib.load()
    _req(ref, torch.int32, "ref"); _req(hyp, torch.int32, "hyp")
    _req(ref_len, torch.int32, "ref_len"); _req(hyp_len, torch.int32, "hyp_len")
    N, R = ref.shape
    Hy = hyp.shape[1]
    dist = torch.empty(N, dtype=torch.int32, device=ref.device)
    prefix = torch.full((N, Hy + 1), -1, dtype=torch.int32, device=ref.device) if want_prefix else None
    st = lib.pgasr_edit_distance(_p(ref) if R else 0, _p(ref_len), R, _p(hyp) if Hy else 0, _p(hyp_len), Hy,
                                 N, _p(dist), _p(prefix), _stream())
    _lib.check(st, "pgasr_edit_distance")
    return (dist, prefix) if want_prefix else dist


def reinforce_grad(scores, path, coef, lengths, out=None, accumulate=False):
    lib = _lib.load()
    _req(scores, torch.float32, "scores"); _req(path, torch.int32, "path")
    _req(coef, torch.float32, "coef"); _req(lengths, torch.int32, "lengths")
    T, B, V = scores.shape
    if out is None:
        if accumulate:
            raise _lib.PgasrError("accumulate=True needs an output tensor")
        out = torch.empty_like(scores)
    _req(out, torch.float32, "out")
    st = lib.pgasr_reinforce_grad(_p(scores), _p(path), _p(coef), _p(lengths), T, B, V, int(bool(accumulate)),
                                  _p(out), _stream())
    _lib.check(st, "pgasr_reinforce_grad")
    return out
