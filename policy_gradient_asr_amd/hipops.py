"""Tensor-level wrappers over the C ABI (one function per entry point of
include/pgasr_hip.h).  torch is used for device memory and the current stream only.
Every wrapper requires CUDA(HIP) tensors and raises otherwise -- no CPU fallback."""
import os

import torch

_os_environ_get = os.environ.get

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

# ---- optional per-kernel timing with HIP events on the launch stream (used by bench.py) ----
_prof_on = False
_prof_events = []


_prof_only = None


def profile_reset(enable, only=None):
    """only: iterable of name prefixes to time (None = every instrumented launch); an event pair costs a few
    microseconds of stream time, so bench.py times just the kernels its roofline is about inside the timed region."""
    global _prof_on, _prof_only
    _prof_on = bool(enable)
    _prof_only = tuple(only) if only else None
    _prof_events.clear()


def profile_pause(paused):
    """Stop / resume recording without dropping what was collected (bench.py instruments every n-th step: each event
    record costs the stream ~5 us)."""
    global _prof_on
    _prof_on = not paused


def profile_collect():
    """name -> (total ms, launches); synchronises."""
    torch.cuda.synchronize()
    out = {}
    for name, e0, e1 in _prof_events:
        ms, n = out.get(name, (0.0, 0))
        out[name] = (ms + e0.elapsed_time(e1), n + 1)
    return out


def profile_phases(step_marks):
    """Mean phases of a train step from the HIP events of its six sweeps (``profile_reset(True, only=("lstm_",))``) and one
    (start, end) event pair per step, all on the main stream: front end (step start -> first forward sweep), forward (first
    forward sweep start -> last forward sweep end, gaps included), loss section (-> first backward sweep), backward sweeps, tail
    (last backward sweep end -> end of the step's Adam).  Synchronises."""
    torch.cuda.synchronize()
    sweeps = [(n, a, b) for n, a, b in _prof_events if n.startswith("lstm_")]
    if not step_marks or len(sweeps) != 6 * len(step_marks):
        return None
    acc = {"front_end": 0.0, "forward_sweeps": 0.0, "loss_section": 0.0, "backward_sweeps": 0.0, "tail": 0.0}
    per_sweep = [0.0] * 6      # launch order: forward layers 1..3, backward layers 3..1
    for i, (s0, s1) in enumerate(step_marks):
        sw = sweeps[6 * i:6 * i + 6]
        if [n for n, _, _ in sw] != ["lstm_fwd_kernel"] * 3 + ["lstm_bwd_kernel"] * 3:
            return None
        acc["front_end"] += s0.elapsed_time(sw[0][1])
        acc["forward_sweeps"] += sw[0][1].elapsed_time(sw[2][2])
        acc["loss_section"] += sw[2][2].elapsed_time(sw[3][1])
        acc["backward_sweeps"] += sw[3][1].elapsed_time(sw[5][2])
        acc["tail"] += sw[5][2].elapsed_time(s1)
        for j in range(6):
            per_sweep[j] += sw[j][1].elapsed_time(sw[j][2])
    out = {k: v / len(step_marks) for k, v in acc.items()}
    out["sweeps_in_launch_order"] = [v / len(step_marks) for v in per_sweep]
    return out


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.on = _prof_on and (_prof_only is None or self.name.startswith(_prof_only))
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True); self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            _prof_events.append((self.name, self.e0, self.e1))
        return False


def _workspace(nbytes, device, tag):
    key = (tag, device, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        # sweep workspaces start zeroed: their first word is a STICKY error flag (lstm_assert_no_timeouts)
        alloc = torch.zeros if tag.startswith("lstm") else torch.empty
        buf = alloc(max(int(nbytes), 256), dtype=torch.uint8, device=device)
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


def ctc_lattice(log_probs, targets, input_lengths, target_lengths, blank=0):
    """First half of ``ctc_loss_grad``: alpha/beta lattice only.  Returns (nll (B,), handle); the handle goes to
    ``ctc_grad_from_lattice`` (which may run on another stream once this one's work is ordered before it)."""
    lib = _lib.load()
    _req(log_probs, torch.float32, "log_probs")
    T, B, V = log_probs.shape
    _req(targets, torch.int32, "targets"); _req(input_lengths, torch.int32, "input_lengths")
    _req(target_lengths, torch.int32, "target_lengths")
    Lmax = targets.shape[1] if targets.dim() == 2 else 0
    if Lmax == 0:
        targets = torch.zeros(B, 1, dtype=torch.int32, device=log_probs.device)
    nbytes = lib.pgasr_ctc_workspace_bytes(T, B, V, Lmax)
    ws = _workspace(nbytes, log_probs.device, "ctc")
    nll = torch.empty(B, dtype=torch.float32, device=log_probs.device)
    st = lib.pgasr_ctc_loss_grad(_p(log_probs), _p(targets), _p(input_lengths), _p(target_lengths),
                                 T, B, V, Lmax, blank, None, None, None, _p(nll), None, _p(ws), ws.numel(), _stream())
    _lib.check(st, "pgasr_ctc_loss_grad")
    return nll, (ws, Lmax, blank)


def _coef_per_frame(pg_coef, T, B):
    """pg_coef (B,): one coefficient per utterance; (T,B): one per frame (pg_step_coefs)."""
    if pg_coef is None or pg_coef.numel() == B and pg_coef.dim() == 1:
        return 0
    if tuple(pg_coef.shape) != (T, B):
        raise _lib.PgasrError(f"pg_coef must be (B,) or (T,B) = ({B},) / ({T},{B}); got {tuple(pg_coef.shape)}")
    return 1


def ctc_grad_from_lattice(log_probs, input_lengths, target_lengths, handle, utt_scale=None, pg_coef=None, pg_path=None):
    lib = _lib.load()
    ws, Lmax, blank = handle
    T, B, V = log_probs.shape
    _req(utt_scale, torch.float32, "utt_scale"); _req(pg_coef, torch.float32, "pg_coef"); _req(pg_path, torch.int32, "pg_path")
    grad = torch.empty_like(log_probs)
    st = lib.pgasr_ctc_grad_from_lattice(_p(log_probs), _p(input_lengths), _p(target_lengths), T, B, V, Lmax, blank,
                                         _p(utt_scale), _p(pg_coef), _p(pg_path), _coef_per_frame(pg_coef, T, B), _p(grad), _p(ws),
                                         ws.numel(), _stream())
    _lib.check(st, "pgasr_ctc_grad_from_lattice")
    return grad


def pg_rewards(dist, target_lengths, lam, inv_global_batch):
    """dist (2B,) int32 [greedy..., sampled...] -> (R_greedy, R_sample, pg_coef, utt_scale), each (B,) fp32."""
    lib = _lib.load()
    _req(dist, torch.int32, "dist"); _req(target_lengths, torch.int32, "target_lengths")
    B = target_lengths.numel()
    if dist.numel() != 2 * B:
        raise _lib.PgasrError("pg_rewards wants 2*B distances")
    out = torch.empty(4, B, dtype=torch.float32, device=dist.device)
    st = lib.pgasr_pg_rewards(_p(dist), _p(target_lengths), B, float(lam), float(inv_global_batch),
                              out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), _stream())
    _lib.check(st, "pgasr_pg_rewards")
    return out[0], out[1], out[2], out[3]


def pg_loss_value(log_probs, path, input_lengths, nll, utt_scale, pg_coef):
    """Per-utterance value of the objective (see include/pgasr_hip.h); sum() it for the loss."""
    lib = _lib.load()
    T, B, V = log_probs.shape
    _req(log_probs, torch.float32, "log_probs"); _req(path, torch.int32, "path"); _req(nll, torch.float32, "nll")
    _req(utt_scale, torch.float32, "utt_scale"); _req(pg_coef, torch.float32, "pg_coef")
    terms = torch.empty(B, dtype=torch.float32, device=log_probs.device)
    st = lib.pgasr_pg_loss_value(_p(log_probs), _p(path), _p(input_lengths), _p(nll), _p(utt_scale), _p(pg_coef),
                                 T, B, V, _coef_per_frame(pg_coef, T, B), _p(terms), _stream())
    _lib.check(st, "pgasr_pg_loss_value")
    return terms


FUSED_HEAD = _os_environ_get("PGASR_FUSED_HEAD", "1") != "0"


def head_logsoftmax_ok(K, V):
    return FUSED_HEAD and V <= 32 and K % 64 == 0 and K <= 1024


def head_logsoftmax(x, weight, bias, want_logits=True):
    """x (rows,K) @ weight (V,K)^T + bias -> (logits or None, log_probs), both (rows,V): the CTC head as one exact-fp32 kernel."""
    lib = _lib.load()
    _req(x, torch.float32, "x"); _req(weight, torch.float32, "weight"); _req(bias, torch.float32, "bias")
    rows, K = x.shape
    V = weight.shape[0]
    logits = torch.empty(rows, V, dtype=torch.float32, device=x.device) if want_logits else None
    lp = torch.empty(rows, V, dtype=torch.float32, device=x.device)
    with _timed("head_logsoftmax"):
        st = lib.pgasr_head_logsoftmax(_p(x), rows, K, x.stride(0), _p(weight), _p(bias), V, _p(logits), _p(lp), _stream())
    _lib.check(st, "pgasr_head_logsoftmax")
    return logits, lp


def pg_step_coefs(paths, input_lengths, prefix_dist, token_lengths, target_lengths, lam, inv_global_batch, blank=0):
    """Per-frame REINFORCE coefficients (T,B) from the per-step rewards (policy_grad.py:10-15; include/pgasr_hip.h):
    paths (2,T,B) greedy then sampled frame labels, prefix_dist (2B, P) / token_lengths (2B) of their collapsed forms."""
    lib = _lib.load()
    _req(paths, torch.int32, "paths"); _req(prefix_dist, torch.int32, "prefix_dist"); _req(token_lengths, torch.int32, "token_lengths")
    _req(input_lengths, torch.int32, "input_lengths"); _req(target_lengths, torch.int32, "target_lengths")
    two, T, B = paths.shape
    if two != 2 or prefix_dist.shape[0] != 2 * B or token_lengths.numel() != 2 * B or target_lengths.numel() != B:
        raise _lib.PgasrError("pg_step_coefs wants paths (2,T,B), prefix_dist (2B,P), token_lengths (2B,), target_lengths (B,)")
    coef = torch.empty(T, B, dtype=torch.float32, device=paths.device)
    st = lib.pgasr_pg_step_coefs(_p(paths), _p(input_lengths), _p(prefix_dist), prefix_dist.shape[1], _p(token_lengths),
                                 _p(target_lengths), T, B, int(blank), float(lam), float(inv_global_batch), _p(coef), _stream())
    _lib.check(st, "pgasr_pg_step_coefs")
    return coef


def frame_argmax_sample(scores, seed=0, offset=0, want_greedy=True, want_sample=True, batch_stride=0, batch_offset=0):
    """batch_stride / batch_offset: the GLOBAL batch and this shard's first utterance in it (data parallel: N ranks with
    one seed then draw what one process holding the whole batch draws); 0 / 0 = the local batch is the whole batch."""
    lib = _lib.load()
    _req(scores, torch.float32, "scores")
    T, B, V = scores.shape
    g = torch.empty(T, B, dtype=torch.int32, device=scores.device) if want_greedy else None
    s = torch.empty(T, B, dtype=torch.int32, device=scores.device) if want_sample else None
    st = lib.pgasr_frame_argmax_sample(_p(scores), T, B, V, int(seed) & (2 ** 64 - 1), int(offset) & 0xFFFFFFFF,
                                       int(batch_stride), int(batch_offset), _p(g), _p(s), _stream())
    _lib.check(st, "pgasr_frame_argmax_sample")
    return g, s


def batch_prep(fmask, tmask, targets):
    """The collated batch (model.py:227-230) -> (in_len (B) int32, tg_len (B) int32, targets (B,L) int32) in one launch."""
    lib = _lib.load()
    _req(fmask, torch.float32, "fmask"); _req(tmask, torch.int64, "tmask"); _req(targets, torch.int64, "targets")
    B, T = fmask.shape
    L = targets.shape[1]
    if tuple(tmask.shape) != (B, L) or targets.shape[0] != B:
        raise _lib.PgasrError("batch_prep: fmask (B,T), tmask (B,L), targets (B,L)")
    dev = fmask.device
    in_len = torch.empty(B, dtype=torch.int32, device=dev)
    tg_len = torch.empty(B, dtype=torch.int32, device=dev)
    tg32 = torch.empty(B, L, dtype=torch.int32, device=dev)
    _lib.check(lib.pgasr_batch_prep(_p(fmask), B, T, _p(tmask), _p(targets), L, _p(in_len), _p(tg_len), _p(tg32), _stream()), "pgasr_batch_prep")
    return in_len, tg_len, tg32


def ctc_collapse(paths, lengths, blank=0, out=None):
    """paths (P,T,B) int32 -> tokens (P,B,T) int32, token_lengths (P,B) int32.
    out = (tokens, token_lengths) to write into (contiguous, zero-filled tokens)."""
    lib = _lib.load()
    _req(paths, torch.int32, "paths"); _req(lengths, torch.int32, "lengths")
    P, T, B = paths.shape
    if out is not None:
        tokens, tl = out
        _req(tokens, torch.int32, "out tokens"); _req(tl, torch.int32, "out lengths")
        if tokens.numel() != P * B * T or tl.numel() != P * B:
            raise _lib.PgasrError("ctc_collapse: out tensors must hold (P,B,T) and (P,B) int32")
    else:
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


# ------------------------------------------------------------------------------------------
# dense contractions
# ------------------------------------------------------------------------------------------
GEMM_XCC_BUSY_PTR = 0   # device address of a sweep's 8 busy counters (0 = plain launch); set around side-stream GEMMs
GEMM_PRECISION = 0   # the model's GEMMs: 0 = fp32-faithful (six-product / exact fp32 MFMA; the default mode "f32"), 1 = bf16x3 split MFMA
LSTM_PLANES = 3      # bf16 planes per fp32 operand in the recurrent sweeps: 3 = hi/mid/lo, 6 products (default, "f32"); 2 = hi/lo, 3 products

# The arithmetic of the whole path is a MODE of the host layer:
#   "bf16x3": (opt-in) every dense product = 3 bf16 MFMA terms of a 2-plane split (~16 operand bits), fp32 accumulate; the input
#             affine (whose sign feeds leaky_relu') on the exact fp32 MFMA.  Within north_star's 1e-3 bar; ~20 % faster.
#   "f32":    (the DEFAULT since round 5) the reference's arithmetic (torch fp32: nn.Linear / nn.LSTM, model.py:38-44): every fp32 operand of a big product
#             as THREE bf16 planes and the product as SIX MFMA terms (every term down to 2^-24) -- the recurrent sweeps
#             (lstm.hip NP = 3) and, since round 4, the hoisted W_ih products and the weight gradients too (gemm_x6.hip), so the
#             mode runs the same feed-ahead and streamed orders as "bf16x3"; the small products (input affine, CTC head) on
#             the exact fp32 MFMA (v_mfma_f32_32x32x2_f32).
PRECISION_MODES = {"bf16x3": (1, 2), "f32": (0, 3)}
_precision = "f32"


def set_precision(mode):
    global GEMM_PRECISION, LSTM_PLANES, _precision
    if mode not in PRECISION_MODES:
        raise ValueError(f"precision mode must be one of {sorted(PRECISION_MODES)}")
    GEMM_PRECISION, LSTM_PLANES = PRECISION_MODES[mode]
    _precision = mode


def get_precision():
    return _precision


class precision:
    """``with hipops.precision("f32"): ...`` -- the mode for everything launched inside."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = _precision
        set_precision(self.mode)
        return self

    def __exit__(self, *a):
        set_precision(self.prev)
        return False


def gemm(A, B, C, M, N, K, transA=False, transB=False, lda=None, ldb=None, ldc=None, alpha=1.0,
         strideA=0, strideB=0, strideC=0, batch=1, sum_batches=False, splitk=1, bias=None, bias2=None,
         act=0, slope=0.01, accumulate=False, dact_y=None, norm_operand=0, shift=None, scale=None,
         a_off=0, b_off=0, c_off=0, precision=None, xcc_busy=None):
    """Raw strided GEMM on device tensors (element offsets a_off/b_off/c_off into A/B/C).
    See include/pgasr_hip.h for the contract."""
    lib = _lib.load()
    if precision is None:
        precision = GEMM_PRECISION
    for t, nm in ((A, "A"), (B, "B"), (C, "C"), (bias, "bias"), (bias2, "bias2"), (dact_y, "dact_y"),
                  (shift, "shift"), (scale, "scale")):
        if t is not None:
            if not t.is_cuda or t.dtype != torch.float32:
                raise _lib.PgasrError(f"gemm operand {nm} must be a float32 GPU tensor")
    lda = lda if lda is not None else (M if transA else K)
    ldb = ldb if ldb is not None else (K if transB else N)
    ldc = ldc if ldc is not None else N
    nbytes = lib.pgasr_gemm_workspace_bytes(M, N, batch, splitk, int(sum_batches))
    ws = _workspace(nbytes, C.device, "gemm")
    with _timed("gemm_f32"):
      st = lib.pgasr_gemm_f32(int(transA), int(transB), M, N, K, float(alpha),
                            A.data_ptr() + 4 * a_off, lda, int(strideA), B.data_ptr() + 4 * b_off, ldb, int(strideB),
                            C.data_ptr() + 4 * c_off, ldc, int(strideC), batch, int(sum_batches), splitk,
                            _p(bias), _p(bias2), act, float(slope), int(accumulate), _p(dact_y),
                            norm_operand, _p(shift), _p(scale), int(precision), int(GEMM_XCC_BUSY_PTR if xcc_busy is None else xcc_busy), _p(ws), ws.numel() if ws is not None else 0, _stream())
    _lib.check(st, "pgasr_gemm_f32")
    return C


X6_PACKED = _os_environ_get("PGASR_X6_PACKED", "1") != "0"     # three-plane weights as ONE buffer in the six-product kernel's LDS-image order


class X6Pack:
    """The three-plane weight operand of the six-product kernels, packed tile by tile (``pgasr_pack_x6w_planes``)."""
    __slots__ = ("pack", "N", "K")

    def __init__(self, pack, N, K):
        self.pack, self.N, self.K = pack, N, K

    def __len__(self):
        return 3

    def __iter__(self):          # tensors to keep alive / to hold across streams
        return iter((self.pack,))


def split_planes(w, transpose=False, planes=None, packed=None):
    """fp32 matrix (rows, cols) -> bf16 planes (int16 storage) of shape (rows, cols), or (cols, rows) with ``transpose``: the
    pre-split weight operand of ``gemm_x3w``.  planes = 2: (hi, lo), the bf16x3 kernels' operand; 3: (hi, mid, lo), the
    six-product kernels' (precision mode "f32"); None: what the current precision mode uses."""
    lib = _lib.load()
    _req(w, torch.float32, "w")
    if w.dim() != 2 or w.stride(1) != 1:
        raise _lib.PgasrError("split_planes wants a row-major 2-D tensor")
    planes = LSTM_PLANES if planes is None else int(planes)
    if planes not in (2, 3):
        raise _lib.PgasrError("split_planes: 2 or 3 planes")
    rows, cols = w.shape
    shape = (cols, rows) if transpose else (rows, cols)
    if planes == 3 and (X6_PACKED if packed is None else packed) and shape[0] % 256 == 0 and shape[1] % 16 == 0:
        pack = torch.empty(shape[0] * shape[1] * 6, dtype=torch.uint8, device=w.device)
        _lib.check(lib.pgasr_pack_x6w_planes(w.data_ptr(), rows, cols, w.stride(0), int(transpose), pack.data_ptr(), _stream()), "pgasr_pack_x6w_planes")
        return X6Pack(pack, shape[0], shape[1])
    out = tuple(torch.empty(shape, dtype=torch.int16, device=w.device) for _ in range(planes))
    if planes == 3:
        st = lib.pgasr_split_bf16_planes3(w.data_ptr(), rows, cols, w.stride(0), int(transpose), *[t.data_ptr() for t in out], _stream())
    else:
        st = lib.pgasr_split_bf16_planes(w.data_ptr(), rows, cols, w.stride(0), int(transpose), *[t.data_ptr() for t in out], _stream())
    _lib.check(st, "pgasr_split_bf16_planes")
    return out


def gemm_x3w_ok(M, N, K, lda=None, planes=2):
    """Shapes the LDS-DMA kernels take (else use ``gemm``): planes = 2 the bf16x3 kernels, 3 the six-product ones."""
    lda = K if lda is None else lda
    if planes == 3:
        Mp = (M + 255) // 256 * 256       # the kernel addresses whole 256-row tiles with 32-bit offsets (x6w_shape_ok, gemm_x6.hip)
        return K % 16 == 0 and K >= 64 and N % 256 == 0 and lda % 4 == 0 and M > 0 and Mp * max(N, lda) * 4 < 2 ** 32
    return K % 32 == 0 and N % 128 == 0 and lda % 4 == 0 and M > 0


def _plane_ptrs(planes):
    """(hi, mid, lo) device addresses for the six-product entry points: a pack travels as hi with mid = lo = NULL."""
    if isinstance(planes, X6Pack):
        return planes.pack.data_ptr(), None, None
    return tuple(t.data_ptr() for t in planes)


def _check_planes(planes, N, K, what):
    if isinstance(planes, X6Pack):
        if (planes.N, planes.K) != (N, K) or not planes.pack.is_cuda:
            raise _lib.PgasrError(f"{what}: packed planes are for a ({planes.N}, {planes.K}) weight, the product wants ({N}, {K})")
        return
    if len(planes) not in (2, 3):
        raise _lib.PgasrError(f"{what}: planes must be a (hi, lo) or (hi, mid, lo) tuple")
    for t in planes:
        if tuple(t.shape) != (N, K) or t.dtype != torch.int16 or not t.is_contiguous() or not t.is_cuda:
            raise _lib.PgasrError(f"{what} planes must be contiguous int16 (N, K) GPU tensors")


def gemm_x3w(A, planes, C, M, N, K, lda=None, ldc=None, bias=None, dact_y=None, slope=0.01):
    """C[M,N] = A[M,K] @ W[N,K]^T (+bias) (*leaky'(dact_y)); W given as ``split_planes`` output: two planes -> the bf16x3
    kernels (three products), three planes -> the fp32-faithful six-product kernel (gemm_x6.hip)."""
    lib = _lib.load()
    for t, nm in ((A, "A"), (C, "C"), (bias, "bias"), (dact_y, "dact_y")):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32):
            raise _lib.PgasrError(f"gemm_x3w operand {nm} must be a float32 GPU tensor")
    _check_planes(planes, N, K, "gemm_x3w")
    with _timed("gemm_x6c" if len(planes) == 3 else "gemm_x3c"):
        if len(planes) == 3:
            st = lib.pgasr_gemm_x6w_f32(M, N, K, A.data_ptr(), K if lda is None else lda, *_plane_ptrs(planes),
                                        C.data_ptr(), N if ldc is None else ldc, _p(bias), _p(dact_y), float(slope), _stream())
        else:
            st = lib.pgasr_gemm_x3w_f32(M, N, K, A.data_ptr(), K if lda is None else lda, planes[0].data_ptr(), planes[1].data_ptr(),
                                        C.data_ptr(), N if ldc is None else ldc, _p(bias), _p(dact_y), float(slope), _stream())
    _lib.check(st, "pgasr_gemm_x6w_f32" if len(planes) == 3 else "pgasr_gemm_x3w_f32")
    return C


def colsum(X, rows, cols, ld, out, out2=None, accumulate=False):
    lib = _lib.load()
    nbytes = lib.pgasr_colsum_workspace_bytes(rows, cols)
    ws = _workspace(nbytes, X.device, "colsum")
    st = lib.pgasr_colsum_f32(_p(X), rows, cols, ld, _p(out), _p(out2), int(accumulate), _p(ws), ws.numel(), _stream())
    _lib.check(st, "pgasr_colsum_f32")
    return out


def instnorm_stats(x, eps=1e-5):
    lib = _lib.load()
    _req(x, torch.float32, "x")
    B, F, T = x.shape
    mean = torch.empty(B, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, dtype=torch.float32, device=x.device)
    _lib.check(lib.pgasr_instnorm_stats(_p(x), B, F, T, float(eps), _p(mean), _p(rstd), _stream()), "pgasr_instnorm_stats")
    return mean, rstd


def log_softmax_rows(logits):
    lib = _lib.load()
    _req(logits, torch.float32, "logits")
    V = logits.shape[-1]
    out = torch.empty_like(logits)
    _lib.check(lib.pgasr_log_softmax_rows(_p(logits), logits.numel() // V, V, _p(out), _stream()), "pgasr_log_softmax_rows")
    return out


# ------------------------------------------------------------------------------------------
# LSTM layer
# ------------------------------------------------------------------------------------------
HID = 256


def lstm_pack(params, in_dim):
    """params: 8 tensors (w_ih, w_hh, b_ih, b_hh) x (fwd, rev) -> (wih_perm, bias_perm, pack_f, pack_b)."""
    lib = _lib.load()
    dev = params[0].device
    for t in params:
        _req(t, torch.float32, "lstm parameter")
    wih_perm = torch.empty(2 * 4 * HID, in_dim, dtype=torch.float32, device=dev)
    bias_perm = torch.empty(2 * 4 * HID, dtype=torch.float32, device=dev)
    nb = lib.pgasr_lstm_pack_bytes(0, LSTM_PLANES)
    pack_f = torch.empty(nb, dtype=torch.uint8, device=dev)
    pack_b = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = lib.pgasr_lstm_pack_weights(*[_p(t) for t in params], in_dim, _p(wih_perm), _p(bias_perm),
                                     _p(pack_f), _p(pack_b), LSTM_PLANES, _stream())
    _lib.check(st, "pgasr_lstm_pack_weights")
    return wih_perm, bias_perm, pack_f, pack_b


def _lstm_flags(pack):
    """Sweep flags for a W_hh pack made by ``lstm_pack``: the pack's size says how many planes it holds."""
    planes = 3 if pack.numel() == _lib.load().pgasr_lstm_pack_bytes(0, 3) else 2
    return LSTM_FLAGS | (2 if planes == 3 else 0)


def lstm_unpack_grads(dwih_perm, dbias_perm, dwhh_perm, in_dim, grads, accumulate=False):
    lib = _lib.load()
    st = lib.pgasr_lstm_unpack_grads(_p(dwih_perm), _p(dbias_perm), _p(dwhh_perm), in_dim,
                                     *[_p(g) for g in grads], int(accumulate), _stream())
    _lib.check(st, "pgasr_lstm_unpack_grads")


_lstm_err_checks = []


def _lstm_ws(T, B, backward, device):
    lib = _lib.load()
    nbytes = lib.pgasr_lstm_workspace_bytes(T, B, int(backward))
    return _workspace(nbytes, device, "lstm_bwd" if backward else "lstm_fwd")


import os as _os
LSTM_FLAGS = int(_os.environ.get("PGASR_LSTM_FLAGS", "0"), 0)   # bit 0: force the write-through (cross-XCD) hand-off protocol; bits 8..: diagnostics


def lstm_layer_fwd(gates, out, cbuf, pack_f, lengths, T, B, fed=None, fed_need=0, out_drop=None, drop=None):
    """fed: int32 (2, ceil(T*B/256)) finished-tile counters of a ``gemm_x3w_feed`` that is launched AFTER this call on
    another stream and fills ``gates`` while the sweep runs (zeroed by the caller before this call).
    out_drop + drop = (p, seed, offset): the sweep also writes dropout(out) -- the next layer's input -- there."""
    lib = _lib.load()
    ws = _lstm_ws(T, B, False, gates.device)
    p, seed, offset = drop if (drop is not None and out_drop is not None) else (0.0, 0, 0)
    if out_drop is not None:
        _req(out_drop, torch.float32, "out_drop")
        if out_drop.numel() != out.numel() or drop is None:
            raise _lib.PgasrError("lstm_layer_fwd: out_drop must have out's size and come with drop = (p, seed, offset)")
    dargs = (_p(out_drop), float(p), int(seed) & (2 ** 64 - 1), int(offset) & 0xFFFFFFFF)
    with _timed("lstm_fwd_kernel"):
        if fed is None:
            st = lib.pgasr_lstm_layer_fwd(_p(gates), _p(out), _p(cbuf), _p(pack_f), _p(lengths), T, B, _lstm_flags(pack_f), *dargs,
                                          _p(ws), ws.numel(), _stream())
        else:
            if fed.dtype != torch.int32 or not fed.is_cuda or fed.numel() < 2 * ((T * B + 255) // 256):
                raise _lib.PgasrError("lstm_layer_fwd: fed must be an int32 GPU tensor of 2 * ceil(T*B/256) words")
            st = lib.pgasr_lstm_layer_fwd_fed(_p(gates), _p(out), _p(cbuf), _p(pack_f), _p(lengths), T, B, _lstm_flags(pack_f),
                                              _p(fed), int(fed_need), *dargs, _p(ws), ws.numel(), _stream())
    _lib.check(st, "pgasr_lstm_layer_fwd_fed" if fed is not None else "pgasr_lstm_layer_fwd")
    return ws


def lstm_fed_ok(T, B):
    """Can a forward sweep of this shape be fed by a concurrent projection GEMM (B <= 32, helpers on)?"""
    return bool(_lib.load().pgasr_lstm_fed_ok(T, B, LSTM_FLAGS))


def x3w_feed_col_tiles(N, planes=2):
    """Column tiles per direction half that ``gemm_x3w_feed`` counts per row tile for an N-column product: the
    ``fed_need`` of the sweep it feeds (0: not a feedable width).  planes: of the W operand (2 / 3, see ``gemm_x3w``)."""
    lib = _lib.load()
    return int(lib.pgasr_gemm_x6w_feed_col_tiles(int(N)) if planes == 3 else lib.pgasr_gemm_x3w_feed_col_tiles(int(N)))


# Tile groups of a feeding GEMM computed IN FRONT of the fed sweep, on its own stream (pgasr_gemm_x3w_feed_f32 phase 1).
# Off by default -- measured round 3 on one box, 40 + 100 steps each: the fed sweeps get shorter (forward 1.15 / 1.11 / 1.09 ->
# 1.10 / 1.08 / 1.06 ms, backward 1.27 -> 1.23-1.25: they no longer sit ~40 us on their first rows) but the head launch itself
# takes the ~30 us it saves, on the same critical stream: 8.49 ms per step without, 8.53 / 8.50 / 8.60 with 2 / 4 / 1 groups.
FEED_HEAD_GROUPS = int(_os_environ_get("PGASR_FEED_HEAD", "0"))


# Six-product feeds (round 5): their K-split head can be a launch of its own (``gemm_x3w_feed(phase=1 / 2, ctrl=...)``,
# pgasr_gemm_x6w_feed_phase_f32) -- right behind the previous sweep instead of behind the consuming sweep's registration, gate and memset
# (the feed otherwise starts ~40 us after the previous sweep's end).  Wired into BLSTMLayerFn twice and measured (tools/dev/r5_side_head.sh, f32
# step, A/B/A/B on one box each): head on the feeding stream 10.00 / 9.90 ms against 9.67 / 9.75 (gate and persistent launch queue behind the
# head kernel: the whole tiles start ~40 us later and the sweep waits for those); head on a stream of its own with caller-zeroed queue words
# 9.79 / 9.785 (9.76 with four or six forward groups in the head) against 9.77 / 9.72 -- nothing.  The wiring was taken out again; the entry
# point stays, covered by test_gemm_x6w_feed_graded_head (same bits as the single launch).


def x3w_feed_head_items(N, K, groups=None, planes=2):
    """Work items of a feed's HEAD launch IN FRONT of the sweep on the sweep's stream (0: this shape / tile structure has none; the
    six-product feeds have none of that kind)."""
    groups = FEED_HEAD_GROUPS if groups is None else groups
    if planes == 3:
        return 0
    return int(_lib.load().pgasr_gemm_x3w_feed_head_items(int(N), int(K), int(groups))) if groups > 0 else 0


def gemm_x3w_feed(A, planes, C, M, N, K, bias, busy_ptr, tiles_done, order=0, phase=0, ws=None, ctrl=None):
    """``gemm_x3w`` in feed-ahead mode on the CURRENT stream (see include/pgasr_hip.h, pgasr_gemm_x3w_feed_f32).
    phase 1 (head, on the sweep's stream in front of the sweep) returns the workspace that the phase-2 call (rest, on
    the feeding stream) must be given as ``ws``: both work on one queue."""
    lib = _lib.load()
    for t, nm in ((A, "A"), (C, "C"), (bias, "bias")):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32):
            raise _lib.PgasrError(f"gemm_x3w_feed operand {nm} must be a float32 GPU tensor")
    _check_planes(planes, N, K, "gemm_x3w_feed")
    if tiles_done.dtype != torch.int32 or tiles_done.numel() < 2 * ((M + 255) // 256):
        raise _lib.PgasrError("gemm_x3w_feed: tiles_done must hold 2 * ceil(M/256) int32 words")
    if len(planes) == 3:
        if phase == 2 and ws is None:
            raise _lib.PgasrError("gemm_x3w_feed: phase 2 continues the queue of a phase-1 call: pass its workspace")
        if ws is None:
            ws = _workspace(lib.pgasr_gemm_x6w_feed_workspace_bytes(), C.device, "x6w_feed")
        with _timed("gemm_feed_x6c"):        # on the feeding stream: the time includes what the persistent workgroups wait for the sweep
            if ctrl is not None and (ctrl.dtype != torch.int32 or ctrl.numel() < 256 or not ctrl.is_contiguous()):
                raise _lib.PgasrError("gemm_x3w_feed: ctrl = 256 zeroed int32 words")
            st = lib.pgasr_gemm_x6w_feed_phase_f32(M, N, K, A.data_ptr(), K, *_plane_ptrs(planes), C.data_ptr(), N, _p(bias),
                                                   busy_ptr, tiles_done.data_ptr(), int(order), int(phase), _p(ctrl), _p(ws), ws.numel(), _stream())
        _lib.check(st, "pgasr_gemm_x6w_feed_phase_f32")
        return ws if phase == 1 else C
    if ws is None:
        ws = _workspace(lib.pgasr_gemm_x3w_feed_workspace_bytes(), C.device, "x3w_feed")
    with _timed("gemm_feed_x3w"):
        st = lib.pgasr_gemm_x3w_feed_f32(M, N, K, A.data_ptr(), K, planes[0].data_ptr(), planes[1].data_ptr(), C.data_ptr(), N, _p(bias),
                                         busy_ptr, tiles_done.data_ptr(), int(order), int(phase), FEED_HEAD_GROUPS if phase else 0,
                                         _p(ws), ws.numel(), _stream())
    _lib.check(st, "pgasr_gemm_x3w_feed_f32")
    return ws if phase == 1 else C


def lstm_layer_bwd(gates, out, cbuf, dout, pack_b, lengths, T, B, want_dbias=False, fed=None, fed_need=0, drop=None, slab=None):
    """gates := d(pre-activation gates).  want_dbias: also returns the (ceil(B/16), 2*4H) per-group bias-gradient
    partial sums the sweep accumulates on the way (sum its rows).
    fed: counters of a ``gemm_x3w_feed(order=1)`` launched AFTER this call that fills ``dout`` while the sweep runs;
    drop = (p, seed, offset): dout arrives without that dropout mask, the sweep's helpers apply it.
    slab: STREAMED sweep -- int32 tensor of 2 * ceil(B/16) words, zeroed by the caller, in which the sweep counts the time
    slabs (``lstm_wgrad_slabs(T)``) whose dgates rows consumers on other XCDs may read (``lstm_wgrads(slab=...)``)."""
    lib = _lib.load()
    ws = _lstm_ws(T, B, True, gates.device)
    part = torch.empty((B + 15) // 16, 2 * 4 * HID, dtype=torch.float32, device=gates.device) if want_dbias else None
    with _timed("lstm_bwd_kernel"):
        if slab is not None:
            words = slab
            if words.dtype != torch.int32 or not words.is_cuda or words.numel() < 2 * ((B + 15) // 16):
                raise _lib.PgasrError("lstm_layer_bwd: slab_done must be an int32 GPU tensor of 2 * ceil(B/16) words")
            p, seed, offset = drop if drop is not None else (0.0, 0, 0)
            st = lib.pgasr_lstm_layer_bwd_streamed(_p(gates), _p(out), _p(cbuf), _p(dout), _p(pack_b), _p(lengths), T, B, _lstm_flags(pack_b),
                                                   _p(part), _p(fed), int(fed_need), float(p), int(seed), int(offset),
                                                   _p(words), _p(ws), ws.numel(), _stream())
        elif fed is None:
            st = lib.pgasr_lstm_layer_bwd(_p(gates), _p(out), _p(cbuf), _p(dout), _p(pack_b), _p(lengths), T, B, _lstm_flags(pack_b),
                                          _p(part), _p(ws), ws.numel(), _stream())
        else:
            if fed.dtype != torch.int32 or not fed.is_cuda or fed.numel() < 2 * ((T * B + 255) // 256):
                raise _lib.PgasrError("lstm_layer_bwd: fed must be an int32 GPU tensor of 2 * ceil(T*B/256) words")
            p, seed, offset = drop if drop is not None else (0.0, 0, 0)
            st = lib.pgasr_lstm_layer_bwd_fed(_p(gates), _p(out), _p(cbuf), _p(dout), _p(pack_b), _p(lengths), T, B, _lstm_flags(pack_b),
                                              _p(part), _p(fed), int(fed_need), float(p), int(seed), int(offset),
                                              _p(ws), ws.numel(), _stream())
    _lib.check(st, "pgasr_lstm_layer_bwd")
    return (ws, part) if want_dbias else ws


def lstm_wgrad_slabs(T):
    """Frame boundaries 0 = h_0 < .. < h_n = T of the time slabs the weight-gradient products are summed over."""
    import ctypes
    lib = _lib.load()
    n = lib.pgasr_lstm_wgrad_slabs(int(T), None, 0)          # the count alone
    edges = (ctypes.c_int * (n + 1))()
    lib.pgasr_lstm_wgrad_slabs(int(T), ctypes.addressof(edges), n + 1)
    return [int(edges[i]) for i in range(n + 1)]


def lstm_wgrads_ok(T, B, in_dim, planes=None):
    """Shapes ``lstm_wgrads`` takes (the 256 x 256 TN kernels: both products in one launch): the time slabs of the sum must be whole
    k-steps of the kernel -- 16 rows for the six-product kernel of gemm_x6.hip (planes = 3, mode "f32": B % 16 == 0), 32 rows for
    gemm_c256.hip's bf16x3 kernel (planes = 2: B % 32 == 0).  planes = None: the current precision mode's."""
    planes = LSTM_PLANES if planes is None else int(planes)
    return T >= 2 and in_dim % 256 == 0 and B > 0 and B % (16 if planes == 3 else 32) == 0 and T * B * 2048 * 4 < 2 ** 31


def lstm_wgrads(dgates, x, out, T, B, in_dim, dwih, dwhh, busy_ptr=0, slab=None, err_ws=None, planes=None):
    """dwih (2*4H, in_dim) = dgates^T x and dwhh (2, 4H, H) = dgates[d]^T h_prev(d) in one launch, summed over the time slabs
    of ``lstm_wgrad_slabs(T)``.  slab: the slab_done words of the STREAMED sweep that is still writing ``dgates`` on another
    stream; err_ws: that sweep's workspace (its error word is set if a wait gives up).
    planes: 2 = bf16x3 products, 3 = the fp32-faithful six-product arithmetic; None = the current precision mode's."""
    lib = _lib.load()
    planes = LSTM_PLANES if planes is None else int(planes)
    for t_, nm in ((dgates, "dgates"), (x, "x"), (out, "out"), (dwih, "dwih"), (dwhh, "dwhh")):
        _req(t_, torch.float32, nm)
    nbytes = lib.pgasr_lstm_wgrads_workspace_bytes(T, in_dim)
    ws = _workspace(nbytes, dgates.device, "gemm")
    err_ptr = 0
    if err_ws is not None:
        import ctypes
        off = ctypes.c_size_t(0)
        _lib.check(lib.pgasr_lstm_error_offset(B, 1, ctypes.byref(off)), "pgasr_lstm_error_offset")
        err_ptr = err_ws.data_ptr() + off.value
    with _timed("gemm_t6" if planes == 3 else "gemm_t256"):      # streamed (slab != None): includes the waits for the sweep's slabs
        st = lib.pgasr_lstm_wgrads_streamed(_p(dgates), _p(x), _p(out), T, B, in_dim, _p(dwih), _p(dwhh), int(busy_ptr),
                                            _p(slab), err_ptr, planes, _p(ws), ws.numel(), _stream())
    _lib.check(st, "pgasr_lstm_wgrads_streamed")


def lstm_check_error(ws, B, backward):
    """Host-side check (synchronises): raises if a persistent sweep timed out."""
    import ctypes
    lib = _lib.load()
    off = ctypes.c_size_t(0)
    _lib.check(lib.pgasr_lstm_error_offset(B, int(backward), ctypes.byref(off)), "pgasr_lstm_error_offset")
    flag = int(ws[off.value:off.value + 4].view(torch.int32).item())
    if flag != 0:
        raise _lib.PgasrError("persistent LSTM sweep timed out waiting for its cluster (status 5)")


def lstm_assert_no_timeouts():
    """Host-side check of EVERY sweep workspace this process has used (synchronises the current stream): a persistent
    sweep that gave up on a bounded wait -- its cluster, its helpers or the GEMM feeding it never showed up -- leaves
    its sticky error word set and its results are invalid.  ``pgasr_lstm_status`` turns the word into
    PGASR_ERR_TIMEOUT, which is raised here.  Call it wherever the host synchronises anyway (the loss print every
    ``log_every`` steps, before a checkpoint is written, at the end of a benchmark); between two calls the guarded
    Adam (``lstm_error_words``) keeps invalid gradients away from the parameters."""
    lib = _lib.load()
    for key, ws in list(_ws_cache.items()):
        if not key[0].startswith("lstm"):
            continue
        st = lib.pgasr_lstm_status(ws.data_ptr(), ws.numel(), 1, int(key[0] == "lstm_bwd"), _stream())
        if st == _lib.TIMEOUT:
            raise _lib.PgasrError(f"persistent LSTM sweep timed out (status {st}: {lib.pgasr_status_string(st).decode()}) "
                                  f"in workspace {key[0]!r}; its outputs and every gradient since are invalid")
        _lib.check(st, "pgasr_lstm_status")


def lstm_error_words(device):
    """Device addresses of the sticky error words of the sweep workspaces used on the CURRENT stream (at most two:
    forward, backward) -- the guards of ``adam_step``."""
    return [w.data_ptr() for w in lstm_error_word_tensors(device)]


def lstm_error_word_tensors(device):
    """The same words as int32 (1,) views, forward workspace first.  The word's place inside a workspace comes from the library
    (``pgasr_lstm_error_offset``: the same call ``lstm_wgrads`` and ``lstm_check_error`` use), not from an assumption here."""
    import ctypes
    lib = _lib.load()
    cur = torch.cuda.current_stream().cuda_stream
    out = []
    for tag, backward in (("lstm_fwd", 0), ("lstm_bwd", 1)):
        ws = _ws_cache.get((tag, device, cur))
        if ws is None:
            continue
        off = ctypes.c_size_t(0)
        _lib.check(lib.pgasr_lstm_error_offset(1, backward, ctypes.byref(off)), "pgasr_lstm_error_offset")
        out.append(ws[off.value:off.value + 4].view(torch.int32))
    return out


# ------------------------------------------------------------------------------------------
# attention context of the reference's seq2seq decoder (model.py:58-94)
# ------------------------------------------------------------------------------------------
def attention_ctx(dec, enc):
    """dec (NQ,H) or (L,B,H) decoder states (row q belongs to utterance q % B), enc (B,T,H) -> ctx with dec's shape:
    Attention.forward of the reference as executed (include/pgasr_hip.h, pgasr_attention_ctx)."""
    lib = _lib.load()
    _req(dec, torch.float32, "dec"); _req(enc, torch.float32, "enc")
    if enc.dim() != 3 or dec.shape[-1] != enc.shape[2]:
        raise _lib.PgasrError("attention_ctx: enc (B,T,H) and dec (..., H) with the same H (model.py:69: the bmm output is (H,H))")
    B, T, H = enc.shape
    NQ = dec.numel() // H
    if NQ % B:
        raise _lib.PgasrError("attention_ctx: dec must hold a multiple of B rows")
    ctx = torch.empty_like(dec)
    _lib.check(lib.pgasr_attention_ctx(_p(dec), _p(enc), NQ, B, T, H, _p(ctx), _stream()), "pgasr_attention_ctx")
    return ctx


def lstm_cell(gh, xp, c, h, h_out=None):
    """One step of the decoder's LSTM cell in place (include/pgasr_hip.h, pgasr_lstm_cell_f32): gh, xp (B,4H) contiguous,
    c, h (B,H) updated, h_out (B,H) optional copy of the new h (the decoder's output row)."""
    lib = _lib.load()
    for n, t in (("gh", gh), ("xp", xp), ("c", c), ("h", h)):
        _req(t, torch.float32, n)
    B, H = c.shape
    if gh.shape != (B, 4 * H) or xp.shape != (B, 4 * H) or h.shape != (B, H) or not (gh.is_contiguous() and xp.is_contiguous() and c.is_contiguous() and h.is_contiguous()):
        raise _lib.PgasrError("lstm_cell: gh, xp (B,4H) and c, h (B,H), all contiguous")
    if h_out is not None:
        _req(h_out, torch.float32, "h_out")
        if h_out.shape != (B, H) or not h_out.is_contiguous():
            raise _lib.PgasrError("lstm_cell: h_out (B,H) contiguous")
    _lib.check(lib.pgasr_lstm_cell_f32(_p(gh), _p(xp), _p(c), _p(h), _p(h_out), B, H, _stream()),
               "pgasr_lstm_cell_f32")


# ------------------------------------------------------------------------------------------
# prefix beam search
# ------------------------------------------------------------------------------------------
def ctc_beam_search(log_probs, lengths=None, beam=5, blank=0, collapse=False, out=None, generic=False):
    """log_probs (T,B,V) fp32 or fp64 natural-log probabilities on the GPU.
    Returns (tokens (B,T) int32, token_lengths (B) int32, score (B) float64 = -log p).
    collapse: the returned tokens have gone through collapse_fn (adjacent duplicates removed, CTCdecoder.py:119-131),
    the form policy_grad.py:8 scores.  out = (tokens, token_lengths) to write into (tokens zero-filled).
    generic: never take the single-wave small-beam kernel (testing)."""
    lib = _lib.load()
    if not log_probs.is_cuda or log_probs.dtype not in (torch.float32, torch.float64):
        raise _lib.PgasrError("log_probs must be a float32/float64 GPU tensor")
    if log_probs.stride(2) != 1:
        raise _lib.PgasrError("log_probs must be contiguous in its last dimension")
    T, B, V = log_probs.shape
    _req(lengths, torch.int32, "lengths")
    nbytes = lib.pgasr_beam_workspace_bytes(T, B, V, beam)
    ws = _workspace(nbytes, log_probs.device, "beam")
    if out is not None:
        tokens, tl = out
        _req(tokens, torch.int32, "out tokens"); _req(tl, torch.int32, "out lengths")
        if tokens.numel() != B * T or tl.numel() != B:
            raise _lib.PgasrError("ctc_beam_search: out tensors must hold (B,T) and (B) int32")
    else:
        tokens = torch.zeros(B, T, dtype=torch.int32, device=log_probs.device)
        tl = torch.empty(B, dtype=torch.int32, device=log_probs.device)
    score = torch.empty(B, dtype=torch.float64, device=log_probs.device)
    with _timed("beam_search"):
        st = lib.pgasr_ctc_beam_search(_p(log_probs), int(log_probs.dtype == torch.float64), log_probs.stride(0),
                                       log_probs.stride(1), _p(lengths), T, B, V, int(beam), int(blank), int(bool(collapse)) | (2 if generic else 0),
                                       _p(tokens), _p(tl), _p(score), _p(ws), ws.numel(), _stream())
    _lib.check(st, "pgasr_ctc_beam_search")
    return tokens, tl, score


# ------------------------------------------------------------------------------------------
# dropout / Adam
# ------------------------------------------------------------------------------------------
def dropout(x, p, seed, offset, out=None, dact_y=None, slope=0.01):
    """Inverted dropout; with ``dact_y`` the result is also multiplied by leaky'(dact_y) (fused backward)."""
    lib = _lib.load()
    _req(x, torch.float32, "x"); _req(dact_y, torch.float32, "dact_y")
    if dact_y is not None and dact_y.numel() != x.numel():
        raise _lib.PgasrError("dropout: dact_y must have x's size")
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.pgasr_dropout(_p(x), _p(out), x.numel(), float(p), int(seed) & (2 ** 64 - 1),
                                 int(offset) & 0xFFFFFFFF, _p(dact_y), float(slope), _stream()), "pgasr_dropout")
    return out


def stream_copy(dst, src, workgroups=8):
    """dst (GPU tensor) <- src: a GPU tensor or a PINNED host tensor (mapped into the device's address space), same
    dtype and element count, both contiguous.  A kernel on the current stream; never blocks the host."""
    lib = _lib.load()
    if not dst.is_cuda or not dst.is_contiguous() or not src.is_contiguous() or dst.dtype != src.dtype or dst.numel() != src.numel():
        raise _lib.PgasrError("stream_copy: contiguous tensors of one dtype and size, destination on the GPU")
    if not src.is_cuda and not src.is_pinned():
        raise _lib.PgasrError("stream_copy: a host source must be pinned (device-mapped) memory")
    nbytes = dst.numel() * dst.element_size()
    if nbytes == 0:
        return dst
    _lib.check(lib.pgasr_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes, int(workgroups), _stream()), "pgasr_stream_copy")
    return dst


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, guards=(),
              applied=None):
    """guards: up to two device addresses of int32 words (``lstm_error_words``, or the word of the gradient buffer that
    carried the error flag through the all-reduce); the update is skipped while one is set.
    applied: int32 (2,) GPU tensor, zeroed once: the count of updates really applied lives there (bias correction then
    ignores skipped calls); ``step`` stays the 1-based count of calls."""
    lib = _lib.load()
    for t, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, torch.float32, nm)
    _req(applied, torch.int32, "applied")
    if applied is not None and applied.numel() != 2:
        raise _lib.PgasrError("adam_step: applied must hold two int32 words")
    g = list(guards)[:2] + [0, 0]
    _lib.check(lib.pgasr_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), int(step),
                                   float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                   g[0], g[1], _p(applied), _stream()), "pgasr_adam_step")


def lstm_busy_ptr(T, B, backward, device, stream=None):
    """Device address of the per-XCD busy counters inside the sweep workspace used on ``stream``
    (default: current stream) -- the hint queue-mode GEMMs read."""
    import ctypes
    lib = _lib.load()
    off = ctypes.c_size_t(0)
    _lib.check(lib.pgasr_lstm_busy_offset(B, int(backward), ctypes.byref(off)), "pgasr_lstm_busy_offset")
    if stream is None:
        ws = _lstm_ws(T, B, backward, device)
    else:
        with torch.cuda.stream(stream):
            ws = _lstm_ws(T, B, backward, device)
    return ws.data_ptr() + off.value


_concurrent = {}
STRICT_CONCURRENCY = False   # bench.py: raise instead of quietly running the (slower) sequential order


def streams_concurrent(other):
    """True iff kernels on ``other`` run while a kernel of the CURRENT stream is resident (probed once per stream pair,
    synchronises).  The feed-ahead GEMMs need this: with serialised kernels (counter-collecting profilers,
    HIP_LAUNCH_BLOCKING, one hardware queue) a fed sweep would wait for a GEMM that cannot start."""
    main = torch.cuda.current_stream()
    key = (main.cuda_stream, other.cuda_stream)
    if key not in _concurrent:
        lib = _lib.load()
        words = torch.zeros(2, dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
        torch.cuda.synchronize()
        _lib.check(lib.pgasr_stream_probe(words.data_ptr(), 200000, _stream()), "pgasr_stream_probe")
        with torch.cuda.stream(other):
            words[0:1].fill_(1)
        torch.cuda.synchronize()
        _concurrent[key] = bool(int(words[1].item()) == 1)
        if not _concurrent[key] and STRICT_CONCURRENCY:
            raise _lib.PgasrError("kernels of different streams do not run concurrently here (serialising profiler / launch-blocking "
                                  "mode / one hardware queue): the feed-ahead GEMMs would fall back to the sequential order; set "
                                  "PGASR_ALLOW_SEQUENTIAL=1 to benchmark that order knowingly")
        if not _concurrent[key]:
            import warnings
            warnings.warn("policy_gradient_asr_amd: kernels of different streams do not run concurrently here "
                          "(serialising profiler / launch-blocking mode / one hardware queue): feed-ahead GEMMs are off")
    return _concurrent[key]


GATE_OPENED_ON_BUSY, GATE_OPENED_ON_PUBLICATION, GATE_TIMED_OUT = 1, 2, 3     # report[0] of pgasr_stream_gate_report


def stream_gate(words_ptr, count=8, timeout_us=60, need=0, running=None, report=None):
    """Hold the current stream until a sweep has registered in the busy counters at ``words_ptr``; need > 0: until ``need``
    clusters have (consumers that wait for the sweep's publications: all of its workgroups must be resident first) -- or until
    the first word of ``running`` (the streamed sweep's slab_done words) is non-zero: the sweep is under way or already over.
    report (need > 0; int32 tensor of >= 2 words): the gate writes how it left (GATE_*) and the microseconds it held the stream."""
    lib = _lib.load()
    if need > 0 and report is not None:
        _req(report, torch.int32, "report")
        if report.numel() < 2:
            raise _lib.PgasrError("report needs two int32 words")
        _lib.check(lib.pgasr_stream_gate_report(words_ptr, count, int(need), _p(running), timeout_us, _p(report), _stream()),
                   "pgasr_stream_gate_report")
    elif need > 0:
        _lib.check(lib.pgasr_stream_gate_sum(words_ptr, count, int(need), _p(running), timeout_us, _stream()), "pgasr_stream_gate_sum")
    else:
        _lib.check(lib.pgasr_stream_gate(words_ptr, count, timeout_us, _stream()), "pgasr_stream_gate")
