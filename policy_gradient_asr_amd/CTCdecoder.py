"""Drop-in for the reference's CTCdecoder.py: ``CTCDecoder(alphabet).decode(probs, beam_size=100,
blank=0) -> (tuple[int], float)`` and ``collapse_fn(str) -> str``, with the search itself running
as a HIP kernel (csrc/beam.hip).  ``greedy_decode`` is the best-path decoder the reference lacks
(SURVEY §8a A9)."""
import numpy as np
import torch

from . import hipops


def _device(device=None):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError("policy_gradient_asr_amd needs the MI355X: there is no CPU decoder")
    return torch.device("cuda", torch.cuda.current_device())


BEAM_KMAX = 128      # csrc/beam.hip


class CTCDecoder:
    def __init__(self, alphabet, device=None):
        self.alphabet = alphabet
        self.NEG_INF = -float("inf")
        self.device = device

    def decode(self, probs, beam_size=100, blank=0):
        """probs: (time x output dim) array of PROBABILITIES (CTCdecoder.py:41-53).
        Returns (label tuple, negative log-likelihood of that prefix).
        Limits of the device search (the reference has none): beam_size <= 128 and at most 64 output symbols --
        a larger request raises instead of silently searching a narrower beam."""
        dev = _device(self.device)
        probs = np.asarray(probs)
        T, V = probs.shape
        if T == 0:
            return tuple(), -0.0
        with np.errstate(divide="ignore"):
            logp = np.log(probs.astype(np.float64))          # like CTCdecoder.py:55
        lp = torch.from_numpy(np.ascontiguousarray(logp)).to(dev).view(T, 1, V)
        if int(beam_size) > BEAM_KMAX:
            raise ValueError(f"beam_size {beam_size} exceeds the device search's limit of {BEAM_KMAX}")
        tokens, tl, score = hipops.ctc_beam_search(lp, None, beam=int(beam_size), blank=int(blank))
        n = int(tl[0].item())
        return tuple(int(x) for x in tokens[0, :n].tolist()), float(score[0].item())

    def decode_batch(self, log_probs, lengths=None, beam_size=5, blank=0):
        """Device-side batched form: log_probs (T,B,V) GPU tensor of natural-log probabilities.
        Returns (tokens (B,T) int32, lengths (B) int32, nll (B) float64) without a host sync."""
        return hipops.ctc_beam_search(log_probs, lengths, beam=int(beam_size), blank=int(blank))


def collapse_fn(preds):
    """Remove adjacent duplicate characters of an already-decoded string (CTCdecoder.py:119-131):
    'aabbcc' -> 'abc', '' -> ''.  Pure host string work."""
    out = []
    for ch in preds:
        if not out or ch != out[-1]:
            out.append(ch)
    return "".join(out)


def greedy_decode(scores, lengths=None, blank=0):
    """Best-path decode on the device: scores (T,B,V) GPU tensor (logits or log-probs) ->
    (tokens (B,T) int32, token_lengths (B) int32): argmax per frame (first max wins), collapse
    repeats, drop blank."""
    T, B, V = scores.shape
    if lengths is None:
        lengths = torch.full((B,), T, dtype=torch.int32, device=scores.device)
    greedy, _ = hipops.frame_argmax_sample(scores.contiguous().float(), want_sample=False)
    tokens, tl = hipops.ctc_collapse(greedy[None].contiguous(), lengths.to(torch.int32).contiguous(), blank=blank)
    return tokens[0], tl[0]
