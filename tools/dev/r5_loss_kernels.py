"""The loss section's kernels (A4 head, A5 CTC lattice + gradient, A9 / A12 frame arg-max + sampler, collapse, A10 edit distance, A7
beam-16 reward hypothesis) stand-alone at the headline shape (T=1000, B=32, V=29, L=100), a few launches each: the target of the
FETCH_SIZE / WRITE_SIZE counter passes of tools/profile_round5.sh (SURVEY §8d asks for their HBM bytes) and, un-profiled, a timing line."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops

dev = torch.device("cuda:0")
T, B, V, L, K = 1000, 32, 29, 100, 512
g = torch.Generator().manual_seed(0)
x = torch.randn(T * B, K, generator=g).to(dev)
w = (torch.randn(V, K, generator=g) * 0.05).to(dev); bias = torch.full((V,), 0.1, device=dev)
targets = torch.randint(1, V, (B, L), generator=g, dtype=torch.int32).to(dev)
il = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), L, dtype=torch.int32, device=dev)
reps = int(os.environ.get("REPS", "5"))
times = {}


def timed(name, fn):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    times[name] = round(e0.elapsed_time(e1) / reps * 1e3, 1)
    return out


logits, lp = timed("head_logsoftmax", lambda: hipops.head_logsoftmax(x, w, bias))
lp = lp.view(T, B, V)
nll, lattice = timed("ctc_lattice", lambda: hipops.ctc_lattice(lp, targets, il, tl))
greedy, sample = timed("frame_argmax_sample", lambda: hipops.frame_argmax_sample(lp, seed=1, offset=1))
coef = torch.full((B,), 0.01, device=dev); scale = torch.full((B,), 1.0 / (L * B), device=dev)
timed("ctc_grad_from_lattice", lambda: hipops.ctc_grad_from_lattice(lp, il, tl, lattice, utt_scale=scale, pg_coef=coef, pg_path=sample))
paths = torch.stack((greedy, sample), 0)
tok, tlen = timed("ctc_collapse", lambda: hipops.ctc_collapse(paths, il))
timed("edit_distance", lambda: hipops.edit_distance(targets.repeat(2, 1), tl.repeat(2), tok.view(2 * B, T), tlen.view(2 * B)))
timed("beam16", lambda: hipops.ctc_beam_search(lp, il, beam=16, collapse=True))
timed("beam5", lambda: hipops.ctc_beam_search(lp, il, beam=5, collapse=True))
print(json.dumps({"shape": {"T": T, "B": B, "V": V, "L": L}, "us_per_launch": times}), flush=True)
