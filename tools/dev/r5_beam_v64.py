"""The single-wave beam kernel at 8 and 16 symbols per lane (round 5: V <= 64 on the in-step kernel): ms for 32 utterances x T = 1000 at
V = 29 / 32 / 33 / 40 / 64, beam 16 and 5, model-like and flat log-probs, beside the generic kernel (development aid)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
T, B = 1000, 32
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 3)
g = torch.Generator().manual_seed(0)
for V in (29, 32, 33, 40, 64):
    for kind, scale in (("model-like", 1.0), ("flat", 0.05)):
        z = torch.randn(T, B, V, generator=g) * 2.0 * scale
        z[:, :, 0] += 1.5 * scale
        lp = torch.log_softmax(z, 2).to(dev)
        row = {"V": V, "logits": kind}
        for beam in (16, 5):
            row[f"beam{beam}_ms"] = t(lambda: hipops.ctc_beam_search(lp, None, beam=beam, collapse=True))
        if kind == "model-like":
            row["generic_beam16_ms"] = t(lambda: hipops.ctc_beam_search(lp, None, beam=16, collapse=True, generic=True), 1)
            a = hipops.ctc_beam_search(lp, None, beam=16, collapse=True); b = hipops.ctc_beam_search(lp, None, beam=16, collapse=True, generic=True)
            row["equal_to_generic"] = bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
        print(json.dumps(row), flush=True)
