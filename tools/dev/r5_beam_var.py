import os, sys, json, torch
sys.path.insert(0, os.getcwd())
from policy_gradient_asr_amd import hipops
dev = torch.device("cuda:0")
T, B, V, K = 1000, 32, 29, 512
g = torch.Generator().manual_seed(0)
x = torch.randn(T * B, K, generator=g).to(dev); w = (torch.randn(V, K, generator=g) * 0.05).to(dev); bias = torch.full((V,), 0.1, device=dev)
lp = hipops.head_logsoftmax(x, w, bias)[1].view(T, B, V)
il = torch.full((B,), T, dtype=torch.int32, device=dev)
def t(fn, reps):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 3)
for name, fn in (("None,nocollapse", lambda: hipops.ctc_beam_search(lp, None, beam=16)),
                 ("il,nocollapse", lambda: hipops.ctc_beam_search(lp, il, beam=16)),
                 ("None,collapse", lambda: hipops.ctc_beam_search(lp, None, beam=16, collapse=True)),
                 ("il,collapse", lambda: hipops.ctc_beam_search(lp, il, beam=16, collapse=True))):
    print(name, "1 rep:", t(fn, 1), "5 reps:", t(fn, 5), "1 rep:", t(fn, 1), flush=True)
import ctypes
from policy_gradient_asr_amd import _lib
lib = _lib.load()
if hasattr(lib, "pgasr_diag_beam_counters"):
    try:
        cnt = (ctypes.c_ulonglong * 4)()
        lib.pgasr_diag_beam_counters(cnt, 1)
        tok, tl, sc = hipops.ctc_beam_search(lp, il, beam=16, collapse=True); torch.cuda.synchronize()
        lib.pgasr_diag_beam_counters(cnt, 1)
        # per-utterance cycles per frame (diagnostic flag bit 2 of the entry point: out_score = cycles / frame)
        nb_ = lib.pgasr_beam_workspace_bytes(T, B, V, 16); ws_ = torch.empty(nb_, dtype=torch.uint8, device=dev)
        tk_ = torch.zeros(B, T, dtype=torch.int32, device=dev); tl_ = torch.empty(B, dtype=torch.int32, device=dev); sc_ = torch.empty(B, dtype=torch.float64, device=dev)
        lib.pgasr_ctc_beam_search(lp.data_ptr(), 0, lp.stride(0), lp.stride(1), il.data_ptr(), T, B, V, 16, 0, 1 | 4, tk_.data_ptr(), tl_.data_ptr(), sc_.data_ptr(),
                                  ws_.data_ptr(), nb_, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        cyc = sc_.cpu().tolist()
        print("cycles per frame by utterance: min %.0f median %.0f max %.0f" % (min(cyc), sorted(cyc)[len(cyc) // 2], max(cyc)), [round(c) for c in cyc[:8]])
        print("counters: frames", cnt[0], "redone", cnt[1], "probes/frame", round(cnt[2] / max(cnt[0], 1), 2), "longest chain", cnt[3], "hyp lens", tl.tolist()[:8])
    except AttributeError:
        pass
