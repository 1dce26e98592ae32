"""How often does the single-wave beam kernel redo a frame with the exact rounds (two lanes share the maximal high word)?  Needs the
diagnostic library (PGASR_HIP_LIB=.../libpgasr_hip_beamdiag.so, -DPGASR_BEAM_DIAG).  Three kinds of log-probs at T=1000, B=32, V=29."""
import os, sys, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
T, B, V, K = 1000, 32, 29, 512
g = torch.Generator().manual_seed(0)
cases = {}
cases["randn*2 (peaked)"] = torch.log_softmax(torch.randn(T, B, V, generator=g) * 2.0, 2).to(dev)
cases["randn*0.05 (flat)"] = torch.log_softmax(torch.randn(T, B, V, generator=g) * 0.05, 2).to(dev)
x = torch.randn(T * B, K, generator=g).to(dev); w = (torch.randn(V, K, generator=g) * 0.05).to(dev); bias = torch.full((V,), 0.1, device=dev)
cases["head kernel on random x (model-like)"] = hipops.head_logsoftmax(x, w, bias)[1].view(T, B, V)
cnt = (ctypes.c_ulonglong * 4)()
have = hasattr(lib, "pgasr_diag_beam_counters")
try:
    lib.pgasr_diag_beam_counters
except AttributeError:
    have = False
    class _Nop:
        def pgasr_diag_beam_counters(self, *a): return 0
    lib = _Nop()
for name, lp in cases.items():
    for beam in (16, 5):
        hipops.ctc_beam_search(lp, None, beam=beam); torch.cuda.synchronize()
        lib.pgasr_diag_beam_counters(cnt, 1)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); hipops.ctc_beam_search(lp, None, beam=beam); e1.record(); torch.cuda.synchronize()
        lib.pgasr_diag_beam_counters(cnt, 1)
        print(json.dumps({"log_probs": name, "beam": beam, "ms": round(e0.elapsed_time(e1), 3), "frames": cnt[0], "frames_redone": cnt[1],
                          "fraction": round(cnt[1] / max(cnt[0], 1), 4),
                          "table_probes_per_frame": round(cnt[2] / max(cnt[0], 1), 2), "longest_probe_chain": cnt[3]}), flush=True)
