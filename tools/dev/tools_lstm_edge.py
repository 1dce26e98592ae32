"""Edge shapes of the BLSTM layer against torch-CPU (development aid; the durable cases live in tests/)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
from policy_gradient_asr_amd import functional as Fh
DEV = "cuda:0"
def rel(a, b): return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
for T, B, lens in [(1, 1, [1]), (1, 5, [1] * 5), (2, 3, [2, 1, 2]), (3, 17, [3] * 9 + [1] * 8), (7, 48, [7] * 20 + [4] * 28), (4, 128, [4] * 128), (33, 33, list(range(33, 0, -1)))]:
    g = torch.Generator().manual_seed(T * 100 + B)
    lstm = torch.nn.LSTM(512, 256, 1, bidirectional=True)
    x = torch.randn(T, B, 512, generator=g); dy = torch.randn(T, B, 512, generator=g)
    lengths = torch.tensor(lens, dtype=torch.int64)
    for b, n in enumerate(lens): dy[n:, b] = 0
    xr = x.clone().requires_grad_(True)
    out, _ = lstm(pack_padded_sequence(xr, lengths, enforce_sorted=False))
    out, _ = pad_packed_sequence(out, total_length=T)
    out.backward(dy)
    params = [getattr(lstm, n).detach().to(DEV).requires_grad_(True) for n in names]
    xg = x.to(DEV).requires_grad_(True)
    y = Fh.blstm_layer(xg, lengths.to(torch.int32).to(DEV), params)
    y.backward(dy.to(DEV)); torch.cuda.synchronize()
    errs = [rel(y.detach().cpu(), out.detach()), rel(xg.grad.cpu(), xr.grad)] + [rel(p.grad.cpu(), getattr(lstm, n).grad) for n, p in zip(names, params)]
    print(f"T={T} B={B}: max rel err {max(errs):.2e}", flush=True)
