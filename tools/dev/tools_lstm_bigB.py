"""Large-batch check of the BLSTM layer forward against torch-CPU (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import functional as Fh, hipops
DEV = "cuda:0"
names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
for T, B in [(4, 64), (4, 80), (4, 96), (4, 112), (4, 128), (40, 128)]:
    g = torch.Generator().manual_seed(T * 100 + B)
    lstm = torch.nn.LSTM(512, 256, 1, bidirectional=True)
    x = torch.randn(T, B, 512, generator=g)
    out, _ = lstm(x)
    params = [getattr(lstm, n).detach().to(DEV) for n in names]
    with torch.no_grad():
        y = Fh.blstm_layer(x.to(DEV), torch.full((B,), T, dtype=torch.int32, device=DEV), params)
    torch.cuda.synchronize()
    ws = hipops._lstm_ws(T, B, False, torch.device(DEV))
    err = int(ws[:4].view(torch.int32).item())
    print(f"T={T} B={B}: rel err {float((y.cpu() - out).abs().max() / out.abs().max()):.2e}  err word {err}", flush=True)
