"""What would ONE bf16 product (instead of the 3-term split) on the hoisted W_ih projections cost in accuracy?  (BASELINE configs[4]
names "bf16 MFMA projections"; development aid, measurement only.)

Emulation on the real kernels: the weight planes' lo part is zeroed and the activation operand is rounded to bf16 before every
x3w product (input projections X W_ih^T and their input gradients dG W_ih), so the kernels compute exactly hi x hi with fp32
accumulation; the recurrent products, the weight gradients and everything else stay bf16x3.  Prints the error of the loss and of
the parameter gradients of one eval-mode CTC step at the headline shape against torch-CPU fp64 (bench.parity_vs_fp64), next to the
default mode's on the same weights."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from policy_gradient_asr_amd import hipops
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Seq2Seq(bench.V, n_feats=bench.F); model.apply(weights); model = model.to(dev).train()
trainer = PolicyGradientTrainer(model, lr=5e-4, lam=1.0, seed=1234)
batch = [t.to(dev) for t in bench.synth_batch(100)]
for _ in range(int(os.environ.get("TRAIN_STEPS", "20"))):      # a few steps away from the initialisation
    trainer.step(*batch)
torch.cuda.synchronize()
ref = bench.parity_vs_fp64(model, trainer, batch, brief=True)
print("bf16x3 (default)      :", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in ref.items()}, flush=True)
with hipops.precision("f32"):
    f32 = bench.parity_vs_fp64(model, trainer, batch, brief=True)
print("f32 mode              :", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in f32.items()}, flush=True)

_split, _x3w, _feed = hipops.split_planes, hipops.gemm_x3w, hipops.gemm_x3w_feed
def split1(w, transpose=False):
    hi, lo = _split(w, transpose)
    return hi, torch.zeros_like(lo)
def r16(a):
    return a.to(torch.bfloat16).to(torch.float32)
hipops.split_planes = split1
hipops.gemm_x3w = lambda A, planes, C, M, N, K, **kw: _x3w(r16(A), planes, C, M, N, K, **kw)
hipops.gemm_x3w_feed = lambda A, planes, C, M, N, K, bias, busy, done, **kw: _feed(r16(A), planes, C, M, N, K, bias, busy, done, **kw)
import policy_gradient_asr_amd.functional as Fh
Fh.FEED_AHEAD = False          # the rounded copy of A is a temporary: keep the products in stream order
bench._fp64_ref.clear()
one = bench.parity_vs_fp64(model, trainer, batch, brief=True)
print("bf16x1 on projections :", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in one.items()}, flush=True)
