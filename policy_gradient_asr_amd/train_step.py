"""The train-step body of model.py:225-239 for the CTC + policy-gradient objective, one process
per GPU, gradients all-reduced over RCCL (replaces nn.DataParallel, model.py:201).

Parameters and gradients live in two flat fp32 buffers (nn.Parameters are views), so the
data-parallel exchange is ONE all-reduce of the whole gradient (19.15 MB at F=80,V=29) and the
optimizer is one Adam over the flat buffer.  The loss is normalised by the GLOBAL batch so 1-GPU
and N-GPU gradients agree to fp32 rounding."""
import torch
import torch.distributed as dist

from .loss import pg_ctc_loss


def flatten_parameters(model):
    """Re-point every parameter (and its .grad) at a slice of one flat buffer."""
    params = [p for p in model.parameters()]
    dev = params[0].device
    total = sum(p.numel() for p in params)
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    gflat = torch.zeros(total, dtype=torch.float32, device=dev)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.view(-1))
            p.data = flat[off:off + n].view_as(p.data)
            p.grad = gflat[off:off + n].view_as(p.data)
            off += n
    return flat, gflat


class PolicyGradientTrainer:
    """step(batch) = H2D-ready batch -> forward -> CTC + REINFORCE loss -> backward -> all-reduce -> Adam.
    Optimizer: Adam(lr=5e-4), the reference's commented choice (model.py:207)."""

    def __init__(self, model, lr=5e-4, lam=1.0, seed=0, blank=0, world_size=1, process_group=None):
        self.model = model
        self.lam = lam
        self.seed = seed
        self.blank = blank
        self.world = world_size
        self.pg = process_group
        self.flat, self.gflat = flatten_parameters(model)
        self.flat_param = torch.nn.Parameter(self.flat)
        self.flat_param.grad = self.gflat
        self.opt = torch.optim.Adam([self.flat_param], lr=lr, fused=self.flat.is_cuda)
        self.nstep = 0
        if self.world > 1:
            dist.broadcast(self.flat, src=0, group=self.pg)

    def step(self, x, targets, fmask, tmask):
        """x (B,F,T) fp32; targets (B,L) int (pad 0); fmask (B,T); tmask (B,L).  Returns the
        detached loss tensor (no host sync)."""
        model = self.model
        B = x.shape[0]
        tg_len = tmask.sum(dim=1).to(torch.int32).contiguous()
        tg = targets.to(torch.int32).contiguous()
        self.gflat.zero_()
        logits, in_len = model.logits(x, fmask)
        loss, nll, R_s, R_g = pg_ctc_loss(logits, in_len, tg, tg_len, lam=self.lam, seed=self.seed,
                                          offset=self.nstep, global_batch=B * self.world, blank=self.blank)
        loss.backward()
        if self.world > 1:
            dist.all_reduce(self.gflat, op=dist.ReduceOp.SUM, group=self.pg)
        self.opt.step()
        self.nstep += 1
        self.last_stats = (nll, R_s, R_g)
        return loss.detach()
