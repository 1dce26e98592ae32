"""One rank of tests/test_dp_rccl_gpu.py (not a test module itself): the real trainer on its own GPU, gradients
all-reduced over RCCL.  argv: rank world port out_dir lam"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def make_batch(global_b, F, T, V, L, seed=7):
    g = torch.Generator().manual_seed(seed)
    lens = [T - (3 * b) % (T // 2) for b in range(global_b)]
    tlens = [max(1, n // 10) for n in lens]
    x = torch.randn(global_b, F, T, generator=g)
    targets = torch.randint(1, V, (global_b, L), generator=g)
    fmask = torch.zeros(global_b, T)
    tmask = torch.zeros(global_b, L, dtype=torch.int64)
    for b, (n, m) in enumerate(zip(lens, tlens)):
        fmask[b, :n] = 1; x[b, :, n:] = 0; tmask[b, :m] = 1; targets[b, m:] = 0
    return x, targets, fmask, tmask


def build(dev, world, rank, lam, pg=None):
    from policy_gradient_asr_amd.model import Seq2Seq, weights
    from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
    torch.manual_seed(0)
    m = Seq2Seq(29, n_feats=80)
    m.apply(weights)
    if rank == 1:      # replicas start different: the trainer must broadcast rank 0's weights
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.5)
    m = m.to(dev).eval()
    return PolicyGradientTrainer(m, lr=1e-3, lam=lam, seed=11, world_size=world, rank=rank, process_group=pg)


def main():
    rank, world, port, out_dir, lam = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], float(sys.argv[5])
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    from policy_gradient_asr_amd import streams
    streams.prime()          # before the communicator takes its stream from torch's pool
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    try:
        from policy_gradient_asr_amd.train_step import shard_slice
        tr = build(dev, world, rank, lam)
        batch = make_batch(8 * world, 80, 60, 29, 6)
        sl = shard_slice(8 * world, rank, world)
        mine = [t[sl].to(dev) for t in batch]
        losses = [float(tr.step(*mine))]
        torch.cuda.synchronize()
        g1 = tr.gflat.cpu().clone()
        losses.append(float(tr.step(*mine)))
        torch.cuda.synchronize()
        torch.save({"flat": tr.flat.cpu(), "gflat": tr.gflat.cpu(), "gflat_step1": g1, "losses": losses,
                    "world": dist.get_world_size()}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
