"""Data-parallel plumbing on CPU with the gloo backend, world_size 2 (the N>1 path of bench.py /
train_step.py): flat-buffer all-reduce + global-batch loss normalisation give the single-process
global-batch gradient and identical replicas."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from policy_gradient_asr_amd.train_step import (FLAG_PAD, DataParallelStep, balance_by_frames, flatten_parameters,
                                                shard_slice)


class ToyStep(DataParallelStep):
    def forward_loss(self, batch, global_batch):
        x, y = batch
        # per-utterance losses summed and divided by the GLOBAL batch
        return ((self.model(x) - y) ** 2).sum() / global_batch


class ToyStepTwoBuckets(ToyStep):
    """The N>1 trainer's bucket split: the upper bucket's all-reduce is started (async) before the lower one."""

    def backward(self, loss):
        loss.backward()
        self.reduce_upper(self.param_offset("2.weight"))
        assert self._early is not None and self._early[0] == FLAG_PAD + 6 * 5 + 5


class ToyStepOneRankFails(ToyStepTwoBuckets):
    """Rank 1 reports invalid gradients (a timed-out sweep) in its SECOND step only: the flag must reach every rank through
    the gradient all-reduce, so that both replicas skip that update together."""
    fail_rank, fail_call = 1, 1

    def write_local_error_flag(self):
        bad = dist.get_rank() == self.fail_rank and self.nstep == self.fail_call
        self.gflat[0] = 1.0 if bad else 0.0          # what pgasr_error_flag writes on the device
        return True


def make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


def make_data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)


def _worker(rank, world, port, q, two_buckets):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = make_model()
        if rank == 1:   # replicas start different: the trainer must broadcast rank 0's weights
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        st = (ToyStepTwoBuckets if two_buckets else ToyStep)(model, lr=1e-2, world_size=world)
        x, y = make_data()
        sl = shard_slice(8, rank, world)
        losses = [float(st.step(x[sl], y[sl])) for _ in range(3)]
        q.put((rank, st.flat.tolist(), st.gflat.tolist(), losses))   # by value: the worker exits right after
    finally:
        dist.destroy_process_group()


def _worker_flag(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = ToyStepOneRankFails(make_model(), lr=1e-2, world_size=world)
        x, y = make_data()
        sl = shard_slice(8, rank, world)
        snaps = []
        for _ in range(3):
            st.step(x[sl], y[sl])
            snaps.append(st.flat.tolist())
        q.put((rank, snaps, st.applied_steps(), st.nstep))
    finally:
        dist.destroy_process_group()


def test_error_flag_on_one_rank_skips_the_update_on_every_rank():
    """the Adam guard must be global.  One rank flags its second step; both ranks must leave their
    parameters untouched in that step, count two applied updates out of three calls, and stay bit-identical replicas."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_flag, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, snaps, applied, calls in res:
        assert calls == 3 and applied == 2, (rank, calls, applied)
        assert snaps[1] == snaps[0]              # the flagged step changed nothing, on the rank that did NOT fail too
        assert snaps[2] != snaps[1]              # the next step is applied again
    assert res[0][1] == res[1][1]                # replicas bit-identical after every step


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("two_buckets", [False, True])
def test_two_rank_gloo_matches_single_process(two_buckets):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, two_buckets)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process, whole batch
    st = ToyStep(make_model(), lr=1e-2, world_size=1)
    x, y = make_data()
    ref_losses = [float(st.step(x, y)) for _ in range(3)]
    res = [(r, torch.tensor(f), torch.tensor(g), l) for r, f, g, l in res]
    for rank, flat, gflat, losses in res:
        torch.testing.assert_close(flat, st.flat, rtol=1e-5, atol=1e-6)       # same parameters after 3 steps
        torch.testing.assert_close(gflat, st.gflat, rtol=1e-5, atol=1e-6)     # summed grad = global-batch grad
    # local losses add up to the global loss
    for i in range(3):
        assert res[0][3][i] + res[1][3][i] == pytest.approx(ref_losses[i], rel=1e-5)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=0, atol=0)          # replicas bit-identical


def test_flatten_parameters_views():
    m = make_model()
    before = [p.detach().clone() for p in m.parameters()]
    flat, gflat = flatten_parameters(m)
    assert flat.numel() == sum(p.numel() for p in m.parameters())
    flat_p, _ = flatten_parameters(make_model(), FLAG_PAD)
    assert flat_p.numel() == flat.numel() + FLAG_PAD and float(flat_p[:FLAG_PAD].abs().sum()) == 0.0
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p.detach(), b)
        assert p.data_ptr() >= flat.data_ptr() and p.grad.data_ptr() >= gflat.data_ptr()
    m(torch.ones(2, 6)).sum().backward()
    assert gflat.abs().sum() > 0      # autograd accumulated into the flat buffer


def test_shard_and_balance():
    assert shard_slice(32, 3, 8) == slice(12, 16)
    with pytest.raises(ValueError):
        shard_slice(30, 0, 8)
    lengths = [1000, 990, 500, 510, 700, 720, 100, 900]
    parts = balance_by_frames(lengths, 4)
    assert sorted(i for p in parts for i in p) == list(range(8)) and all(len(p) == 2 for p in parts)
    loads = [sum(lengths[i] for i in p) for p in parts]
    contiguous = [sum(lengths[i:i + 2]) for i in range(0, 8, 2)]
    assert max(loads) < max(contiguous) and max(loads) <= 1490
