"""The N > 1 path on real hardware: two ranks, one GPU each, the gradient all-reduce over RCCL (backend "nccl") in the
trainer's two-bucket form.  Skipped on a one-GPU box (RCCL refuses two ranks on one device); tests/test_dp_gloo_cpu.py
covers the same plumbing on the CPU.  The assertion is the gloo test's: summed gradients and updated parameters equal
the single-process global batch (model.py:201's nn.DataParallel semantics), replicas bit-identical."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("lam", [0.0, 1.0])
def test_two_rank_rccl_step_matches_single_process_global_batch(tmp_path, lam):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL does not take two ranks on one device)")
    world, port = 2, _free_port()
    worker = os.path.join(ROOT, "tests", "dp_rccl_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(tmp_path), str(lam)])
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    assert all(r["world"] == world for r in res)
    assert torch.equal(res[0]["flat"], res[1]["flat"])                 # replicas bit-identical after two steps
    assert torch.equal(res[0]["gflat"], res[1]["gflat"])               # both hold the all-reduced gradient
    if True:
        # compare with ONE process holding the whole batch -- for lambda = 1 too: all ranks share the sampling seed and
        # address their draws by global utterance index (pgasr_frame_argmax_sample ctr_stride / ctr_base), so the sampled
        # paths, the rewards and the REINFORCE term of the two shards are those of the global batch
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import dp_rccl_worker as w
        dev = torch.device("cuda", 0)
        tr = w.build(dev, 1, 0, lam)
        batch = [t.to(dev) for t in w.make_batch(8 * world, 80, 60, 29, 6)]
        ref_loss = float(tr.step(*batch))
        torch.cuda.synchronize()
        # first step: the summed gradient is the global-batch gradient to fp32 rounding (Adam's first update is
        # lr * sign(g), so parameters are compared through the gradient, not after the update)
        g, want = res[0]["gflat_step1"], tr.gflat.cpu()
        assert float((g - want).abs().max() / want.abs().max()) < 1e-5
        assert res[0]["losses"][0] + res[1]["losses"][0] == pytest.approx(ref_loss, rel=1e-5)   # local losses are / global batch
    assert all(torch.isfinite(r["flat"]).all() for r in res)
