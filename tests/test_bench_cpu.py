"""bench.py's host-side pieces that need no GPU: the self-launcher refuses a world it cannot realise, the synthetic
batches have the collate_custom layout, the bucketed pool deals equal counts and near-equal frames to the ranks."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_exits_nonzero_without_enough_gpus():
    if torch.cuda.device_count() >= 2:
        return
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("PGASR_BENCH_REHEARSE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "GPU(s) are visible" in r.stderr


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "does not match" in r.stderr


def test_synth_batch_layout():
    x, t, fmask, tmask = bench.synth_batch(0, [1000, 500, 737])
    assert x.shape == (3, 80, 1000) and t.shape == (3, 100) and fmask.shape == (3, 1000) and tmask.shape == (3, 100)
    assert fmask.sum(1).tolist() == [1000, 500, 737] and tmask.sum(1).tolist() == [100, 50, 73]
    assert float(x[1, :, 500:].abs().sum()) == 0 and int(t[1, 50:].abs().sum()) == 0 and int(t.min()) == 0 and int(t[0].min()) >= 1


def test_bucketed_pool_balances_ranks():
    world = 4
    pools = [bench.bucketed_pool(r, world, n_batches=8, seed=0) for r in range(world)]
    assert all(len(p) == 8 for p in pools)
    for step in range(8):
        frames = [sum(pools[r][step]) for r in range(world)]
        assert all(len(pools[r][step]) == bench.B_PER_GPU for r in range(world))
        assert all(bench.T // 2 <= n <= bench.T for r in range(world) for n in pools[r][step])
        assert max(frames) - min(frames) <= 0.01 * max(frames)          # ranks reach the all-reduce together
        # length bucketing: a batch's utterances are of similar length (half the corpus range at most)
        alln = [n for r in range(world) for n in pools[r][step]]
        assert max(alln) - min(alln) <= (bench.T // 2) // 2 + 8
