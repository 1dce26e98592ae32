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


def test_gpu_count_comes_from_sysfs_not_from_hip(tmp_path, monkeypatch):
    """The parent of a multi-rank run counts GPUs in the KFD topology (simd_count > 0) and applies the *_VISIBLE_DEVICES
    lists; without sysfs it leaves the check to the ranks.  bench.launch_ranks never calls torch.cuda."""
    import inspect
    assert "torch.cuda" not in inspect.getsource(bench.launch_ranks) and "torch.cuda" not in inspect.getsource(bench.visible_gpu_count)
    nodes = tmp_path / "nodes"
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text(f"cpu_cores_count 0\nsimd_count {simd}\nmem_banks_count 1\n")
    real_listdir, real_open = os.listdir, open
    monkeypatch.setattr(bench.os, "listdir", lambda p: real_listdir(str(nodes)) if "kfd" in p else real_listdir(p))
    import builtins
    monkeypatch.setattr(builtins, "open", lambda p, *a, **k: real_open(str(p).replace("/sys/class/kfd/kfd/topology/nodes", str(nodes)), *a, **k))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count() == 2


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


def test_roofline_record_counts_launches_per_sampled_step():
    """launches_per_step is launches / SAMPLED steps (only every n-th timed step carries HIP events), 3 per
    step for a three-layer model, and the achieved rate follows from the average launch time."""
    prof = {"lstm_fwd_kernel": (5 * 3 * 1.10, 15), "lstm_bwd_kernel": (5 * 3 * 1.30, 15), "gemm_f32": (9.0, 100)}
    rl = bench.sweep_roofline(prof, n_sampled=5, frames_per_step=bench.B_PER_GPU * bench.T)
    assert rl["kernel"] == "lstm_bwd_kernel" and rl["launches_per_step"] == 3
    assert abs(rl["avg_launch_ms"] - 1.30) < 1e-12
    assert abs(rl["flops_per_launch"] - 2 * 2 * 32 * 256 * 1024 * 1000) < 1
    assert abs(rl["achieved"] - rl["flops_per_launch"] / 1.30e-3 / 1e12) < 1e-9 and abs(rl["peak"] - 2500 / 3) < 1e-9


def test_committed_bench_line_keeps_the_contract():
    """profiles/r05_bench.json is one line of `python bench.py` on an MI355X (default precision f32): the driver's contract fields, the
    roofline (latency-bound sweep with its hand-off floor), the MFMA-bound kernels' roofline_gemm, the CPU baseline and the round's extra
    legs are all there and consistent with each other; profiles/r05_bench_bf16x3.json is the same command with --precision bf16x3."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_gemm", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == bench.METRIC and d["dtype"].startswith("f32") and d["precision"] == "f32"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6           # utterances/sec of B = 32 per GPU
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "latency" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] > 0
    assert abs(r["peak"] - 2500.0 / 6) < 1e-9                                              # six bf16 products per flop in the f32 mode
    assert r["launches_per_step"] == 3.0 and r["handoff_floor_us"] == 0.5
    assert abs(r["per_step_us"] - r["avg_launch_ms"]) < 1e-9                               # T = 1000 steps per launch
    assert abs(r["floor_frac"] - 0.5 / r["per_step_us"]) < 1e-9 and 0.2 < r["floor_frac"] < 0.6
    g = d["roofline_gemm"]
    assert g["bound"] == "mfma" and g["peak"] == 2500.0 and g["products_per_flop"] == 6
    for role in ("input_projection", "input_gradient", "weight_gradient_ih"):
        k = g["kernels"][role]
        assert abs(k["issued_bf16_tflops"] - 6 * k["gflop_fp32"] / k["avg_launch_us"] * 1e3) < 1e-6
        assert abs(k["frac_of_2500"] - k["issued_bf16_tflops"] / 2500.0) < 1e-9 and 0.3 < k["frac_of_2500"] < 0.8
        assert 0.4 < k["pmc_mfma_busy"] < 1.0 and "profiles/" in k["pmc_source"]
        assert abs(k["frac_of_sustained"] - k["issued_bf16_tflops"] / g["sustained_bare_mfma_tflops"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    for leg in ("long_run", "inputs_resident", "bucketed", "precision_variants", "max_rel_err_vs_fp64"):
        assert leg in d, leg
    assert d["inputs_resident"]["ms_per_step"] <= d["long_run"]["ms_per_step"] * 1.02      # staging costs time, never saves it
    assert d["max_rel_err_vs_fp64"]["param_grads_frobenius"] < 1e-5 and d["max_rel_err_vs_fp64"]["loss"] < 1e-5
    pv = d["precision_variants"]
    assert pv["f32"]["max_rel_err_vs_fp64"]["param_grads_frobenius"] < pv["bf16x3"]["max_rel_err_vs_fp64"]["param_grads_frobenius"] < 1e-3
    assert pv["f32"]["ms_per_step"] <= 11.0 and pv["bf16x3"]["ms_per_step"] < pv["f32"]["ms_per_step"]
    b = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_bf16x3.json")))
    assert b["dtype"].startswith("bf16x3") and abs(b["roofline"]["peak"] - 2500.0 / 3) < 1e-9 and b["roofline_gemm"]["products_per_flop"] == 3
    assert b["ms_per_step"] < d["ms_per_step"] and b["precision"] == "bf16x3"
    assert d["bucketed"]["ms_per_step"] < 12.6            # configs[4]: the beam-16 reward hypothesis (round 4: 13.06 ms)
    # the N > 1 plumbing rehearsal (two ranks on ONE GPU over gloo) ran every leg with world 2 and says what it is
    reh = json.load(open(os.path.join(ROOT, "profiles", "r05_rehearse.json")))
    assert reh["n_gpus"] == 2 and reh["rccl_world_size"] == 2 and "NOT a scaling number" in reh["collective_backend"]
    for leg in ("long_run", "inputs_resident", "bucketed"):
        assert reh[leg]["ms_per_step"] > 0
