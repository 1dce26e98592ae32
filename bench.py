#!/usr/bin/env python3
"""bench.py --gpus N --steps K --warmup W : utterances/sec of one policy-gradient train step
(BASELINE.json metric) on synthetic (B=32 per GPU, T=1000, F=80, V=29) data, one process per GPU.

`python3 bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks ITSELF
(fresh child processes, before this process makes any GPU call); under torchrun (WORLD_SIZE set) it is
one of the ranks.  Rank 0 prints ONE JSON line (contract in the task statement).

Workloads:
  headline  configs[2]+[1]: CTC + REINFORCE step, greedy (self-critical) baseline, lengths = T
  bucketed  configs[4]   : the reference's own reward hypothesis (prefix beam search, beam 16,
                           policy_grad.py:6-8) + variable lengths U[T/2,T] in length-bucketed batches,
                           ranks balanced by frames
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The step uses three streams (main, weight-gradient side stream, CTC lattice) and RCCL adds its own: HIP's default of
# 4 hardware queues makes streams share a queue and serialise behind each other.  Read by the runtime at start-up.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL across processes)

import torch  # noqa: E402  (importing torch does not touch the GPU)
import torch.distributed as dist  # noqa: E402

B_PER_GPU, T, F, V, L = 32, 1000, 80, 29, 100
METRIC = "utterances/sec (B=32,T=1000,F=80) policy-grad step, 1/2/4/8 MI355X"
# The arithmetic of the timed step is a mode of the host layer (hipops.PRECISION_MODES).  The DEFAULT is "f32": the reference computes in
# torch fp32 (model.py:38-44), and that is what the headline `value` measures; "bf16x3" (within north_star's 1e-3 bar, ~1.35x
# faster) is timed beside it in precision_variants.
DTYPES = {
    "f32": ("f32 (fp32-faithful: fp32 storage and accumulate; every big product -- recurrent sweeps, W_ih projections, input and weight "
            "gradients -- on the bf16 MFMA as SIX terms of a 3-plane split hi/mid/lo, every term down to 2^-24 of the product; the small "
            "products on the exact fp32 MFMA; the faster bf16x3 mode of the same step is timed beside it: precision_variants.bf16x3)"),
    "bf16x3": ("bf16x3 (fp32 storage and accumulate; every dense product = 3 bf16 MFMA terms hi*hi+hi*lo+lo*hi of a 2-plane split; the "
               "fp32-faithful mode of the same step is timed beside it: precision_variants.f32)"),
}
PRODUCTS = {"f32": 6, "bf16x3": 3}       # bf16 MFMA products issued per algorithmic fp32 multiply-add
BF16_DENSE_PEAK_TF = 2500.0       # MI355X_MICROARCH.md, dense
FP32_MFMA_PEAK_TF = 157.3
STEP_GFLOP_PER_UTT = 28.65        # dense contraction forward + backward, T=1000


# ------------------------------------------------------------------------------------------------
# self-launch: N ranks as child processes of a parent that never touches the GPU
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpu_count():
    """GPUs this process may use, WITHOUT any HIP / ROCr / amdsmi call: the KFD topology in sysfs (a GPU node has
    simd_count > 0, a CPU node 0), cut down by the *_VISIBLE_DEVICES lists.  None when sysfs does not say (the children
    then validate their own device index).  The parent of a multi-rank run must never initialise the GPU: a process that
    has may not start children that exec."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as fi:
                for line in fi:
                    k, _, v = line.partition(" ")
                    if k == "simd_count":
                        n += int(v) > 0
                        break
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        lst = os.environ.get(var)
        if lst is not None:
            n = min(n, len([x for x in lst.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n):
    """Parent of a `--gpus n` run without torchrun: start n children (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set), pass
    their output through, exit with the first non-zero code.  No HIP call is made here -- the GPU count comes from
    sysfs (visible_gpu_count) -- and nothing is exec'ed from a process that has initialised the GPU."""
    rehearse = os.environ.get("PGASR_BENCH_REHEARSE", "") == "1"
    ndev = visible_gpu_count()
    if ndev is not None and ndev < n and not rehearse:
        print(f"[bench] --gpus {n} but only {ndev} GPU(s) are visible", file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PGASR_BENCH_CHILD="1")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's peer mappings need it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                deadline = time.time() + 20          # a dead rank leaves the others stuck in a collective
                for q in alive:
                    q.terminate()
                for q in alive:
                    try:
                        q.wait(max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
    return rc


# ------------------------------------------------------------------------------------------------
# synthetic batches: features N(0,1), targets U{1..V-1}, L = T/10
# ------------------------------------------------------------------------------------------------
def synth_batch(seed, lengths=None, pin=False):
    """Host-side batch in the collate_custom layout after model.py:227-230's squeeze: x (B,F,Tmax) fp32 zero past each
    length, targets (B,Lmax) int64 pad 0, fmask (B,Tmax), tmask (B,Lmax)."""
    g = torch.Generator().manual_seed(seed)
    if lengths is None:
        lengths = [T] * B_PER_GPU
    nb = len(lengths)
    tmax = max(lengths)
    tl = [max(1, n // 10) for n in lengths]
    lmax = max(tl)
    x = torch.randn(nb, F, tmax, generator=g)
    targets = torch.randint(1, V, (nb, lmax), generator=g)
    fmask = torch.zeros(nb, tmax)
    tmask = torch.zeros(nb, lmax, dtype=torch.int64)
    for b, (n, m) in enumerate(zip(lengths, tl)):
        fmask[b, :n] = 1
        x[b, :, n:] = 0
        tmask[b, :m] = 1
        targets[b, m:] = 0
    out = [x, targets, fmask, tmask]
    return [t.pin_memory() for t in out] if pin else out


def bucketed_pool(rank, world, n_batches, seed=0):
    """configs[4]: a corpus of n_batches * global_batch utterances with lengths U[T/2,T]; global batches from the
    length-bucketing sampler (data.LengthBucketSampler, replaces shuffle=True of model.py:221); each global batch is
    dealt to the ranks by train_step.balance_by_frames.  Returns this rank's per-step length lists."""
    from policy_gradient_asr_amd.data import LengthBucketSampler
    from policy_gradient_asr_amd.train_step import balance_by_frames
    g = torch.Generator().manual_seed(seed)
    gb = B_PER_GPU * world
    lens = torch.randint(T // 2, T + 1, (n_batches * gb,), generator=g).tolist()
    sampler = LengthBucketSampler(lens, gb, bucket_batches=max(1, n_batches // 2), seed=seed, drop_last=True)
    out = []
    for idx in sampler:
        mine = balance_by_frames([lens[i] for i in idx], world)[rank]
        out.append([lens[idx[i]] for i in mine])
    return out


class BatchFeeder:
    """The H2D copy of every step's batch happens INSIDE the timed region: pinned host tensors, copied on a copy stream
    into one of two device buffer sets one step ahead of their use (DataLoader(pin_memory) + non_blocking, the
    production pattern), or on the main stream at the start of the step (`serial`)."""

    def __init__(self, host_batches, dev, mode, trainer):
        self.host, self.dev, self.mode = host_batches, dev, mode
        # prefetch: no stream of its own (see PolicyGradientTrainer.staging_stream); dma_prefetch: the textbook copy stream
        self.copy = trainer.staging_stream() if mode == "prefetch" else (torch.cuda.Stream(device=dev) if mode == "dma_prefetch" else None)
        self.kernel_copy = not mode.startswith("dma")
        mx = [max(hb[i].numel() for hb in host_batches) for i in range(4)]      # flat buffers: every batch is a contiguous view
        self.bufs = [[torch.empty(mx[i], dtype=host_batches[0][i].dtype, device=dev) for i in range(4)] for _ in range(2)]
        self.ready = [None, None]
        self.freed = [None, None]
        self.k = 0
        if self.copy is not None:
            self._issue(0)

    def _views(self, slot, hb):
        return [self.bufs[slot][i][:hb[i].numel()].view(hb[i].shape) for i in range(4)]

    def _copy(self, slot, hb):
        from policy_gradient_asr_amd import hipops
        for dst, src in zip(self._views(slot, hb), hb):
            if self.kernel_copy:
                hipops.stream_copy(dst, src)                # pgasr_stream_copy: asynchronous for the host
            else:
                dst.copy_(src, non_blocking=True)           # hipMemcpyAsync: returns when the DMA has run (measured)

    def _issue(self, k):
        slot, hb = k % 2, self.host[k % len(self.host)]
        with torch.cuda.stream(self.copy):
            if self.freed[slot] is not None and self.mode == "dma_prefetch":
                self.copy.wait_event(self.freed[slot])      # the step that last used this buffer set has finished
            # (the staging stream is ordered behind the previous step by construction: it joined the main stream there)
            self._copy(slot, hb)
            ev = torch.cuda.Event()
            ev.record()
        self.ready[slot] = ev

    def next(self):
        k, slot = self.k, self.k % 2
        hb = self.host[k % len(self.host)]
        main = torch.cuda.current_stream()
        if self.copy is not None:
            main.wait_event(self.ready[slot])
            self.pending = k + 1                            # staged by done(), once this step has been enqueued
        elif self.mode.endswith("serial"):
            self._copy(slot, hb)
        elif k < 2:                                         # "resident": inputs copied once, outside the timed steps
            for dst, src in zip(self._views(slot, hb), hb):
                dst.copy_(src)
        self.k += 1
        return self._views(slot, hb)

    def done(self):
        """Call after the step that consumed next()'s tensors has been enqueued."""
        ev = torch.cuda.Event()
        ev.record()
        self.freed[(self.k - 1) % 2] = ev
        if self.copy is not None:
            self._issue(self.pending)


# ------------------------------------------------------------------------------------------------
# CPU leg (rank 0, N=1 only): the oracle's train step timed on the host cores + one fp64 parity step
# ------------------------------------------------------------------------------------------------
def _cpu_cores():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))   # the GPU box gives one GPU a 16-core share


def _cpu_step(model_ref, decode_ref, p, opt, x, targets, fmask, il, tl, packed, s):
    import numpy as np
    nb, lt = x.shape[0], targets.shape[1]
    opt.zero_grad()
    enc = model_ref.encoder_forward_torch(p, x, fmask, packed=packed)
    logits = model_ref.head_logits_torch(p, enc)
    lp = torch.log_softmax(logits, 2)
    ctc = torch.nn.functional.ctc_loss(lp, targets, il, tl, blank=0, reduction="mean")
    with torch.no_grad():
        ln = logits.detach().numpy()
        paths, _, _ = decode_ref.sample_paths(ln, seed=0, offset=s)
        greedy = decode_ref.greedy_decode(ln)
        adv = np.zeros(nb)
        for b in range(nb):
            y = targets[b].tolist()
            rs = -decode_ref.edit_dist(y, decode_ref.collapse_path(paths[:, b]))[0] / lt
            rg = -decode_ref.edit_dist(y, greedy[b])[0] / lt
            adv[b] = rs - rg
    lps = lp.gather(2, torch.from_numpy(paths).unsqueeze(-1)).squeeze(-1).sum(0)
    loss = ctc - (torch.from_numpy(adv).float() * lps).sum() / nb
    loss.backward()
    opt.step()


def cpu_baseline(steps=3):
    """The oracle's train step (torch-CPU model + ctc_loss + numpy sampler/greedy/edit distance + Adam) on this box's
    host cores, both variants of BASELINE.md §3:
      (ii) unpacked LSTM at the full headline shape (lengths are all T, so it is the same arithmetic as the
           reference's packed call) -- `value`, the honest bar;
      (i)  the packed-sequence LSTM exactly as model.py:52-55 writes it, on a BOUNDED sample (T=250): torch's packed
           CPU backward is O(T^2) (121 s per step at T=1000 on 8 cores), so the full shape cannot be timed
           inside a default run."""
    from oracle import model_ref, decode_ref
    cores = _cpu_cores()
    torch.set_num_threads(cores)
    out = {}
    for name, packed, tt in (("unpacked", False, T), ("packed", True, T // 4)):
        lt = tt // 10
        p = {k: v.clone().requires_grad_(True) for k, v in model_ref.init_params(F, V, 0).items()}
        opt = torch.optim.Adam(list(p.values()), lr=5e-4)
        x, targets, fmask, _ = synth_batch(0, [tt] * B_PER_GPU)
        il = torch.full((B_PER_GPU,), tt, dtype=torch.long)
        tl = torch.full((B_PER_GPU,), lt, dtype=torch.long)
        times = []
        for s in range(steps + 1):
            t0 = time.perf_counter()
            _cpu_step(model_ref, decode_ref, p, opt, x, targets, fmask, il, tl, packed, s)
            dt = time.perf_counter() - t0
            print(f"[bench] cpu baseline ({name}, T={tt}) step {s}: {dt:.2f} s", file=sys.stderr, flush=True)
            if s > 0:
                times.append(dt)
        times.sort()
        out[name] = (times[len(times) // 2], tt, len(times))
    sec, _, n = out["unpacked"]
    psec, ptt, pn = out["packed"]
    return {"value": B_PER_GPU / sec, "unit": "utterances/sec", "cores": cores, "kind": "port",
            "sample": f"median of {n} full steps at B={B_PER_GPU},T={T},F={F},V={V} after 1 warm-up; torch-CPU LSTM "
                      f"(unpacked, lengths=T: same arithmetic as the packed call) + ctc_loss + numpy sampler/greedy/"
                      f"edit distance + Adam; {sec:.2f} s/step",
            "variants": {
                "unpacked_T1000": {"value": B_PER_GPU / sec, "s_per_step": sec, "steps": n},
                "packed_as_written_T250_sample": {
                    "value": B_PER_GPU / psec, "s_per_step": psec, "steps": pn,
                    "note": f"model.py:52-55 pack_padded_sequence path, bounded sample B={B_PER_GPU},T={ptt},L={ptt // 10}; its "
                            "CPU backward is O(T^2): at T=1000 the survey measured 121 s/step on 8 cores (0.26 utt/s)"}}}


_fp64_ref = {}


def parity_vs_fp64(model, trainer, batch, brief=False):
    """One UNTIMED eval-mode CTC step (lambda = 0, no sampling) of the benchmarked model on the GPU against the oracle in
    fp64 on the same weights and inputs: relative error of the loss and max-norm relative error over all parameter
    gradients.  Travels with the throughput number as its precision statement."""
    from oracle import model_ref
    from policy_gradient_asr_amd.loss import pg_ctc_loss
    x, targets, fmask, tmask = batch
    was_training = model.training
    model.eval()
    trainer.gflat.zero_()
    logits, in_len = model.logits(x, fmask)
    tg_len = tmask.sum(1).to(torch.int32).contiguous()
    loss, _, _, _ = pg_ctc_loss(logits, in_len, targets.to(torch.int32).contiguous(), tg_len, lam=0.0)
    loss.backward()
    torch.cuda.synchronize()
    g_gpu = {k: v.grad.detach().double().cpu() for k, v in model.named_parameters()}
    model.train(was_training)
    key = (trainer.nstep, x.data_ptr())       # the two precision variants are checked on the SAME weights: one fp64 CPU step serves both
    if _fp64_ref.get("key") != key:
        torch.set_num_threads(_cpu_cores())
        p = {(k[len("encoder."):] if k.startswith("encoder.") else k): v.detach().double().cpu().requires_grad_(True)
             for k, v in model.named_parameters()}
        t0 = time.perf_counter()
        xc, fm = x.double().cpu(), fmask.cpu()
        lens = fm.sum(1).long()
        enc = model_ref.encoder_forward_torch(p, xc, fm, packed=bool((lens != xc.shape[2]).any()))
        lp = torch.log_softmax(model_ref.head_logits_torch(p, enc), 2)
        ref = torch.nn.functional.ctc_loss(lp, targets.cpu().long(), lens, tg_len.cpu().long(), blank=0, reduction="mean")
        ref.backward()
        print(f"[bench] fp64 parity step on the CPU: {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
        _fp64_ref.update(key=key, p=p, ref=ref)
    p, ref = _fp64_ref["p"], _fp64_ref["ref"]
    worst, worst_l2, worst_name = 0.0, 0.0, ""
    for k, v in g_gpu.items():
        rk = k[len("encoder."):] if k.startswith("encoder.") else k
        r = p[rk].grad
        e = float((v - r).abs().max() / (r.abs().max() + 1e-300))
        worst_l2 = max(worst_l2, float((v - r).norm() / (r.norm() + 1e-300)))
        if e > worst:
            worst, worst_name = e, rk
    res = {"loss": abs(float(loss.detach()) - float(ref.detach())) / abs(float(ref.detach())), "param_grads_maxnorm": worst,
           "param_grads_maxnorm_tensor": worst_name, "param_grads_frobenius": worst_l2}
    if brief:
        return res
    return {**res,
            "what": "eval-mode CTC step (lambda=0) of the benchmarked model (after the timed steps) vs torch-CPU fp64 on the same "
                    "weights and batch; worst tensor.  The input layer's two tensors sit behind leaky_relu': a pre-activation "
                    "that is zero to rounding changes side between two fp32 evaluations and moves one summand by 100x"}


# ------------------------------------------------------------------------------------------------
def sweep_roofline(prof, n_sampled, frames_per_step, products=3):
    """The dominant kernel's roofline figures from the HIP-event record of the SAMPLED steps: prof maps kernel name ->
    (total ms, launches) over n_sampled instrumented steps.  Algorithmic flops of one sweep launch: h (B,256) x W_hh^T
    (256,1024), two directions, one multiply-add per frame of the chain.  products: bf16 MFMA products
    the precision mode issues per algorithmic flop (3 or 6): the peak the algorithmic rate is priced against is 2500 / products."""
    sweeps = {k: v for k, v in prof.items() if k.startswith("lstm_")}
    name, (tot_ms, calls) = max(sweeps.items(), key=lambda kv: kv[1][0]) if sweeps else ("none", (0.0, 1))
    tavg = frames_per_step / B_PER_GPU                                # average steps of a sweep's chain
    flops_per_launch = 2.0 * 2 * B_PER_GPU * 256 * 1024 * tavg
    avg_ms = tot_ms / max(calls, 1)
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    return {"kernel": name, "avg_launch_ms": avg_ms, "achieved": achieved,
            "peak": BF16_DENSE_PEAK_TF / products,     # bf16 MFMA products per algorithmic fp32 flop: 3 (bf16x3) or 6 (f32)
            "flops_per_launch": flops_per_launch, "launches_per_step": calls / max(n_sampled, 1),
            # what really bounds the kernel: a chain of T dependent cross-CU hand-offs (floor measured with tools/allgather.hip)
            "per_step_us": avg_ms * 1e3 / tavg if tavg > 0 else 0.0, "handoff_floor_us": 0.5}


def gemm_probe(precision, reps=5):
    """roofline_gemm: the path's three big products in the timed precision mode as STAND-ALONE whole-chip launches (outside the timed
    region; inside the step the same kernels run beside sweeps on the XCDs those leave free and partly wait for them): HIP-event
    average, issued bf16 MFMA rate against the 2.5 PF dense peak and against what the chip sustains on bare MFMAs."""
    from policy_gradient_asr_amd import hipops
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator().manual_seed(7)
    M = B_PER_GPU * T
    npl, prods = (3, 6) if precision == "f32" else (2, 3)

    def timeit(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    out = {}
    pmc = {}
    try:       # optional evidence written by tools/profile_round5.sh (round 4: profile_round4.sh); never required to run
        gp = next(f for f in ("r05_gemm_pmc.json", "r04_gemm_pmc.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
        with open(os.path.join(ROOT, "profiles", gp)) as fi:
            pmc = json.load(fi)
    except Exception:  # noqa: BLE001
        pmc = {}
    with hipops.precision(precision):
        for role, (N, K) in (("input_projection", (2048, 512)), ("input_gradient", (512, 2048))):
            A = torch.randn(M, K, generator=g).to(dev)
            W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
            C = torch.empty(M, N, device=dev)
            planes = hipops.split_planes(W, planes=npl)
            us = timeit(lambda: hipops.gemm_x3w(A, planes, C, M, N, K))
            gf = 2.0 * M * N * K / 1e9
            name = "x6c::gemm_x6c_kernel" if npl == 3 else "c256::gemm_x3c_kernel"
            out[role] = {"kernel": name, "shape_MNK": [M, N, K], "gflop_fp32": gf, "avg_launch_us": us,
                         "issued_bf16_tflops": prods * gf / us * 1e3, "frac_of_2500": prods * gf / us * 1e3 / BF16_DENSE_PEAK_TF}
            del A, W, C, planes
        Mw, Nw, Kw = 2048, 512, M
        A = torch.randn(Kw, Mw, generator=g).to(dev); Bm = torch.randn(Kw, Nw, generator=g).to(dev)
        C = torch.empty(Mw, Nw, device=dev)
        us = timeit(lambda: hipops.gemm(A, Bm, C, Mw, Nw, Kw, transA=True, splitk=16, precision=2 if npl == 3 else 1))
        gf = 2.0 * Mw * Nw * Kw / 1e9
        out["weight_gradient_ih"] = {"kernel": "t6::gemm_t6_kernel" if npl == 3 else "t256::gemm_t256_kernel", "shape_MNK": [Mw, Nw, Kw],
                                     "split_k": 16, "gflop_fp32": gf, "avg_launch_us": us,
                                     "issued_bf16_tflops": prods * gf / us * 1e3, "frac_of_2500": prods * gf / us * 1e3 / BF16_DENSE_PEAK_TF}
    for rec in out.values():
        k = rec["kernel"].split("::")[-1]
        if k in pmc.get("kernels", {}):
            rec["pmc_mfma_busy"] = pmc["kernels"][k].get("mfma_busy")
            rec["pmc_source"] = pmc.get("source")
    sustained = pmc.get("sustained_bare_mfma_tflops")
    if sustained:
        for rec in out.values():
            rec["frac_of_sustained"] = rec["issued_bf16_tflops"] / sustained
    return {"bound": "mfma", "peak": BF16_DENSE_PEAK_TF, "unit": "TFLOP/s (issued bf16 MFMA)", "products_per_flop": prods,
            "sustained_bare_mfma_tflops": sustained,
            "note": "stand-alone whole-chip launches after the timed region, HIP events over %d launches; peak = 2.5 PF dense bf16; "
                    "sustained_bare_mfma_tflops = what every CU issuing bare v_mfma_f32_32x32x16_bf16 on random operands reaches on "
                    "this chip (tools/mfma_peak.hip, clock ~1.7-1.9 GHz under that load), from the committed profile" % reps,
            "kernels": out}


def allreduce_probe(trainer, dev, reps=10):
    """Wall time of the step's two gradient buckets as stand-alone RCCL all-reduces (barrier-bracketed, outside the
    timed region): what the early bucket hides under the tail of backward and what the late one adds."""
    out = {}
    split = trainer.upper_split or 0
    scratch = torch.zeros_like(trainer.gflat)
    for name, view in (("upper_bucket", scratch[split:]), ("rest_bucket", scratch[:split]), ("whole", scratch)):
        if view.numel() == 0:
            continue
        for _ in range(2):
            dist.all_reduce(view)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(view)
        torch.cuda.synchronize()
        dt = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], device=dev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        out[name] = {"MB": view.numel() * 4 / 1e6, "ms": float(dt.item())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("headline", "bucketed"), default="headline")
    ap.add_argument("--event-every", type=int, default=4, help="HIP-event timing of the sweeps and phases on every n-th timed step")
    ap.add_argument("--h2d", choices=("prefetch", "serial", "dma_prefetch", "dma_serial", "resident"), default="prefetch",
                    help="how each step's batch reaches HBM inside the timed region (resident: not at all, diagnostic)")
    ap.add_argument("--long-steps", type=int, default=200, help="second, longer timed region after the headline one (0 = off)")
    ap.add_argument("--precision", choices=("f32", "bf16x3"), default="f32",
                    help="arithmetic of the timed step: f32 = the reference's (fp32-faithful six-product kernels), bf16x3 = the faster mode within 1e-3")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the configs[4] and other-precision legs that follow the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    # Rank 0 prints ONE JSON line on stdout -- and nothing else may: RCCL writes its version banner to stdout from C when a
    # communicator is created (every rank would add five lines to the driver's capture).  File descriptor 1 is pointed at
    # stderr for the whole run; the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} does not match the realised world size {world}", file=sys.stderr)
        sys.exit(2)
    # rehearsal switch for a 1-GPU box (never used by the driver): all ranks on cuda:0, gloo collectives staged
    # through the host -- it exercises the launcher and the plumbing, it is NOT a scaling number
    rehearse = os.environ.get("PGASR_BENCH_REHEARSE", "") == "1"
    if rehearse:
        local_rank = 0
        # two processes with persistent kernels on ONE card can starve each other: say where the host is if a leg stops moving
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("PGASR_BENCH_STUCK_S", "240")), repeat=True, file=sys.stderr)
    if local_rank >= torch.cuda.device_count():      # a rank validates its own device (the parent does not touch the GPU)
        print(f"[bench] rank {rank} wants device {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    # plumbing rehearsal for a one-GPU box (never used by the driver): a 1-rank RCCL group whose all-reduces (identities)
    # are really issued, buckets, stream and event waits included -- what a collective stream does to the step's own streams
    # can be seen without a second GPU
    solo_collective = world == 1 and os.environ.get("PGASR_BENCH_SOLO_COLLECTIVE", "") == "1"
    if world > 1 or solo_collective:
        # before the communicator takes its stream from torch's pool: see streams.prime()
        from policy_gradient_asr_amd import streams
        streams.prime()
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif solo_collective:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)

    from policy_gradient_asr_amd import hipops
    if rehearse:
        hipops.LSTM_FLAGS |= 4      # ranks share one GPU here: two sweeps' clusters per XCD leave no room for helper workgroups
    from policy_gradient_asr_amd.model import Seq2Seq, weights
    from policy_gradient_asr_amd.train_step import PolicyGradientTrainer

    torch.manual_seed(0)
    model = Seq2Seq(V, n_feats=F)
    model.apply(weights)
    model = model.to(dev).train()   # dropout on (model.py:45,51 p=0.5; model.py:42 p=0.3), as in training
    bucketed = args.workload == "bucketed"
    trainer = PolicyGradientTrainer(model, lr=5e-4, lam=1.0, seed=1234, world_size=world, rank=rank, precision=args.precision,
                                    **({"reward_decoder": "beam", "beam_size": 16} if bucketed else {}))
    if solo_collective:
        trainer.collective = True
    # a silently slower configuration (feed-ahead fallen back to the sequential order because kernels of different
    # streams do not overlap here) must not be benchmarked unnoticed; counter-collecting profiler passes serialise
    # kernels by design and say so with PGASR_ALLOW_SEQUENTIAL=1
    hipops.STRICT_CONCURRENCY = os.environ.get("PGASR_ALLOW_SEQUENTIAL", "") != "1"
    if rehearse and world > 1:
        _ar = dist.all_reduce
        def _cpu_all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):   # gloo has no device tensors here
            c = t.cpu(); _ar(c, op=op, group=group); t.copy_(c)                      # (synchronous either way)
        dist.all_reduce = _cpu_all_reduce
        _bc = dist.broadcast
        def _cpu_broadcast(t, src=0, group=None):
            c = t.cpu(); _bc(c, src=src, group=group); t.copy_(c)
        dist.broadcast = _cpu_broadcast

    if bucketed:
        pool = bucketed_pool(rank, world, n_batches=8, seed=0)
        host = [synth_batch(1000 + 17 * i + rank, lens, pin=True) for i, lens in enumerate(pool)]
        frames = [sum(lens) for lens in pool]
    else:
        host = [synth_batch(100 + rank, pin=True)]
        frames = [B_PER_GPU * T]
    feeder = BatchFeeder(host, dev, args.h2d, trainer)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step_marks = []      # (start, end) events of the timed steps: with the sweeps' events they give the phases of a step

    def one_step(mark=False):
        batch = feeder.next()
        if mark:
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
        loss = trainer.step(*batch)
        if mark:
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            step_marks.append((e0, e1))
        feeder.done()
        return loss

    for i in range(args.warmup):
        one_step()
        torch.cuda.synchronize()
        if rank == 0:
            print(f"[bench] warmup step {i} done", file=sys.stderr, flush=True)
    barrier()
    # live HIP-event timing of the dominant kernels (6 launches a step) and of the step's phases -- on every
    # --event-every-th step of the timed region: an event record costs the stream it is issued on ~5 us
    # (tools/dev/tools_marker_cost.py), 14 of them per instrumented step
    hipops.profile_reset(True, only=("lstm_",))
    every = max(1, args.event_every)
    n_sampled = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        sampled = (i + 1) % every == 0 or (args.steps < every and i == args.steps - 1)   # never the step right behind the barrier: its front end is host-bound
        hipops.profile_pause(not sampled)
        n_sampled += int(sampled)
        loss = one_step(mark=sampled)
    hipops.profile_pause(False)
    host_enqueue_s = time.perf_counter() - t0       # how long the host needed to ENQUEUE the steps (it runs ahead of the GPU)
    barrier()
    dt_local = time.perf_counter() - t0
    if rank == 0:
        print(f"[bench] timed region: {dt_local:.3f} s for {args.steps} steps", file=sys.stderr, flush=True)
    phases = hipops.profile_phases(step_marks)
    prof = hipops.profile_collect()
    # the other instrumented kernels (GEMMs, beam search) are timed in two extra steps OUTSIDE the timed region
    hipops.profile_reset(True)
    for _ in range(2):
        one_step()
    extra = {k: (v[0] * n_sampled / 2.0, v[1] * n_sampled / 2.0) for k, v in hipops.profile_collect().items() if k not in prof}
    prof.update(extra)
    hipops.profile_reset(False)
    hipops.lstm_assert_no_timeouts()      # every rank: a timed-out sweep would make the number meaningless

    def timed_leg(fd, steps, warm, name=""):
        """ms per step (max over ranks) of `steps` steps fed by `fd`, barrier-bracketed like the headline region."""
        for _ in range(warm):
            trainer.step(*fd.next()); fd.done()
        barrier()
        t_0 = time.perf_counter()
        for _ in range(steps):
            trainer.step(*fd.next()); fd.done()
        barrier()
        sec = time.perf_counter() - t_0
        if world > 1:
            tmax = torch.tensor([sec], device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            sec = float(tmax.item())
        hipops.lstm_assert_no_timeouts()
        if rank == 0:
            print(f"[bench] leg {name}: {sec / steps * 1e3:.3f} ms per step over {steps} steps", file=sys.stderr, flush=True)
        return sec / steps * 1e3

    # ---- legs that follow the headline region (all outside it; every rank runs them: they contain collectives) ----
    long_run = None
    if args.long_steps > 0:
        lms = timed_leg(feeder, args.long_steps, 0, "long_run")
        long_run = {"steps": args.long_steps, "ms_per_step": lms, "value": B_PER_GPU * world / (lms * 1e-3),
                    "note": "same workload and feeder as the headline region, timed again over a longer run"}
    resident_leg = None
    if not bucketed and not args.no_extra_legs and args.h2d != "resident":
        # the same step with its inputs ALREADY in HBM (no staging copy in the step): what the PCIe-inclusive headline costs
        rms = timed_leg(BatchFeeder(host, dev, "resident", trainer), 100, 5, "inputs_resident")
        resident_leg = {"ms_per_step": rms, "utt_per_s": B_PER_GPU * world / (rms * 1e-3), "steps": 100, "warmup": 5,
                        "note": "inputs resident in HBM before the timed region; the headline `value` stages every step's batch from pinned "
                                "host memory inside the timed region (model.py:227-230's .to(device) is part of the reference's step)"}
    bucketed_leg = None
    if not bucketed and not args.no_extra_legs:
        # configs[4] in the same driver-run line: the reference's reward hypothesis (prefix beam search, beam 16) on
        # length-bucketed batches U[T/2,T], measured after the headline with the same model and trainer
        pool = bucketed_pool(rank, world, n_batches=8, seed=0)
        bhost = [synth_batch(1000 + 17 * i + rank, lens, pin=True) for i, lens in enumerate(pool)]
        trainer.reward_decoder, trainer.beam_size = "beam", 16
        bms = timed_leg(BatchFeeder(bhost, dev, args.h2d, trainer), 40, 8, "bucketed")
        trainer.reward_decoder = "greedy"
        bframes = sum(sum(lens) for lens in pool) / len(pool)
        bucketed_leg = {"ms_per_step": bms, "utt_per_s": B_PER_GPU * world / (bms * 1e-3), "steps": 40, "warmup": 8,
                        "frames_per_step_per_gpu": bframes,
                        "workload": "configs[4]: beam-16 prefix-search reward hypothesis + collapse_fn + edit distance, lengths "
                                    "U[500,1000] in length-bucketed batches balanced by frames, B=32/GPU, train mode, Adam, H2D inside"}
    precision_variants, parity_batch = None, None
    other = "bf16x3" if args.precision == "f32" else "f32"
    if world == 1 and not bucketed and not args.no_extra_legs:
        # the same step in the OTHER precision mode, outside the headline region (f32 = the reference's own arithmetic, torch fp32;
        # bf16x3 = two planes / three products, within north_star's 1e-3 bar)
        head_ms = (long_run or {}).get("ms_per_step", dt_local / args.steps * 1e3)
        precision_variants = {args.precision: {"ms_per_step": head_ms, "utt_per_s": B_PER_GPU / (head_ms * 1e-3), "timed_as": "headline (long_run)"}}
        trainer.precision = other
        oms = timed_leg(BatchFeeder(host, dev, args.h2d, trainer), 40, 5)
        precision_variants[other] = {"ms_per_step": oms, "utt_per_s": B_PER_GPU / (oms * 1e-3), "steps": 40, "warmup": 5}
        precision_variants["f32"]["arithmetic"] = ("every big product (recurrent sweeps, W_ih projections, input / weight gradients) = six bf16 MFMA terms of a "
                                                   "3-plane split (all terms >= 2^-24), fed + streamed orders; small products on the exact fp32 MFMA")
        precision_variants["bf16x3"]["arithmetic"] = "every dense product = three bf16 MFMA terms of a 2-plane split (input affine on the exact fp32 MFMA)"
        if not args.no_parity:
            parity_batch = [t.to(dev) for t in host[0]]
            with hipops.precision(other):
                precision_variants[other]["max_rel_err_vs_fp64"] = parity_vs_fp64(model, trainer, parity_batch, brief=True)
        trainer.precision = args.precision
    dt = dt_local
    per_rank_ms = [dt_local / args.steps * 1e3]
    ar = None
    if world > 1:
        tt = torch.tensor([dt_local], device=dev)
        if not rehearse:
            gathered = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(gathered, tt)
            per_rank_ms = [float(g.item()) / args.steps * 1e3 for g in gathered]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        if not rehearse:
            ar = allreduce_probe(trainer, dev)
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = B_PER_GPU * world * args.steps / dt
        prods = PRODUCTS[args.precision]
        rl = sweep_roofline(prof, n_sampled, sum(frames) / len(frames), products=prods)
        name, avg_ms, achieved, peak, flops_per_launch = rl["kernel"], rl["avg_launch_ms"], rl["achieved"], rl["peak"], rl["flops_per_launch"]
        traffic, traffic_src = None, None         # HBM bytes per launch of that kernel from the committed PMC passes
        try:       # written by tools/pmc_summary5.py (round 4: pmc_summary4.py); the file is optional evidence, never required to run
            pmc_file = next(f for f in ("r05_pmc.json", "r04_pmc.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            with open(os.path.join(ROOT, "profiles", pmc_file)) as fi:
                rec = json.load(fi)[args.precision]["kernels"][name]
            traffic = rec["hbm_bytes_per_launch"]
            traffic_src = ("from_committed_profile profiles/" + pmc_file + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                           "gfx950 FETCH x2 correction) on the sweep kernels of this precision mode (a counter pass serialises kernels: "
                           "the sequential order of the same kernels); read %.0f MB + write %.0f MB per launch against 262 + 392 MB "
                           "(forward sweep) / 392 + 262 MB (backward sweep) algorithmic"
                           % (rec["hbm_read_bytes_per_launch"] / 1e6, rec["hbm_write_bytes_per_launch"] / 1e6))
        except Exception:  # noqa: BLE001
            traffic = None
        gflop_step_gpu = STEP_GFLOP_PER_UTT * (sum(frames) / len(frames) / T)     # per GPU: 28.65 GFLOP per 1000 frames
        step_tf = gflop_step_gpu / (ms * 1e-3) / 1e3
        out = {
            "metric": METRIC,
            "value": value, "unit": "utterances/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPES[args.precision], "precision": args.precision, "data": "synthetic",
            "config": {"workload": ("configs[2]+[1]: CTC + REINFORCE train step (greedy baseline, sampled path, WER-style "
                                    "edit-distance reward), B=32/GPU, T=1000, F=80, V=29, L=100, Adam" if not bucketed else
                                    "configs[4]: CTC + REINFORCE train step with the reference's reward hypothesis (prefix beam "
                                    "search, beam 16, collapse_fn, edit distance), lengths U[500,1000] in length-bucketed "
                                    "batches balanced by frames across ranks, B=32/GPU, F=80, V=29, L=T/10, Adam"),
                       "global_batch": B_PER_GPU * world, "parallelism": f"dp{world}", "dropout": "on (train mode)",
                       "h2d": {"prefetch": "inside the timed region: the next step's pinned host batch is staged by pgasr_stream_copy on the CTC lattice's side stream, beside this step's backward pass",
                               "serial": "inside the timed region: pinned host batch copied by pgasr_stream_copy on the main stream at step start",
                               "dma_prefetch": "inside the timed region: hipMemcpyAsync one step ahead on a copy stream (blocks the host)",
                               "dma_serial": "inside the timed region: hipMemcpyAsync on the main stream at step start (blocks the host)",
                               "resident": "NOT in the timed region (diagnostic)"}[args.h2d],
                       "frames_per_step_per_gpu": sum(frames) / len(frames)},
            "rccl_world_size": (dist.get_world_size() if world > 1 else 1),
            "collective_backend": ("none" if world == 1 else ("gloo-rehearsal (NOT a scaling number)" if rehearse else "nccl (RCCL)")),
            "ms_per_step_per_rank": per_rank_ms,
            "roofline": {"bound": "latency", "kernel": name, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": avg_ms, "launches_per_step": rl["launches_per_step"],
                         "algorithmic_gflop_per_launch": flops_per_launch / 1e9,
                         "per_step_us": rl["per_step_us"], "handoff_floor_us": rl["handoff_floor_us"],
                         "floor_frac": rl["handoff_floor_us"] / rl["per_step_us"] if rl["per_step_us"] > 0 else None,
                         "peak_note": "algorithmic fp32 flops against the bf16 dense MFMA peak / %d (the kernel issues %d bf16 products "
                                      "per flop in this precision mode): identical to issued bf16 flops / 2500 TF" % (prods, prods),
                         "issued_bf16_tflops": prods * achieved, "issued_bf16_frac_of_2500": prods * achieved / BF16_DENSE_PEAK_TF,
                         "frac_vs_fp32_mfma_peak_157.3": achieved / FP32_MFMA_PEAK_TF,
                         "note": "bound = latency: a serial chain of T dependent cross-CU hand-offs -- neither the MFMA nor the HBM roofline "
                                 "binds; floor_frac = the measured hand-off floor of this chip (0.5 us for the 16-way all-gather, "
                                 "tools/allgather.hip) / the per-step time: the fraction that says how close the chain is to its bound; "
                                 "frac is kept for continuity.  The MFMA-bound kernels of the step have their fractions in roofline_gemm"},
            "roofline_gemm": gemm_probe(args.precision) if (world == 1 and not args.no_extra_legs) else None,
            "roofline_step": {"algorithmic_tflops": step_tf, "gflop_per_step_per_gpu": gflop_step_gpu,
                              "frac_vs_bf16_peak_over_products": step_tf / peak, "frac_vs_fp32_mfma_peak_157.3": step_tf / FP32_MFMA_PEAK_TF,
                              "note": "whole-step dense contraction flops (SURVEY §8d: 28.65 GFLOP per 1000-frame utterance) per GPU / ms_per_step"},
            "kernel_ms_per_step": {k: v[0] / max(n_sampled, 1) for k, v in prof.items()},
            "event_timed_steps": n_sampled,
            "host_enqueue_ms_per_step": host_enqueue_s / args.steps * 1e3,
            "phase_ms_per_step": phases,
            "loss": float(loss.item()),
        }
        if ar is not None:
            out["allreduce_ms"] = ar
        if long_run is not None:
            out["long_run"] = long_run
        if resident_leg is not None:
            out["inputs_resident"] = resident_leg
        if bucketed_leg is not None:
            out["bucketed"] = bucketed_leg
        if world == 1 and not args.no_parity:
            with hipops.precision(args.precision):
                out["max_rel_err_vs_fp64"] = parity_vs_fp64(model, trainer, parity_batch if parity_batch is not None else [t.to(dev) for t in host[0]])
        if precision_variants is not None:
            if "max_rel_err_vs_fp64" in out:
                precision_variants[args.precision]["max_rel_err_vs_fp64"] = {k: out["max_rel_err_vs_fp64"][k] for k in
                                                                         ("loss", "param_grads_maxnorm", "param_grads_maxnorm_tensor", "param_grads_frobenius")}
            out["precision_variants"] = precision_variants
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    elif solo_collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
