"""Side streams that do not slow down the stream they serve.

Measured on MI355X (tools/dev/tools_blocked_queues.py): while a stream sits on an unsatisfied event wait -- a barrier packet at
the head of its hardware queue, which is where every side stream of the train step spends most of its time, because the host
enqueues a whole step ~3 ms ahead of the GPU -- kernel-to-kernel dispatch on ONE particular other hardware queue slows from
~6 us to ~70 us per transition (six dependent 35-us kernels on the default stream: 210 us, with the wrong stream blocked
400-580 us; any of the other streams blocked: 205-220 us; in a forward pass with its weight packs on that stream the head GEMM
started 420 us late).  Which HIP stream lands on the hardware queue that is coupled to the
main stream's depends on the order in which streams were created in the process (ROCclr deals streams onto GPU_MAX_HW_QUEUES
queues round-robin), so it is MEASURED: candidates are created, each is blocked in turn behind a short chain of tiny kernels on
the serving stream, and the ones that stretch the chain are never handed out (they stay allocated, so that later streams do not
take their place in the deal).  The probe runs once per (device, serving stream), synchronises, and takes ~30 ms.
"""
import os

import torch

from . import hipops

VET = os.environ.get("PGASR_VET_STREAMS", "1") != "0"     # 0: hand out streams in creation order (A/B switch)
_CANDIDATES = 32          # torch.cuda.Stream() draws round-robin from a pool of 32 HIP streams per device: one full turn
_CHAIN = 8                # tiny kernels on the serving stream per measurement
_state = {}               # (device index, serving stream handle) -> {"good": [...], "named": {...}, "report": [...]}


def _chain_us(main, blocked, tiny, zero_words, reps=1):
    """Time of _CHAIN dependent tiny kernels on ``main`` (current stream) that start behind a 300-us sleeper -- so that the host
    has enqueued everything, the blocked stream's wait included, before the first of them is dispatched."""
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        with torch.cuda.stream(main):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            hipops.stream_gate(zero_words.data_ptr(), timeout_us=300)
            e0.record()
            for _k in range(_CHAIN):
                tiny.add_(1)
            e1.record()
            later = torch.cuda.Event()
            later.record()
        if blocked is not None:
            blocked.wait_event(later)
            with torch.cuda.stream(blocked):
                tiny.add_(1)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        best = us if best is None else min(best, us)
    return best


def _vet(main):
    """One full turn of torch's stream pool is taken and measured, so that (a) the streams handed out are known not to be
    coupled to ``main`` and (b) the NEXT stream anybody creates (the process group's collective stream, for one) is pool
    stream 0 again, whose verdict is known: if it is a coupled one, the turn is advanced past it."""
    dev = torch.device("cuda", torch.cuda.current_device())
    tiny = torch.zeros(1, device=dev)
    zero_words = torch.zeros(8, dtype=torch.int32, device=dev)
    cands = [torch.cuda.Stream() for _ in range(_CANDIDATES)]
    for s in cands:                      # a stream's hardware queue exists from its first use
        with torch.cuda.stream(s):
            tiny.add_(1)
    base = _chain_us(main, None, tiny, zero_words, reps=3)
    times = [_chain_us(main, s, tiny, zero_words) for s in cands]
    typical = sorted(times)[len(times) // 2]
    for i, s in enumerate(cands):        # a slow reading is confirmed before it condemns a stream
        if times[i] >= 1.5 * typical:
            times[i] = min(times[i], _chain_us(main, s, tiny, zero_words, reps=2))
    good, report, verdicts = [], [("nothing blocked", base, "")], []
    for i, (s, us) in enumerate(zip(cands, times)):
        # measured with 8 tiny kernels: nothing blocked 17 us, an ordinary stream blocked 28 us, the coupled one 53 us (with
        # 250-workgroup kernels on the serving stream the same stream costs +75 us PER KERNEL)
        # ... and 17 us again for a stream that shares the serving stream's OWN hardware queue: its wait costs nothing, but
        # nothing on it can ever run beside the serving stream
        coupled, same_queue = us >= 1.5 * typical, us < 0.5 * (base + typical)
        ok = not coupled and not same_queue
        verdicts.append(ok)
        report.append((f"pool stream {i} blocked", us, "ok" if ok else ("coupled to the serving stream's queue: never handed out" if coupled
                                                                       else "shares the serving stream's hardware queue: never handed out")))
        if ok:
            good.append((i, s))
    burnt = []
    for ok in verdicts:                  # the pool's next stream is candidate 0 again, then 1, ...
        if ok:
            break
        burnt.append(torch.cuda.Stream())
    if len(good) < 4:                    # never seen; do not fail a training run over a scheduling heuristic
        good = list(enumerate(cands))
    if os.environ.get("PGASR_DEBUG"):
        import sys
        for r in report:
            print("[pgasr streams]", *r, file=sys.stderr)
    return {"good": good, "keep": cands + burnt, "named": {}, "report": report, "taken": [], "typical": typical,
            "probe": (tiny, zero_words)}


def prime():
    """Run the probe for the CURRENT stream now.  Call it before ``dist.init_process_group("nccl", device_id=...)``: the
    process group takes ITS stream from the same pool when the communicator is created, and a collective stream is blocked
    on event waits for most of a step -- it should be one of the streams this module has looked at, not the one that happens
    to be coupled to the stream the step runs on."""
    side_stream("weight_gradients")


def side_stream(name):
    """The side stream called ``name`` of the CURRENT stream (one per name and serving stream; created and vetted on first use)."""
    main = torch.cuda.current_stream()
    key = (torch.cuda.current_device(), main.cuda_stream)
    st = _state.get(key)
    if st is None:
        if VET:
            st = _vet(main)
        else:
            st = {"good": None, "keep": [], "named": {}, "report": [], "taken": []}
        _state[key] = st
    s = st["named"].get(name)
    if s is None:
        if st["good"]:
            # the coupling is pairwise and periodic (tools/dev/tools_blocked_queues.py with SERVE=i: pool stream i is slowed by
            # stream i + 4 of the 8 hardware queues, whichever of the two serves): two streams of one step never sit 4 apart
            # ... the index rule is a first guess; what counts is measured: the newcomer must not stretch a chain served by a
            # stream already handed out, nor the other way round
            while True:
                pick = next((k for k, (i, _) in enumerate(st["good"]) if all((i - t) % 8 != 4 for t in st["taken"])), 0)
                i, s = st["good"].pop(pick)
                tiny, zero_words = st["probe"]
                clash = any(_chain_us(a, b, tiny, zero_words) >= 1.5 * st["typical"]
                            for o in st["named"].values() for a, b in ((o, s), (s, o)))
                if not clash or not st["good"]:
                    break
                st["report"].append((f"pool stream {i}", 0.0, "coupled to another side stream of the step: not handed out"))
            st["taken"].append(i)
        else:
            s = torch.cuda.Stream()
        st["named"][name] = s
    return s


# ---- tensors that cross streams -------------------------------------------------------------------------------------
# ``tensor.record_stream(s)`` makes the caching allocator record an event on ``s`` when the tensor is FREED -- one marker packet
# per tensor, wherever stream ``s`` happens to be at that moment.  The weight packs of a step (30 tensors made on a side stream,
# read on the main stream) die when the forward pass returns: 75-150 us of markers sat on the main stream between the last
# forward sweep and the head GEMM (tools/dev/tools_gap_bisect.py: 117 us -> 49 us without them).  Inside a managed step the
# tensors are instead kept alive until the NEXT step begins: by then every side stream of the step has been joined into the
# main stream, and each side stream starts its next piece of work with a wait for the main stream, so whichever stream a block
# returns to, its reuse is ordered behind the last reader.  Outside a managed step (eval, tests of single ops) it is
# ``record_stream`` as before.
_held = []
_depth = 0
HOLD = os.environ.get("PGASR_HOLD_TENSORS", "1") != "0"


def hold(tensor, stream):
    """``tensor`` (allocated on another stream) is used on ``stream``."""
    if _depth > 0 and HOLD:
        _held.append(tensor)
    else:
        tensor.record_stream(stream)


class managed_step:
    """``with managed_step():`` around ONE train step whose side streams are all joined into the calling stream before it
    ends (PolicyGradientTrainer.step).  Entering releases what the previous step held."""

    def __enter__(self):
        global _depth
        if _depth == 0:
            del _held[:]
        _depth += 1
        return self

    def __exit__(self, *exc):
        global _depth
        _depth -= 1
        return False


def release():
    """Drop the tensors held for the last step (call after a synchronisation, e.g. at the end of an epoch)."""
    del _held[:]


def report():
    """[(label, chain microseconds, verdict)] of every probe run so far (bench.py prints it with PGASR_DEBUG)."""
    return [r for st in _state.values() for r in st["report"]]
