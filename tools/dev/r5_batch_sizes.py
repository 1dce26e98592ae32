"""Train-step time against the local batch size (development aid, round 5): which orders a batch size takes and what it costs.
B <= 32: feed-ahead; B % 16 == 0 ("f32") / B % 32 == 0 ("bf16x3"): streamed weight gradients; other sizes <= 32 are padded with empty
utterances (PolicyGradientTrainer.pad_ragged_batches); B > 32: the sequential order (the sweeps occupy 6 or 8 XCDs).  PREC, STEPS."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops, functional as Fh
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
import bench

prec = os.environ.get("PREC", "f32")
steps = int(os.environ.get("STEPS", "12"))
dev = torch.device("cuda:0")
full = [v.to(dev) for v in bench.synth_batch(100)]
for B, pad in [(int(b), p) for b in os.environ.get("BATCHES", "32,16,8,24,48,64").split(",") for p in ((True, False) if int(b) % 16 else (True,))]:
    torch.manual_seed(0)
    m = Seq2Seq(bench.V, n_feats=bench.F); m.apply(weights); m = m.to(dev).train()
    tr = PolicyGradientTrainer(m, lr=5e-4, lam=1.0, seed=1234, precision=prec)
    tr.pad_ragged_batches = pad
    reps = -(-B // full[0].shape[0])
    batch = [torch.cat([v] * reps, dim=0)[:B].contiguous() for v in full]
    for _ in range(3):
        tr.step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(*batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    hipops.lstm_assert_no_timeouts()
    Bp = tr._padded(*batch)[0].shape[0]
    with hipops.precision(prec):
        fed, streamed = bool(hipops.lstm_fed_ok(1000, Bp)), bool(hipops.lstm_wgrads_ok(1000, Bp, 512) and hipops.lstm_fed_ok(1000, Bp))
    print(json.dumps({"precision": prec, "B": B, "padded_to": Bp, "feed_ahead": fed and Fh.FEED_AHEAD, "streamed_weight_gradients": streamed and Fh.STREAM_DW,
                      "ms_per_step": round(ms, 3), "utt_per_s": round(B / ms * 1e3, 1), "us_per_utterance": round(ms / B * 1e3, 1)}), flush=True)
    del tr, m
