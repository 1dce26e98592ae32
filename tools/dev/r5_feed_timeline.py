"""When does the LAST fed backward sweep of an f32 train step wait, and when were its row tiles ready?  (development aid, round 5)
Needs the diagnostic library (make -C policy_gradient_asr_amd/csrc lstmdiag; PGASR_HIP_LIB=.../libpgasr_hip_diag.so).  One step after a
warm-up; prints, per cluster of the layer-0 backward sweep, the steps at which member 0's loader found the staging ring late (step, us
waited, us since the loader's first request) and, for the feed GEMM of that sweep, when each of the first row tiles of either direction was
counted complete (us since the same origin)."""
import os, sys, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from policy_gradient_asr_amd import hipops, _lib
from policy_gradient_asr_amd.model import Seq2Seq, weights
from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
from bench import synth_batch, V, F
lib = _lib.load()
if not hasattr(lib, "pgasr_diag_lstm_late"):
    sys.exit("needs the -DPGASR_LSTM_DIAG library (make lstmdiag, PGASR_HIP_LIB)")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Seq2Seq(V, n_feats=F); m.apply(weights); m = m.to(dev).train()
tr = PolicyGradientTrainer(m, lr=0.0, lam=1.0, seed=1234, precision=os.environ.get("PREC", "f32"))
if os.environ.get("NOWG"):                    # no weight-gradient products (results invalid; lr = 0)
    hipops.lstm_wgrads = lambda *a, **k: None
batch = [v.to(dev) for v in synth_batch(100)]
for i in range(5):
    tr.step(*batch)
torch.cuda.synchronize()
late = (ctypes.c_uint * 512)(); feed = (ctypes.c_uint * 4096)()
lib.pgasr_diag_lstm_late(late, 1); lib.pgasr_diag_x6_feed(feed, 1)
tr.step(*batch)
torch.cuda.synchronize()
lib.pgasr_diag_lstm_late(late, 0); lib.pgasr_diag_x6_feed(feed, 0)
T, B = 1000, 32
mt = (T * B + 255) // 256
org = min(late[64 * c + 60] for c in range(4) if late[64 * c + 60])
def us(w):
    return round(((w - org) & 0xFFFFFFFF) / 100.0, 1) if w else None
print(json.dumps({"origin": "first loader request of the sweep (100 MHz wall clock)", "first_four_quarter_items_drawn_us": [us(feed[4 + i]) for i in range(4)],
                  "parked_us": [us(feed[8 + i]) for i in range(4)], "first_four_whole_tiles_drawn_us": [us(feed[1024 + i]) for i in range(4)],
                  "counted_us": [us(feed[1028 + i]) for i in range(4)]}))
for c in range(4):
    rows = []
    for k in range(16):
        s, cyc, w = late[64 * c + 3 * k], late[64 * c + 3 * k + 1], late[64 * c + 3 * k + 2]
        if w:
            rows.append({"step": s, "waited_us": round(cyc * 16 / 2400.0, 1), "at_us": us(w)})
    marks = [us(late[64 * c + 48 + i]) for i in range(8)]
    print(json.dumps({"cluster": c, "direction": c & 1, "loader_first_request_us": us(late[64 * c + 60]), "late_waits": rows,
                      "loader_asks_for_step_0_128_.._896_at_us": marks,
                      "us_per_step_between_marks": [round((b - a) / 128.0, 3) if (a is not None and b is not None) else None for a, b in zip(marks, marks[1:])],
                      "shader_clock_MHz_between_marks": [round(((late[64 * c + 41 + i] - late[64 * c + 40 + i]) & 0xFFFFFFFF) / max((((late[64 * c + 49 + i] - late[64 * c + 48 + i]) & 0xFFFFFFFF) / 100.0), 1e-9))
                                                         for i in range(7)]}))
for d in range(2):
    # direction 0's backward sweep walks down in time: its first row tile is the LAST one
    order = range(mt - 1, mt - 41, -1) if d == 0 else range(0, 40)
    print(json.dumps({"direction": d, "row_tile_ready_us_in_consumption_order(8 steps each)": [us(feed[16 + d * mt + i]) for i in order]}))
