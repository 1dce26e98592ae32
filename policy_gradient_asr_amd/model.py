"""Drop-in for the model-side names of the reference's model.py, running on the HIP kernels.

Kept names and positional signatures: ``weights``, ``nan_to_num``, ``Encoder``,
``Attention``, ``Decoder``, ``Seq2Seq``.  Documented divergences:
  * ``Encoder(n_feats=120)``: the reference hard-codes 120 MFCC features (model.py:37-38); the
    keyword makes the F=80 benchmark shape reachable while ``Encoder()`` is unchanged.
  * ``Seq2Seq.forward`` returns (T,B,V) log-probs from a CTC head ``nn.Linear(512, V)`` +
    log_softmax (the reference's Decoder prints a shape and returns None, model.py:117; the
    consumer contract is model.py:323).  ``Attention.forward`` / ``Decoder.forward`` (model.py:58-117) run
    as the reference executes them (pinned by its own outputs); a seq2seq TRAINING mode through them does not exist in
    the reference and is out of scope.
  * dropout (model.py:45,51 p=0.5; model.py:42 p=0.3) is applied only in train() mode like the
    reference; eval() is the parity mode.
"""
import torch
import torch.nn as nn

from . import functional as Fh
from . import hipops
from . import streams

HID = 256


def weights(m):
    """Xavier-normal on nn.Linear weights, bias 0.1 (model.py:19-25).  Used via model.apply."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight.data)
        nn.init.constant_(m.bias.data, 0.1)


def nan_to_num(t, mynan=0.):
    """Non-finite entries -> mynan (model.py:27-32 does this recursively, element by element)."""
    if torch.all(torch.isfinite(t)):
        return t
    return torch.where(torch.isfinite(t), t, torch.as_tensor(mynan, dtype=t.dtype, device=t.device))


class Encoder(nn.Module):
    """InstanceNorm -> Linear(F,512)+leaky_relu(+dropout) -> 3-layer BLSTM(256) (model.py:34-56).

    Parameters carry the reference's state_dict names: input_layer.{weight,bias},
    blstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l{0,1,2}{,_reverse}."""

    def __init__(self, n_feats=120):
        super().__init__()
        self.n_feats = n_feats
        self.inst_norm = nn.InstanceNorm2d(n_feats)   # parameter-free; kept for state_dict/API parity
        self.input_layer = nn.Linear(n_feats, 512)
        self.blstm = nn.LSTM(input_size=512, hidden_size=HID, num_layers=3, dropout=0.3,
                             bidirectional=True, batch_first=True)
        self.drop = nn.Dropout()
        self.dropout_seed = 0x5EED          # masks are functions of (seed, call counter, index)
        self._drop_calls = 0

    def _layer_params(self, l):
        out = []
        for sfx in ("", "_reverse"):
            for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                out.append(getattr(self.blstm, f"{k}_l{l}{sfx}"))
        return out

    def forward_time_major(self, x, mask, lengths=None):
        """x (B,F,T), mask (B,T) -> (T,B,512) and int32 lengths (``lengths``: mask.sum(1) already on the device as int32)."""
        if not x.is_cuda:
            raise RuntimeError("policy_gradient_asr_amd.Encoder runs on the MI355X only (no CPU fallback)")
        if lengths is None:
            lengths = mask.sum(dim=1).to(torch.int32).contiguous()   # stays on the device: no host sync
        training = self.training
        # weight repacking (gate-permuted W_ih, its bf16 planes, the register-resident W_hh packs) depends on the
        # parameters only: all three layers' packs are made on a side stream while the front end runs
        packs = Fh.prepack_blstm_layers([self._layer_params(l) for l in range(3)], [512, 512, 512],
                                        rows=x.shape[0] * x.shape[2])
        # the affine's leaky'(y) factor is applied by whoever consumes y in the backward pass: the dropout that
        # follows it (train) or the first layer's input-gradient GEMM epilogue (eval)
        y = Fh.InstNormAffineFn.apply(x.float(), self.input_layer.weight, self.input_layer.bias, True)
        h = y
        if training:                                   # nn.Dropout(), model.py:45,51
            h = Fh.DropoutFn.apply(h, self.drop.p, self.dropout_seed, self._next_drop_offset(), True)
        for l in range(3):
            # nn.LSTM(dropout=0.3), model.py:42: the outputs of layers 0 and 1 are dropped on their way to the next layer --
            # by the sweep itself (its storer waves write h_t and dropout(h_t))
            od = (self.blstm.dropout, self.dropout_seed, self._next_drop_offset()) if (training and l < 2 and self.blstm.dropout > 0) else None
            h = Fh.blstm_layer(h, lengths, self._layer_params(l), dact_y=y if (l == 0 and not training) else None,
                               sweep_follows=(l > 0), prepacked=packs[l], out_dropout=od)
        return h, lengths

    def _next_drop_offset(self):
        self._drop_calls += 1
        return self._drop_calls

    def forward(self, x, mask):
        h, _ = self.forward_time_major(x, mask)
        return h.transpose(0, 1)   # (B,T,512) like model.py:56


class Attention(nn.Module):
    """model.py:58-94 as executed, on the MI355X (csrc/attention.hip, ``pgasr_attention_ctx``): per encoder frame the (H,H) outer
    product exp(d[r] e_i[k]) divided by its row sums lined up with the LAST axis (model.py:73's broadcast -- entry [r,k] by row k's
    sum; defect recorded in DESIGN.md), times the frame, summed over frames and over r.  dec_t (B,H), enc_out (B,T,H) -> (B,H).
    Forward only: the reference never trains through it (``Decoder.forward`` returns None, model.py:117)."""

    def __init__(self):
        super().__init__()

    def forward(self, dec_t, enc_out):
        from . import hipops
        return hipops.attention_ctx(dec_t.contiguous(), enc_out.contiguous())


class Decoder(nn.Module):
    """model.py:99-117: Decoder(alphabet_size, hidden_size) -- embedding(128) -> one-layer LSTM(128 -> hidden) -> per decoder step
    the attention context -> cat(dec_out[:, t], c_t).  Same parameter names / shapes as the reference (``embed_layer.weight``,
    ``lstm.*_l0``; ``nn.LSTM`` is a parameter container here, never called: its cuDNN/MIOpen path is not used).
    DOCUMENTED DIVERGENCE (like ``Seq2Seq.forward``): the reference builds the list `preds`, prints the shape of its stack and returns
    None (model.py:112-117); this returns that stack, (L, B, 2 * hidden), and prints nothing.  Forward only.
    The recurrent part is L dependent steps of (B,hidden) x (hidden,4 hidden) on the exact fp32 MFMA GEMM plus one cell kernel
    (``pgasr_lstm_cell_f32``) -- the BLSTM sweep kernels are built for H = 256 bidirectional layers and do not take this shape."""

    def __init__(self, alphabet_size, hidden_size):
        super().__init__()
        self.alphabet_size = alphabet_size
        self.hidden_size = hidden_size
        self.embed_layer = nn.Embedding(alphabet_size, 128)
        self.lstm = nn.LSTM(input_size=128, hidden_size=hidden_size, num_layers=1, batch_first=True)   # reference: dropout=0.3, a no-op with one layer
        self.attn = Attention()

    def forward(self, target_inputs, encoder_outputs, device=None):
        from . import hipops
        H = self.hidden_size
        w_ih, w_hh = self.lstm.weight_ih_l0.detach().contiguous(), self.lstm.weight_hh_l0.detach().contiguous()
        bias = (self.lstm.bias_ih_l0 + self.lstm.bias_hh_l0).detach().contiguous()
        if not w_ih.is_cuda:
            raise RuntimeError("Decoder.forward runs on the MI355X only (no CPU path)")
        B, L = target_inputs.shape
        x = self.embed_layer.weight.detach()[target_inputs.to(w_ih.device)].transpose(0, 1).contiguous()      # (L,B,128), time-major
        xp = torch.empty(L * B, 4 * H, dtype=torch.float32, device=w_ih.device)
        hipops.gemm(x.view(L * B, 128), w_ih, xp, M=L * B, N=4 * H, K=128, transB=True, bias=bias, precision=0)
        xp = xp.view(L, B, 4 * H)
        h = torch.zeros(B, H, dtype=torch.float32, device=w_ih.device)
        c = torch.zeros_like(h)
        g = torch.empty(B, 4 * H, dtype=torch.float32, device=w_ih.device)
        dec_out = torch.empty(L, B, H, dtype=torch.float32, device=w_ih.device)
        for t in range(L):
            hipops.gemm(h, w_hh, g, M=B, N=4 * H, K=H, transB=True, precision=0)
            hipops.lstm_cell(g, xp[t], c, h, dec_out[t])          # gates, c, h in one kernel (csrc/attention.hip)
        ctx = hipops.attention_ctx(dec_out, encoder_outputs.to(w_ih.device).contiguous())                       # (L,B,H): row q = t * B + b
        return torch.cat((dec_out, ctx), dim=2)


class Seq2Seq(nn.Module):
    """Seq2Seq(alphabet_size).forward(x, t, fmask, device) (model.py:174-183) with a CTC head."""

    def __init__(self, alphabet_size, n_feats=120):
        super().__init__()
        self.encoder = Encoder(n_feats=n_feats)
        self.head = nn.Linear(2 * HID, alphabet_size)

    def logits(self, x, fmask, lengths=None):
        """(T,B,V) pre-softmax scores and int32 lengths -- what the fused loss consumes."""
        h, lengths = self.encoder.forward_time_major(x, fmask, lengths)
        if hipops.head_logsoftmax_ok(self.head.in_features, self.head.out_features):
            # head + log-softmax in one kernel; the log-probs ride along for the fused loss (pg_ctc_loss takes them from the attribute)
            z, lp = Fh.HeadFn.apply(h, self.head.weight, self.head.bias)
            z.log_probs = lp
            z.log_probs_version = z._version      # an in-place edit of z afterwards (temperature, masking) invalidates lp: pg_ctc_loss checks
            return z, lengths
        return Fh.LinearFn.apply(h, self.head.weight, self.head.bias), lengths

    def forward(self, x, t, fmask, device=None):
        z, _ = self.logits(x, fmask)
        return Fh.LogSoftmaxFn.apply(z)        # differentiable log-probs for callers outside the fused loss


# ------------------------------------------------------------------------------------------
# driver shells: same positional signatures as model.py:186 / :277
# ------------------------------------------------------------------------------------------
def _read_alphabet(alphabet_path):
    with open(alphabet_path, "r") as fo:
        alphabet = ["<pad>"] + fo.readlines()                       # model.py:194-195
    char2ind = {alphabet[i].replace("\n", ""): i for i in range(len(alphabet))}   # model.py:197
    return alphabet, char2ind


def _to_device(batch, device):
    x = batch["feat"].to(device)                                    # model.py:227-230
    t = batch["trans"].to(device)
    fmask = batch["fmask"].squeeze(1).to(device)
    tmask = batch["tmask"].squeeze(1).to(device)
    return x, t, fmask, tmask


FEATURE_DIMS = {"mfcc": 120, "logmel80": 80}      # what data.collate_custom(features=...) produces from a waveform


def _check_features(features, n_feats, dataset):
    """A waveform-carrying dataset meets the model through the front end: its width must be the model's n_feats (a mismatch
    used to surface as a shape error at the instance norm of the first step)."""
    if features not in FEATURE_DIMS:
        raise ValueError(f"features must be one of {sorted(FEATURE_DIMS)}")
    try:
        item = dataset[0] if dataset is not None and len(dataset) > 0 else None
    except Exception:
        item = None
    if isinstance(item, dict) and ("wave" in item or "aud" in item) and "feat" not in item:
        if n_feats != FEATURE_DIMS[features]:
            raise ValueError(f"features='{features}' gives {FEATURE_DIMS[features]} features per frame but n_feats={n_feats}")


def train(corpus_path, model_path, num_epochs, batch_size, device, train_dataset=None, dev_dataset=None,
          n_feats=120, lam=1.0, lr=5e-4, resume=True, log_every=10, seed=0, bucket_by_length=True, features="mfcc",
          precision="f32"):
    """Epoch loop of model.py:186-274 on the MI355X path: per-epoch train loss -> train_loss.npy,
    validation CTC loss -> val_losses.npy, model_best.pth / model_last.pth (state_dicts, reference
    names), plus checkpoint_last.pth (model + Adam moments + epoch) from which ``resume`` restarts
    -- the reference saves weights only.  ``train_dataset`` / ``dev_dataset`` default to the
    reference's Data(train.tsv / dev.tsv, clips) and accept any Dataset of collate_custom items.
    features: "mfcc" (the reference's 120 MFCC + delta features, n_feats=120) or "logmel80" (80-band log-mel, n_feats=80) for
    items that carry waveforms / audio paths; the collate function leaves the features on the GPU (data.collate_custom(device=...)):
    from the front end to the trainer they never visit the host.
    precision: "f32" (default: the reference's torch-fp32 arithmetic, model.py:38-44) or "bf16x3" (opt-in, ~20 % faster,
    within 1e-3 on loss and gradients) -- hipops.PRECISION_MODES."""
    import os
    import numpy as np
    import torch.utils.data as tud
    import functools
    from .data import Data, LengthBucketSampler, dataset_lengths
    from .data import collate_custom as _collate
    from .loss import pg_ctc_loss
    from .train_step import PolicyGradientTrainer

    print("Num epochs:", num_epochs, "Batch size:", batch_size)
    alphabet, char2ind = _read_alphabet(os.path.join(corpus_path, "alphabet.txt"))
    dev = torch.device("cuda", device if isinstance(device, int) else 0) if not isinstance(device, torch.device) else device
    os.makedirs(model_path, exist_ok=True)
    collate_custom = functools.partial(_collate, device=dev, features=features)
    if train_dataset is None:
        train_dataset = Data(os.path.join(corpus_path, "train.tsv"), os.path.join(corpus_path, "clips"), char2ind)
    if dev_dataset is None and os.path.exists(os.path.join(corpus_path, "dev.tsv")):
        dev_dataset = Data(os.path.join(corpus_path, "dev.tsv"), os.path.join(corpus_path, "clips"), char2ind)

    torch.manual_seed(seed)
    model = Seq2Seq(alphabet_size=len(char2ind), n_feats=n_feats)
    model.apply(weights)                                            # model.py:202
    model = model.to(dev)
    _check_features(features, n_feats, train_dataset)
    trainer = PolicyGradientTrainer(model, lr=lr, lam=lam, seed=seed, precision=precision)
    losses, val_losses, best, start_epoch = [], [], 9999999.0, 1
    ckpt = os.path.join(model_path, "checkpoint_last.pth")
    if resume and os.path.exists(ckpt):
        st = torch.load(ckpt, map_location=dev)
        model.load_state_dict(st["model"])
        from .train_step import FLAG_PAD
        # the checkpoint holds the moments of the PARAMETERS (the flat buffers' leading flag words are not state)
        trainer.exp_avg[FLAG_PAD:].copy_(st["exp_avg"]); trainer.exp_avg_sq[FLAG_PAD:].copy_(st["exp_avg_sq"]); trainer.nstep = st["nstep"]
        trainer.set_applied_steps(st.get("applied_steps", st["nstep"]))     # Adam's bias correction counts applied updates
        # the dropout masks are functions of (seed, call counter): restore both, or a resumed run replays the first
        # epoch's masks and differs from an uninterrupted one
        model.encoder._drop_calls = st.get("drop_calls", 0)
        model.encoder.dropout_seed = st.get("dropout_seed", model.encoder.dropout_seed)
        for k, want in (("lr", lr), ("lam", lam)):
            if k in st and st[k] != want:
                print("Warning: resuming with {}={} but the checkpoint was written with {}".format(k, want, st[k]))
        losses, val_losses, best, start_epoch = st["losses"], st["val_losses"], st["best"], st["epoch"] + 1
        print("Resumed from epoch", st["epoch"])

    print("Start training...")
    for epoch in range(start_epoch, num_epochs + 1):
        model.train()
        lens = dataset_lengths(train_dataset) if bucket_by_length else None
        if lens is not None:      # similar lengths per batch instead of model.py:221's shuffle=True
            sampler = LengthBucketSampler(lens, batch_size, seed=seed)
            sampler.set_epoch(epoch)
            loader = tud.DataLoader(train_dataset, batch_sampler=sampler, collate_fn=collate_custom)
        else:
            loader = tud.DataLoader(train_dataset, batch_size=batch_size, shuffle=True, collate_fn=collate_custom)
        acc = torch.zeros((), device=dev)
        for step, batch in enumerate(loader, 1):
            loss = trainer.step(*_to_device(batch, dev))
            acc += loss
            if log_every and step % log_every == 0:
                val = float(loss)                      # the host synchronises here anyway: check the sweeps' error words
                hipops.lstm_assert_no_timeouts()       # (until then the guarded Adam has skipped every invalid update)
                print("Step {}/{}. Loss: {:>4f}".format(step, len(loader), val))
        losses.append(float(acc) / max(len(loader), 1))
        hipops.lstm_assert_no_timeouts()          # .. and before anything of this epoch is written to disk
        streams.release()                         # the host has synchronised: nothing of the last step needs keeping alive
        np.save(os.path.join(model_path, "train_loss.npy"), np.array(losses))
        print("Epoch:{}/{} Training loss:{:>4f}".format(epoch, num_epochs, losses[-1]))

        curr = losses[-1]
        if dev_dataset is not None:                                 # validation (model.py:246-268)
            model.eval()
            vl = torch.zeros((), device=dev)
            vloader = tud.DataLoader(dev_dataset, batch_size=batch_size, shuffle=False, collate_fn=collate_custom)
            with torch.no_grad():
                for batch in vloader:
                    x, t, fmask, tmask = _to_device(batch, dev)
                    logits, in_len = model.logits(x, fmask)
                    l, _, _, _ = pg_ctc_loss(logits, in_len, t.to(torch.int32).contiguous(),
                                             tmask.sum(1).to(torch.int32).contiguous(), lam=0.0)
                    vl += l
            curr = float(vl) / max(len(vloader), 1)
            val_losses.append(curr)
            np.save(os.path.join(model_path, "val_losses.npy"), np.array(val_losses))
            print("Epoch:{}/{} Validation loss:{:>4f}".format(epoch, num_epochs, curr))
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        if curr < best:                                             # model selection (model.py:270-274)
            torch.save(sd, os.path.join(model_path, "model_best.pth"))
            best = curr
        torch.save(sd, os.path.join(model_path, "model_last.pth"))
        from .train_step import FLAG_PAD
        torch.save({"model": sd, "exp_avg": trainer.exp_avg[FLAG_PAD:], "exp_avg_sq": trainer.exp_avg_sq[FLAG_PAD:], "nstep": trainer.nstep,
                    "applied_steps": trainer.applied_steps(),
                    "losses": losses, "val_losses": val_losses, "best": best, "epoch": epoch,
                    "drop_calls": model.encoder._drop_calls, "dropout_seed": model.encoder.dropout_seed,
                    "lr": lr, "lam": lam}, ckpt)
    return losses, val_losses


def predict(test_path, aud_path, alphabet_path, model_path, batch_size, maxlen=None, maxlent=None, device_id=0,
            test_dataset=None, n_feats=120, beam_size=5, features="mfcc"):
    """model.py:277-339: load model_best.pth, forward, beam=5 prefix search (device side, batched),
    collapse_fn, CER/WER, predicted.txt.  Frames are cut by the FEATURE mask (the reference cuts the
    time axis by the target mask, model.py:322 -- a listed defect).  Returns (CER, WER).
    features: the front end for items that carry waveforms, as in ``train`` ("mfcc": n_feats=120, "logmel80": n_feats=80);
    the features stay on the device."""
    import os
    import functools
    import torch.utils.data as tud
    from .CTCdecoder import CTCDecoder, collapse_fn
    from .data import Data
    from .data import collate_custom as _collate
    from .metrics import evaluate, save_predictions

    alphabet, char2ind = _read_alphabet(alphabet_path)
    ind2char = {char2ind[k]: k for k in char2ind}
    dev = torch.device("cuda", device_id)
    model = Seq2Seq(alphabet_size=len(alphabet), n_feats=n_feats)
    model.load_state_dict(torch.load(os.path.join(model_path, "model_best.pth"), map_location="cpu"))
    model = model.to(dev).eval()
    if test_dataset is None:
        test_dataset = Data(test_path, aud_path, char2ind)
    _check_features(features, n_feats, test_dataset)
    collate_custom = functools.partial(_collate, device=dev, features=features)
    loader = tud.DataLoader(test_dataset, batch_size=batch_size, shuffle=False, collate_fn=collate_custom)
    decoder = CTCDecoder(alphabet)
    targets, predicted, tot_cer, tot_wer, n = [], [], 0.0, 0.0, 0
    print("Total number of examples: ", len(test_dataset))
    with torch.no_grad():
        for step, batch in enumerate(loader, 1):
            print("Decoding step {}/{}...".format(step, len(loader)))
            x, t, fmask, tmask = _to_device(batch, dev)
            logits, in_len = model.logits(x, fmask)
            lp = Fh.LogSoftmaxFn.apply(logits)
            tok, tl, _ = decoder.decode_batch(lp, in_len, beam_size=beam_size)
            tok, tl, t, tmask = tok.cpu(), tl.cpu(), t.cpu(), tmask.cpu()
            for i in range(tok.shape[0]):
                seq = collapse_fn("".join(ind2char[int(k)] for k in tok[i, :tl[i]]))
                target = "".join(ind2char[int(k)] for k in t[i][:int(tmask[i].sum())])
                targets.append(target); predicted.append(seq)
                cer, wer = evaluate(target, seq)
                tot_cer += cer; tot_wer += wer; n += 1
    save_predictions(targets, predicted, model_path)
    cer, wer = tot_cer / max(n, 1), tot_wer / max(n, 1)
    print("CER: {:>4f} WER: {:>4f}".format(cer, wer))
    return cer, wer
