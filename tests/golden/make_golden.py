#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by RUNNING the reference's importable modules.

Run only in the authoring container (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is recorded is data only: seeded inputs and the outputs the reference's own
functions returned for them.  No reference source travels.

  CTCdecoder.CTCDecoder.decode / collapse_fn   (CTCdecoder.py:41,119)
  metrics.edit_dist / evaluate                 (metrics.py:4,23)
  loss.customNLLLoss                           (loss.py:5)
  policy_grad.reward                           (policy_grad.py:4)  -- raises as written
  model.Encoder                                (model.py:34)

model.py imports torchaudio / torchsummary (and data.py imports cvutils), none of
which is installed here and none of which the Encoder touches; SURVEY.md §8c
prescribes registering three EMPTY placeholder modules so the import statement
succeeds.  That is done below, in this harness only.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(1, ROOT)

import CTCdecoder as ref_dec      # noqa: E402
import metrics as ref_metrics     # noqa: E402
import loss as ref_loss           # noqa: E402
import policy_grad as ref_pg      # noqa: E402


def make_probs(rng, T, V, kind):
    if kind == "flat":
        p = rng.random((T, V)) + 1e-3
    elif kind == "peaky":
        p = np.exp(rng.normal(size=(T, V)) * 4.0)
    elif kind == "ctc_like":
        # mostly blank with bursts of repeated symbols, like a trained CTC model
        logit = rng.normal(size=(T, V))
        logit[:, 0] += 3.0
        t = 0
        while t < T:
            run = int(rng.integers(1, 4))
            sym = int(rng.integers(1, V))
            logit[t:t + run, sym] += 6.0
            t += run + int(rng.integers(0, 4))
        p = np.exp(logit)
    elif kind == "zeros":
        p = rng.random((T, V))
        p[rng.random((T, V)) < 0.3] = 0.0
        p[:, 0] += 1e-2
    else:
        raise ValueError(kind)
    return p / p.sum(axis=1, keepdims=True)


def gen_beam():
    dec = ref_dec.CTCDecoder([chr(97 + i) for i in range(29)])
    arrays, cases = {}, []
    cid = 0
    for (T, V) in [(1, 5), (8, 5), (8, 29), (50, 29), (200, 29)]:
        for kind in ["flat", "peaky", "ctc_like", "zeros"]:
            rng = np.random.default_rng(1000 + cid)
            probs = make_probs(rng, T, V, kind)
            key = f"p{cid}"
            arrays[key] = probs
            cid += 1
            for beam in [1, 5, 16, 100]:
                if T == 200 and beam == 100:
                    continue
                with np.errstate(divide="ignore"):
                    prefix, nll = dec.decode(probs, beam_size=beam)
                cases.append({"key": key, "T": T, "V": V, "kind": kind, "beam": beam,
                              "prefix": [int(x) for x in prefix], "nll": float(nll)})
    np.savez_compressed(os.path.join(HERE, "beam_inputs.npz"), **arrays)
    return cases


def gen_text_tables():
    collapse_in = ["", "a", "aa", "hello", "aabbcc", "abab", "aa  bb", "a  a", "zzzzz", "the  cat   sat", "abba"]
    collapse = [[s, ref_dec.collapse_fn(s)] for s in collapse_in]
    ed_pairs = [("kitten", "sitting"), ("", "abc"), ("abc", ""), ("abc", "abc"), ("flaw", "lawn"),
                ("sunday", "saturday"), ("a b c", "a x c"), ("the cat sat", "the cat sat on the mat"),
                ("abcdefghij", "jihgfedcba"), ("aaaa", "aa"), ("", "")]
    rng = np.random.default_rng(7)
    for _ in range(20):
        n1, n2 = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        a = "".join(chr(97 + int(c)) for c in rng.integers(0, 5, n1))
        b = "".join(chr(97 + int(c)) for c in rng.integers(0, 5, n2))
        ed_pairs.append((a, b))
    edit = [[a, b, list(ref_metrics.edit_dist(a, b))] for a, b in ed_pairs]
    # list-of-token inputs (the WER form)
    edit_tok = []
    for a, b in [("a b c", "a x c"), ("the cat sat", "cat sat down now"), ("x", "x")]:
        edit_tok.append([a.split(" "), b.split(" "), list(ref_metrics.edit_dist(a.split(" "), b.split(" ")))])
    ev_pairs = [("a b c", "a x c"), ("hello world", "hello word"), ("the cat sat", "the cat sat"),
                ("one two three four", "one three four five"), ("abc", "")]
    evals = [[a, b, list(ref_metrics.evaluate(a, b))] for a, b in ev_pairs]
    return {"collapse_fn": collapse, "edit_dist": edit, "edit_dist_tokens": edit_tok, "evaluate": evals}


def gen_nll():
    out = []
    for seed, (L, B, V) in enumerate([(5, 3, 7), (12, 4, 29), (1, 1, 4)]):
        g = torch.Generator().manual_seed(100 + seed)
        inp = torch.log_softmax(torch.randn(L, B, V, generator=g, dtype=torch.float64), dim=2)
        tgt = torch.randint(0, V, (B, L), generator=g)
        tgt[:, -1] = 0  # make sure padding index 0 is present
        row = {"seed": 100 + seed, "L": L, "B": B, "V": V,
               "inp": inp.tolist(), "target": tgt.tolist()}
        for name, ig in [("none", None), ("zero", 0), ("two", 2)]:
            if ig == 2 and V <= 2:
                continue
            if ig == 2 and not bool((tgt != 2).any(dim=0).all()):
                continue  # a column fully ignored gives nan in torch; skip
            row["loss_ignore_" + name] = float(ref_loss.customNLLLoss(ignore_index=ig)(inp, tgt))
        out.append(row)
    return out


def gen_reward_defect():
    dec = ref_dec.CTCDecoder(["<pad>", "a", "b", "c"])
    ind2char = {0: "<pad>", 1: "a", 2: "b", 3: "c"}
    rng = np.random.default_rng(5)
    probs = make_probs(rng, 12, 4, "ctc_like")
    res = {}
    for t in [0, 1, 2, 5]:
        try:
            r = ref_pg.reward("abc", probs, t, ind2char, dec)
            res[str(t)] = {"returned": int(r)}
        except Exception as e:  # noqa: BLE001 -- recording the defect is the point
            res[str(t)] = {"raises": type(e).__name__}
    seq, _ = dec.decode(probs, beam_size=5)
    s = ref_dec.collapse_fn("".join(ind2char[i] for i in seq))
    return {"probs": probs.tolist(), "true_y": "abc", "as_written": res, "decoded_collapsed": s}


def _placeholders():
    for name in ("torchaudio", "torchsummary", "cvutils"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "torchsummary":
                m.summary = None
            if name == "cvutils":
                m.Validator = None
                m.Alphabet = None
            sys.modules[name] = m


def gen_attention():
    """model.Attention.forward (model.py:62-92) on seeded (B,H,T) cases and the `preds` list model.Decoder.forward builds
    (model.py:109-116; the reference only prints its shape and returns None -- recorded) -> attention_cases.npz."""
    import contextlib
    import io
    _placeholders()
    import model as ref_model  # noqa: E402
    attn = ref_model.Attention()
    cases = {}
    shapes = [(2, 64, 3), (2, 64, 17), (3, 128, 17), (2, 512, 50), (1, 512, 3), (4, 96, 50)]
    for cid, (B, H, T) in enumerate(shapes):
        g = torch.Generator().manual_seed(300 + cid)
        d = torch.randn(B, H, generator=g) * 0.5
        e = torch.randn(B, T, H, generator=g) * 0.5
        with torch.no_grad():
            c = attn(d, e)
        cases[f"d{cid}"] = d.numpy(); cases[f"e{cid}"] = e.numpy(); cases[f"c{cid}"] = c.numpy()
    # Decoder: V = 29, hidden 64 / 128 (H must equal the encoder feature size; small cases keep the fixture small); preds captured by running the sub-modules in the
    # reference's own order (model.py:110-116), forward() itself called once to record its return value and what it prints
    dec_meta = []
    for did, (V, H, B, L, T) in enumerate([(29, 64, 2, 5, 7), (29, 128, 3, 4, 6)]):
        torch.manual_seed(400 + did)
        with warnings_off():
            dec = ref_model.Decoder(V, H)
        dec.eval()
        g = torch.Generator().manual_seed(500 + did)
        tgt = torch.randint(0, V, (B, L), generator=g)
        enc = torch.randn(B, T, H, generator=g) * 0.5
        buf = io.StringIO()
        with torch.no_grad(), contextlib.redirect_stdout(buf):
            ret = dec(tgt, enc)
            x = dec.embed_layer(tgt)
            dec_out, _ = dec.lstm(x)
            preds = torch.stack([torch.cat((dec_out[:, t, :], dec.attn(dec_out[:, t, :], enc)), 1) for t in range(dec_out.shape[1])])
        for k, v in dec.state_dict().items():
            cases[f"dec{did}.{k}"] = v.numpy()
        cases[f"dec{did}.targets"] = tgt.numpy(); cases[f"dec{did}.enc"] = enc.numpy(); cases[f"dec{did}.preds"] = preds.numpy()
        dec_meta.append({"V": V, "H": H, "B": B, "L": L, "T": T, "forward_returns": repr(ret), "forward_prints": buf.getvalue().strip(),
                         "state_dict": {k: list(v.shape) for k, v in dec.state_dict().items()}})
    np.savez_compressed(os.path.join(HERE, "attention_cases.npz"), **cases)
    return {"attention_shapes_BHT": shapes, "decoder": dec_meta}


@__import__("contextlib").contextmanager
def warnings_off():
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # nn.LSTM(num_layers=1, dropout=0.3) warns (model.py:103-106)
        yield


def gen_encoder():
    for name in ("torchaudio", "torchsummary", "cvutils"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            # names the import statements bind (model.py:10, data.py:10); never called
            if name == "torchsummary":
                m.summary = None
            if name == "cvutils":
                m.Validator = None
                m.Alphabet = None
            sys.modules[name] = m
    import model as ref_model  # noqa: E402
    from oracle import model_ref

    p = model_ref.init_params(n_feats=120, vocab=29, seed=0)
    enc = ref_model.Encoder()
    sd = {k: v for k, v in p.items() if not k.startswith("head.")}
    missing = enc.load_state_dict(sd, strict=True)
    enc.eval()
    cases = {}
    for cid, (B, T, lens) in enumerate([(2, 16, [16, 11]), (3, 24, [24, 24, 24]), (4, 20, [7, 20, 1, 13])]):
        g = torch.Generator().manual_seed(200 + cid)
        x = torch.randn(B, 120, T, generator=g)
        mask = torch.zeros(B, T)
        for b, n in enumerate(lens):
            mask[b, :n] = 1
            x[b, :, n:] = 0  # zero padded features like data.py:71-72
        with torch.no_grad():
            y = enc(x, mask)
        cases[f"x{cid}"] = x.numpy()
        cases[f"mask{cid}"] = mask.numpy()
        cases[f"y{cid}"] = y.numpy()
    n_params = sum(v.numel() for v in enc.state_dict().values())
    cases["n_params"] = np.array([n_params])
    cases["names"] = np.array(list(enc.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, "encoder_cases.npz"), **cases)
    return {"n_cases": 3, "n_params": int(n_params), "param_seed": 0, "load_state_dict": str(missing)}


def main():
    meta = {
        "generator": "tests/golden/make_golden.py",
        "torch": torch.__version__, "numpy": np.__version__,
        "beam": gen_beam(),
        "text": gen_text_tables(),
        "custom_nll": gen_nll(),
        "reward_defect": gen_reward_defect(),
        "encoder": gen_encoder(),
        "attention": gen_attention(),
    }
    with open(os.path.join(HERE, "reference_vectors.json"), "w") as fo:
        json.dump(meta, fo, indent=0)
    print("beam cases:", len(meta["beam"]), " encoder:", meta["encoder"])


if __name__ == "__main__":
    main()
