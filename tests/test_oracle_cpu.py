"""The oracle against the reference's own outputs (tests/golden, produced by
tests/golden/make_golden.py) and against torch-CPU for the pieces the reference
lacks.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ctc_ref, decode_ref, model_ref


@pytest.fixture(scope="module")
def vectors(golden_dir):
    with open(os.path.join(golden_dir, "reference_vectors.json")) as fi:
        return json.load(fi)


@pytest.fixture(scope="module")
def beam_inputs(golden_dir):
    return np.load(os.path.join(golden_dir, "beam_inputs.npz"))


def test_prefix_beam_matches_reference(vectors, beam_inputs):
    n = 0
    for case in vectors["beam"]:
        if case["T"] == 200 and case["beam"] == 16:
            continue  # slow in pure python; covered by the GPU parity test
        probs = beam_inputs[case["key"]]
        prefix, nll = decode_ref.prefix_beam_search(probs, beam_size=case["beam"])
        assert list(prefix) == case["prefix"], case
        assert nll == pytest.approx(case["nll"], rel=1e-12, abs=1e-12), case
        n += 1
    assert n >= 60


def test_collapse_fn_table(vectors):
    for s, want in vectors["text"]["collapse_fn"]:
        assert decode_ref.collapse_fn(s) == want


def test_edit_dist_tables(vectors):
    for a, b, want in vectors["text"]["edit_dist"]:
        assert list(decode_ref.edit_dist(a, b)) == want
    for a, b, want in vectors["text"]["edit_dist_tokens"]:
        assert list(decode_ref.edit_dist(a, b)) == want
    assert decode_ref.edit_dist("kitten", "sitting") == (3, 6)
    assert decode_ref.edit_dist("", "abc") == (3, 0)


def test_evaluate_table(vectors):
    for a, b, want in vectors["text"]["evaluate"]:
        got = decode_ref.evaluate(a, b)
        assert got[0] == pytest.approx(want[0]) and got[1] == pytest.approx(want[1])
    with pytest.raises(ZeroDivisionError):
        decode_ref.evaluate("", "abc")


def test_custom_nll(vectors):
    for row in vectors["custom_nll"]:
        inp = torch.tensor(row["inp"], dtype=torch.float64)
        tgt = torch.tensor(row["target"])
        assert float(model_ref.custom_nll(inp, tgt, None)) == pytest.approx(row["loss_ignore_none"], rel=1e-12)
        # ignore_index=0 is falsy in loss.py:9 -> padding is NOT ignored
        assert float(model_ref.custom_nll(inp, tgt, 0)) == pytest.approx(row["loss_ignore_zero"], rel=1e-12)
        assert row["loss_ignore_zero"] == pytest.approx(row["loss_ignore_none"], rel=1e-12)
        if "loss_ignore_two" in row:
            assert float(model_ref.custom_nll(inp, tgt, 2)) == pytest.approx(row["loss_ignore_two"], rel=1e-12)


def test_reward_defect_and_intent(vectors):
    rd = vectors["reward_defect"]
    # the reference raises for every t
    assert rd["as_written"]["0"]["raises"] == "UnboundLocalError"
    for t in ("1", "2", "5"):
        assert rd["as_written"][t]["raises"] == "TypeError"
    probs = np.array(rd["probs"])
    ind2char = {0: "<pad>", 1: "a", 2: "b", 3: "c"}
    seq, _ = decode_ref.prefix_beam_search(probs, beam_size=5)
    s = decode_ref.collapse_fn("".join(ind2char[i] for i in seq))
    assert s == rd["decoded_collapsed"]
    y = rd["true_y"]
    # telescoping property of the intended reward
    total = sum(decode_ref.reward_from_string(y, s, t) for t in range(1, len(s) + 2))
    assert total == len(y) - decode_ref.edit_dist(y, s)[0]
    with pytest.raises(ValueError):
        decode_ref.reward_from_string(y, s, 0)


def test_encoder_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "encoder_cases.npz"))
    assert int(z["n_params"][0]) == 4792832
    assert list(z["names"]) == model_ref.param_names()
    p = model_ref.init_params(n_feats=120, vocab=29, seed=0)
    for cid in range(3):
        x = torch.from_numpy(z[f"x{cid}"]); mask = torch.from_numpy(z[f"mask{cid}"])
        with torch.no_grad():
            y = model_ref.encoder_forward_torch(p, x, mask).numpy()
        np.testing.assert_allclose(y, z[f"y{cid}"], rtol=1e-5, atol=1e-6)
        # zeros past each length
        lens = mask.sum(1).int().tolist()
        for b, n in enumerate(lens):
            assert np.all(y[b, n:] == 0)


def test_fast_packed_oracle_equals_packed_sequence_path(golden_dir):
    """model_ref.encoder_forward_torch(fast_packed=True) -- the packed semantics of model.py:52-55 by reversing every utterance
    within its own length instead of torch's PackedSequence path (which needs minutes per full-size case on the CPU) -- against (a)
    the reference's own Encoder outputs (the golden cases, ragged lengths included) and (b) the PackedSequence path in fp64:
    outputs and every parameter gradient, lengths down to 1."""
    z = np.load(os.path.join(golden_dir, "encoder_cases.npz"))
    p = model_ref.init_params(n_feats=120, vocab=29, seed=0)
    for cid in range(3):
        x = torch.from_numpy(z[f"x{cid}"]); mask = torch.from_numpy(z[f"mask{cid}"])
        with torch.no_grad():
            y = model_ref.encoder_forward_torch(p, x, mask, fast_packed=True).numpy()
        np.testing.assert_allclose(y, z[f"y{cid}"], rtol=1e-5, atol=1e-6)
        for b, n in enumerate(mask.sum(1).int().tolist()):
            assert np.all(y[b, n:] == 0)
    g = torch.Generator().manual_seed(3)
    B, F, T = 5, 120, 23
    lens = [23, 1, 17, 8, 2]
    x = torch.randn(B, F, T, generator=g, dtype=torch.float64)
    mask = torch.zeros(B, T)
    for b, n in enumerate(lens):
        mask[b, :n] = 1; x[b, :, n:] = 0
    dy = torch.randn(B, T, 512, generator=g, dtype=torch.float64)
    res = []
    for fast in (False, True):
        pr = {k: v.double().requires_grad_(True) for k, v in p.items() if not k.startswith("head.")}
        y = model_ref.encoder_forward_torch(pr, x, mask, fast_packed=fast)
        y.backward(dy)
        res.append((y.detach(), {k: v.grad for k, v in pr.items()}))
    np.testing.assert_allclose(res[1][0].numpy(), res[0][0].numpy(), rtol=1e-12, atol=1e-13)
    for k in res[0][1]:
        np.testing.assert_allclose(res[1][1][k].numpy(), res[0][1][k].numpy(), rtol=1e-9, atol=1e-12, err_msg=k)


def test_blstm_numpy_matches_packed_torch(golden_dir):
    z = np.load(os.path.join(golden_dir, "encoder_cases.npz"))
    p = model_ref.init_params(n_feats=120, vocab=29, seed=0)
    x = torch.from_numpy(z["x2"]); mask = torch.from_numpy(z["mask2"])
    h = model_ref.instance_norm(x).transpose(1, 2)
    h = torch.nn.functional.leaky_relu(torch.nn.functional.linear(h, p["input_layer.weight"], p["input_layer.bias"]))
    pn = {k: v.double().numpy() for k, v in p.items()}
    y = model_ref.blstm_numpy(pn, h.double().numpy(), mask.sum(1).numpy())
    np.testing.assert_allclose(y, z["y2"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("T,B,V,L", [(12, 3, 5, 4), (30, 4, 29, 6), (7, 2, 4, 3), (5, 1, 3, 0)])
def test_ctc_oracle_matches_torch(T, B, V, L):
    g = torch.Generator().manual_seed(T * 100 + V)
    logits = torch.randn(T, B, V, generator=g, dtype=torch.float64, requires_grad=True)
    Lmax = max(L, 1)
    targets = torch.randint(1, V, (B, Lmax), generator=g)
    if L >= 2:
        targets[0, 1] = targets[0, 0]  # force a repeat
    in_len = torch.tensor([max(T - 2 * b, 2 * L + 1) for b in range(B)])
    tg_len = torch.tensor([max(L - b, 0) for b in range(B)]) if L else torch.zeros(B, dtype=torch.long)
    lp = torch.log_softmax(logits, dim=2)
    loss = torch.nn.functional.ctc_loss(lp, targets, in_len, tg_len, blank=0, reduction="none")
    loss.sum().backward()
    nll, grad = ctc_ref.ctc_loss_and_grad(logits.detach().numpy(), targets.numpy(), in_len.numpy(), tg_len.numpy())
    np.testing.assert_allclose(nll, loss.detach().numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(grad, logits.grad.numpy(), rtol=1e-8, atol=1e-10)
    mean = torch.nn.functional.ctc_loss(lp, targets, in_len, tg_len, blank=0, reduction="mean")
    assert ctc_ref.ctc_reduce_mean(nll, tg_len.numpy()) == pytest.approx(float(mean.detach()), rel=1e-10)


def test_greedy_differs_from_beam1_and_collapses():
    sc = np.full((6, 1, 4), -5.0)
    for t, k in enumerate([1, 1, 0, 1, 2, 2]):
        sc[t, 0, k] = 0.0
    assert decode_ref.greedy_decode(sc) == [[1, 1, 2]]
    assert decode_ref.greedy_decode(sc, lengths=[2]) == [[1]]
    # first max wins
    tie = np.zeros((1, 1, 3))
    assert decode_ref.greedy_decode(tie) == [[]]


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    z = np.zeros(1, dtype=np.uint32)
    out = decode_ref.philox4x32_10(z, z, z, z, 0, 0)
    assert [int(o[0]) for o in out] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = np.full(1, 0xFFFFFFFF, dtype=np.uint32)
    out = decode_ref.philox4x32_10(f, f, f, f, 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(o[0]) for o in out] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    a = decode_ref.philox4x32_10(np.array([0x243f6a88], dtype=np.uint32), np.array([0x85a308d3], dtype=np.uint32),
                                 np.array([0x13198a2e], dtype=np.uint32), np.array([0x03707344], dtype=np.uint32),
                                 0xa4093822, 0x299f31d0)
    assert [int(o[0]) for o in a] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_sampler_and_reinforce_grad():
    rng = np.random.default_rng(0)
    logits = rng.normal(size=(40, 3, 6)) * 2
    paths, cdf, u = decode_ref.sample_paths(logits, seed=1234)
    assert paths.shape == (40, 3) and paths.min() >= 0 and paths.max() < 6
    # empirical check of the inverse CDF rule
    for t in range(40):
        for b in range(3):
            k = paths[t, b]
            lo = 0.0 if k == 0 else cdf[t, b, k - 1]
            assert lo <= u[t, b] < cdf[t, b, k] or k == 5
    coef = np.array([0.5, -1.0, 2.0]); lens = np.array([40, 10, 0])
    g = decode_ref.reinforce_grad(logits, paths, coef, lens)
    lt = torch.tensor(logits, requires_grad=True)
    lp = torch.log_softmax(lt, dim=2)
    picked = lp.gather(2, torch.tensor(paths)[..., None])[..., 0]
    mask = (torch.arange(40)[:, None] < torch.tensor(lens)[None, :]).double()
    obj = -(picked * mask * torch.tensor(coef)[None, :]).sum()
    obj.backward()
    np.testing.assert_allclose(g, lt.grad.numpy(), rtol=1e-9, atol=1e-12)


def test_feature_oracle_stages_are_pinned():
    """oracle/features_ref.py restates torchaudio (absent here): pin the stages that have an independent
    implementation in this image -- STFT against torch.stft, DCT against scipy, deltas against a convolution."""
    import scipy.fft
    from oracle import features_ref as fr
    rng = np.random.default_rng(0)
    w = rng.standard_normal(5000) * 0.1
    p = fr.power_spectrogram(w)
    st = torch.stft(torch.from_numpy(w), n_fft=400, hop_length=200, win_length=400,
                    window=torch.hann_window(400, dtype=torch.float64), center=True, pad_mode="reflect",
                    normalized=False, onesided=True, return_complex=True)
    np.testing.assert_allclose(p, (st.abs() ** 2).T.numpy(), rtol=1e-10, atol=1e-14)
    x = rng.standard_normal((7, 128))
    np.testing.assert_allclose(x @ fr.dct_matrix(40, 128), scipy.fft.dct(x, type=2, norm="ortho", axis=1)[:, :40], atol=1e-12)
    y = rng.standard_normal((3, 11))
    pad = np.pad(y, ((0, 0), (2, 2)), mode="edge")
    want = sum(m * pad[:, 2 + m:2 + m + 11] for m in range(-2, 3)) / 10.0
    np.testing.assert_allclose(fr.compute_deltas(y), want, atol=1e-12)
    fb = fr.mel_filterbank()
    assert fb.shape == (201, 128) and (fb >= 0).all() and fb.max() <= 1.0
    f = fr.mfcc_deltas(w)
    assert f.shape == (120, 26)
    feats, mask = fr.extract_feats([w, w[:3000]])
    assert feats.shape == (2, 120, 26) and mask[1, 0].sum() == 16 and np.all(feats[1, :, 16:] == 0)


def test_length_bucket_sampler():
    from policy_gradient_asr_amd.data import LengthBucketSampler
    g = torch.Generator().manual_seed(0)
    lengths = torch.randint(50, 1000, (203,), generator=g).tolist()
    s = LengthBucketSampler(lengths, batch_size=8, bucket_batches=4, seed=3)
    batches = list(s)
    assert len(batches) == len(s) == 26
    assert sorted(i for b in batches for i in b) == list(range(203))          # every utterance exactly once
    spread = max(max(lengths[i] for i in b) - min(lengths[i] for i in b) for b in batches if len(b) == 8)
    assert spread < 0.35 * (max(lengths) - min(lengths))                       # similar lengths inside a batch
    assert [b for b in s] == batches                                            # deterministic per epoch
    s.set_epoch(1)
    assert [b for b in s] != batches
    assert len(list(LengthBucketSampler(lengths, 8, drop_last=True))) == 25


def test_attention_decoder_oracle_matches_reference(golden_dir, vectors):
    """N4 oracle pin: oracle/attn_ref.py against the outputs of the reference's own ``model.Attention`` / ``model.Decoder`` (model.py:58-117)
    recorded by tests/golden/make_golden.py -- the broadcasting quirk of model.py:73 included -- and what the reference's
    ``Decoder.forward`` itself returns (None) and prints (the shape of the stack)."""
    from oracle import attn_ref
    z = np.load(os.path.join(golden_dir, "attention_cases.npz"))
    meta = vectors["attention"]
    assert len(meta["attention_shapes_BHT"]) == 6
    for cid, (B, H, T) in enumerate(meta["attention_shapes_BHT"]):
        d, e, c = z[f"d{cid}"], z[f"e{cid}"], z[f"c{cid}"]
        assert d.shape == (B, H) and e.shape == (B, T, H) and c.shape == (B, H)
        assert np.abs(attn_ref.attention_ctx(d, e) - c).max() / np.abs(c).max() < 1e-5
    for did, m in enumerate(meta["decoder"]):
        assert m["forward_returns"] == "None" and m["forward_prints"] == f"torch.Size([{m['L']}, {m['B']}, {2 * m['H']}])"
        pre = f"dec{did}."
        params = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and k[len(pre):] not in ("targets", "enc", "preds")}
        assert {k: list(v.shape) for k, v in params.items()} == m["state_dict"]
        got = attn_ref.decoder_preds(params, z[pre + "targets"], z[pre + "enc"])
        want = z[pre + "preds"]
        assert got.shape == want.shape and np.abs(got - want).max() / np.abs(want).max() < 1e-5
    with pytest.raises(ValueError):
        attn_ref.attention_ctx(np.zeros((2, 8)), np.zeros((2, 3, 16)))


def test_logmel80_oracle_bank_properties():
    """The 80-band HTK bank of oracle/features_ref.log_mel (the build's F = 80 front end; no reference call site): triangles that
    start at 0 Hz, peak in mel-equidistant order, overlap only with their neighbours; log_mel = the MFCC chain's dB spectrogram."""
    from oracle import features_ref as fr
    fb = fr.mel_filterbank(n_mels=80)
    assert fb.shape == (201, 80) and np.all(fb >= 0) and fb.max() <= 1.0
    peaks = fb.argmax(axis=0)
    assert np.all(np.diff(peaks) >= 0) and peaks[0] >= 0 and peaks[-1] < 201
    assert np.all((fb > 0).sum(axis=0) >= 1)
    # partition of unity between the first and the last centre (HTK triangles without area normalisation)
    inner = fb[peaks[0]:peaks[-1] + 1].sum(axis=1)
    assert np.abs(inner - 1.0).max() < 1e-9
    rng = np.random.default_rng(0)
    w = rng.standard_normal(3000) * 0.1
    lm = fr.log_mel(w, 80)
    assert lm.shape == (80, 16) and lm.max() - lm.min() <= 80.0 + 1e-9
    p = fr.power_spectrogram(w) @ fb
    np.testing.assert_allclose(lm.T, np.maximum(10 * np.log10(np.maximum(p, 1e-10)), (10 * np.log10(np.maximum(p, 1e-10))).max() - 80.0))


def test_step_rewards_and_reward_to_go_properties():
    """The per-step rewards of policy_grad.py:10-15 (intent) telescope to |y| - ED(y, yhat); the per-frame reward-to-go used by the
    "per_step" reward mode is their suffix sum laid on the frames where the characters start."""
    rng = np.random.default_rng(3)
    for trial in range(20):
        T, V = int(rng.integers(5, 60)), 6
        path = rng.integers(0, V, size=T)
        path[rng.random(T) < 0.4] = 0
        y = list(rng.integers(1, V, size=int(rng.integers(0, 12))))
        G, seq, rho = decode_ref.reward_to_go(path, y)
        ed = decode_ref.edit_dist(y, seq)[0]
        assert rho == [decode_ref.edit_dist(y, seq[:j - 1])[0] - decode_ref.edit_dist(y, seq[:j])[0] for j in range(1, len(seq) + 1)]
        assert G[0] == len(y) - ed and len(rho) == len(seq)
        starts = [t for t in range(T) if path[t] != 0 and (t == 0 or path[t] != path[t - 1])]
        drop = np.append(G[:-1] - G[1:], G[-1])                   # what frame t's own character earns
        want = np.zeros(T); want[starts] = rho
        np.testing.assert_array_equal(drop, want)
        r = decode_ref.step_rewards(y, seq)
        if len(seq) >= 2:
            assert r[0] == rho[0] + rho[1] and r[1:len(seq) - 1] == rho[2:] and r[-1] == 0      # slices past the end saturate
            assert sum(r) == len(y) - ed
        # frames before the first character all carry the whole reward; frames after the last start carry none
        first = next((t for t, k in enumerate(path) if k != 0), T)
        assert np.all(G[:first + 1] == G[0])
