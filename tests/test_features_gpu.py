"""N3: the GPU feature front end (data.py:44-79) against the CPU restatement of torchaudio's defaults."""
import numpy as np
import pytest
import torch

from oracle import features_ref as fr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _waves(lengths, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i, n in enumerate(lengths):
        t = torch.arange(n, dtype=torch.float64) / 16000.0
        tone = 0.3 * torch.sin(2 * np.pi * (200.0 + 150.0 * i) * t) + 0.1 * torch.sin(2 * np.pi * 3100.0 * t)
        out.append((tone + 0.02 * torch.randn(n, generator=g, dtype=torch.float64)).float())
    return out


@pytest.mark.parametrize("lengths", [[16000], [4000, 201, 12345], [32000, 31999, 16001, 800]])
def test_mfcc_deltas_match_oracle(lengths):
    from policy_gradient_asr_amd.features import MFCCDeltas
    waves = _waves(lengths, seed=len(lengths))
    feat, fmask = MFCCDeltas(DEV)(waves)
    want, wmask = fr.extract_feats([w.double().numpy() for w in waves])
    assert tuple(feat.shape) == want.shape and tuple(fmask.shape) == wmask.shape
    np.testing.assert_array_equal(fmask.cpu().numpy(), wmask)
    got = feat.cpu().numpy()
    # dB-domain quantities: absolute tolerance relative to the 80 dB range the transform keeps
    assert np.abs(got - want).max() < 2e-3, np.abs(got - want).max()
    for b, n in enumerate(lengths):
        T = 1 + n // 200
        assert np.all(got[b, :, T:] == 0)


@pytest.mark.parametrize("lengths", [[16000], [4000, 201, 12345]])
def test_logmel80_matches_oracle_and_stays_on_device(lengths):
    """N3: the 80-band log-mel front end (F = 80 of the benchmark) against the CPU restatement; outputs live on the GPU."""
    from policy_gradient_asr_amd.features import LogMel
    waves = _waves(lengths, seed=7 + len(lengths))
    feat, fmask = LogMel(80, DEV)(waves)
    assert feat.is_cuda and fmask.is_cuda
    want, wmask = fr.extract_logmel([w.double().numpy() for w in waves], 80)
    assert tuple(feat.shape) == want.shape and tuple(fmask.shape) == wmask.shape
    np.testing.assert_array_equal(fmask.cpu().numpy(), wmask)
    got = feat.cpu().numpy()
    assert np.abs(got - want).max() < 2e-3, np.abs(got - want).max()
    for b, n in enumerate(lengths):
        assert np.all(got[b, :, 1 + n // 200:] == 0)


def test_device_collate_feeds_the_trainer_without_a_host_copy():
    """collate_custom(device=...) leaves feat / fmask on the GPU and a train step consumes them as they are."""
    import functools
    from policy_gradient_asr_amd.data import collate_custom
    from policy_gradient_asr_amd.model import Seq2Seq, _to_device
    from policy_gradient_asr_amd.train_step import PolicyGradientTrainer
    waves = _waves([6000, 4500, 5200, 6100], seed=9)
    char2ind = {"<pad>": 0, "a": 1, "b": 2, "c": 3}
    batch = [{"wave": w, "trans": t, "charmap": char2ind} for w, t in zip(waves, ["ab", "c", "ba", "abc"])]
    out = functools.partial(collate_custom, device=torch.device(DEV), features="logmel80")(batch)
    assert out["feat"].is_cuda and out["fmask"].is_cuda and out["trans"].is_cuda and tuple(out["feat"].shape) == (4, 80, 31)
    x, t, fmask, tmask = _to_device(out, torch.device(DEV))
    assert x.data_ptr() == out["feat"].data_ptr()            # .to(device) of a resident tensor is the tensor itself
    torch.manual_seed(0)
    m = Seq2Seq(len(char2ind), n_feats=80).to(DEV).train()
    tr = PolicyGradientTrainer(m, lr=1e-3, lam=1.0, seed=1)
    loss = tr.step(x, t, fmask, tmask)
    assert np.isfinite(float(loss))
    cpu = collate_custom(batch, features="logmel80")           # the reference's contract: CPU tensors
    assert not cpu["feat"].is_cuda and torch.equal(cpu["feat"], out["feat"].cpu())


def test_silence_and_floor():
    """All-zero audio hits the 1e-10 floor everywhere (-100 dB), the top_db clamp is then inactive."""
    from policy_gradient_asr_amd.features import MFCCDeltas
    feat, _ = MFCCDeltas(DEV)([torch.zeros(3000)])
    want, _ = fr.extract_feats([np.zeros(3000)])
    assert np.abs(feat.cpu().numpy() - want).max() < 1e-5 * np.abs(want).max() + 1e-3      # c0 = -100*sqrt(128)


def test_collate_from_waveforms_and_wav_files(tmp_path):
    import wave
    from policy_gradient_asr_amd.data import collate_custom
    waves = _waves([6000, 4500], seed=5)
    char2ind = {"<pad>": 0, "a": 1, "b": 2}
    batch = [{"wave": waves[0], "trans": "ab", "charmap": char2ind}, {"wave": waves[1], "trans": "b", "charmap": char2ind}]
    out = collate_custom(batch)
    assert tuple(out["feat"].shape) == (2, 120, 31) and tuple(out["fmask"].shape) == (2, 1, 31)
    assert not out["feat"].is_cuda and out["trans"].tolist() == [[1, 2], [2, 0]]
    # the same audio through 16-bit WAV files ("aud" items, data.py:53)
    items = []
    for i, w in enumerate(waves):
        path = str(tmp_path / f"u{i}.wav")
        with wave.open(path, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes((w.clamp(-1, 1) * 32767).round().to(torch.int16).numpy().tobytes())
        items.append({"aud": path, "trans": "a", "charmap": char2ind})
    out2 = collate_custom(items)
    assert (out2["feat"] - out["feat"]).abs().max() < 0.5      # 16-bit quantisation noise only
    with pytest.raises(ValueError):
        collate_custom([{"wave": torch.zeros(100), "trans": "a", "charmap": char2ind}])
