"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- CPU restatement of the reference's feature front end.

data.py:44-62 computes, per utterance,
    mfcc   = torchaudio.transforms.MFCC()(waveform)               # (1, 40, T)
    deltas = torchaudio.transforms.ComputeDeltas()(mfcc)
    ddelta = torchaudio.transforms.ComputeDeltas()(deltas)
    feat   = cat((mfcc, deltas, ddelta), dim=1).squeeze(0)          # (120, T)
The arithmetic lives in the third-party dependency torchaudio (requirements.txt, unpinned; ABSENT from this
image), so this file restates torchaudio's published defaults:
    MFCC(sample_rate=16000, n_mfcc=40, dct_type=2, norm="ortho", log_mels=False) over
    MelSpectrogram(n_fft=400, win_length=400, hop_length=200, f_min=0, f_max=sr/2, n_mels=128, window=hann
    (periodic), power=2, center=True, pad_mode="reflect", norm=None, mel_scale="htk"),
    AmplitudeToDB("power", top_db=80) = 10*log10(max(x, 1e-10)) floored at (per-waveform max - 80),
    create_dct(40, 128, "ortho"), ComputeDeltas(win_length=5, mode="replicate").
PARITY UNPINNED by the reference (torchaudio cannot be imported here); the stages that have an independent
implementation in this image are pinned against it in tests/test_oracle_cpu.py: the STFT against torch.stft, the
DCT against scipy.fft.dct, the delta filter against a direct convolution."""
import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 200
N_MELS = 128
N_MFCC = 40
TOP_DB = 80.0
AMIN = 1e-10


def hann_periodic(n=N_FFT):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def mel_filterbank(n_freqs=N_FFT // 2 + 1, f_min=0.0, f_max=SAMPLE_RATE / 2.0, n_mels=N_MELS, sample_rate=SAMPLE_RATE):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") -> (n_freqs, n_mels)."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * np.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def dct_matrix(n_mfcc=N_MFCC, n_mels=N_MELS):
    """torchaudio.functional.create_dct(n_mfcc, n_mels, norm="ortho") -> (n_mels, n_mfcc)."""
    n = np.arange(n_mels, dtype=np.float64)
    k = np.arange(n_mfcc, dtype=np.float64)[:, None]
    dct = np.cos(np.pi / n_mels * (n + 0.5) * k)
    dct[0] *= 1.0 / np.sqrt(2.0)
    dct *= np.sqrt(2.0 / n_mels)
    return dct.T


def n_frames(n_samples):
    return 1 + n_samples // HOP          # center=True


def power_spectrogram(wave):
    """|STFT|^2, (T, 201): frame t covers reflect-padded samples [t*HOP - 200, t*HOP + 200)."""
    wave = np.asarray(wave, dtype=np.float64)
    pad = N_FFT // 2
    if wave.shape[0] <= pad:
        raise ValueError("reflect padding needs more than n_fft/2 samples")
    x = np.pad(wave, (pad, pad), mode="reflect")
    T = n_frames(wave.shape[0])
    idx = np.arange(T)[:, None] * HOP + np.arange(N_FFT)[None, :]
    frames = x[idx] * hann_periodic()[None, :]
    spec = np.fft.rfft(frames, n=N_FFT, axis=1)
    return spec.real ** 2 + spec.imag ** 2


def compute_deltas(x):
    """(C, T) -> (C, T): sum_{m=-2..2} m * x[clamp(t+m)] / 10 (replicate padding)."""
    T = x.shape[1]
    out = np.zeros_like(x)
    for m in range(-2, 3):
        idx = np.clip(np.arange(T) + m, 0, T - 1)
        out += m * x[:, idx]
    return out / 10.0


def mfcc_deltas(wave):
    """1-D waveform -> (120, T) float64: MFCC(40), delta, delta-delta stacked as data.py:54-59."""
    p = power_spectrogram(wave)                                   # (T, 201)
    mel = p @ mel_filterbank()                                    # (T, 128)
    db = 10.0 * np.log10(np.maximum(mel, AMIN))
    db = np.maximum(db, db.max() - TOP_DB)
    mfcc = (db @ dct_matrix()).T                                  # (40, T)
    d1 = compute_deltas(mfcc)
    d2 = compute_deltas(d1)
    return np.concatenate((mfcc, d1, d2), axis=0)


def log_mel(wave, n_mels=80):
    """1-D waveform -> (n_mels, T) float64: the dB-scaled mel spectrogram the MFCC chain computes in front of its DCT, with an
    n_mels-band HTK bank (the build's 80-band F = 80 front end, features.LogMel; not a reference call site)."""
    p = power_spectrogram(wave)
    mel = p @ mel_filterbank(n_mels=n_mels)
    db = 10.0 * np.log10(np.maximum(mel, AMIN))
    return np.maximum(db, db.max() - TOP_DB).T


def extract_logmel(waves, n_mels=80):
    feats = [log_mel(w, n_mels) for w in waves]
    tmax = max(f.shape[1] for f in feats)
    out = np.zeros((len(feats), n_mels, tmax))
    mask = np.zeros((len(feats), 1, tmax))
    for i, f in enumerate(feats):
        out[i, :, :f.shape[1]] = f
        mask[i, 0, :f.shape[1]] = 1.0
    return out, mask


def extract_feats(waves):
    """list of waveforms -> (B,120,Tmax) zero padded, (B,1,Tmax) masks (data.py:64-79)."""
    feats = [mfcc_deltas(w) for w in waves]
    tmax = max(f.shape[1] for f in feats)
    out = np.zeros((len(feats), feats[0].shape[0], tmax))
    mask = np.zeros((len(feats), 1, tmax))
    for i, f in enumerate(feats):
        out[i, :, :f.shape[1]] = f
        mask[i, 0, :f.shape[1]] = 1.0
    return out, mask
