"""Feature front end of the reference on the GPU (data.py:44-79).

``LogMel(80)(waves)`` is the same front end stopped in front of the DCT with an 80-band HTK mel bank: (B,80,Tmax) dB-scaled
log-mel features -- the "mel-spectrogram" input (F = 80) the benchmark's shapes are quoted on; feat / fmask stay on the device.
``MFCCDeltas()(waves)`` returns what ``extract_feats`` returns there: ``feat`` (B,120,Tmax) fp32 zero padded and
``fmask`` (B,1,Tmax) -- MFCC(40) + delta + delta-delta with torchaudio's default parameters, which the reference
uses unchanged (``torchaudio.transforms.MFCC()``, ``ComputeDeltas()``).  torchaudio itself is not needed: framing,
power, the dB map and the delta filters are HIP kernels (csrc/features.hip), the DFT / mel / DCT contractions run
on the matrix cores through ``pgasr_gemm_f32`` in exact-fp32 mode.  No CPU path."""
import math

import torch

from . import _lib, hipops

SAMPLE_RATE, N_FFT, HOP, N_MELS, N_MFCC, TOP_DB = 16000, 400, 200, 128, 40, 80.0
N_BINS = N_FFT // 2 + 1


def _constants(device, n_mels=N_MELS):
    """DFT basis (400, 402) = [cos | -sin], HTK mel bank (201, n_mels), orthonormal DCT-II (n_mels, 40); built in fp64."""
    N_MELS = n_mels
    n = torch.arange(N_FFT, dtype=torch.float64)[:, None]
    k = torch.arange(N_BINS, dtype=torch.float64)[None, :]
    ang = 2.0 * math.pi * n * k / N_FFT
    dft = torch.cat((torch.cos(ang), -torch.sin(ang)), dim=1)
    all_freqs = torch.linspace(0, SAMPLE_RATE // 2, N_BINS, dtype=torch.float64)
    m_max = 2595.0 * math.log10(1.0 + (SAMPLE_RATE / 2.0) / 700.0)
    m_pts = torch.linspace(0.0, m_max, N_MELS + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    fb = torch.clamp(torch.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)
    nn_ = torch.arange(N_MELS, dtype=torch.float64)[None, :]
    kk = torch.arange(N_MFCC, dtype=torch.float64)[:, None]
    dct = torch.cos(math.pi / N_MELS * (nn_ + 0.5) * kk)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / N_MELS)
    return tuple(t.to(torch.float32).contiguous().to(device) for t in (dft, fb, dct.t()))


class _FrontEnd:
    """Shared part: waveforms -> dB-scaled mel spectrogram rows (B * Tmax, n_mels) on the device."""

    def __init__(self, device="cuda:0", n_mels=N_MELS):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PgasrError("the feature front end runs on the MI355X only; there is no CPU path")
        self.n_mels = int(n_mels)
        self.dft, self.fb, self.dct = _constants(self.device, self.n_mels)

    def mel_db(self, waves):
        lib = _lib.load()
        dev = self.device
        B = len(waves)
        ns = [int(w.numel()) for w in waves]
        if min(ns) <= N_FFT // 2:
            raise ValueError("reflect-centred framing needs more than n_fft/2 = 200 samples per utterance")
        nmax = (max(ns) + 3) // 4 * 4
        wave = torch.zeros(B, nmax, dtype=torch.float32, device=dev)
        for i, w in enumerate(waves):
            wave[i, :ns[i]] = w.reshape(-1).to(device=dev, dtype=torch.float32)
        n_samples = torch.tensor(ns, dtype=torch.int32, device=dev)
        nf = [1 + n // HOP for n in ns]
        n_frames = torch.tensor(nf, dtype=torch.int32, device=dev)
        Tmax = max(nf)
        rows = B * Tmax
        st = torch.cuda.current_stream().cuda_stream
        frames = torch.empty(rows, N_FFT, dtype=torch.float32, device=dev)
        _lib.check(lib.pgasr_feat_frames(wave.data_ptr(), n_samples.data_ptr(), n_frames.data_ptr(), B, nmax, Tmax,
                                         frames.data_ptr(), st), "pgasr_feat_frames")
        spec = torch.empty(rows, 2 * N_BINS, dtype=torch.float32, device=dev)
        hipops.gemm(frames, self.dft, spec, M=rows, N=2 * N_BINS, K=N_FFT, precision=0)
        power = torch.empty(rows, N_BINS, dtype=torch.float32, device=dev)
        _lib.check(lib.pgasr_feat_power(spec.data_ptr(), rows, power.data_ptr(), st), "pgasr_feat_power")
        mel = torch.empty(rows, self.n_mels, dtype=torch.float32, device=dev)
        hipops.gemm(power, self.fb, mel, M=rows, N=self.n_mels, K=N_BINS, precision=0)
        _lib.check(lib.pgasr_feat_db(mel.data_ptr(), n_frames.data_ptr(), B, Tmax, self.n_mels, TOP_DB, st), "pgasr_feat_db")
        return mel, n_frames, B, Tmax


class MFCCDeltas(_FrontEnd):
    """waves: list of 1-D float tensors (any device; moved to ``device``) -> (feat (B,120,Tmax), fmask (B,1,Tmax))."""

    def __init__(self, device="cuda:0"):
        super().__init__(device, N_MELS)

    def __call__(self, waves):
        lib = _lib.load()
        dev = self.device
        mel, n_frames, B, Tmax = self.mel_db(waves)
        rows = B * Tmax
        st = torch.cuda.current_stream().cuda_stream
        mfcc = torch.empty(rows, N_MFCC, dtype=torch.float32, device=dev)
        hipops.gemm(mel, self.dct, mfcc, M=rows, N=N_MFCC, K=N_MELS, precision=0)
        feat = torch.empty(B, 3 * N_MFCC, Tmax, dtype=torch.float32, device=dev)
        fmask = torch.empty(B, 1, Tmax, dtype=torch.float32, device=dev)
        _lib.check(lib.pgasr_feat_deltas_stack(mfcc.data_ptr(), n_frames.data_ptr(), B, Tmax, N_MFCC, feat.data_ptr(),
                                               fmask.data_ptr(), st), "pgasr_feat_deltas_stack")
        return feat, fmask


class LogMel(_FrontEnd):
    """waves -> (feat (B,n_mels,Tmax), fmask (B,1,Tmax)) on the device: the dB-scaled mel spectrogram the MFCC chain computes in
    front of its DCT (torchaudio's MelSpectrogram + AmplitudeToDB("power", top_db=80) defaults), with an ``n_mels``-band HTK bank --
    80 bands give the F = 80 input of the benchmark (``Encoder(n_feats=80)``), feeding the instance norm / affine (A1-A2) directly."""

    def __init__(self, n_mels=80, device="cuda:0"):
        super().__init__(device, n_mels)

    def __call__(self, waves):
        lib = _lib.load()
        mel, n_frames, B, Tmax = self.mel_db(waves)
        feat = torch.empty(B, self.n_mels, Tmax, dtype=torch.float32, device=self.device)
        fmask = torch.empty(B, 1, Tmax, dtype=torch.float32, device=self.device)
        _lib.check(lib.pgasr_feat_stack(mel.data_ptr(), n_frames.data_ptr(), B, Tmax, self.n_mels, feat.data_ptr(), fmask.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "pgasr_feat_stack")
        return feat, fmask


def read_wav(path):
    """PCM-16 / PCM-32 / float32 RIFF WAV -> (1-D float32 tensor in [-1,1), sample rate): the stand-in for
    ``torchaudio.load`` (data.py:53) where torchaudio is absent; first channel only, like ``.squeeze(0)`` on mono."""
    import wave as _wave
    import numpy as np
    with _wave.open(path, "rb") as f:
        sr, nch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"unsupported sample width {width}")
    return torch.from_numpy(x.reshape(-1, nch)[:, 0].copy()), sr
