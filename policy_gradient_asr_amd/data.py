"""Batch format of the reference's data.py (row A0 of SURVEY §8a) -- host-side only.

``collate_custom`` returns the reference's batch dict {"feat" (B,F,Tmax) fp32 zero padded,
"fmask" (B,1,Tmax), "trans" (B,Lmax) int64 pad 0, "tmask" (B,Lmax)} (data.py:107-116).  Feature
extraction (``extract_feats``, data.py:44-79: MFCC(40)+delta+delta-delta, torchaudio's defaults) runs on the GPU
(features.py); torchaudio is only used to decode audio files where it exists (RIFF WAV is read
without it), and ``SyntheticSpeech`` supplies ready-made features for the benchmarks."""
import os

import torch
import torch.nn as nn
import torch.utils.data as data


def pad_feats(feats):
    """list of (F,T_i) tensors -> (B,F,Tmax) zero padded, (B,1,Tmax) masks (data.py:64-79)."""
    tmax = max(f.shape[1] for f in feats)
    padded, masks = [], []
    for f in feats:
        mask = torch.ones(1, f.shape[1])
        padded.append(nn.functional.pad(f, (0, tmax - f.shape[1], 0, 0)))
        masks.append(nn.functional.pad(mask, (0, tmax - f.shape[1], 0, 0)))
    return torch.stack(padded), torch.stack(masks)


_front_end = {}


def extract_feats(batch, device="cuda:0", features="mfcc", keep_on_device=False):
    """data.py:44-79: MFCC(40) + delta + delta-delta per utterance, zero padded to (B,120,Tmax) + (B,1,Tmax) masks.
    An item carries precomputed features ("feat", (F,T)), a waveform tensor ("wave"), or a path ("aud", read with
    torchaudio where it exists, else as RIFF WAV).  Waveforms go through the GPU front end: ``features`` = "mfcc"
    (features.MFCCDeltas, the reference's 120 features) or "logmel80" (features.LogMel(80): the benchmark's F = 80).
    keep_on_device: return the tensors where the front end left them (no host round trip on the way to the trainer);
    default: on the CPU like the reference's collate output."""
    if all("feat" in inst for inst in batch):
        feat, fmask = pad_feats([inst["feat"] for inst in batch])
        return (feat.to(device), fmask.to(device)) if keep_on_device else (feat, fmask)
    if any("feat" in inst for inst in batch):
        raise ValueError("a batch mixes precomputed features and waveforms")
    from .features import LogMel, MFCCDeltas, read_wav
    waves = []
    for inst in batch:
        if "wave" in inst:
            waves.append(inst["wave"])
            continue
        try:
            import torchaudio
            waveform, _sr = torchaudio.load(inst["aud"])          # data.py:53
            waves.append(waveform[0])
        except ImportError:
            waves.append(read_wav(inst["aud"])[0])
    if features not in ("mfcc", "logmel80"):
        raise ValueError('features must be "mfcc" or "logmel80"')
    fe = _front_end.get((str(device), features))
    if fe is None:
        fe = _front_end[(str(device), features)] = MFCCDeltas(device) if features == "mfcc" else LogMel(80, device)
    feat, fmask = fe(waves)
    return (feat, fmask) if keep_on_device else (feat.cpu(), fmask.cpu())


def encode_trans(batch):
    """char -> index with pad 0, mask = (id > 0) (data.py:82-104)."""
    char2ind = batch[0]["charmap"]
    enc = [torch.tensor([char2ind[c] for c in inst["trans"]], dtype=torch.int64) for inst in batch]
    lmax = max(max((e.shape[0] for e in enc), default=1), 1)
    out = torch.stack([nn.functional.pad(e, (0, lmax - e.shape[0])) for e in enc])
    return out, (out > 0).to(torch.int64)


def collate_custom(batch, device=None, features="mfcc"):
    """data.py:107-116.  device = None: the reference's contract -- everything on the CPU.  device = a GPU: "feat" / "fmask" stay
    where the front end computed them and "trans" / "tmask" are put beside them, so ``PolicyGradientTrainer.step`` consumes the
    batch without the features ever visiting the host (use ``functools.partial(collate_custom, device=...)`` as collate_fn)."""
    if device is None:
        feats, fmasks = extract_feats(batch, features=features)
        trans, tmasks = encode_trans(batch)
    else:
        feats, fmasks = extract_feats(batch, device=device, features=features, keep_on_device=True)
        trans, tmasks = (t.to(device) for t in encode_trans(batch))
    return {"feat": feats, "fmask": fmasks, "trans": trans, "tmask": tmasks}


class Data(data.Dataset):
    """Data(csv_path, aud_path, char2ind) (data.py:118-132): CommonVoice TSV rows."""

    def __init__(self, csv_path, aud_path, char2ind):
        import pandas as pd
        self.df = pd.read_csv(csv_path, sep="\t")
        self.char2ind = char2ind
        self.fnames = [os.path.join(aud_path, f) for f in self.df["path"]]
        self.transcrpts = self.df["sentence"]

    def __len__(self):
        return len(self.df)

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        return {"aud": self.fnames[idx], "trans": self.transcrpts[idx], "charmap": self.char2ind}


class SyntheticSpeech(data.Dataset):
    """Seeded stand-in corpus: each item carries a precomputed (F,T_i) feature matrix whose frames
    are noisy one-hot-ish codes of its transcript (learnable by CTC), so the driver shells can be
    exercised without audio files."""

    def __init__(self, n_items, char2ind, n_feats=80, max_chars=8, frames_per_char=6, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.char2ind = char2ind
        chars = [c for c, i in char2ind.items() if i > 0]
        proto = torch.randn(len(char2ind), n_feats, generator=g)
        self.items = []
        for _ in range(n_items):
            n = int(torch.randint(2, max_chars + 1, (1,), generator=g))
            text = "".join(chars[int(torch.randint(0, len(chars), (1,), generator=g))] for _ in range(n))
            frames = []
            for ch in text:
                k = int(torch.randint(frames_per_char - 2, frames_per_char + 3, (1,), generator=g))
                frames.append(proto[char2ind[ch]].unsqueeze(1).repeat(1, k))
            feat = torch.cat(frames, dim=1)
            feat = feat + 0.3 * torch.randn(feat.shape, generator=g)
            self.items.append({"feat": feat, "trans": text, "charmap": char2ind})

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        return self.items[idx]


class LengthBucketSampler(data.Sampler):
    """Batch sampler for variable-length utterances (replaces ``shuffle=True`` of model.py:221):
    utterances are sorted by length, cut into buckets of ``bucket_batches`` batches, shuffled inside each bucket and
    the resulting batches are shuffled -- every batch holds utterances of similar length, so little of the
    (B, F, Tmax) tensor is padding and the persistent LSTM sweeps (whose time is Tmax steps whatever the lengths)
    do little idle work.  Deterministic per (seed, epoch); ``set_epoch`` like a DistributedSampler."""

    def __init__(self, lengths, batch_size, bucket_batches=16, seed=0, drop_last=False):
        self.lengths = [int(n) for n in lengths]
        self.batch_size, self.bucket_batches, self.seed, self.drop_last = int(batch_size), int(bucket_batches), int(seed), drop_last
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        n = len(self.lengths)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed * 100003 + self.epoch)
        order = sorted(range(len(self.lengths)), key=lambda i: self.lengths[i])
        span = self.batch_size * self.bucket_batches
        batches = []
        for s in range(0, len(order), span):
            bucket = order[s:s + span]
            perm = torch.randperm(len(bucket), generator=g).tolist()
            bucket = [bucket[i] for i in perm]
            for b in range(0, len(bucket), self.batch_size):
                batch = bucket[b:b + self.batch_size]
                if len(batch) == self.batch_size or not self.drop_last:
                    batches.append(batch)
        for i in torch.randperm(len(batches), generator=g).tolist():
            yield batches[i]


def dataset_lengths(dataset):
    """Frame counts of a dataset whose items carry precomputed features or waveforms (None if unknown, e.g. paths)."""
    out = []
    for i in range(len(dataset)):
        item = dataset[i]
        if "feat" in item:
            out.append(int(item["feat"].shape[1]))
        elif "wave" in item:
            out.append(1 + int(item["wave"].numel()) // 200)
        else:
            return None
    return out
