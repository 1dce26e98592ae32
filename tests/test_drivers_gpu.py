"""Driver shells: train() with checkpoints + resume, predict() with CER/WER."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_resume_predict(tmp_path):
    from policy_gradient_asr_amd.data import SyntheticSpeech, collate_custom
    from policy_gradient_asr_amd.model import train, predict
    corpus = tmp_path / "corpus"; out = tmp_path / "run"
    corpus.mkdir()
    (corpus / "alphabet.txt").write_text("a\nb\nc\nd\n \n")
    char2ind = {"<pad>": 0, "a": 1, "b": 2, "c": 3, "d": 4, " ": 5}
    ds = SyntheticSpeech(48, char2ind, n_feats=20, seed=1)
    dv = SyntheticSpeech(16, char2ind, n_feats=20, seed=2)
    b = collate_custom([ds[0], ds[1], ds[2]])
    assert b["feat"].shape[0] == 3 and b["fmask"].shape[1] == 1 and b["trans"].dtype == torch.int64
    assert (b["tmask"] == (b["trans"] > 0)).all()
    l1, v1 = train(str(corpus), str(out), 6, 16, 0, train_dataset=ds, dev_dataset=dv, n_feats=20, lam=0.0, lr=3e-3,
                   log_every=0)
    for f in ("train_loss.npy", "val_losses.npy", "model_best.pth", "model_last.pth", "checkpoint_last.pth"):
        assert os.path.exists(out / f), f
    assert len(np.load(out / "train_loss.npy")) == 6 and l1[-1] < l1[0]
    # resume continues from epoch 7 with the saved optimizer state
    l2, v2 = train(str(corpus), str(out), 8, 16, 0, train_dataset=ds, dev_dataset=dv, n_feats=20, lam=0.0, lr=3e-3,
                   log_every=0)
    assert len(l2) == 8 and l2[:6] == pytest.approx(l1)
    cer, wer = predict(None, None, str(corpus / "alphabet.txt"), str(out), 8, test_dataset=dv, n_feats=20)
    lines = open(out / "predicted.txt").read().splitlines()
    assert len(lines) == 16 and all("|" in ln for ln in lines)
    assert 0.0 <= cer and np.isfinite(wer)
