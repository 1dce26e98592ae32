"""Parity of the HIP kernels (called through the C ABI via ctypes) against the CPU oracle.
Tolerances: CTC loss / gradients 1e-3 relative (north_star); integer outputs bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ctc_ref, decode_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from policy_gradient_asr_amd import hipops
    return hipops


def _ctc_case(T, B, V, L, seed, ragged=True, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(T, B, V, generator=g) * scale
    targets = torch.randint(1, V, (B, max(L, 1)), generator=g, dtype=torch.int32)
    if L >= 2:
        targets[0, 1] = targets[0, 0]
    if ragged:
        il = torch.tensor([max(T - 3 * b, min(T, 2 * L + 1)) for b in range(B)], dtype=torch.int32)
        tl = torch.tensor([max(L - b, 0) for b in range(B)], dtype=torch.int32)
    else:
        il = torch.full((B,), T, dtype=torch.int32)
        tl = torch.full((B,), L, dtype=torch.int32)
    return logits, targets, il, tl


@pytest.mark.parametrize("T,B,V,L,ragged", [
    (12, 3, 5, 4, True), (50, 4, 29, 6, True), (200, 4, 29, 20, False), (7, 2, 4, 3, True),
    (5, 1, 3, 0, False), (300, 2, 29, 140, False), (64, 3, 64, 10, True),
])
def test_ctc_loss_grad_vs_oracle(ops, dev, T, B, V, L, ragged):
    logits, targets, il, tl = _ctc_case(T, B, V, L, seed=T * 7 + V, ragged=ragged)
    lp = torch.log_softmax(logits, 2)
    nll, grad = ops.ctc_loss_grad(lp.to(dev), targets.to(dev), il.to(dev), tl.to(dev))
    nll_ref, grad_ref = ctc_ref.ctc_loss_and_grad(logits.numpy(), targets.numpy(), il.numpy(), tl.numpy())
    np.testing.assert_allclose(nll.cpu().numpy(), nll_ref, rtol=1e-4, atol=1e-5)
    g = grad.cpu().numpy()
    # 1e-3 relative to the gradient scale of each utterance (north_star tolerance)
    for b in range(B):
        den = np.abs(grad_ref[:, b]).max() + 1e-12
        assert np.abs(g[:, b] - grad_ref[:, b]).max() / den < 1e-3
    # frames past the length get exactly zero
    for b in range(B):
        assert np.all(g[int(il[b]):, b] == 0)


def test_ctc_headline_size_properties(ops, dev):
    """B=32,T=1000,V=29,L=100: oracle on 2 utterances + size-independent properties."""
    T, B, V, L = 1000, 32, 29, 100
    logits, targets, il, tl = _ctc_case(T, B, V, L, seed=1, ragged=False)
    lp = torch.log_softmax(logits, 2).to(dev)
    scale = torch.full((B,), 1.0 / (L * B), device=dev)
    nll, grad = ops.ctc_loss_grad(lp, targets.to(dev), il.to(dev), tl.to(dev), utt_scale=scale)
    nll2, grad2 = ops.ctc_loss_grad(lp, targets.to(dev), il.to(dev), tl.to(dev), utt_scale=scale)
    assert torch.equal(nll, nll2) and torch.equal(grad, grad2)  # run-to-run reproducible
    g = grad.cpu().numpy().astype(np.float64) * (L * B)
    # each frame's gradient sums to zero over the vocabulary (softmax - posterior, both sum to 1)
    assert np.abs(g.sum(axis=2)).max() < 2e-4
    sub = [0, 17]
    nll_ref, grad_ref = ctc_ref.ctc_loss_and_grad(logits[:, sub].numpy(), targets[sub].numpy(),
                                                  il[sub].numpy(), tl[sub].numpy())
    np.testing.assert_allclose(nll.cpu().numpy()[sub], nll_ref, rtol=1e-5)
    for i, b in enumerate(sub):
        den = np.abs(grad_ref[:, i]).max()
        assert np.abs(g[:, b] - grad_ref[:, i]).max() / den < 1e-3
    # agrees with torch's CPU ctc_loss (fp64) on the whole batch for the loss
    ref = torch.nn.functional.ctc_loss(torch.log_softmax(logits.double(), 2), targets.long(), il.long(), tl.long(),
                                       blank=0, reduction="none")
    np.testing.assert_allclose(nll.cpu().numpy(), ref.numpy(), rtol=1e-5)


def test_ctc_infeasible_and_errors(ops, dev):
    from policy_gradient_asr_amd import _lib
    logits, targets, il, tl = _ctc_case(4, 2, 5, 4, seed=3, ragged=False)  # T=4 < 2L+1 needs no repeats..
    targets[:] = 2  # all repeats: needs T >= 2L-1+... = 7 > 4 -> infeasible
    lp = torch.log_softmax(logits, 2).to(dev)
    nll, grad = ops.ctc_loss_grad(lp, targets.to(dev), il.to(dev), tl.to(dev))
    assert torch.isinf(nll).all() and (grad == 0).all()
    with pytest.raises(_lib.PgasrError):
        ops.ctc_loss_grad(lp, torch.zeros(2, 1100, dtype=torch.int32, device=dev), il.to(dev), tl.to(dev))


def test_ctc_fused_reinforce_term(ops, dev):
    T, B, V, L = 40, 3, 29, 5
    logits, targets, il, tl = _ctc_case(T, B, V, L, seed=9)
    lp = torch.log_softmax(logits, 2).to(dev)
    _, path = ops.frame_argmax_sample(lp, seed=5, want_greedy=False)
    coef = torch.tensor([0.3, -0.7, 1.5], device=dev)
    _, g_ctc = ops.ctc_loss_grad(lp, targets.to(dev), il.to(dev), tl.to(dev))
    _, g_all = ops.ctc_loss_grad(lp, targets.to(dev), il.to(dev), tl.to(dev), pg_coef=coef, pg_path=path)
    g_pg = ops.reinforce_grad(lp, path, coef, il.to(dev))
    torch.testing.assert_close(g_all, g_ctc + g_pg, rtol=1e-5, atol=1e-6)


def test_ctc_split_lattice_then_grad_is_the_same_call(ops, dev):
    """pgasr_ctc_loss_grad(grad=NULL) + pgasr_ctc_grad_from_lattice == the one-call form, bit for bit, with the lattice
    on another stream than the gradient pass (how loss.py runs it)."""
    T, B, V, L = 120, 5, 29, 11
    logits, targets, il, tl = _ctc_case(T, B, V, L, seed=21)
    lp = torch.log_softmax(logits, 2).to(dev)
    targets, il, tl = targets.to(dev), il.to(dev), tl.to(dev)
    _, path = ops.frame_argmax_sample(lp, seed=3, want_greedy=False)
    coef = torch.linspace(-1, 1, B, device=dev)
    us = torch.linspace(0.1, 0.5, B, device=dev)
    nll1, g1 = ops.ctc_loss_grad(lp, targets, il, tl, utt_scale=us, pg_coef=coef, pg_path=path)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        nll2, handle = ops.ctc_lattice(lp, targets, il, tl)
    torch.cuda.current_stream().wait_stream(side)
    g2 = ops.ctc_grad_from_lattice(lp, il, tl, handle, utt_scale=us, pg_coef=coef, pg_path=path)
    torch.cuda.synchronize()
    assert torch.equal(nll1, nll2) and torch.equal(g1, g2)


def test_pg_rewards_and_loss_value(ops, dev):
    """pgasr_pg_rewards / pgasr_pg_loss_value against the oracle's arithmetic (numpy fp64)."""
    T, B, V = 90, 6, 29
    g = torch.Generator().manual_seed(4)
    lp = torch.log_softmax(torch.randn(T, B, V, generator=g), 2)
    path = torch.randint(0, V, (T, B), generator=g, dtype=torch.int32)
    il = torch.tensor([90, 77, 1, 50, 90, 13], dtype=torch.int32)
    tl = torch.tensor([5, 0, 1, 9, 3, 7], dtype=torch.int32)          # a zero-length target: divisor clamps to 1
    dist = torch.randint(0, 40, (2 * B,), generator=g, dtype=torch.int32)
    nll = torch.rand(B, generator=g) * 50
    lam, inv_bg = 0.7, 1.0 / 24
    R_g, R_s, coef, us = ops.pg_rewards(dist.to(dev), tl.to(dev), lam, inv_bg)
    Lf = np.maximum(tl.numpy(), 1).astype(np.float64)
    rg = -dist[:B].numpy() / Lf; rs = -dist[B:].numpy() / Lf
    np.testing.assert_allclose(R_g.cpu().numpy(), rg, rtol=1e-6)
    np.testing.assert_allclose(R_s.cpu().numpy(), rs, rtol=1e-6)
    np.testing.assert_allclose(coef.cpu().numpy(), lam * inv_bg * (rs - rg), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(us.cpu().numpy(), inv_bg / Lf, rtol=1e-6)
    terms = ops.pg_loss_value(lp.to(dev), path.to(dev), il.to(dev), nll.to(dev), us, coef)
    lpn = lp.double().numpy()
    want = np.zeros(B)
    for b in range(B):
        s = sum(lpn[t, b, int(path[t, b])] for t in range(int(il[b])))
        want[b] = float(nll[b]) * (inv_bg / Lf[b]) - lam * inv_bg * (rs[b] - rg[b]) * s
    np.testing.assert_allclose(terms.cpu().numpy(), want, rtol=2e-5, atol=1e-6)
    again = ops.pg_loss_value(lp.to(dev), path.to(dev), il.to(dev), nll.to(dev), us, coef)
    assert torch.equal(terms, again)                                   # fixed-order reduction


@pytest.mark.parametrize("T,B,V", [(1, 1, 2), (17, 3, 29), (1000, 32, 29), (33, 5, 64)])
def test_argmax_bit_exact_and_sampler(ops, dev, T, B, V):
    g = torch.Generator().manual_seed(T + V)
    scores = torch.randn(T, B, V, generator=g) * 3
    scores[0, 0, :] = 0.5  # exact tie -> first max wins
    if V > 3:
        scores[-1, -1, 1] = scores[-1, -1, 3] = 9.0
    greedy, sample = ops.frame_argmax_sample(scores.to(dev), seed=0xDEADBEEF1234, offset=3)
    assert np.array_equal(greedy.cpu().numpy(), np.argmax(scores.numpy(), axis=2))
    paths, cdf, u = decode_ref.sample_paths(scores.numpy(), seed=0xDEADBEEF1234, offset=3)
    s = sample.cpu().numpy()
    assert s.min() >= 0 and s.max() < V
    # identical unless u sits within fp32 rounding of a CDF boundary
    diff = np.argwhere(s != paths)
    for t, b in diff:
        k = min(s[t, b], paths[t, b])
        assert abs(cdf[t, b, k] - u[t, b]) < 2e-6, (t, b, s[t, b], paths[t, b])
    assert len(diff) <= max(1, s.size // 1000)


def test_sampler_distribution(ops, dev):
    V = 8
    p = np.array([0.4, 0.05, 0.2, 0.0, 0.1, 0.15, 0.07, 0.03])
    with np.errstate(divide="ignore"):
        scores = torch.tensor(np.log(p), dtype=torch.float32).expand(20000, 4, V).contiguous()
    _, s = ops.frame_argmax_sample(scores.to(dev), seed=42, want_greedy=False)
    freq = np.bincount(s.cpu().numpy().ravel(), minlength=V) / s.numel()
    assert np.abs(freq - p).max() < 0.01 and freq[3] == 0


@pytest.mark.parametrize("T,B", [(1, 1), (9, 2), (300, 5), (1000, 32)])
def test_collapse_bit_exact(ops, dev, T, B):
    rng = np.random.default_rng(T)
    paths = rng.integers(0, 4, size=(2, T, B)).astype(np.int32)
    paths[1] = np.repeat(rng.integers(0, 3, size=((T + 3) // 4, B)), 4, axis=0)[:T]  # long runs
    lengths = rng.integers(0, T + 1, size=B).astype(np.int32)
    lengths[0] = T
    tok, tl = ops.ctc_collapse(torch.from_numpy(paths).to(dev), torch.from_numpy(lengths).to(dev))
    tok, tl = tok.cpu().numpy(), tl.cpu().numpy()
    for p in range(2):
        for b in range(B):
            want = decode_ref.collapse_path(paths[p, :lengths[b], b])
            assert tl[p, b] == len(want)
            assert list(tok[p, b, :tl[p, b]]) == want


def test_greedy_decode_matches_oracle(ops, dev):
    T, B, V = 120, 6, 29
    g = torch.Generator().manual_seed(4)
    scores = torch.randn(T, B, V, generator=g)
    scores[:, :, 0] += 1.5
    lengths = torch.tensor([120, 100, 1, 0, 77, 120], dtype=torch.int32)
    greedy, _ = ops.frame_argmax_sample(scores.to(dev), want_sample=False)
    tok, tl = ops.ctc_collapse(greedy[None].contiguous(), lengths.to(dev))
    want = decode_ref.greedy_decode(scores.numpy(), lengths.numpy())
    for b in range(B):
        assert list(tok[0, b, :tl[0, b]].cpu().numpy()) == want[b]


def test_edit_distance_reference_table(ops, dev, golden_dir):
    vec = json.load(open(os.path.join(golden_dir, "reference_vectors.json")))
    pairs = [(a, b, w[0]) for a, b, w in vec["text"]["edit_dist"]]
    R = max(max(len(a) for a, _, _ in pairs), 1); Hy = max(max(len(b) for _, b, _ in pairs), 1)
    ref = np.zeros((len(pairs), R), np.int32); hyp = np.zeros((len(pairs), Hy), np.int32)
    rl = np.zeros(len(pairs), np.int32); hl = np.zeros(len(pairs), np.int32)
    for i, (a, b, _) in enumerate(pairs):
        ref[i, :len(a)] = [ord(c) for c in a]; hyp[i, :len(b)] = [ord(c) for c in b]
        rl[i], hl[i] = len(a), len(b)
    d = ops.edit_distance(*(torch.from_numpy(x).to(dev) for x in (ref, rl, hyp, hl)))
    assert list(d.cpu().numpy()) == [w for _, _, w in pairs]


@pytest.mark.parametrize("N,R,Hy,alpha", [(7, 1, 1, 2), (33, 70, 90, 4), (64, 100, 300, 28), (5, 500, 40, 3), (3, 1200, 700, 5),
                                          (300, 40, 60, 6)])     # more than 128 pairs: the bulk path, many waves per CU (no LDS reservation)
def test_edit_distance_random_and_prefix(ops, dev, N, R, Hy, alpha):
    rng = np.random.default_rng(N * R)
    ref = rng.integers(1, alpha + 1, size=(N, R)).astype(np.int32)
    hyp = rng.integers(1, alpha + 1, size=(N, Hy)).astype(np.int32)
    rl = rng.integers(0, R + 1, size=N).astype(np.int32); hl = rng.integers(0, Hy + 1, size=N).astype(np.int32)
    rl[0], hl[0] = R, Hy
    if N > 2:
        rl[1], hl[2] = 0, 0
    d, pre = ops.edit_distance(*(torch.from_numpy(x).to(dev) for x in (ref, rl, hyp, hl)), want_prefix=True)
    d, pre = d.cpu().numpy(), pre.cpu().numpy()
    check = range(N) if R * Hy < 50000 else [0, 1, 2]
    for n in check:
        a, b = list(ref[n, :rl[n]]), list(hyp[n, :hl[n]])
        assert d[n] == decode_ref.edit_dist(a, b)[0]
        for i in sorted(set([0, 1, hl[n] // 2, hl[n]])):
            if i <= hl[n]:
                assert pre[n, i] == decode_ref.edit_dist(a, b[:i])[0]
    # properties at any size: |len diff| <= d <= max(len), prefix[0] = len(ref)
    assert np.all(d >= np.abs(rl - hl)) and np.all(d <= np.maximum(rl, hl))
    assert np.all(pre[:, 0] == rl)


def test_reinforce_grad_vs_oracle(ops, dev):
    T, B, V = 60, 4, 29
    rng = np.random.default_rng(2)
    scores = (rng.normal(size=(T, B, V)) * 2).astype(np.float32)
    path = rng.integers(0, V, size=(T, B)).astype(np.int32)
    coef = np.array([0.5, -1.25, 0.0, 3.0], np.float32); lens = np.array([60, 10, 33, 0], np.int32)
    g = ops.reinforce_grad(*(torch.from_numpy(x).to(dev) for x in (scores, path, coef, lens)))
    want = decode_ref.reinforce_grad(scores, path, coef, lens)
    np.testing.assert_allclose(g.cpu().numpy(), want, rtol=1e-4, atol=1e-6)
    acc = torch.ones(T, B, V, device=dev)
    ops.reinforce_grad(*(torch.from_numpy(x).to(dev) for x in (scores, path, coef, lens)), out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), want + 1.0, rtol=1e-4, atol=1e-6)


def test_batch_prep_matches_torch(ops, dev):
    """fmask / tmask / targets of the collate_custom batch -> lengths and int32 targets in one launch."""
    g = torch.Generator().manual_seed(5)
    B, T, L = 7, 333, 41
    lens = torch.randint(0, T + 1, (B,), generator=g); tl = torch.randint(0, L + 1, (B,), generator=g)
    fmask = (torch.arange(T)[None, :] < lens[:, None]).float()
    tmask = (torch.arange(L)[None, :] < tl[:, None]).long()
    targets = torch.randint(1, 29, (B, L), generator=g) * tmask
    in_len, tg_len, tg32 = ops.batch_prep(fmask.to(dev), tmask.to(dev), targets.to(dev))
    assert in_len.dtype == tg_len.dtype == tg32.dtype == torch.int32
    assert torch.equal(in_len.cpu(), lens.int()) and torch.equal(tg_len.cpu(), tl.int()) and torch.equal(tg32.cpu(), targets.int())


def test_sampler_draws_are_addressed_globally_across_shards():
    """Data parallel (model.py:201): a rank samples utterances [base, base + B_local) of a global batch of `stride` with
    the SAME seed and gets exactly the paths a single process holding the whole batch samples for them."""
    from policy_gradient_asr_amd import hipops as ops
    T, Bg, V = 37, 12, 29
    g = torch.Generator().manual_seed(3)
    scores = torch.randn(T, Bg, V, generator=g).to("cuda:0")
    _, whole = ops.frame_argmax_sample(scores, seed=99, offset=7, want_greedy=False)
    for base, nb in ((0, 4), (4, 4), (8, 4), (0, 12), (5, 7)):
        shard = scores[:, base:base + nb].contiguous()
        _, part = ops.frame_argmax_sample(shard, seed=99, offset=7, want_greedy=False, batch_stride=Bg, batch_offset=base)
        assert torch.equal(part, whole[:, base:base + nb])
    _, local = ops.frame_argmax_sample(scores[:, 4:8].contiguous(), seed=99, offset=7, want_greedy=False)
    assert not torch.equal(local, whole[:, 4:8])          # without the layout a shard draws its own (local) counters
    # utterances beyond the global batch (a ragged batch padded with empty utterances, round 5): the real ones keep their draws, the
    # padding draws from a disjoint domain
    padded = torch.cat((scores[:, :5], scores[:, :3]), dim=1).contiguous()
    _, part = ops.frame_argmax_sample(padded, seed=99, offset=7, want_greedy=False, batch_stride=5, batch_offset=0)
    _, five = ops.frame_argmax_sample(scores[:, :5].contiguous(), seed=99, offset=7, want_greedy=False)
    assert torch.equal(part[:, :5], five) and not torch.equal(part[:, 5:], five[:, :3])
    import pytest
    with pytest.raises(Exception):
        ops.frame_argmax_sample(scores[:, :4].contiguous(), seed=1, want_greedy=False, batch_stride=6, batch_offset=6)   # base outside the batch
