"""GPU: greedy decode + CER, minibatch assembly, running statistics, CMN and white noise through the C ABI,
against oracle/ and the reference's goldens."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fft as offt
from oracle import text as otext

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _text():
    with open(os.path.join(GOLD, "text.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.mark.parametrize("T,B,V", [(37, 5, 29), (300, 3, 3000), (1, 2, 7)])
def test_argmax_and_collapse(device, T, B, V):
    from asr import _ops
    rs = np.random.RandomState(T + V)
    x = rs.randn(T, B, V).astype(np.float32)
    x[rs.rand(T, B) < 0.5, 0] += 6.0                 # plenty of blanks
    if T > 2:
        x[1] = x[0]                                  # forced repeats
    x[0, 0, :] = 1.0                                 # a full tie: np.argmax takes the first index
    xd = torch.from_numpy(x).to(device)
    ids = _ops.argmax_rows(xd)
    want = np.argmax(x, axis=2).T                    # (B, T)
    np.testing.assert_array_equal(ids.cpu().numpy(), want)
    lens = torch.tensor(rs.randint(1, T + 1, size=B), dtype=torch.int32, device=device)
    for L in (None, lens):
        out, n = _ops.ctc_collapse(ids, L, 0, True)
        out, n = out.cpu().numpy(), n.cpu().numpy()
        for b in range(B):
            frames = want[b] if L is None else want[b, :int(lens[b])]
            ref = otext.collapse_greedy(frames, 0)
            assert n[b] == len(ref) and list(out[b, :n[b]]) == ref and (out[b, n[b]:] == 0).all()
    out, n = _ops.ctc_collapse(ids, None, 0, False)          # blanks only
    for b in range(B):
        ref = [int(i) for i in want[b] if i != 0]
        assert list(out[b, :int(n[b])].cpu().numpy()) == ref


def test_greedy_decode_wrapper(device):
    from asr.error import greedy_decode
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.randn(50, 4, 11).astype(np.float32)).to(device)
    ids, n = greedy_decode(x, blank=0)
    want = np.argmax(x.cpu().numpy(), axis=2).T
    for b in range(4):
        assert list(ids[b, :int(n[b])].cpu().numpy()) == otext.collapse_greedy(want[b], 0)


def test_edit_distance(device):
    from asr import _ops
    rs = np.random.RandomState(0)
    pairs = []
    for n, m in [(0, 0), (0, 5), (5, 0), (1, 1), (7, 3), (40, 55), (300, 280), (120, 120)]:
        r = rs.randint(1, 6, size=n).tolist()
        h = (r[: m] + rs.randint(1, 6, size=max(0, m - n)).tolist()) if rs.rand() < 0.5 else rs.randint(1, 6, size=m).tolist()
        pairs.append((r, h[:m]))
    Lr, Lh = max(1, max(len(r) for r, _ in pairs)), max(1, max(len(h) for _, h in pairs))
    R = np.zeros((len(pairs), Lr), np.int32)
    H = np.zeros((len(pairs), Lh), np.int32)
    for i, (r, h) in enumerate(pairs):
        R[i, :len(r)] = r
        H[i, :len(h)] = h
    d = _ops.edit_distance(torch.from_numpy(R).to(device), torch.tensor([len(r) for r, _ in pairs], dtype=torch.int32, device=device),
                           torch.from_numpy(H).to(device), torch.tensor([len(h) for _, h in pairs], dtype=torch.int32, device=device))
    d = d.cpu().numpy()
    for i, (r, h) in enumerate(pairs):
        want = len(h) if len(r) == 0 else otext.levenshtein(r, h)
        assert d[i] == want, (i, len(r), len(h))


def test_minibatch_error_matches_reference(device):
    from asr import error, vocab
    g = _text()
    ids, inv = vocab.get_unigram_ids()
    y, t = np.asarray(g["y"]), np.asarray(g["t"])
    assert abs(error.compute_minibatch_error(y, t, 0, ids, inv) - g["cer_mean"]) < 1e-12
    yd, td = torch.from_numpy(y).to(device), torch.from_numpy(t).to(device)
    for b in range(len(y)):
        assert abs(error.compute_minibatch_error(yd[b:b + 1], td[b:b + 1], 0, ids, inv) - g["cer_each"][b]) < 1e-12
    for r, h, want in zip(g["lev_r"], g["lev_h"], g["lev"]):
        assert error.compute_character_error_rate(r, h) == want


def test_minibatch_error_with_bigram_inventory(device):
    """a Gram-CTC inventory holds tokens spelled by two unigrams: predictions are re-tokenised (asr/error.py:49-53)"""
    from asr import error, vocab
    ids, inv = vocab.get_unigram_ids()
    ids = dict(ids)
    big = "キョ" + "ウ"
    ids[big] = len(ids)
    inv = {v: k for k, v in ids.items()}
    y = np.array([[0, ids[big], ids[big], 0, ids["ア"], 0]])
    t = np.array([[ids["キョ"], ids["ウ"], ids["ア"], 0]])
    assert error.compute_minibatch_error(y, t, 0, ids, inv) == 0.0


def test_features_to_minibatch_against_reference(device):
    from asr.data.processing import Processor
    mb = np.load(os.path.join(GOLD, "minibatch.npz"))
    with open(os.path.join(GOLD, "minibatch.json"), encoding="utf-8") as f:
        meta = json.load(f)
    feats = [tuple(mb["feat%d_%d" % (i, c)] for c in range(3)) for i in range(len(meta["sentences"]))]
    proc = Processor(device=device)
    x, xl, t, tl, bg = proc.features_to_minibatch(feats, meta["sentences"], int(mb["lens"].max()), int(mb["max_sentence_length"]),
                                                  meta["token_ids"], 0)
    np.testing.assert_array_equal(x.cpu().numpy(), mb["x"])
    assert list(xl) == list(mb["x_length"]) and list(tl) == list(mb["t_length"])
    np.testing.assert_array_equal(t, mb["t"])
    np.testing.assert_array_equal(bg, mb["bigram"])


def test_loader_running_stats_and_normalisation(device):
    """Loader.features_to_minibatch: update the running statistics with this minibatch, then normalise with them
    (asr/data/loaders/base.py:16-33,64-80) -- against the reference's own recursion (tests/golden/stats.npz)."""
    from asr.data.loaders.base import Loader
    st = np.load(os.path.join(GOLD, "stats.npz"))
    chunks = [st[k] for k in sorted((k for k in st.files if k.startswith("chunk")), key=lambda s: int(s[5:]))]
    ld = Loader()
    for c in chunks:
        ld._update_stats_recursively(c)
    mean, std = ld.get_mean_and_std()
    np.testing.assert_allclose(mean.cpu().numpy()[0, ..., 0], st["mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ld.stats_nvar.cpu().numpy(), st["nvar"], rtol=1e-4)
    np.testing.assert_allclose(mean.cpu().numpy(), st["bmean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(std.cpu().numpy(), st["bstd"], rtol=1e-4)
    assert ld.stats_total == sum(c.shape[2] for c in chunks)
    # a batch in one kernel call == utterance by utterance
    ld2 = Loader()
    T = max(c.shape[2] for c in chunks)
    xb = np.zeros((len(chunks),) + chunks[0].shape[:2] + (T,), np.float32)
    for i, c in enumerate(chunks):
        xb[i, ..., :c.shape[2]] = c
    ld2._update_stats_batch(torch.from_numpy(xb).to(device), [c.shape[2] for c in chunks])
    np.testing.assert_allclose(ld2.stats_mean.cpu().numpy(), ld.stats_mean.cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(ld2.stats_nvar.cpu().numpy(), ld.stats_nvar.cpu().numpy(), rtol=1e-10)
    # normalisation of the padded array, padding included (:24)
    from asr import _ops
    xd = torch.from_numpy(xb).to(device)
    _ops.normalize_bcmt(xd, ld2._mean32, ld2._std32)
    want = (xb - ld2._mean32.cpu().numpy()[None, ..., None]) / ld2._std32.cpu().numpy()[None, ..., None]
    np.testing.assert_allclose(xd.cpu().numpy(), want, rtol=1e-6, atol=1e-6)


def test_cmn_and_noise(device):
    from asr import _ops
    from asr.data.processing import Processor
    rs = np.random.RandomState(2)
    B, Fmax, nb = 3, 20, 257
    nfr = np.array([20, 13, 1], np.int32)
    p = (rs.rand(B, Fmax, nb).astype(np.float32) + 0.05)
    pd = torch.from_numpy(p.copy()).to(device)
    _ops.cmn_pspec(pd, torch.from_numpy(nfr).to(device))
    got = pd.cpu().numpy()
    for b in range(B):
        np.testing.assert_allclose(got[b, :nfr[b]], otext.cmn_power_spectrum(p[b, :nfr[b]].astype(np.float64)), rtol=2e-5)
        np.testing.assert_array_equal(got[b, nfr[b]:], p[b, nfr[b]:])
    # white noise: integer valued, zero mean, standard deviation = gain (up to truncation), untouched beyond the length
    N = 200000
    sig = torch.zeros((2, N), dtype=torch.float32, device=device)
    lens = torch.tensor([N, N // 2], dtype=torch.int32, device=device)
    gain = torch.tensor([200.0, 50.0], device=device)
    _ops.add_white_noise(sig, lens, gain, 1234)
    s = sig.cpu().numpy()
    assert (s == np.trunc(s)).all() and (s[1, N // 2:] == 0).all()
    assert abs(s[0].mean()) < 2.0 and abs(s[0].std() / 200.0 - 1.0) < 0.02
    assert abs(s[1, :N // 2].std() / 50.0 - 1.0) < 0.03
    assert abs(np.corrcoef(s[0, :N // 2], s[1, :N // 2])[0, 1]) < 0.02
    # end to end: the feature path with CMN equals the oracle's log-mel of the normalised spectrum
    class Aug(object):
        add_noise = False

        def using_augmentation(self):
            return False
    proc = Processor(device=device)
    sigs = [(rs.randn(16000 + 800 * i) * 3000).astype(np.int16) for i in range(2)]
    feats, sents, maxf, maxs = proc.extract_batch_features([(s_, "ア") for s_ in sigs], Aug(), apply_cmn=True)
    for i, s_ in enumerate(sigs):
        spec = offt.get_specgram(s_, 16000, 0.032, 0.01, 512, 0.97, np.hanning)
        spec = otext.cmn_power_spectrum(spec)
        lm = offt.compute_logmel(spec, proc.fbank)
        lm, dl, dd = offt.compute_deltas(lm)
        np.testing.assert_allclose(feats[i][0].cpu().numpy(), lm.T, rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(feats[i][1].cpu().numpy(), dl.T, rtol=2e-3, atol=2e-3)


def test_eval_loop_on_a_model(device):
    """run/ctc/cnn/dev.py:100-108 with this package: model(x, split_into_variables=False) -> argmax -> CER"""
    import asr.functions as F
    from asr import error, vocab
    from asr.model import ds2
    from asr.data.synthetic import synthetic_batch
    cfg = ds2.configure()
    cfg.vocab_size = 119
    cfg.ndim_rnn = 64
    cfg.num_rnn_layers = 1
    torch.manual_seed(0)
    model = ds2.Model(cfg).to_gpu(0)
    x, labels, x_len, l_len = synthetic_batch(4, 60, 119, 3, 9, seed=1)
    with torch.no_grad():
        y = model(x.to(device), split_into_variables=False)
    assert tuple(y.shape) == (4, 60, 119)
    ids = F.argmax(y, axis=2)
    np.testing.assert_array_equal(ids.cpu().numpy(), np.argmax(y.float().cpu().numpy(), axis=2))
    tok, inv = vocab.get_unigram_ids()
    got = error.compute_minibatch_error(ids, labels.to(device), 0, tok, inv)
    want = otext.minibatch_error(ids.cpu().numpy(), labels.numpy(), 0)
    assert abs(got - want) < 1e-12


def test_augment_specgram_against_reference(device):
    """asr/fft.py:21-50 under the reference's seeds (tests/golden/augment.npz): the PUBLIC asr.fft.augment_specgram, seeded through
    np.random.seed as the reference is, returns the reference's array bit for bit; then the batched kernel entry underneath"""
    from asr import _ops, fft
    g = np.load(os.path.join(GOLD, "augment.npz"))
    pspec = g["pspec"].astype(np.float32)
    pd1 = torch.from_numpy(pspec).to(device)
    for seed, want, rate, tract in ((int(g["seed_both"]), g["aug_both"], True, True), (int(g["seed_speed"]), g["aug_speed"], True, False)):
        np.random.seed(seed)
        got = fft.augment_specgram(pd1, change_speech_rate=rate, change_vocal_tract=tract)
        assert got.shape == want.shape
        np.testing.assert_array_equal(got.cpu().numpy(), want.astype(np.float32))
    np.random.seed(int(g["seed_both"]))
    a = fft.augment_specgram(pd1)                                    # defaults = both, as in the reference
    np.testing.assert_array_equal(a.cpu().numpy(), g["aug_both"].astype(np.float32))
    state = np.random.get_state()[1].copy()
    assert fft.augment_specgram(pd1, False, False) is pd1 and (np.random.get_state()[1] == state).all()      # nothing drawn, nothing done
    np.random.seed(5)
    r = max(min(np.random.normal(1, 0.15), 1.2), 0.8)
    np.random.seed(5)
    v = fft.augment_specgram(pd1, False, True).cpu().numpy()         # (NameError in the reference) vocal tract only: length kept
    idx = np.minimum((np.arange(pspec.shape[1]) * r).astype(np.int64), pspec.shape[1] - 1)
    np.testing.assert_array_equal(v, pspec[:, idx])
    for seed, want, use_ratio in ((int(g["seed_both"]), g["aug_both"], True), (int(g["seed_speed"]), g["aug_speed"], False)):
        rs = np.random.RandomState(seed)
        speed = max(min(rs.normal(1, 0.15), 1.2), 0.8)
        ratio = max(min(rs.normal(1, 0.15), 1.2), 0.8) if use_ratio else 1.0
        n_out = int(len(pspec) / speed)
        assert n_out == want.shape[0]
        pd = torch.from_numpy(np.stack([pspec, pspec])).to(device)
        out = _ops.augment_specgram(pd, torch.tensor([n_out, n_out], dtype=torch.int32, device=device),
                                    torch.tensor([speed, 1.0], dtype=torch.float64, device=device),
                                    torch.tensor([ratio, 1.0], dtype=torch.float64, device=device), n_out + 3)
        got = out.cpu().numpy()
        np.testing.assert_array_equal(got[0, :n_out], want.astype(np.float32))
        assert (got[0, n_out:] == 0).all()
        np.testing.assert_array_equal(got[1, :n_out], pspec[:n_out])          # speed = ratio = 1: identity


def test_extract_batch_features_with_warps(device):
    from asr.data.augment import AugmentationOption
    from asr.data.processing import Processor
    rs = np.random.RandomState(9)
    sigs = [(rs.randn(12000 + 1600 * i) * 2000).astype(np.int16) for i in range(3)]
    aug = AugmentationOption()
    aug.change_speech_rate = aug.change_vocal_tract = True
    proc = Processor(device=device)
    np.random.seed(77)
    feats, sents, maxf, maxs = proc.extract_batch_features([(s_, "アイ") for s_ in sigs], aug)
    np.random.seed(77)
    for i, s_ in enumerate(sigs):
        spec = offt.get_specgram(s_, 16000, 0.032, 0.01, 512, 0.97, np.hanning)
        spec = offt.augment_specgram(spec, True, True)              # draws speed, ratio from np.random in the same order
        lm, dl, dd = offt.compute_deltas(offt.compute_logmel(spec, proc.fbank))
        assert feats[i][0].shape[1] == lm.shape[0]
        np.testing.assert_allclose(feats[i][0].cpu().numpy(), lm.T, rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(feats[i][2].cpu().numpy(), dd.T, rtol=2e-3, atol=4e-3)


def test_device_prefetcher(device):
    """asr/data/prefetch.py: minibatches prepared on a side stream equal the ones prepared inline"""
    from asr import vocab
    from asr.data.loaders.base import Loader
    from asr.data.prefetch import DevicePrefetcher
    from asr.data.processing import Processor

    class L(Loader):
        def __init__(self):
            super(L, self).__init__()
            self.processor = Processor(device=device)
            self.token_ids, _ = vocab.get_unigram_ids()
            self.id_blank = 0
    rs = np.random.RandomState(4)
    raw = [[((rs.randn(9000 + 700 * j + 300 * i) * 1500).astype(np.int16), "アイウ"[: 1 + (i + j) % 3]) for j in range(3)] for i in range(4)]
    want = []
    inline = L()
    for b in raw:
        f, s, mf, ms = inline.extract_batch_features(b)
        want.append(inline.features_to_minibatch(f, s, mf, ms))
    torch.cuda.synchronize()
    got = list(DevicePrefetcher(raw, L(), depth=2))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        for a, b in zip(g, w):
            assert torch.equal(a.cpu(), b.cpu())


def test_bucket_loader_end_to_end(device, tmp_path):
    """asr/data/loaders/buckets.py: disk -> Reader -> Processor (GPU) -> minibatch tensors; the dev iterator yields every
    development utterance exactly once, in chunks of batchsizes[bucket]"""
    import sys
    sys.path.insert(0, GOLD)
    import bucket_fixture
    from asr import vocab
    from asr.data.loaders.buckets import Loader
    root = bucket_fixture.build(str(tmp_path))
    ids, _ = vocab.get_unigram_ids()
    ld = Loader(root, batchsizes_train=[3, 4, 2], batchsizes_dev=[2, 2, 2], vocab_token_to_id=ids, dev_split=0.25, seed=3)
    assert ld.get_num_buckets() == 3 and ld.get_total_training_iterations() == 14 and ld.get_total_dev_iterations() == 6
    np.random.seed(5)
    x, xl, t, tl, bg, bucket = ld.sample_minibatch()
    assert x.is_cuda and x.dtype == torch.float32 and x.shape[1:3] == (3, 40) and x.shape[0] == len(xl) == t.shape[0] == 3
    assert bucket == 0 and int(xl.max()) == x.shape[3] and (tl > 0).all()
    seen = 0
    for x, xl, t, tl, bg, bucket in ld.get_development_batch_iterator([2, 2, 2]):
        assert x.shape[0] <= 2 and x.shape[0] == t.shape[0]
        seen += x.shape[0]
    assert seen == sum(len(ix) for pieces in ld.reader.buckets_indices_dev for ix in pieces) == 8
    it = ld.get_training_batch_iterator([3, 4, 2])
    assert it.get_total_iterations() == 14 and len(next(it)) == 6
    # statistics switch on once initialised: the minibatch is then normalised by the running mean / std
    ld._update_stats_batch(x, [int(v) for v in xl])
    x2, *_ = ld.sample_minibatch()
    assert ld.stats_total > 0 and torch.isfinite(x2).all()
