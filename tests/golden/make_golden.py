#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference's NumPy code paths.

Runs ONLY in the build container (it needs /root/reference); the GPU box and the test-suite never
execute it -- they read the committed .npz files.  Nothing of the reference's source is copied:
the files hold inputs and the outputs the reference computed for them.

The reference imports Chainer / CuPy / python_speech_features / acoustics / jaconv at module
scope; none is installed here.  Empty ``types.ModuleType`` stand-ins (no arithmetic) are
registered so the module bodies execute; every number below is produced by the reference's own
NumPy statements:

  G1  asr/fft.py:68-82   get_filterbanks(40, 512, 16000)
  G2  asr/fft.py:58-66,6-19,90-99   compute_logmel + compute_deltas on a seeded power spectrum
  G3  asr/loss/gram_ctc.py:219-297  GramCTC.forward / .backward  (bigram == -1  ==>  standard CTC)
  G4  asr/loss/gram_ctc.py:66-140   forward / backward connection matrices
  G5  asr/nn/sru.py:289-324         SRUFunction.forward_cpu
  G6  asr/vocab.py:107-126, asr/error.py:7-68   tokeniser + greedy-collapse CER
  G7  asr/data/loaders/base.py:64-80,39-41      running mean / n*var statistics
  G8  asr/fft.py:21-50              augment_specgram under a fixed NumPy seed
  G9  asr/nn/layernorm.py:33-48     NormalizeLayer.forward (4-d and 3-d inputs; no epsilon);  asr/nn/convolution_2d.py:21-25,
      62-64  _norm and the weight-normalised W = g * V / _norm(V)

usage:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    import numpy.ma  # noqa: F401  (must be imported before the alias below is installed)
    import scipy.fftpack  # noqa: F401
    np.bool = bool  # alias removed from NumPy >= 1.24, used at asr/loss/gram_ctc.py:83,121,122

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    class _Anything(object):
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return _Anything()

        def __call__(self, *a, **k):
            return _Anything()

    cuda = mod("chainer.cuda", get_array_module=lambda *a: np, cudnn_enabled=False, to_gpu=lambda x: x)
    class _Function(object):        # chainer.Function's book-keeping calls, no arithmetic
        def retain_inputs(self, indexes):
            pass

        def retain_outputs(self, indexes):
            pass

    function = mod("chainer.function", Function=_Function)
    utils = mod("chainer.utils", force_array=np.asarray)
    type_check = mod("chainer.utils.type_check", expect=lambda *a, **k: None)
    conv_nd = mod("chainer.utils.conv_nd")
    conv = mod("chainer.utils.conv")
    utils.type_check, utils.conv_nd, utils.conv = type_check, conv_nd, conv
    variable = mod("chainer.variable", Variable=_Anything, Parameter=_Anything)
    link = mod("chainer.link", Link=object, Chain=object)
    initializers = mod("chainer.initializers", _get_initializer=lambda *a: None)
    configuration = mod("chainer.configuration")
    functions = mod("chainer.functions")
    f_array = mod("chainer.functions.array")
    f_array.broadcast = mod("chainer.functions.array.broadcast", _backward_one=None)    # backward only: never called here
    f_conn = mod("chainer.functions.connection")
    f_conn.convolution_2d = mod("chainer.functions.connection.convolution_2d", Convolution2DFunction=_Function)
    functions.array, functions.connection = f_array, f_conn
    links = mod("chainer.links")
    serializers = mod("chainer.serializers")
    optimizers = mod("chainer.optimizers")
    mod("chainer", cuda=cuda, function=function, utils=utils, variable=variable, link=link,
        initializers=initializers, configuration=configuration, Function=object, Variable=_Anything,
        functions=functions, links=links, serializers=serializers, optimizers=optimizers,
        Chain=object, Link=object, is_debug=lambda: False)
    compiler = mod("cupy.cuda.compiler", compile_using_nvrtc=lambda *a: b"")
    cfunction = mod("cupy.cuda.function")
    ccuda = mod("cupy.cuda", compiler=compiler, function=cfunction)
    mod("cupy", cuda=ccuda)
    sigproc = mod("python_speech_features.sigproc")
    mod("python_speech_features", sigproc=sigproc)
    mod("acoustics")
    mod("jaconv")


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def g_fft(fft):
    fbank = fft.get_filterbanks(nfilt=40, nfft=512, samplerate=16000)
    rs = np.random.RandomState(0)
    pspec = rs.rand(52, 257)
    pspec[3, :] = 0.0       # an all-zero frame exercises the `feat == 0 -> eps` branch (asr/fft.py:64)
    logmel_full = fft.compute_logmel(pspec, 16000, fbank=fbank, nfft=512, nfilt=40)
    logmel, delta, delta_delta = fft.compute_deltas(logmel_full)
    # full-size statistics quoted in SURVEY.md section 8c
    big = np.random.RandomState(0).rand(1002, 257)
    lb, db, ddb = fft.compute_deltas(fft.compute_logmel(big, 16000, fbank=fbank, nfft=512, nfilt=40))
    np.savez_compressed(os.path.join(OUT, "fft.npz"), fbank=fbank, pspec=pspec, logmel_full=logmel_full,
                        logmel=logmel, delta=delta, delta_delta=delta_delta,
                        big_stats=np.array([lb.mean(), db.std(), ddb.std()]),
                        hz2mel_1000=np.array(fft.hz2mel(1000.0)), mel2hz_1000=np.array(fft.mel2hz(1000.0)))
    # G8 augmentation under a fixed seed
    np.random.seed(1234)
    aug = fft.augment_specgram(pspec.copy(), True, True)
    np.random.seed(77)
    aug_speed = fft.augment_specgram(pspec.copy(), True, False)
    np.savez_compressed(os.path.join(OUT, "augment.npz"), pspec=pspec, seed_both=1234, aug_both=aug,
                        seed_speed=77, aug_speed=aug_speed)


def _ctc_case(gc, rs, B, T, L, V, n_uni, bigram_mode, ragged, reduce, repeats=False, scale=1.0):
    """One Gram-CTC case.  Returns dict of inputs and the reference's outputs."""
    xs = (rs.randn(T, B, V) * scale).astype(np.float32)
    uni = rs.randint(1, n_uni, size=(B, L)).astype(np.int32)
    if repeats:
        uni[:, 1::2] = uni[:, 0::2][:, :uni[:, 1::2].shape[1]]     # u_i == u_{i-1} on odd positions
    if bigram_mode == "none":
        big = np.full((B, L), -1, dtype=np.int32)
    else:
        big = rs.randint(n_uni, V, size=(B, L)).astype(np.int32)
        if bigram_mode == "mixed":
            big[rs.rand(B, L) < 0.3] = -1
        elif bigram_mode == "repeat2":
            big[:, 2:] = big[:, :-2]                                 # b_i == b_{i-2}
        big[:, 0] = -1                                               # asr/data/processing.py:139
    if ragged == "len1":
        tl = np.ones(B, dtype=np.int32)
        tl[0] = L
        xl = np.full(B, T, dtype=np.int32)
        xl[-1] = 3
    elif ragged:
        tl = rs.randint(max(1, L // 2), L + 1, size=B).astype(np.int32)
        xl = np.array([rs.randint(min(T, 3 * int(l) + 2), T + 1) for l in tl], dtype=np.int32)
        tl[0], xl[0] = L, T
    else:
        tl = np.full(B, L, dtype=np.int32)
        xl = np.full(B, T, dtype=np.int32)
    for b in range(B):      # padding as the reference's loader leaves it (asr/data/processing.py:125-126)
        uni[b, tl[b]:] = 0
        big[b, tl[b]:] = 0
    f = gc.GramCTC(0, reduce)
    inputs = (xl.copy(), tl.copy(), uni.copy(), big.copy()) + tuple(xs[t].copy() for t in range(T))
    loss, = f.forward(inputs)
    loss = np.asarray(loss, dtype=np.float32)
    gy = np.float32(1.0) if reduce == "mean" else rs.rand(B).astype(np.float32)
    grads = f.backward(inputs, (gy,))
    gx = np.stack(grads[4:]).astype(np.float32)
    return dict(xs=xs, uni=uni, big=big, xl=xl, tl=tl, loss=loss, gy=np.asarray(gy, np.float32), gx=gx,
                reduce=np.array(reduce), prob_trans0=np.asarray(f.prob_trans[0], np.float32))


def g_ctc(gc):
    rs = np.random.RandomState(20261003)
    cases = {
        # name: (B, T, L, V, n_uni, bigram_mode, ragged, reduce, repeats, scale)
        "ctc_small": (3, 20, 4, 9, 9, "none", True, "mean", True, 1.0),
        "ctc_noreduce": (4, 30, 6, 12, 12, "none", True, "no", False, 2.0),
        "ctc_full": (2, 25, 5, 7, 7, "none", False, "mean", False, 1.0),
        "ctc_v300": (3, 60, 12, 300, 300, "none", True, "mean", True, 1.0),
        "ctc_v3000": (2, 80, 15, 3000, 3000, "none", True, "mean", False, 1.0),
        # Lmax must be >= 2: asr/loss/gram_ctc.py:115 indexes column 4 of a (B, 3*Lmax+1) mask
        "ctc_len1": (3, 8, 2, 6, 6, "none", "len1", "mean", False, 1.0),
        "gram_mixed": (3, 24, 5, 40, 12, "mixed", True, "mean", False, 1.0),
        "gram_all": (2, 30, 6, 30, 10, "all", False, "mean", True, 1.0),
        "gram_repeat2": (3, 36, 7, 30, 8, "repeat2", True, "no", False, 1.5),
        "gram_v3000": (2, 70, 12, 3000, 119, "mixed", True, "mean", True, 1.0),
        "gram_len1": (2, 9, 2, 20, 8, "all", "len1", "mean", False, 1.0),
    }
    out = {}
    for name, c in cases.items():
        r = _ctc_case(gc, rs, *c)
        for k, v in r.items():
            out["%s.%s" % (name, k)] = v
    out["names"] = np.array(sorted(cases.keys()))
    np.savez_compressed(os.path.join(OUT, "gram_ctc.npz"), **out)

    # G4 connection matrices (log 0 / zero_padding) for one mixed case
    uni = np.array([[3, 3, 5, 2], [4, 1, 1, 0]], dtype=np.int32)
    big = np.array([[-1, 20, -1, 22], [-1, 25, 21, 0]], dtype=np.int32)
    tl = np.array([4, 3], dtype=np.int32)
    plen = tl * 3 + 1
    N = uni.shape[1] * 3 + 1
    fwd = gc._create_forward_connection_matrix(uni, big, plen, N, np.float32, np, -1e10)
    runi = gc._reverse_path(uni, tl, np)
    rbig = gc._reverse_path(big, tl, np)
    bwd = gc._create_backward_connection_matrix(runi, rbig, plen, N, np.float32, np, -1e10)
    path = gc._label_to_path(uni, big, 0, np)
    np.savez_compressed(os.path.join(OUT, "gram_ctc_connection.npz"), uni=uni, big=big, tl=tl, fwd=fwd, bwd=bwd,
                        path=path, rev_path=gc._reverse_path(path, plen, np))


def g_sru(sru):
    rs = np.random.RandomState(7)
    out = {}
    for name, (B, D, T, tanh, masked) in {"tanh": (3, 8, 11, True, False), "linear": (2, 6, 9, False, False),
                                          "masked": (4, 8, 7, True, True)}.items():
        X = rs.randn(B, D, T).astype(np.float32)
        W = (rs.randn(3 * D, D) * 0.4).astype(np.float32)
        Bias = (rs.randn(2 * D) * 0.3).astype(np.float32)
        c0 = rs.randn(B, D).astype(np.float32)
        mask = (rs.rand(B, D) >= 0.3).astype(np.float32) if masked else np.ones((B, D), np.float32)
        # forward_cpu multiplies x by the mask only in the highway term (asr/nn/sru.py:301); U uses raw X
        H, C, cT = sru.SRUFunction(tanh).forward_cpu((X, W, Bias, c0, mask))
        for k, v in dict(X=X, W=W, B=Bias, c0=c0, mask=mask, H=H, C=C, cT=cT, use_tanh=np.array(tanh)).items():
            out["%s.%s" % (name, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "sru.npz"), **out)


def g_text(vocab, error):
    ids, inv = vocab.get_unigram_ids()
    sents = ["キョウワイイテンキデス", "ヴァイオリン", "ファッション", "シェークスピア", "ア", "ヴェルディ", "クヮルテット", "スィート",
             "テュニス", "ミェヴ", "フュージョン"]
    toks = [vocab.convert_sentence_to_unigram_tokens(s) for s in sents]
    tok_ids = [vocab.convert_sentence_to_unigram_ids(s, ids) for s in sents]
    rs = np.random.RandomState(3)
    y = rs.randint(0, 6, size=(5, 40))          # argmax sequences with blanks (0) and repeats
    t = np.zeros((5, 12), dtype=np.int64)
    for b in range(5):
        n = rs.randint(3, 12)
        t[b, :n] = rs.randint(1, 6, size=n)
    cer = error.compute_minibatch_error(y, t, 0, ids, inv)
    per = []
    for b in range(5):
        per.append(error.compute_minibatch_error(y[b:b + 1], t[b:b + 1], 0, ids, inv))
    pairs_r = [[1, 2, 3, 4], [1, 2], [5, 5, 5], [1, 2, 3]]
    pairs_h = [[1, 3, 4], [1, 2], [], [3, 2, 1, 1]]
    lev = [error.compute_character_error_rate(r, h) for r, h in zip(pairs_r, pairs_h)]
    import json
    with open(os.path.join(OUT, "text.json"), "w") as f:
        json.dump(dict(vocab_size=len(ids), blank=vocab.ID_BLANK, unigram_tokens=list(vocab.UNIGRAM_TOKENS),
                       sutegana=list(vocab.SUTEGANA), collapse=dict(vocab.UNIGRAM_COLLAPSE),
                       sentences=sents, tokens=toks, token_ids=tok_ids,
                       y=y.tolist(), t=t.tolist(), cer_mean=cer, cer_each=per, lev_r=pairs_r, lev_h=pairs_h, lev=lev),
                  f, ensure_ascii=False, indent=1)


def g_minibatch(vocab):
    """asr/data/processing.py:113-173 (Processor.features_to_minibatch) called unbound on a stand-in self: padding,
    unigram / bigram ids, and the truncation of labels that cannot be aligned in x_length frames."""
    spec = importlib.util.spec_from_file_location("refasr.data.processing", os.path.join(REF, "asr/data/processing.py"))
    proc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(proc)
    ids, _ = vocab.get_unigram_ids()
    ids = dict(ids)
    for tok in ("キョ" + "ウ", "イ" + "イ", "テ" + "ン"):          # a few bigram entries, the rest stay out of vocabulary (-1)
        ids[tok] = len(ids)
    sents = ["キョウワイイテンキデス", "アアアアアア", "ファッション", "ア", "ヴァイオリンノオト"]
    lens = [40, 9, 6, 3, 11]           # utterances 1, 2 and 4 are too short for their transcription
    rs = np.random.RandomState(11)
    feats = [tuple(rs.randn(40, n).astype(np.float64) for _ in range(3)) for n in lens]

    class Self(object):
        using_delta = True
        using_delta_delta = True
        num_mel_filters = 40
    maxs = max(len(s) for s in sents)
    x, xl, t, tl, bg = proc.Processor.features_to_minibatch(Self(), feats, sents, max(lens), maxs, ids, 0)
    out = dict(x=x, x_length=np.asarray(xl), t=t, t_length=np.asarray(tl), bigram=bg, lens=np.asarray(lens), max_sentence_length=np.asarray(maxs))
    for i, f in enumerate(feats):
        for c in range(3):
            out["feat%d_%d" % (i, c)] = f[c]
    np.savez_compressed(os.path.join(OUT, "minibatch.npz"), **out)
    import json
    with open(os.path.join(OUT, "minibatch.json"), "w") as f:
        json.dump(dict(sentences=sents, token_ids=ids), f, ensure_ascii=False, indent=1)


def g_buckets():
    """asr/data/readers/buckets.py:26-209 on the synthetic corpus of bucket_fixture.py: the seeded train / dev split, the
    iteration counts and a few sampled minibatches (identified by signal length and sentence)."""
    import json
    import tempfile
    sys.path.insert(0, OUT)
    import bucket_fixture
    sys.modules["refasr.data.readers"] = types.ModuleType("refasr.data.readers")
    sys.modules["refasr.data.readers"].__path__ = [os.path.join(REF, "asr/data/readers")]
    spec = importlib.util.spec_from_file_location("refasr.data.readers.buckets", os.path.join(REF, "asr/data/readers/buckets.py"))
    rb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rb)
    with tempfile.TemporaryDirectory() as root:
        bucket_fixture.build(root)
        reader = rb.Reader(root, buckets_limit=None, buckets_cache_size=2, dev_split=0.25, seed=3)
        out = dict(train=[[list(map(int, ix)) for ix in pieces] for pieces in reader.buckets_indices_train],
                   dev=[[list(map(int, ix)) for ix in pieces] for pieces in reader.buckets_indices_dev],
                   distribution=[float(v) for v in reader.bucket_distribution],
                   train_iterations=reader.calculate_total_training_iterations_with_batchsizes([3, 4, 2]),
                   dev_iterations=reader.calculate_total_dev_iterations_with_batchsizes([3, 4, 2]))
        np.random.seed(5)
        samples = []
        for _ in range(6):
            batch, b, p = reader.sample_minibatch([3, 4, 2])
            samples.append(dict(bucket=int(b), piece=int(p), items=[[int(len(sig)), sent] for sig, sent in batch]))
        out["samples"] = samples
        out["statistics"] = reader.get_statistics()
    with open(os.path.join(OUT, "buckets.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False, indent=1)


def g_stats():
    """asr/data/loaders/base.py:64-80,39-41 -- executed from the reference file with its package imports stubbed."""
    for n in ("asr", "asr.data", "asr.data.readers", "asr.data.readers.buckets", "asr.data.processing", "asr.utils",
              "asr.data.iterators", "asr.data.loaders"):
        m = types.ModuleType(n)
        m.__path__ = []
        sys.modules.setdefault(n, m)
    sys.modules["asr.data.readers.buckets"].Reader = object
    sys.modules["asr.data.processing"].Processor = object
    u = sys.modules["asr.utils"]
    u.stdout, u.printb, u.Object = None, print, object
    sys.modules["asr.data"].iterators = sys.modules["asr.data.iterators"]
    spec = importlib.util.spec_from_file_location("asr.data.loaders.base", os.path.join(REF, "asr/data/loaders/base.py"))
    base = importlib.util.module_from_spec(spec)
    base.__package__ = "asr.data.loaders"
    spec.loader.exec_module(base)
    ld = base.Loader()
    rs = np.random.RandomState(11)
    chunks = [(rs.randn(3, 40, n) * (1 + 0.1 * i) + 0.3 * i).astype(np.float32) for i, n in enumerate([17, 5, 33, 8])]
    for c in chunks:
        ld._update_stats_recursively(c)
    mean, std = ld.get_mean_and_std()
    np.savez_compressed(os.path.join(OUT, "stats.npz"), mean=ld.stats_mean, nvar=ld.stats_nvar, total=ld.stats_total,
                        bmean=mean, bstd=std, **{"chunk%d" % i: c for i, c in enumerate(chunks)})


def g_norm():
    """G9: the reference's own NumPy statements for the layer normalisation forward and the weight-norm scale"""
    ln = _load("ref_layernorm", "asr/nn/layernorm.py")
    c2d = _load("ref_convolution_2d", "asr/nn/convolution_2d.py")
    rs = np.random.RandomState(9)
    out = {}
    for name, shape, dtype in (("x4", (3, 6, 5, 7), np.float32), ("x4_f64", (2, 4, 3, 5), np.float64), ("x3", (2, 11, 6), np.float32),
                               ("x4_wide", (2, 3000, 1, 4), np.float32)):
        x = (rs.randn(*shape) * 2.0 + 0.5).astype(dtype)
        f = ln.NormalizeLayer()
        y, = f.forward((x,))
        out[name], out[name + ".y"], out[name + ".std"] = x, np.asarray(y), np.asarray(f.std)
    V = (rs.randn(5, 3, 3, 5) * 0.3).astype(np.float32)
    g = (rs.rand(5, 1, 1, 1) + 0.5).astype(np.float32)
    norm = c2d._norm(V)
    out["V"], out["g"], out["norm"], out["W"] = V, g, np.asarray(norm), np.asarray(g * (V / norm))    # asr/nn/convolution_2d.py:62-64
    np.savez_compressed(os.path.join(OUT, "norm.npz"), **out)


def main():
    _install_stubs()
    g_norm()
    fft = _load("ref_fft", "asr/fft.py")
    gc = _load("ref_gram_ctc", "asr/loss/gram_ctc.py")
    sru = _load("ref_sru", "asr/nn/sru.py")
    vocab = _load("ref_vocab", "asr/vocab.py")
    g_fft(fft)
    g_ctc(gc)
    g_sru(sru)
    # asr/error.py uses package-relative imports: load it as part of a synthetic package
    pkg = types.ModuleType("refasr")
    pkg.__path__ = [os.path.join(REF, "asr")]
    sys.modules["refasr"] = pkg
    sys.modules["refasr.vocab"] = vocab
    sys.modules["refasr.utils"] = types.ModuleType("refasr.utils")
    for n in ("printb", "printr", "printc", "stdout"):
        setattr(sys.modules["refasr.utils"], n, print)
    spec = importlib.util.spec_from_file_location("refasr.error", os.path.join(REF, "asr/error.py"))
    error = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(error)
    g_text(vocab, error)
    sys.modules["refasr.data"] = types.ModuleType("refasr.data")
    sys.modules["refasr.data"].__path__ = [os.path.join(REF, "asr/data")]
    sys.modules["refasr.fft"] = fft
    g_minibatch(vocab)
    g_buckets()
    g_stats()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


def g_hdf5():
    """tests/golden/tiny.hdf5: the bytes this package's own HDF5 writer (asr/hdf5_lite.py) produces for a fixed table -- not a
    reference output (h5py does not exist here): a pin against accidental changes of the on-disk structures"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(OUT)), "chainer-speech-recognition_amd"))
    from asr import hdf5_lite
    table = {"layer_0/W": np.arange(24, dtype=np.float32).reshape(2, 3, 4) / 8, "layer_0/b": np.array([1.5, -2.0], np.float32),
             "layer_5_0/W": np.ones((1, 2), np.float32), "N": np.array(3, np.int64)}
    hdf5_lite.write(os.path.join(OUT, "tiny.hdf5"), table)


if __name__ == "__main__":
    if "--hdf5" in sys.argv:
        g_hdf5()
    else:
        main()
