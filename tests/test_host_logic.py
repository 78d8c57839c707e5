"""Host-side logic of the drop-in API that needs no GPU: structure, naming, configuration, optimiser bookkeeping."""
import json
import os

import pytest
import torch


def test_stream_and_module_register_links_with_reference_names():
    """parameter names follow asr/nn/nn.py:304-320 (layer_%d, layer_%d_%d) and :341-392 (_sequential_%d...)."""
    import asr.nn as nn
    s = nn.Stream()
    s.layer(nn.Convolution2D(3, 8, (3, 5), pad=(0, 4)), lambda x: x[..., :-4], nn.Maxout(2))
    s.layer(nn.Residual(nn.Convolution2D(4, 8, (3, 5), pad=(1, 4)), nn.Maxout(2)))
    s.layer(nn.GLU(4, 4), nn.LayerNormalization(4))
    names = [n for n, _ in s.namedparams()]
    assert "/layer_0/W" in names and "/layer_0/b" in names
    assert "/layer_3_0/W" in names                       # Residual at index 3, conv is its element 0
    assert "/layer_4/W" in names                          # GLU registers its conv
    assert "/layer_5/gamma" in names and "/layer_5/beta" in names
    assert len(s.layers) == 6
    m = nn.Module()
    m.add(nn.Convolution1D(8, 16), nn.Maxout(2))
    m.add(nn.SRU(8), nn.Dropout(0))
    assert [n for n, _ in m.namedparams()] == ["/_sequential_0/W", "/_sequential_0/b", "/_sequential_2/W", "/_sequential_2/B"]
    assert len(m.blocks) == 2 and len(m.blocks[1]) == 2
    outer = nn.Module()
    outer.inner = m
    assert m._locked and any(n.startswith("/inner/") for n, _ in outer.namedparams())
    with pytest.raises(AssertionError):
        m.extra = nn.Convolution1D(4, 4)                 # owned module refuses new links (asr/nn/nn.py:363)


def test_parameter_shapes_and_lazy_initialisation():
    import asr.nn as nn
    c = nn.Convolution2D(3, 16, (3, 5), pad=(0, 4))
    assert tuple(c.W.shape) == (16, 3, 3, 5) and tuple(c.b.shape) == (16,)
    assert float(c.b.abs().sum()) == 0.0
    assert nn.Convolution1D(None, 32).W.numel() == 0      # sized at first call (asr/nn/convolution_1d.py:33-35)
    assert nn.LayerNormalization(None).gamma.numel() == 0
    ln = nn.LayerNormalization(7)
    assert torch.equal(ln.gamma.data, torch.ones(7)) and torch.equal(ln.beta.data, torch.zeros(7))
    sru = nn.SRU(12)
    assert tuple(sru.W.shape) == (36, 12) and tuple(sru.B.shape) == (24,)      # asr/nn/sru.py:455-458
    gru = nn.BiGRU(20, 32)
    assert tuple(gru.w_ih.shape) == (2, 96, 20) and tuple(gru.w_hh.shape) == (2, 96, 32)
    wn = nn.Convolution2D(3, 8, (3, 5), pad=(0, 4), weightnorm=True)
    assert tuple(wn.V.shape) == (8, 3, 3, 5) and wn.g.numel() == 0             # g, b come from the first batch


def test_initialisers_follow_the_reference_recipes():
    import math
    from asr.link import HeNormal, LeCunNormal, Normal
    torch.manual_seed(0)
    w = Normal(math.sqrt(1.0 / 128 / 3 / 5))((256, 128, 3, 5))          # run/ctc/cnn/model.py:162
    assert abs(w.std().item() - math.sqrt(1.0 / 1920)) < 2e-4
    w = LeCunNormal()((64, 100))
    assert abs(w.std().item() - 0.1) < 5e-3
    w = HeNormal(1.0)((64, 200))
    assert abs(w.std().item() - 0.1) < 5e-3


def test_configuration_roundtrip(tmp_path):
    from asr.model import cnn, sru, ds2
    c = cnn.configure()
    assert (c.ndim_h, c.ndim_dense, c.num_conv_layers, c.architecture, c.num_mel_filters) == (128, 256, 5, "zhang", 40)
    with pytest.raises(AssertionError):
        c.save(str(tmp_path / "c.json"))                    # vocab_size must be set (asr/model/cnn.py:22-24)
    c.vocab_size = 119
    c.kernel_size = (3, 5)
    c.save(str(tmp_path / "c.json"))
    d = json.load(open(tmp_path / "c.json"))
    assert d["vocab_size"] == 119 and d["frame_width"] == 0.032 and d["bucket_split_sec"] == 0.5
    c2 = cnn.configure()
    assert c2.load(str(tmp_path / "c.json")) is True and c2.vocab_size == 119 and tuple(c2.kernel_size) == (3, 5)
    assert c2.load(str(tmp_path / "missing.json")) is None
    assert sru.configure().num_rnn_layers == 2 and ds2.configure().ndim_rnn == 512


def test_optimizer_factory_and_learning_rate_helpers():
    """asr/optimizers.py:3-75"""
    from asr import optimizers as O
    adam = O.get_optimizer("adam", 1e-3, 0.9)
    assert isinstance(adam, O.Adam) and adam.alpha == 1e-3 and adam.beta1 == 0.9 and adam.beta2 == 0.999 and adam.eps == 1e-8
    assert isinstance(O.get_optimizer("sgd", 0.1, 0.9), O.SGD)
    assert O.get_optimizer("msgd", 0.1, 0.8).momentum == 0.8 and isinstance(O.get_optimizer("nesterov", 0.1, 0.9), O.NesterovAG)
    with pytest.raises(NotImplementedError):
        O.get_optimizer("rmsprop", 0.1, 0.9)
    assert O.get_learning_rate(adam) == 1e-3
    O.decay_learning_rate(adam, 0.5, 1e-6)
    assert adam.alpha == 5e-4
    O.set_learning_rate(adam, 1e-6)
    assert O.decay_learning_rate(adam, 0.5, 1e-6) == 1e-6 and adam.alpha == 1e-6       # floor (asr/optimizers.py:70-72)
    O.set_momentum(adam, 0.5)
    assert adam.beta1 == 0.5
    adam.add_hook(O.GradientClipping(1))
    adam.add_hook(O.WeightDecay(1e-5))
    assert [h.name for h in adam._hooks] == ["GradientClipping", "WeightDecay"]


def test_views_cost_nothing_and_keep_reference_semantics():
    from asr import functions as F
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    assert F.swapaxes(x, 1, 3).shape == (2, 5, 4, 3) and F.swapaxes(x, 1, 3).data_ptr() == x.data_ptr()
    parts = F.split_axis(x.reshape(2, 60), 5, axis=1)
    assert len(parts) == 5 and parts[0].shape == (2, 12)
    assert F.squeeze(x[:, :, :1], axis=2).shape == (2, 3, 5)
    assert F.expand_dims(x, 1).shape == (2, 1, 3, 4, 5)
    assert F.reshape(x, (2, -1, 5)).shape == (2, 12, 5)
    phys = torch.zeros(5, 2, 4, 3, dtype=torch.bfloat16)            # (T, B, H, C)
    logical = phys.permute(1, 3, 2, 0)                                # (B, C, H, T) view
    merged = F.reshape(logical, (2, -1, 5))
    assert merged.shape == (2, 12, 5) and merged.data_ptr() == phys.data_ptr()       # no copy


def test_logit_time_steps_reach_the_loss_without_a_copy():
    """asr/model/cnn.py:41-49 hands T Variables (B, V) to the loss; here they are views of one buffer and the loss must
    find that buffer again (a torch.stack of 1000 views costs 0.4 ms per train step)"""
    from asr.model._acoustic import split_output, TimeSteps
    from asr.loss.ctc import _as_tbv
    phys = torch.randn(7, 2, 1, 5, requires_grad=True) * 1.0          # (T, B, 1, V) buffer of the last layer
    steps = split_output(phys.permute(1, 3, 2, 0), 2, 7, True)         # logical (B, V, 1, T)
    assert isinstance(steps, (tuple, TimeSteps)) and len(steps) == 7 and steps[3].shape == (2, 5)
    assert torch.equal(steps[3], phys[3, :, 0])
    assert _as_tbv(steps).data_ptr() == phys.data_ptr() and _as_tbv(steps).shape == (7, 2, 5)
    plain = tuple(steps)                                               # a user re-wrapping the tuple: still no copy
    assert _as_tbv(plain).data_ptr() == phys.data_ptr()
    foreign = [torch.randn(2, 5) for _ in range(7)]                    # the reference's own list of arrays: stacked
    assert _as_tbv(foreign).shape == (7, 2, 5)
    assert torch.equal(_as_tbv(foreign)[4], foreign[4])


def test_ctc_argument_checks_match_reference():
    """asr/loss/gram_ctc.py:224-227,301-308"""
    from asr.loss import connectionist_temporal_classification, gram_ctc
    x = torch.zeros(4, 2, 5)
    t = torch.ones(2, 2, dtype=torch.int32)
    with pytest.raises(TypeError):
        gram_ctc(5, t, t, 0)
    with pytest.raises(TypeError):
        gram_ctc(x, t, t, 0.0)
    with pytest.raises(ValueError):
        connectionist_temporal_classification(x, t, 0, reduce="sum")
    with pytest.raises(AssertionError):
        connectionist_temporal_classification(x, t, 7)
    with pytest.raises(AssertionError):
        gram_ctc(x, t, torch.ones(2, 3, dtype=torch.int32), 0)


def test_feature_host_constants():
    from asr import fft
    assert fft.num_frames(160672, 512, 160) == 1002
    assert fft.num_frames(400, 512, 160) == 1 and fft.num_frames(513, 512, 160) == 2
    fb = fft.get_filterbanks(40, 512, 16000)
    assert fb.shape == (40, 257) and fb.sum() == pytest.approx(247.0)


@pytest.mark.parametrize("arch", ["zhang", "zhang+fc_relu", "zhang+residual", "zhang+layernorm", "glu", "relu+layernorm",
                                  "relu+layernorm+residual"])
def test_build_model_architectures(arch):
    """layer counts, channel plan and parameter names of run/ctc/cnn/model.py:11-332"""
    from asr.model import cnn
    from asr.model.architectures import build_model
    import asr.nn as nn
    cfg = cnn.configure()
    cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers, cfg.architecture = 119, 3, 128, 320, 4, arch
    m = build_model(cfg)
    names = [n for n, _ in m.namedparams()]
    assert "/layer_0/W" in names
    assert tuple(m.layer_0.W.shape) == ((256 if arch.startswith("zhang") or arch == "glu" else 128), 3, 3, 5)
    assert isinstance(m.layers[-1], nn.LayerNormalization)            # every recipe ends Conv 1x1 -> LayerNormalization
    if arch == "zhang+residual":
        # 3 residual blocks + 1 plain conv block (run/ctc/cnn/model.py:160-176), then 3 dense blocks
        assert sum(isinstance(l, nn.Residual) for l in m.layers) == 3
        macs = 256 * 38 * 3 * 15 + 4 * 256 * 13 * 128 * 15 + 640 * 128 * 13 + 640 * 320 + 119 * 320
        assert abs(macs / 1e6 - 27.3) < 0.05                           # 27.3 MMAC / frame
        assert "/layer_5_0/W" in names and tuple(m.layer_5_0.W.shape) == (256, 128, 3, 5)
    if arch == "relu+layernorm+residual":
        assert sum(isinstance(l, nn.Residual) for l in m.layers) == 4
        assert isinstance(m.layers[6].layers[0], nn.LayerNormalization)       # pre-activation block (:315-323)
    cfg.num_conv_layers = 6
    if arch == "zhang+residual":
        wide = build_model(cfg)                                         # "VGG-deep" branch: 4th layer 128->512, +4 layers 256->512
        shapes = [tuple(p.shape) for n, p in wide.namedparams() if n.endswith("/W") and p.dim() == 4 and p.shape[2:] == (3, 5)]
        assert shapes.count((512, 256, 3, 5)) == 4 and (512, 128, 3, 5) in shapes
    cfg.architecture = "unknown"
    with pytest.raises(NotImplementedError):
        build_model(cfg)


def test_bucket_reader_matches_reference(tmp_path):
    """asr/data/readers/buckets.py: the seeded train / dev split, iteration counts and sampled minibatches equal the
    reference Reader's on the same synthetic corpus (tests/golden/buckets.json, made by make_golden.py)."""
    import json
    import os
    import sys
    import numpy as np
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold_dir)
    import bucket_fixture
    from asr.data.readers.buckets import Reader
    with open(os.path.join(gold_dir, "buckets.json"), encoding="utf-8") as f:
        g = json.load(f)
    root = bucket_fixture.build(str(tmp_path))
    reader = Reader(root, buckets_limit=None, buckets_cache_size=2, dev_split=0.25, seed=3)
    assert [[list(map(int, ix)) for ix in pieces] for pieces in reader.buckets_indices_train] == g["train"]
    assert [[list(map(int, ix)) for ix in pieces] for pieces in reader.buckets_indices_dev] == g["dev"]
    np.testing.assert_allclose(reader.bucket_distribution, g["distribution"])
    assert reader.calculate_total_training_iterations_with_batchsizes([3, 4, 2]) == g["train_iterations"]
    assert reader.calculate_total_dev_iterations_with_batchsizes([3, 4, 2]) == g["dev_iterations"]
    assert reader.get_num_buckets() == 3 and reader.buckets_num_data == [[7, 4], [12], [5, 9, 3]]
    np.random.seed(5)
    for want in g["samples"]:
        batch, b, p = reader.sample_minibatch([3, 4, 2])
        assert (int(b), int(p)) == (want["bucket"], want["piece"])
        assert [[int(len(sig)), sent] for sig, sent in batch] == want["items"]
    assert reader.get_statistics() == g["statistics"]
    limited = Reader(root, buckets_limit=2, dev_split=0.25, seed=3)
    assert limited.get_num_buckets() == 2


# ---------------------------------------------------------------------------------------------- checkpoints (row f4)
def _ds2_small():
    from asr.model import ds2
    cfg = ds2.configure()
    cfg.vocab_size, cfg.ndim_conv, cfg.ndim_rnn, cfg.ndim_dense, cfg.num_rnn_layers = 17, 8, 32, 16, 2
    return cfg, ds2.Model(cfg)


def _materialise_ds2(cfg, m):
    m.rnn_blocks.layers[0]._initialize_params(cfg.ndim_conv * 6)
    for i in range(1, cfg.num_rnn_layers):
        m.rnn_blocks.layers[2 * i]._initialize_params(cfg.ndim_rnn)
    m.dense_blocks.layers[0]._initialize_params(cfg.ndim_rnn)
    m.dense_blocks.layers[7].norm._initialize_params(cfg.vocab_size)


def test_checkpoint_names_follow_the_reference(tmp_path):
    """Chainer's serialisers store a parameter under the attribute path that leads to it; the reference's containers name
    those attributes layer_%d / layer_%d_%d (asr/nn/nn.py:304-320) and _module_<ns>_sequential_%d (:355-392)"""
    import torch
    from asr import serializers
    from asr.model import cnn
    from asr.model.architectures import build_model
    torch.manual_seed(0)
    cfg = cnn.configure()
    cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers, cfg.architecture = 19, 3, 8, 12, 3, "zhang+residual"
    model = build_model(cfg)
    names = sorted(serializers.to_table(model).keys())
    # first block: conv at 0; residual blocks at layer index 5 and 6 hold their conv at inner index 0; plain block conv at 7
    assert "layer_0/W" in names and "layer_0/b" in names
    assert "layer_5_0/W" in names and "layer_6_0/W" in names and "layer_7/W" in names
    assert all("/" in n and "." not in n for n in names)
    cfg.architecture = "glu"
    glu = build_model(cfg)
    gl = sorted(serializers.to_table(glu).keys())
    assert "layer_5/W" in gl and "layer_7/b" in gl          # GLU registers its convolution under the layer's own index (:312-313)
    cfg2, m = _ds2_small()
    _materialise_ds2(cfg2, m)
    t = serializers.to_table(m)
    for key in ("_module_conv_blocks_sequential_0/W", "_module_rnn_blocks_sequential_0/w_ih", "_module_rnn_blocks_sequential_2/w_hh",
                "_module_dense_blocks_sequential_6/W", "_module_dense_blocks_sequential_7/norm/gamma"):
        assert key in t, (key, sorted(t)[:8])


def test_checkpoint_round_trip_on_a_fresh_model(tmp_path):
    """build_model(config); model.load(path) -- the reference's resume / eval flow (run/ctc/cnn/train.py:
    105-106) -- on a model whose lazily sized parameters are still empty"""
    import numpy as np
    import torch
    from asr import serializers
    torch.manual_seed(1)
    cfg, m = _ds2_small()
    _materialise_ds2(cfg, m)
    path = str(tmp_path / "model.hdf5")            # the reference's file name; the container is sniffed on load
    m.save(path)
    assert serializers.sniff(path) == "hdf5"              # the reference's container (asr/model/cnn.py:51-56), h5py or not
    torch.manual_seed(2)
    _, fresh = _ds2_small()
    assert fresh.rnn_blocks.layers[0].w_ih.numel() == 0
    assert fresh.load(path) is True
    a, b = m.state_dict(), fresh.state_dict()
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert fresh.load(str(tmp_path / "missing.hdf5")) is False
    # the merged (channel, height) columns of the first recurrent layer are stored in the reference's (c, h) order
    C, H = m._merged
    w = m.rnn_blocks.layers[0].w_ih.detach().numpy()
    stored = serializers.read_hdf5_table(path)["_module_rnn_blocks_sequential_0/w_ih"]
    h, c = 3, 5
    assert np.array_equal(stored[..., c * H + h], w[..., h * C + c])
    # a torch.save'd state_dict of round 1 still loads (own names, own column order)
    old = str(tmp_path / "old.pt")
    torch.save(m.state_dict(), old)
    _, again = _ds2_small()
    assert again.load(old) is True
    assert torch.equal(again.rnn_blocks.layers[0].w_ih, m.rnn_blocks.layers[0].w_ih)
    # torch's own load_state_dict also sizes the lazy parameters now
    _, third = _ds2_small()
    third.load_state_dict(m.state_dict())
    assert torch.equal(third.dense_blocks.layers[7].norm.gamma, m.dense_blocks.layers[7].norm.gamma)


def test_strict_load_refuses_an_incomplete_file_before_touching_the_model(tmp_path):
    """a checkpoint that lacks an entry raised KeyError AFTER the entries it did have had been copied in"""
    import torch
    from asr import serializers
    torch.manual_seed(3)
    cfg, m = _ds2_small()
    _materialise_ds2(cfg, m)
    table = serializers.to_table(m)
    del table["_module_dense_blocks_sequential_6/W"]
    torch.manual_seed(4)
    _, other = _ds2_small()
    _materialise_ds2(cfg, other)
    before = {k: v.clone() for k, v in other.state_dict().items()}
    with pytest.raises(KeyError):
        serializers.from_table(other, table, strict=True)
    for k, v in other.state_dict().items():
        assert torch.equal(v, before[k]), k
    unused = serializers.from_table(other, table, strict=False)
    assert unused == [] and not torch.equal(other.state_dict()["conv_blocks._sequential_0.W"], before["conv_blocks._sequential_0.W"])


# ---------------------------------------------------------------------------------------------- HDF5 container (row f4)
def _hdf5_table(rs):
    import numpy as np
    table = {"layer_%d/W" % i: rs.randn(3, 4, 2).astype(np.float32) for i in range(40)}          # 40 links: 5 symbol-table nodes
    table.update({"layer_%d/b" % i: rs.randn(5).astype(np.float32) for i in range(40)})
    table["_module_rnn_blocks_sequential_0/w_ih"] = rs.randn(2, 6, 7).astype(np.float32)
    table["_module_dense_blocks_sequential_7/norm/gamma"] = rs.randn(9).astype(np.float32)
    table["N"] = np.array(7, dtype=np.int64)                       # a persistent scalar, as BatchNormalization's counter
    table["ids"] = np.arange(10, dtype=np.int32)
    table["f64"] = rs.randn(3, 3)
    table["transposed"] = rs.randn(6, 5).astype(np.float32).T      # not contiguous in memory
    return table


@pytest.mark.parametrize("compression,shuffle", [(None, False), (4, False), (9, True)])
def test_hdf5_container_round_trip(tmp_path, compression, shuffle):
    """asr/hdf5_lite.py: groups + contiguous datasets (what model.save writes) and chunked + gzip datasets (what
    chainer.serializers.save_hdf5 writes through h5py with its default compression=4) read back bit for bit"""
    import numpy as np
    from asr import hdf5_lite
    table = _hdf5_table(np.random.RandomState(0))
    path = str(tmp_path / "t.hdf5")
    hdf5_lite.write(path, table, compression, shuffle)
    head = open(path, "rb").read(8)
    assert head == b"\x89HDF\r\n\x1a\n"
    back = hdf5_lite.read(path)
    assert set(back) == set(table)
    for k, v in table.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    try:
        import h5py                                               # where the real library exists it must agree both ways
    except ImportError:
        return
    with h5py.File(path, "r") as f:
        for k, v in table.items():
            assert np.array_equal(np.asarray(f[k]), v), k
    real = str(tmp_path / "real.hdf5")
    with h5py.File(real, "w") as f:
        for k, v in table.items():
            f.create_dataset(k, data=v, compression=(compression if v.ndim else None), shuffle=bool(shuffle and v.ndim))
    back = hdf5_lite.read(real)
    for k, v in table.items():
        assert np.array_equal(back[k], v), k


def test_hdf5_group_with_more_links_than_one_btree_node_takes(tmp_path):
    """Chainer's flattened `_module_*_link_*` names all land in the owner's group; past 2 LEAF_K x 2 INTERNAL_K = 256 links
    a single leaf B-tree node no longer holds the symbol-table nodes.  The writer adds levels: 700 links = 88 symbol-table nodes = 3
    leaf nodes (chained by sibling addresses, keys ascending) under one level-1 root; everything reads back, in name order."""
    import struct
    import numpy as np
    from asr import hdf5_lite
    table = {"g/_module_%03d_link_%d" % (i // 2, i % 2): np.full((2,), i, np.float32) for i in range(700)}
    table["top"] = np.arange(3, dtype=np.int32)
    raw = hdf5_lite.dumps(table)
    path = str(tmp_path / "many.hdf5")
    open(path, "wb").write(raw)
    back = hdf5_lite.read(path)
    assert list(back) == sorted(table, key=lambda s: [c.encode("utf-8") for c in s.split("/")])
    for k, v in table.items():
        assert np.array_equal(back[k], v), k
    # structure: the group's root node is at level 1 with 3 children; the leaves are chained left to right
    rd = hdf5_lite._Reader(raw)
    trees = [i for i in range(0, len(raw) - 4, 8) if raw[i:i + 4] == b"TREE" and raw[i + 4] == 0]
    roots = [i for i in trees if raw[i + 5] == 1]
    assert len(roots) == 1 and struct.unpack_from("<H", raw, roots[0] + 6)[0] == 3
    leaves = [struct.unpack_from("<Q", raw, roots[0] + 24 + 8 + 16 * i)[0] for i in range(3)]
    undef = 0xFFFFFFFFFFFFFFFF
    assert [struct.unpack_from("<QQ", raw, a + 8) for a in leaves] == [(undef, leaves[1]), (leaves[0], leaves[2]), (leaves[1], undef)]
    assert [struct.unpack_from("<H", raw, a + 6)[0] for a in leaves] == [32, 32, 24]
    try:
        import h5py
    except ImportError:
        return
    with h5py.File(path, "r") as f:
        assert len(f["g"]) == 700 and np.array_equal(np.asarray(f["g/_module_349_link_1"]), table["g/_module_349_link_1"])


def test_hdf5_reader_on_a_file_written_by_the_hdf5_library():
    """an independent producer: SciPy ships a MATLAB v7.3 file, i.e. HDF5 written by the HDF5 library itself (superblock behind a
    512-byte user block, version-1 object headers, local heap, B-tree, symbol-table node, old-style data layout message):
    `testdouble` = 0 : pi/4 : 2 pi"""
    import numpy as np
    import scipy.io
    from asr import hdf5_lite, serializers
    path = os.path.join(os.path.dirname(scipy.io.__file__), "matlab", "tests", "data", "testhdf5_7.4_GLNX86.mat")
    if not os.path.isfile(path):
        pytest.skip("SciPy's test data are not installed")
    assert serializers.sniff(path) == "hdf5"
    table = hdf5_lite.read(path)
    assert list(table) == ["testdouble"]
    a = table["testdouble"]
    assert a.dtype == np.float64 and a.shape == (9, 1)
    np.testing.assert_allclose(a[:, 0], np.arange(9) * np.pi / 4, rtol=0, atol=1e-15)


def test_hdf5_writer_output_is_stable(golden_dir):
    """the writer's bytes for a fixed table are pinned by a committed fixture (tests/golden/tiny.hdf5, written by
    tests/golden/make_golden.py --hdf5): a change of the on-disk structures cannot slip in unnoticed; the fixture also reads back"""
    import numpy as np
    from asr import hdf5_lite
    table = {"layer_0/W": np.arange(24, dtype=np.float32).reshape(2, 3, 4) / 8, "layer_0/b": np.array([1.5, -2.0], np.float32),
             "layer_5_0/W": np.ones((1, 2), np.float32), "N": np.array(3, np.int64)}
    path = os.path.join(golden_dir, "tiny.hdf5")
    want = open(path, "rb").read()
    assert hdf5_lite.dumps(table) == want
    back = hdf5_lite.read(path)
    for k, v in table.items():
        assert np.array_equal(back[k], v) and back[k].dtype == v.dtype


@pytest.mark.parametrize("arch,nconv,wn", [("zhang", 3, False), ("zhang", 6, False), ("zhang+fc_relu", 2, False), ("zhang+residual", 4, False),
                                           ("zhang+residual", 6, True), ("zhang+layernorm", 2, False), ("glu", 3, True),
                                           ("relu+layernorm", 2, False), ("relu+layernorm+residual", 3, False)])
def test_recipe_parameter_names_agree_with_the_oracles_bookkeeping(arch, nconv, wn):
    """asr.model.architectures.build_model and oracle/cnn.py both replay run/ctc/cnn/model.py's layer bookkeeping
    (``layer_%d`` / ``layer_%d_%d``, asr/nn/nn.py:304-320) -- independently; their parameter names must coincide"""
    from asr.model import cnn
    from asr.model.architectures import build_model
    from oracle import cnn as ocnn
    cfg = cnn.configure()
    cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers = 19, 3, 8, 12, nconv
    cfg.architecture, cfg.weightnorm = arch, wn
    model = build_model(cfg)
    mine = sorted({n.rsplit(".", 1)[0] for n, _ in model.named_parameters()})
    theirs = sorted({name for _, name, _ in ocnn.program(arch, cfg) if name is not None})
    assert mine == theirs


def test_every_function_layer_of_the_reference_has_a_mirror():
    """VERDICT r2 (missing 6): the names asr/nn/nn.py defines (:9-231) all exist in asr.nn with the reference's argument names"""
    import asr.nn as nn
    for name in ("ClippedReLU", "CReLU", "ELU", "HardSigmoid", "LeakyReLU", "LogSoftmax", "Maxout", "ReLU", "Sigmoid", "Softmax", "Softplus", "Tanh",
                 "AveragePooling2D", "AveragePoolingND", "MaxPooling2D", "MaxPoolingND", "SpatialPyramidPooling2D", "Unpooling2D", "UpSampling2D",
                 "BroadcastTo", "ExpandDims", "Flatten", "Reshape", "RollAxis", "Squeeze", "SwapAxes", "Tile", "Transpose", "Dropout", "GaussianNoise",
                 "Convolution2D", "LayerNormalization", "GLU", "Residual", "Stream", "Module", "Convolution1D", "SRU"):
        assert hasattr(nn, name), name
    assert nn.CReLU().axis == 1 and nn.Softmax(axis=1).axis == 1
    p = nn.Unpooling2D((2, 1), outsize=(9, 1))
    assert (p.ksize, p.stride, p.pad, p.outsize, p.cover_all) == ((2, 1), None, 0, (9, 1), True)
    u = nn.UpSampling2D("idx", (2, 1))
    assert u.indexes == "idx" and u.ksize == (2, 1)
    g = nn.GaussianNoise(0, 0.5)
    assert (g.mean, g.std) == (0, 0.5)
    assert callable(nn.LogSoftmax())


def test_environment_is_reloaded_on_sigusr1(tmp_path):
    """asr/training/environment.py:13-37 + its use in run/ctc/cnn/train.py:115-131: the script declares what can be steered, save()
    writes it, an edit of the file + SIGUSR1 re-reads it (only declared attributes, nested option objects walked) and calls back"""
    import io
    import json
    import signal
    from asr.training import Environment, Iteration
    from asr.data.augment import AugmentationOption
    seen = []
    keep = signal.getsignal(signal.SIGUSR1)
    try:
        env = Environment(str(tmp_path / "env.json"), lambda: seen.append((env.learning_rate, env.augmentation.add_noise)))
        env.learning_rate, env.momentum, env.augmentation = 1e-3, 0.9, AugmentationOption(change_speech_rate=True)
        env.save()
        on_disk = json.load(open(str(tmp_path / "env.json")))
        assert on_disk == {"learning_rate": 1e-3, "momentum": 0.9,
                           "augmentation": {"add_noise": False, "change_speech_rate": True, "change_vocal_tract": False}}
        on_disk["learning_rate"] = 5e-4
        on_disk["augmentation"]["add_noise"] = True
        on_disk["not_declared"] = 7                 # ignored: the file cannot add attributes
        json.dump(on_disk, open(str(tmp_path / "env.json"), "w"))
        os.kill(os.getpid(), signal.SIGUSR1)        # delivered to the main thread before the next byte code
        assert seen == [(5e-4, True)] and env.momentum == 0.9 and not hasattr(env, "not_declared")
        buf = io.StringIO()
        env.dump(buf)
        assert "learning_rate:\t0.0005" in buf.getvalue() and "add_noise:\tTrue" in buf.getvalue()
        open(str(tmp_path / "env.json"), "w").write("{ not json")
        with pytest.raises(AssertionError):
            env.load()
    finally:
        signal.signal(signal.SIGUSR1, keep)
    assert list(Iteration(3)) == [1, 2, 3]
