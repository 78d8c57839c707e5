"""Checkpoints with the reference's parameter naming.

The reference saves a model with ``chainer.serializers.save_hdf5(tmp, self)`` + ``os.rename`` and loads it with
``load_hdf5`` (asr/model/cnn.py:51-63, asr/nn/nn.py:394-406).  Chainer's serialisers walk the link tree and store every
parameter / persistent value under the path of attribute names that leads to it: ``layer_0/W``, ``layer_5_0/b``
(Stream.layer, asr/nn/nn.py:304-320), ``_sequential_3/W`` (Module.add, :341-353), and -- because Module.__setattr__
registers the links of a sub-module on the OWNER under flattened names (:355-392) -- ``_module_<ns>_sequential_<i>/W`` and
``_module_<ns>_link_<name>/W``.  ``chainer_name`` reproduces those paths from this package's module tree, so a file written
here has the reference's keys and a reference-trained file is found key by key.

Container.  ``model.save("model.hdf5")`` writes real HDF5 -- groups = links, datasets = parameters, exactly the tree
``chainer.serializers.save_hdf5`` builds -- and ``load`` reads it back, including files h5py wrote with Chainer's default gzip
compression.  ``h5py`` is used when it is importable; this image does not have it, so the file format itself is implemented in
``asr/hdf5_lite.py`` (the subset the serialiser uses: classic groups, contiguous or chunked + deflate datasets of numbers).
A file name ending in ``.npz`` selects the layout of ``chainer.serializers.save_npz`` instead.  ``load`` recognises the
container from the file's first bytes (HDF5 signature, zip with .npy members, or a round-1 ``torch.save`` archive).

Layout differences that a checkpoint must not see:
* a layer that consumes the merged (channel, height) axis of an image -- ``reshape(out, (B, -1, T))``,
  run/ctc/sru/model.py:114 -- sees its input columns in (c, h) order in the reference and in (h, c) order here (DESIGN.md
  section 2).  Models declare those parameters (``column_permutations()`` -> {name: (C, H)}); they are stored in the
  reference's order and permuted back on load.
* lazily sized parameters (LayerNormalization(None), Convolution1D(None, ..), BiGRU(None, ..), weight-norm g / b) are empty
  until the first forward pass; loading sizes them from the file, as Chainer's deserialiser initialises uninitialised
  parameters from the stored array.
"""
import io
import os
import uuid
import zipfile

import numpy as np
import torch

_HDF5_MAGIC = b"\x89HDF\r\n\x1a\n"


def have_h5py():
    try:
        import h5py  # noqa: F401
        return True
    except ImportError:
        return False


# ---------------------------------------------------------------------------------------------- names
def chainer_name(model, torch_name):
    """``conv_blocks._sequential_0.W`` -> ``_module_conv_blocks_sequential_0/W``; ``layer_5_0.b`` -> ``layer_5_0/b``"""
    from .nn.nn import Module
    parts = torch_name.split(".")
    obj, i, ns = model, 0, []
    while i < len(parts) - 1 and isinstance(getattr(obj, parts[i], None), Module):
        ns.append(parts[i])
        obj = getattr(obj, parts[i])
        i += 1
    if ns and i < len(parts) - 1:
        head = parts[i] if parts[i].startswith("_sequential_") else "_link_" + parts[i]
        first = "_module_" + "_".join(ns) + head
        rest = parts[i + 1:]
    else:
        first, rest = parts[0], parts[1:]
    return "/".join([first] + rest)


def _entries(model):
    """(chainer path, tensor, is_parameter) for every parameter and persistent buffer"""
    out = []
    for name, p in model.named_parameters():
        out.append((chainer_name(model, name), name, p))
    for name, b in model.named_buffers():
        out.append((chainer_name(model, name), name, b))
    return out


def _permutations(model):
    fn = getattr(model, "column_permutations", None)
    return dict(fn()) if fn is not None else {}


def _to_reference_columns(a, C, H):
    """last axis in this package's (h, c) order -> the reference's (c, h) order"""
    lead = a.shape[:-1]
    return np.ascontiguousarray(a.reshape(lead + (H, C)).swapaxes(-1, -2).reshape(lead + (C * H,)))


def _from_reference_columns(a, C, H):
    lead = a.shape[:-1]
    return np.ascontiguousarray(a.reshape(lead + (C, H)).swapaxes(-1, -2).reshape(lead + (H * C,)))


def to_table(model):
    """{chainer path: float32 / int numpy array in the reference's conventions}; uninitialised parameters are left out"""
    perm = _permutations(model)
    table = {}
    for path, name, t in _entries(model):
        if t.numel() == 0:
            continue
        a = t.detach().cpu().numpy()
        if name in perm:
            C, H = perm[name]
            a = _to_reference_columns(a, C, H)
        table[path] = a
    return table


def from_table(model, table, strict=True, reference_layout=True):
    """copy a {chainer path: array} table into the model; returns the list of paths of the file that were not used.
    strict: a file that lacks an entry of the model is refused BEFORE anything is copied (the model stays as it was)."""
    from .link import bump_weight_epoch
    perm = _permutations(model) if reference_layout else {}
    used, missing = set(), []
    if strict:
        lacking = [path for path, _, _ in _entries(model) if path not in table]
        if lacking:
            raise KeyError("checkpoint lacks %d entries, e.g. %s" % (len(lacking), lacking[:3]))
    with torch.no_grad():
        for path, name, t in _entries(model):
            if path not in table:
                missing.append(path)
                continue
            a = np.asarray(table[path])
            if name in perm:
                C, H = perm[name]
                a = _from_reference_columns(a, C, H)
            src = torch.from_numpy(np.ascontiguousarray(a)).to(t.dtype)
            if t.numel() == 0 and src.numel() > 0:
                t.data = torch.empty(src.shape, dtype=t.dtype, device=t.device)     # lazily sized: take the file's shape
            if tuple(t.shape) != tuple(src.shape):
                raise ValueError("checkpoint entry %s has shape %s, the model expects %s" % (path, tuple(src.shape), tuple(t.shape)))
            t.data.copy_(src)
            used.add(path)
    bump_weight_epoch()
    if strict and missing:
        raise KeyError("checkpoint lacks %d entries, e.g. %s" % (len(missing), missing[:3]))
    return sorted(set(table.keys()) - used)


# ---------------------------------------------------------------------------------------------- containers
def save_npz(filename, model):
    """chainer.serializers.save_npz layout: one .npy member per path"""
    buf = io.BytesIO()
    np.savez(buf, **to_table(model))
    with open(filename, "wb") as f:
        f.write(buf.getvalue())


def load_npz(filename, model, strict=True):
    with np.load(filename) as z:
        return from_table(model, {k: z[k] for k in z.files}, strict)


def save_hdf5(filename, model):
    """chainer.serializers.save_hdf5 layout (groups = links, datasets = parameters): through h5py where it exists, else through
    this package's own writer of the same format (asr/hdf5_lite.py)"""
    table = to_table(model)
    if have_h5py():
        import h5py
        with h5py.File(filename, "w") as f:
            for path, a in table.items():
                f.create_dataset(path, data=a)
        return
    from . import hdf5_lite
    hdf5_lite.write(filename, table)


def read_hdf5_table(filename):
    if have_h5py():
        import h5py
        table = {}
        with h5py.File(filename, "r") as f:
            def visit(name, obj):
                if isinstance(obj, h5py.Dataset):
                    table[name] = np.asarray(obj)
            f.visititems(visit)
        return table
    from . import hdf5_lite
    return hdf5_lite.read(filename)


def load_hdf5(filename, model, strict=True):
    return from_table(model, read_hdf5_table(filename), strict)


def sniff(filename):
    with open(filename, "rb") as f:
        head = f.read(8)
        if head != _HDF5_MAGIC:             # an HDF5 superblock may sit behind a user block of 512, 1024, ... bytes
            pos = 512
            while True:
                f.seek(pos)
                probe = f.read(8)
                if len(probe) < 8:
                    break
                if probe == _HDF5_MAGIC:
                    head = probe
                    break
                pos *= 2
                if pos > (1 << 20):
                    break
    if head == _HDF5_MAGIC:
        return "hdf5"
    if head[:2] == b"PK":
        with zipfile.ZipFile(filename) as z:
            names = z.namelist()
        return "npz" if any(n.endswith(".npy") for n in names) else "torch"
    return "unknown"


def save(filename, model):
    """what ``model.save(filename)`` does: HDF5, the reference's format (a name ending in .npz: the npz container); written to a
    temporary name and renamed (asr/model/cnn.py:51-56)"""
    tmp = filename + "." + str(uuid.uuid4())
    if filename.endswith(".npz"):
        save_npz(tmp, model)
    else:
        save_hdf5(tmp, model)
    if os.path.isfile(filename):
        os.remove(filename)
    os.rename(tmp, filename)


def load(filename, model, strict=True):
    kind = sniff(filename)
    if kind == "npz":
        return load_npz(filename, model, strict)
    if kind == "hdf5":
        return load_hdf5(filename, model, strict)
    if kind == "torch":         # round-1 checkpoints: a torch.save'd state_dict with this package's own names
        state = torch.load(filename, map_location="cpu")
        table = {chainer_name(model, k): v.numpy() for k, v in state.items()}
        return from_table(model, table, strict, reference_layout=False)
    raise ValueError("unrecognised checkpoint container: %s" % filename)
