"""A dependency-free reader / writer for the subset of HDF5 that ``chainer.serializers.save_hdf5`` / ``load_hdf5`` use
(reference: asr/model/cnn.py:51-63, asr/nn/nn.py:394-406 -- ``h5py`` is not available in this image).

What Chainer's serialiser emits through h5py with the library's default ("earliest") file format, and therefore what this
module covers:

* superblock version 0, 8-byte offsets and lengths; groups as version-1 object headers with a Symbol Table message, a
  version-1 B-tree of symbol-table nodes (SNOD) and a local heap for the link names;
* datasets as version-1 object headers with Dataspace (version 1), Datatype (version 1: fixed-point and IEEE floating point,
  either byte order), Fill Value and Data Layout (version 3) messages; layout *contiguous* (what this writer emits) or *chunked*
  with the deflate (gzip) and shuffle filters (what ``save_hdf5(..., compression=4)``, Chainer's default, makes h5py write): chunk
  index = version-1 B-tree of type 1; compact layout and header continuation blocks are read as well.

The writer produces the plain form of the same structures -- contiguous little-endian datasets, no filters, no attributes --
which ``h5py`` / ``chainer.serializers.load_hdf5`` read like any other HDF5 file.  Not covered (never produced by the
reference's path): version-2 object headers / fractal heaps (``libver='latest'``), attributes, variable-length and compound
types, external storage.  Field layouts follow the HDF5 File Format Specification, version 1.1 (superblock 0) -- the reader is
also exercised on an HDF5 file written by the real library that ships with SciPy's test data (tests/test_host_logic.py).
"""
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16            # symbol-table node holds 2 * LEAF_K entries, a B-tree node 2 * INTERNAL_K children
HEAP_FREE_NULL = 1                    # H5HL_FREE_NULL: "no further free block" in a local heap's free list

MSG_NIL, MSG_DATASPACE, MSG_DATATYPE, MSG_FILL_OLD, MSG_FILL, MSG_LAYOUT, MSG_FILTERS = 0x0, 0x1, 0x3, 0x4, 0x5, 0x8, 0xB
MSG_CONTINUATION, MSG_SYMBOL_TABLE = 0x10, 0x11


class Hdf5Error(ValueError):
    pass


# ====================================================================================================================== reader
class _Reader(object):
    def __init__(self, data):
        self.d = data
        self.base = None
        pos = 0
        while pos + 8 <= len(data):        # the superblock sits at 0, 512, 1024, 2048, ... (a user block may precede it)
            if data[pos:pos + 8] == SIGNATURE:
                self.base = pos
                break
            pos = 512 if pos == 0 else pos * 2
        if self.base is None:
            raise Hdf5Error("not an HDF5 file (no superblock signature)")
        sb = self.base
        version = data[sb + 8]
        if version not in (0, 1):
            raise Hdf5Error("superblock version %d (libver='latest' files) is outside this reader's subset" % version)
        self.so, self.sl = data[sb + 13], data[sb + 14]
        if (self.so, self.sl) != (8, 8):
            raise Hdf5Error("only 8-byte offsets / lengths are supported")
        off = sb + 24 + (4 if version == 1 else 0)
        self.base_address = self.u(off, 8, absolute=True)
        self.root_ste = off + 32               # base, free-space, end-of-file, driver-info addresses precede it

    # -- primitive access (file addresses are relative to the base address)
    def u(self, off, n, absolute=False):
        return int.from_bytes(self.d[off:off + n], "little")

    def at(self, addr):
        if addr == UNDEF:
            raise Hdf5Error("undefined address followed")
        return addr + self.base_address

    # -- groups
    def heap_name(self, heap_addr, name_off):
        h = self.at(heap_addr)
        if self.d[h:h + 4] != b"HEAP":
            raise Hdf5Error("local heap signature missing")
        seg = self.at(self.u(h + 24, 8))
        start = seg + name_off
        end = self.d.index(b"\0", start)
        return self.d[start:end].decode("utf-8")

    def group_entries(self, btree_addr, heap_addr):
        """[(name, object header address)] of a group, in B-tree (= name) order"""
        out = []
        n = self.at(btree_addr)
        if self.d[n:n + 4] != b"TREE":
            raise Hdf5Error("B-tree signature missing")
        node_type, level, used = self.d[n + 4], self.d[n + 5], self.u(n + 6, 2)
        if node_type != 0:
            raise Hdf5Error("group B-tree expected")
        p = n + 24
        for i in range(used):
            child = self.u(p + 8, 8)           # key i (8 bytes), child i (8 bytes), ...
            p += 16
            if level > 0:
                out += self.group_entries(child, heap_addr)
            else:
                s = self.at(child)
                if self.d[s:s + 4] != b"SNOD":
                    raise Hdf5Error("symbol table node signature missing")
                for k in range(self.u(s + 6, 2)):
                    e = s + 8 + 40 * k
                    out.append((self.heap_name(heap_addr, self.u(e, 8)), self.u(e + 8, 8)))
        return out

    # -- object headers
    def messages(self, addr):
        h = self.at(addr)
        if self.d[h] != 1:
            raise Hdf5Error("object header version %d is outside this reader's subset" % self.d[h])
        nmsg, size = self.u(h + 2, 2), self.u(h + 8, 4)
        blocks = [(h + 16, size)]
        msgs = []
        while blocks and len(msgs) < nmsg:
            p, left = blocks.pop(0)
            while left >= 8 and len(msgs) < nmsg:
                mtype, msize, flags = self.u(p, 2), self.u(p + 2, 2), self.d[p + 4]
                body = p + 8
                msgs.append((mtype, body, msize, flags))
                if mtype == MSG_CONTINUATION:
                    blocks.append((self.at(self.u(body, 8)), self.u(body + 8, 8)))
                p += 8 + msize
                left -= 8 + msize
        return msgs

    def walk(self, addr, prefix, table):
        msgs = self.messages(addr)
        st = [m for m in msgs if m[0] == MSG_SYMBOL_TABLE]
        if st:
            body = st[0][1]
            for name, child in self.group_entries(self.u(body, 8), self.u(body + 8, 8)):
                self.walk(child, prefix + [name], table)
            return
        if any(m[0] == MSG_LAYOUT for m in msgs):
            table["/".join(prefix)] = self.dataset(msgs)

    # -- datasets
    def datatype(self, body):
        cls, ver = self.d[body] & 0x0F, self.d[body] >> 4
        bits = self.u(body + 1, 3)
        size = self.u(body + 4, 4)
        if ver not in (1, 2, 3):
            raise Hdf5Error("datatype version %d" % ver)
        order = ">" if (bits & 1) else "<"
        if cls == 0:
            kind = "i" if (bits & 0x08) else "u"
        elif cls == 1:
            kind = "f"
        else:
            raise Hdf5Error("datatype class %d (only fixed-point and floating-point numbers are supported)" % cls)
        if size not in ((1, 2, 4, 8) if cls == 0 else (2, 4, 8)):
            raise Hdf5Error("unsupported element size %d" % size)
        return np.dtype("%s%s%d" % (order if size > 1 else "|", kind, size))

    def dataspace(self, body):
        ver, rank, flags = self.d[body], self.d[body + 1], self.d[body + 2]
        if ver == 1:
            p = body + 8
        elif ver == 2:
            if self.d[body + 3] == 2:
                raise Hdf5Error("null dataspace")
            p = body + 4
        else:
            raise Hdf5Error("dataspace version %d" % ver)
        return tuple(self.u(p + 8 * i, 8) for i in range(rank))

    def filters(self, body):
        ver, n = self.d[body], self.d[body + 1]
        if ver not in (1, 2):
            raise Hdf5Error("filter pipeline version %d" % ver)
        p = body + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid = self.u(p, 2)
            if ver == 1 or fid >= 256:
                name_len = self.u(p + 2, 2)
                p += 2
            else:
                name_len = 0
            ncd = self.u(p + 4, 2)
            p += 6
            p += (name_len + 7) // 8 * 8 if ver == 1 else name_len
            cd = [self.u(p + 4 * i, 4) for i in range(ncd)]
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    def chunks(self, btree_addr, rank):
        """[(offsets, file address, stored size, filter mask)] from a version-1 B-tree of type 1"""
        n = self.at(btree_addr)
        if self.d[n:n + 4] != b"TREE" or self.d[n + 4] != 1:
            raise Hdf5Error("chunk B-tree expected")
        level, used = self.d[n + 5], self.u(n + 6, 2)
        key = 8 + 8 * (rank + 1)
        p = n + 24
        out = []
        for _ in range(used):
            size, mask = self.u(p, 4), self.u(p + 4, 4)
            offs = tuple(self.u(p + 8 + 8 * i, 8) for i in range(rank))
            child = self.u(p + key, 8)
            p += key + 8
            if level > 0:
                out += self.chunks(child, rank)
            else:
                out.append((offs, child, size, mask))
        return out

    def dataset(self, msgs):
        get = lambda t: [m for m in msgs if m[0] == t]
        dt = self.datatype(get(MSG_DATATYPE)[0][1])
        shape = self.dataspace(get(MSG_DATASPACE)[0][1])
        body = get(MSG_LAYOUT)[0][1]
        ver, count = self.d[body], int(np.prod(shape, dtype=np.int64)) if shape else 1
        if ver in (1, 2):       # the form library versions before 1.6.3 wrote: rank, class, 5 reserved bytes, address, 32-bit sizes
            rank, cls = self.d[body + 1], self.d[body + 2]
            if cls != 1:
                raise Hdf5Error("data layout version %d, class %d" % (ver, cls))
            addr = self.u(body + 8, 8)
            if addr == UNDEF:
                return np.zeros(shape, dtype=dt.newbyteorder("="))
            raw = self.d[self.at(addr):self.at(addr) + count * dt.itemsize]
            return np.frombuffer(raw, dtype=dt, count=count).reshape(shape).astype(dt.newbyteorder("="))
        if ver != 3:
            raise Hdf5Error("data layout version %d" % ver)
        cls = self.d[body + 1]
        if cls == 0:            # compact: the data sit in the message
            n = self.u(body + 2, 2)
            raw = self.d[body + 4:body + 4 + n]
        elif cls == 1:          # contiguous
            addr, n = self.u(body + 2, 8), self.u(body + 10, 8)
            raw = b"" if addr == UNDEF else self.d[self.at(addr):self.at(addr) + n]
            if addr == UNDEF:
                return np.zeros(shape, dtype=dt.newbyteorder("="))
        elif cls == 2:          # chunked
            rank = self.d[body + 2] - 1
            btree = self.u(body + 3, 8)
            cdims = tuple(self.u(body + 11 + 4 * i, 4) for i in range(rank))
            pipeline = self.filters(get(MSG_FILTERS)[0][1]) if get(MSG_FILTERS) else []
            out = np.zeros(shape, dtype=dt)
            if btree != UNDEF:
                for offs, addr, size, mask in self.chunks(btree, rank):
                    buf = self.d[self.at(addr):self.at(addr) + size]
                    for k, (fid, cd) in reversed(list(enumerate(pipeline))):
                        if mask & (1 << k):
                            continue
                        if fid == 1:
                            buf = zlib.decompress(buf)
                        elif fid == 2:          # shuffle: bytes of equal significance were stored together
                            es = cd[0] if cd else dt.itemsize
                            a = np.frombuffer(buf, dtype=np.uint8)
                            m = a.size // es
                            buf = a[:m * es].reshape(es, m).T.tobytes() + a[m * es:].tobytes()
                        elif fid == 3:          # fletcher32 checksum trails the chunk
                            buf = buf[:-4]
                        else:
                            raise Hdf5Error("filter %d is not supported" % fid)
                    chunk = np.frombuffer(buf, dtype=dt, count=int(np.prod(cdims))).reshape(cdims)
                    sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                    out[sel] = chunk[tuple(slice(0, s.stop - s.start) for s in sel)]
            return out.astype(dt.newbyteorder("="))
        else:
            raise Hdf5Error("data layout class %d" % cls)
        return np.frombuffer(raw, dtype=dt, count=count).reshape(shape).astype(dt.newbyteorder("="))

    def table(self):
        ste = self.root_ste
        table = {}
        self.walk(self.u(ste + 8, 8), [], table)
        return table


def read(filename):
    """{path: numpy array} of every dataset in the file (paths are '/'-joined link names, e.g. 'layer_0/W')"""
    with open(filename, "rb") as f:
        return _Reader(f.read()).table()


# ====================================================================================================================== writer
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _message(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _object_header(messages):
    data = b"".join(messages)
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(data)) + data


def _datatype_message(dt):
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (2, 4, 8):
        exp_bits, mant_bits = {2: (5, 10), 4: (8, 23), 8: (11, 52)}[dt.itemsize]
        bits = 0x20 | ((dt.itemsize * 8 - 1) << 8)              # little-endian, mantissa normalisation 2 (implied msb), sign bit
        props = struct.pack("<HHBBBBI", 0, dt.itemsize * 8, mant_bits, exp_bits, 0, mant_bits, (1 << (exp_bits - 1)) - 1)
        head = struct.pack("<B3sI", 0x11, bits.to_bytes(3, "little"), dt.itemsize)
    elif dt.kind in "iu" and dt.itemsize in (1, 2, 4, 8):
        bits = 0x08 if dt.kind == "i" else 0x00
        props = struct.pack("<HH", 0, dt.itemsize * 8)
        head = struct.pack("<B3sI", 0x10, bits.to_bytes(3, "little"), dt.itemsize)
    else:
        raise Hdf5Error("dtype %s cannot be written" % dt)
    return _message(MSG_DATATYPE, head + props, flags=1)


class _Writer(object):
    def __init__(self, compression=None, shuffle=False):
        self.buf = bytearray(96)            # the superblock is filled in last
        self.compression, self.shuffle = compression, shuffle

    def alloc(self, data, align=8):
        self.buf += b"\0" * (-len(self.buf) % align)
        addr = len(self.buf)
        self.buf += data
        return addr

    def dataset(self, a, compression=None, shuffle=False):
        a = np.asarray(a)
        if a.dtype.byteorder == ">" or (a.dtype.byteorder == "=" and not np.little_endian):
            a = a.astype(a.dtype.newbyteorder("<"))
        raw = a.tobytes(order="C")              # (np.ascontiguousarray would turn a scalar into a 1-element vector)
        if compression is not None and a.ndim > 0 and 0 < len(raw) < (1 << 32):
            return self.chunked_dataset(a, raw, int(compression), shuffle)
        addr = self.alloc(raw) if raw else UNDEF
        space = struct.pack("<BBB5x", 1, a.ndim, 0) + b"".join(struct.pack("<Q", n) for n in a.shape)
        msgs = [_message(MSG_DATASPACE, space),
                _datatype_message(a.dtype),
                _message(MSG_FILL, struct.pack("<BBBB", 2, 2, 2, 0)),       # version 2, allocate late, write if set, undefined
                _message(MSG_LAYOUT, struct.pack("<BBQQ", 3, 1, addr, len(raw)))]
        return self.alloc(_object_header(msgs))

    def chunked_dataset(self, a, raw, level, shuffle):
        """what h5py writes for create_dataset(..., compression=level) (chainer.serializers.save_hdf5's default is gzip level 4):
        chunked layout, filter pipeline [shuffle,] deflate, a version-1 B-tree of type 1 as the chunk index -- here with ONE chunk
        that spans the array"""
        es, rank = a.dtype.itemsize, a.ndim
        filters = []
        buf = raw
        if shuffle:
            m = len(buf) // es
            buf = np.frombuffer(buf, dtype=np.uint8).reshape(m, es).T.tobytes()
            filters.append((2, [es]))
        buf = zlib.compress(buf, level)
        filters.append((1, [level]))
        chunk_addr = self.alloc(buf)
        key = lambda size, offs: struct.pack("<II", size, 0) + b"".join(struct.pack("<Q", o) for o in offs) + struct.pack("<Q", 0)
        node = b"TREE" + struct.pack("<BBHQQ", 1, 0, 1, UNDEF, UNDEF)
        node += key(len(buf), [0] * rank) + struct.pack("<Q", chunk_addr) + key(0, list(a.shape))
        node += b"\0" * (24 + (2 * 32 + 1) * (16 + 8 * rank) + 2 * 32 * 8 - len(node))        # full node for the default K = 32
        btree = self.alloc(node)
        pipeline = struct.pack("<BB6x", 1, len(filters))
        for fid, cd in filters:
            pipeline += struct.pack("<HHHH", fid, 0, 1, len(cd)) + b"".join(struct.pack("<I", v) for v in cd)
            if len(cd) % 2:
                pipeline += b"\0" * 4
        layout = struct.pack("<BBBQ", 3, 2, rank + 1, btree) + b"".join(struct.pack("<I", n) for n in a.shape) + struct.pack("<I", es)
        space = struct.pack("<BBB5x", 1, rank, 0) + b"".join(struct.pack("<Q", n) for n in a.shape)
        msgs = [_message(MSG_DATASPACE, space), _datatype_message(a.dtype), _message(MSG_FILL, struct.pack("<BBBB", 2, 3, 2, 0)),
                _message(MSG_FILTERS, pipeline, flags=1), _message(MSG_LAYOUT, layout)]
        return self.alloc(_object_header(msgs))

    def group(self, tree):
        """tree: {name: ndarray or dict}; returns (object header address, B-tree address, heap address)"""
        names = sorted(tree, key=lambda s: s.encode("utf-8"))          # the library orders links by strcmp
        children = {}
        for name in names:
            v = tree[name]
            children[name] = self.group(v)[0] if isinstance(v, dict) else self.dataset(v, self.compression, self.shuffle)
        # local heap: offset 0 holds the empty string, names are NUL-terminated and 8-byte aligned, one trailing free block
        seg = bytearray(8)
        offs = {}
        for name in names:
            offs[name] = len(seg)
            seg += _pad8(name.encode("utf-8") + b"\0")
        free_at = len(seg)
        seg += struct.pack("<QQ", HEAP_FREE_NULL, 16 + 64) + b"\0" * 64
        seg_addr = self.alloc(bytes(seg))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(seg), free_at, seg_addr))
        # symbol-table nodes of at most 2 * LEAF_K entries, each allocated at its full size
        nodes = [names[i:i + 2 * LEAF_K] for i in range(0, len(names), 2 * LEAF_K)] or [[]]
        snods = []
        for part in nodes:
            body = b"SNOD" + struct.pack("<BxH", 1, len(part))
            for name in part:
                body += struct.pack("<QQII16x", offs[name], children[name], 0, 0)
            body += b"\0" * (8 + 40 * 2 * LEAF_K - len(body))
            snods.append(self.alloc(body))
        # version-1 B-tree over the symbol-table nodes: key 0 = the empty string, key i + 1 = the largest name below child i.  A node
        # takes 2 * INTERNAL_K children; more than that (Chainer's flattened `_module_*_link_*` names all land in the owner's group,
        # so a deep recipe can pass 2 * LEAF_K * 2 * INTERNAL_K = 256 links) adds levels, nodes of one level chained by their sibling
        # addresses as the library writes them
        level, below = 0, [(addr, offs[part[-1]]) for part, addr in zip(nodes, snods) if part]
        while True:
            runs = [below[i:i + 2 * INTERNAL_K] for i in range(0, len(below), 2 * INTERNAL_K)] or [[]]
            addrs = []
            for run in runs:
                node = b"TREE" + struct.pack("<BBHQQ", 0, level, len(run), UNDEF, UNDEF) + struct.pack("<Q", 0)
                for child, last in run:
                    node += struct.pack("<QQ", child, last)
                node += b"\0" * (24 + 8 * (2 * INTERNAL_K + 1) + 8 * 2 * INTERNAL_K - len(node))
                addrs.append(self.alloc(node))
            for i, a in enumerate(addrs):           # left / right sibling
                if i > 0:
                    self.buf[a + 8:a + 16] = struct.pack("<Q", addrs[i - 1])
                if i + 1 < len(addrs):
                    self.buf[a + 16:a + 24] = struct.pack("<Q", addrs[i + 1])
            # (key 0 of a node that is not the leftmost of its level = the largest name to its left, as the library keeps it)
            for i in range(1, len(addrs)):
                self.buf[addrs[i] + 24:addrs[i] + 32] = struct.pack("<Q", runs[i - 1][-1][1])
            if len(addrs) == 1:
                btree_addr = addrs[0]
                break
            below = [(a, run[-1][1]) for a, run in zip(addrs, runs)]
            level += 1
        header = self.alloc(_object_header([_message(MSG_SYMBOL_TABLE, struct.pack("<QQ", btree_addr, heap_addr))]))
        return header, btree_addr, heap_addr

    def finish(self, root):
        header, btree, heap = root
        eof = len(self.buf)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, header, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == 96
        self.buf[0:96] = sb
        return bytes(self.buf)


def _nest(table):
    tree = {}
    for path, a in table.items():
        parts = [p for p in path.split("/") if p]
        if not parts:
            raise Hdf5Error("empty dataset path")
        node = tree
        for p in parts[:-1]:
            node = node.setdefault(p, {})
            if not isinstance(node, dict):
                raise Hdf5Error("%s is both a dataset and a group" % p)
        if parts[-1] in node:
            raise Hdf5Error("duplicate path %s" % path)
        node[parts[-1]] = np.asarray(a)
    return tree


def dumps(table, compression=None, shuffle=False):
    """the bytes of an HDF5 file holding {path: array} (groups made from the '/'-separated path components).
    compression: None = contiguous datasets; 0..9 = chunked + gzip at that level, as chainer.serializers.save_hdf5 asks h5py for
    (its default is 4)"""
    w = _Writer(compression, shuffle)
    return w.finish(w.group(_nest(table)))


def write(filename, table, compression=None, shuffle=False):
    with open(filename, "wb") as f:
        f.write(dumps(table, compression, shuffle))
