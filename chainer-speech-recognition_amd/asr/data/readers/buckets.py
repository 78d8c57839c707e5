"""Length-bucketed corpus on disk (reference: asr/data/readers/buckets.py:26-209; files written by
tools/preprocess/bucket.py:55-74).

Layout: ``<data_path>/signal/<bucket>_<piece>_<count>.bucket`` and the same name under ``sentence/``; each file is a
pickled list (int16 waveforms / transcriptions) of ``count`` utterances whose length falls into bucket ``bucket``
(``bucket_split_sec`` seconds per bucket).  Every piece is split once into a development head and a training tail by a
seeded shuffle -- the same draws in the same order as the reference, so a given seed selects the same utterances.

Host-side file handling only; the minibatches it yields are turned into features on the GPU by the Loader / Processor.
"""
import math
import os
import pickle
import re
from collections import OrderedDict

import numpy as np

_NAME = re.compile(r"([0-9]+)_([0-9]+)_([0-9]+)\.bucket")


class _Piece(object):
    """one pickled file pair and its train / dev index split"""
    __slots__ = ("count", "train", "dev", "updates", "signals", "sentences")

    def __init__(self, count):
        self.count, self.updates = count, 0
        self.train = self.dev = None
        self.signals = self.sentences = None


class Reader(object):
    def __init__(self, data_path, buckets_limit=None, buckets_cache_size=200, dev_split=0.01, seed=0, sampling_rate=16000,
                 bucket_split_sec=0.5):
        self.data_path, self.buckets_limit = data_path, buckets_limit
        self.buckets_cache_size, self.dev_split, self.bucket_split_sec = buckets_cache_size, dev_split, bucket_split_sec
        found = {}
        names_signal = os.listdir(os.path.join(data_path, "signal"))
        names_sentence = os.listdir(os.path.join(data_path, "sentence"))
        if not names_signal or len(names_signal) != len(names_sentence):
            raise Exception("Run preprocess/buckets.py before starting training.")
        for name in names_signal:
            m = _NAME.match(name)
            if m:
                found[(int(m.group(1)), int(m.group(2)))] = int(m.group(3))
        n_buckets = 1 + max(b for b, _ in found) if found else 0
        table = []
        for b in range(n_buckets):
            n_pieces = 1 + max([p for bb, p in found if bb == b], default=-1)
            table.append([_Piece(found.get((b, p), 0)) for p in range(n_pieces)])
        if buckets_limit is not None:
            table = table[:buckets_limit]
        # the split: one shuffle per piece, buckets then pieces in order, from a freshly seeded global NumPy stream
        np.random.seed(seed)
        for pieces in table:
            for piece in pieces:
                order = np.arange(piece.count)
                np.random.shuffle(order)
                n_dev = int(piece.count * dev_split)
                piece.train = order[n_dev:]
                piece.dev = order[:n_dev] if n_dev else []
        self._table = table
        self._cached = OrderedDict()
        self.total_buckets = len(table)
        self.total_pieces = sum(len(p) for p in table)
        self.bucket_distribution = np.asarray([len(p) for p in table]) / max(self.total_pieces, 1)

    # -- the reference's list-of-lists views ---------------------------------------------------------------------
    @property
    def buckets_indices_train(self):
        return [[piece.train for piece in pieces] for pieces in self._table]

    @property
    def buckets_indices_dev(self):
        return [[piece.dev for piece in pieces] for pieces in self._table]

    @property
    def buckets_num_data(self):
        return [[piece.count for piece in pieces] for pieces in self._table]

    @property
    def buckets_num_updates(self):
        return [[piece.updates for piece in pieces] for pieces in self._table]

    @property
    def buckets_num_pieces(self):
        return [len(pieces) for pieces in self._table]

    def get_num_buckets(self):
        return len(self._table)

    # -- file access with a bounded cache --------------------------------------------------------------------------
    def _read(self, kind, bucket_id, piece_id):
        piece = self._table[bucket_id][piece_id]
        name = "{}_{}_{}.bucket".format(bucket_id, piece_id, piece.count)
        with open(os.path.join(self.data_path, kind, name), "rb") as fp:
            return pickle.load(fp)

    def get_signals_by_bucket_and_piece(self, bucket_id, piece_id):
        piece = self._table[bucket_id][piece_id]
        if piece.signals is None:
            piece.signals = self._read("signal", bucket_id, piece_id)
        if self.buckets_cache_size > 0:         # keep at most that many pieces' waveforms in memory, oldest first out
            self._cached[(bucket_id, piece_id)] = True
            self._cached.move_to_end((bucket_id, piece_id))
            while len(self._cached) > self.buckets_cache_size:
                (b, p), _ = self._cached.popitem(last=False)
                self._table[b][p].signals = None
                self._table[b][p].sentences = None
        return piece.signals

    def get_sentences_by_bucket_and_piece(self, bucket_id, piece_id):
        piece = self._table[bucket_id][piece_id]
        if piece.sentences is None:
            piece.sentences = self._read("sentence", bucket_id, piece_id)
        return piece.sentences

    def increment_num_updates(self, bucket_id, piece_id):
        self._table[bucket_id][piece_id].updates += 1

    # -- sampling ----------------------------------------------------------------------------------------------------
    def sample_minibatch(self, batchsizes):
        """a random piece of a bucket drawn by piece count; its training indices are reshuffled and the first
        batchsizes[bucket] taken (asr/data/readers/buckets.py:139-160)"""
        bucket_id = np.random.choice(np.arange(len(self._table)), size=1, p=self.bucket_distribution)[0]
        piece_id = np.random.choice(np.arange(len(self._table[bucket_id])), size=1)[0]
        signals = self.get_signals_by_bucket_and_piece(bucket_id, piece_id)
        sentences = self.get_sentences_by_bucket_and_piece(bucket_id, piece_id)
        self.increment_num_updates(bucket_id, piece_id)
        order = self._table[bucket_id][piece_id].train
        np.random.shuffle(order)
        chosen = order[:min(batchsizes[bucket_id], len(order))]
        return [(signals[i], sentences[i]) for i in chosen], bucket_id, piece_id

    # -- bookkeeping -------------------------------------------------------------------------------------------------
    def _iterations(self, which, batchsizes):
        total = 0
        for pieces, batchsize in zip(self._table, batchsizes):
            total += sum(int(math.ceil(len(getattr(piece, which)) / batchsize)) for piece in pieces)
        return total

    def calculate_total_training_iterations_with_batchsizes(self, batchsizes):
        return self._iterations("train", batchsizes)

    def calculate_total_dev_iterations_with_batchsizes(self, batchsizes):
        return self._iterations("dev", batchsizes)

    def get_statistics(self):
        lines = []
        for bucket_id, pieces in enumerate(self._table):
            counts = [piece.updates for piece in pieces]
            lines += ["bucket {}".format(bucket_id + 1), str(counts), str(sum(counts) / len(counts))]
        return "\n".join(lines) + "\n"

    def dump(self):
        print("\tbucket\t#train\t#dev\tsec")
        totals = [0, 0]
        for bucket_id, pieces in enumerate(self._table):
            n_train, n_dev = sum(len(p.train) for p in pieces), sum(len(p.dev) for p in pieces)
            totals[0] += n_train
            totals[1] += n_dev
            print("\t{}\t{:>6}\t{:>4}\t{:>6.3f}".format(bucket_id + 1, n_train, n_dev, self.bucket_split_sec * (bucket_id + 1)))
        print("\ttotal\t{:>6}\t{:>4}".format(*totals))
