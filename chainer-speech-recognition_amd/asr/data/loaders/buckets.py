"""Loader over a bucketed corpus: Reader (disk) + Processor (GPU features) + the running statistics of base.Loader
(reference: asr/data/loaders/buckets.py:10-83 and asr/data/iterators/buckets/{train,dev}.py)."""
from ..processing import Processor
from ..readers.buckets import Reader
from . import base


class Loader(base.Loader):
    def __init__(self, data_path, batchsizes_train, batchsizes_dev=None, buckets_limit=None, bucket_split_sec=0.5,
                 buckets_cache_size=200, vocab_token_to_id=None, dev_split=0.01, seed=0, id_blank=0, apply_cmn=False,
                 sampling_rate=16000, frame_width=0.032, frame_shift=0.01, num_mel_filters=40, window_func="hanning",
                 using_delta=True, using_delta_delta=True):
        if not isinstance(vocab_token_to_id, dict):
            raise TypeError("vocab_token_to_id: the token -> id dictionary is required")
        super().__init__()
        self.batchsizes_train, self.batchsizes_dev = batchsizes_train, batchsizes_dev
        self.token_ids, self.id_blank, self.apply_cmn = vocab_token_to_id, id_blank, apply_cmn
        self.processor = Processor(sampling_rate=sampling_rate, frame_width=frame_width, frame_shift=frame_shift,
                                   num_mel_filters=num_mel_filters, window_func=window_func, using_delta=using_delta,
                                   using_delta_delta=using_delta_delta)
        self.reader = Reader(data_path=data_path, buckets_limit=buckets_limit, buckets_cache_size=buckets_cache_size,
                             dev_split=dev_split, seed=seed, sampling_rate=sampling_rate, bucket_split_sec=bucket_split_sec)

    def _minibatch(self, batch, augmentation, gpu):
        features, sentences, max_frames, max_tokens = self.extract_batch_features(batch, augmentation=augmentation)
        return self.features_to_minibatch(features, sentences, max_frames, max_tokens, gpu=gpu)

    def sample_minibatch(self, augmentation=None, gpu=True):
        batch, bucket_id, _ = self.reader.sample_minibatch(self.batchsizes_train)
        return self._minibatch(batch, augmentation, gpu) + (bucket_id,)

    def get_total_training_iterations(self):
        return self.reader.calculate_total_training_iterations_with_batchsizes(self.batchsizes_train)

    def get_total_dev_iterations(self):
        return self.reader.calculate_total_dev_iterations_with_batchsizes(self.batchsizes_dev)

    def get_num_buckets(self):
        return self.reader.get_num_buckets()

    def set_batchsizes_train(self, batchsizes):
        self.batchsizes_train = batchsizes

    def set_batchsizes_dev(self, batchsizes):
        self.batchsizes_dev = batchsizes

    def get_training_batch_iterator(self, batchsizes, augmentation=None, gpu=True):
        return TrainIterator(self, batchsizes, augmentation, gpu)

    def get_development_batch_iterator(self, batchsizes, augmentation=None, gpu=True):
        return DevIterator(self, batchsizes, augmentation, gpu)

    def get_statistics(self):
        return self.reader.get_statistics()

    def dump(self):
        print("[Dataset]")
        self.reader.dump()


class TrainIterator(object):
    """get_total_training_iterations() random minibatches (asr/data/iterators/buckets/train.py; like the reference's, it
    samples without augmentation whatever was asked for)"""

    def __init__(self, loader, batchsizes, augmentation=None, gpu=True):
        self.loader, self.batchsizes, self.gpu = loader, batchsizes, gpu
        self.augmentation = None
        self.total_itr = loader.get_total_training_iterations()
        self.itr = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.itr >= self.total_itr:
            raise StopIteration
        self.itr += 1
        return self.loader.sample_minibatch(self.augmentation, self.gpu)

    def get_total_iterations(self):
        return self.total_itr


class DevIterator(object):
    """every development utterance once, bucket by bucket, piece by piece, in chunks of batchsizes[bucket]
    (asr/data/iterators/buckets/dev.py)"""

    def __init__(self, loader, batchsizes, augmentation=None, gpu=True):
        self.loader, self.batchsizes, self.gpu = loader, batchsizes, gpu
        self.augmentation = None
        self.total_itr = loader.get_total_dev_iterations()
        self._chunks = self._walk()

    def _walk(self):
        reader = self.loader.reader
        for bucket_id, pieces in enumerate(reader.buckets_indices_dev):
            for piece_id, indices in enumerate(pieces):
                step = self.batchsizes[bucket_id]
                for pos in range(0, len(indices), step):
                    signals = reader.get_signals_by_bucket_and_piece(bucket_id, piece_id)
                    sentences = reader.get_sentences_by_bucket_and_piece(bucket_id, piece_id)
                    yield bucket_id, [(signals[i], sentences[i]) for i in indices[pos:pos + step]]

    def __iter__(self):
        return self

    def __next__(self):
        bucket_id, batch = next(self._chunks)
        return self.loader._minibatch(batch, self.augmentation, self.gpu) + (bucket_id,)

    def get_total_iterations(self):
        return self.total_itr
