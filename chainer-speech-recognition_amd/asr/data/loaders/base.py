"""asr/data/loaders/base.py:9-106: running feature statistics, normalisation, minibatch hand-over -- on the GPU."""
import os
import pickle

import numpy as np
import torch

from ... import _ops


class Loader(object):
    """Subclasses set ``processor``, ``token_ids``, ``id_blank`` (and ``reader`` for update_stats), as in the reference."""

    def __init__(self):
        self.stats_total = 0
        self.stats_mean = None      # (C, M) float64 on the GPU
        self.stats_nvar = None
        self._mean32 = self._std32 = None
        self.apply_cmn = False

    # -- statistics ------------------------------------------------------------------------------------------
    def _ensure_state(self, x):
        if self.stats_mean is None:
            shape = tuple(x.shape[1:-1])
            self.stats_mean = torch.zeros(shape, dtype=torch.float64, device=x.device)
            self.stats_nvar = torch.zeros(shape, dtype=torch.float64, device=x.device)
        if self._mean32 is None:
            self._mean32 = torch.zeros(self.stats_mean.shape, dtype=torch.float32, device=x.device)
            self._std32 = torch.ones(self.stats_mean.shape, dtype=torch.float32, device=x.device)

    def _update_stats_batch(self, x_batch, x_lengths):
        """asr/data/loaders/base.py:64-80 applied to every utterance x[b, ..., :length] in turn, in one kernel."""
        self._ensure_state(x_batch)
        lens = torch.as_tensor(np.asarray(x_lengths, dtype=np.int32)).to(x_batch.device)
        _ops.running_stats_update(x_batch.contiguous(), lens, self.stats_total, self.stats_mean, self.stats_nvar, self._mean32, self._std32)
        self.stats_total += int(np.sum(np.minimum(np.asarray(x_lengths), x_batch.shape[-1])))

    def _update_stats_recursively(self, x):
        """x: one utterance (C, M, T) -- the reference's entry point."""
        xx = torch.as_tensor(x).to(torch.float32)
        if xx.device.type != "cuda":
            xx = xx.cuda()
        self._update_stats_batch(xx.unsqueeze(0), [xx.shape[-1]])

    def get_mean_and_std(self):
        """(1, C, M, 1) mean and unbiased standard deviation (:39-41)."""
        return self._mean32[None, ..., None], self._std32[None, ..., None]

    def update_stats(self, iteration, batchsizes, augmentation=None):
        for _ in range(iteration):
            batch, bucket_idx, piece_id = self.reader.sample_minibatch(batchsizes)
            features, sentences, max_feature_length, max_sentence_length = self.extract_batch_features(batch, augmentation=augmentation)
            x_batch, x_length_batch, _, _, _ = self.processor.features_to_minibatch(features, sentences, max_feature_length,
                                                                                   max_sentence_length, self.token_ids, self.id_blank)
            self._update_stats_batch(x_batch, x_length_batch)

    def save_stats(self, directory):
        os.makedirs(directory, exist_ok=True)
        np.save(os.path.join(directory, "mean.npy"), self.stats_mean.cpu().numpy())
        np.save(os.path.join(directory, "nvar.npy"), self.stats_nvar.cpu().numpy())
        with open(os.path.join(directory, "total.count"), mode="wb") as f:
            pickle.dump(self.stats_total, f)

    def load_stats(self, directory):
        names = [os.path.join(directory, n) for n in ("mean.npy", "nvar.npy", "total.count")]
        if not all(os.path.isfile(n) for n in names):
            return False
        dev = torch.device("cuda", torch.cuda.current_device())
        self.stats_mean = torch.from_numpy(np.load(names[0]).astype(np.float64)).to(dev)
        self.stats_nvar = torch.from_numpy(np.load(names[1]).astype(np.float64)).to(dev)
        with open(names[2], mode="rb") as f:
            self.stats_total = pickle.load(f)
        mean, nvar = np.load(names[0]).astype(np.float64), np.load(names[1]).astype(np.float64)
        self._mean32 = torch.from_numpy(mean.astype(np.float32)).to(dev)
        self._std32 = torch.from_numpy(np.sqrt(nvar / max(self.stats_total - 1, 1)).astype(np.float32)).to(dev)
        return True

    # -- minibatch ---------------------------------------------------------------------------------------------
    def extract_batch_features(self, batch, augmentation=None):
        return self.processor.extract_batch_features(batch, augmentation, self.apply_cmn)

    def features_to_minibatch(self, features, sentences, max_feature_length, max_sentence_length, gpu=True):
        x_batch, x_length_batch, t_batch, t_length_batch, bigram_batch = self.processor.features_to_minibatch(
            features, sentences, max_feature_length, max_sentence_length, self.token_ids, self.id_blank)
        if self.stats_total > 0:        # :20-24: this minibatch first updates the statistics, then is normalised by them
            x_batch = x_batch.contiguous().clone()
            self._update_stats_batch(x_batch, x_length_batch)
            _ops.normalize_bcmt(x_batch, self._mean32, self._std32)
        dev = x_batch.device
        t_batch = torch.from_numpy(np.ascontiguousarray(t_batch, dtype=np.int32))
        bigram_batch = torch.from_numpy(np.ascontiguousarray(bigram_batch, dtype=np.int32))
        x_length_batch = torch.from_numpy(np.asarray(x_length_batch, dtype=np.int32))
        t_length_batch = torch.from_numpy(np.asarray(t_length_batch, dtype=np.int32))
        if gpu:
            t_batch, bigram_batch = t_batch.to(dev), bigram_batch.to(dev)
            x_length_batch, t_length_batch = x_length_batch.to(dev), t_length_batch.to(dev)
        else:
            x_batch = x_batch.cpu()
        return x_batch, x_length_batch, t_batch, t_length_batch, bigram_batch
