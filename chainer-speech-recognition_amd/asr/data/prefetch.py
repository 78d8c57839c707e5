"""Minibatch prefetch for the train loop -- the role of the preloading process of run/ctc/cnn/train_async.py:25-31,178-203.

The reference prepares the next 50 minibatches on the CPU in a second process.  Here the feature extraction is itself a
handful of GPU kernels, so the prefetcher is a host thread that pulls raw (signal, sentence) batches from any iterator
(e.g. ``Reader.sample_minibatch``), enqueues ``Loader.extract_batch_features`` + ``features_to_minibatch`` on its own HIP
stream and hands the device tensors over with an event: the consumer's stream waits for that event (no host
synchronisation), and the feature kernels run in the gaps of the latency-bound GRU recurrences of the current step
(co-residency).
"""
import queue
import threading

import torch


class DevicePrefetcher(object):
    def __init__(self, batches, loader, depth=2, augmentation=None, device=None):
        """batches: iterable of lists of (int16 signal, sentence); loader: an asr.data.loaders.base.Loader (with
        processor, token_ids, id_blank); depth: minibatches kept ready on the device."""
        self.loader = loader
        self.augmentation = augmentation
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self._q = queue.Queue(maxsize=depth)
        self._it = iter(batches)
        self._err = None
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._thread.start()

    def _work(self):
        try:
            torch.cuda.set_device(self.device)
            for batch in self._it:
                with torch.cuda.stream(self.stream):
                    feats, sents, max_f, max_s = self.loader.extract_batch_features(batch, augmentation=self.augmentation)
                    out = self.loader.features_to_minibatch(feats, sents, max_f, max_s, gpu=True)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                self._q.put((out, ev))
        except BaseException as e:       # surfaces in the consumer
            self._err = e
        self._q.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is None:
            if self._err is not None:
                raise self._err
            raise StopIteration
        out, ev = item
        torch.cuda.current_stream(self.device).wait_event(ev)
        for t in out:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(torch.cuda.current_stream(self.device))     # the allocator must not recycle it early
        return out
