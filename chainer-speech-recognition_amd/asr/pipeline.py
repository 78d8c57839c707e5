"""Two half batches side by side on one GPU.

The recurrences of the train step are latency chains whose step time does not depend on the number of utterances (DESIGN.md
section 5), and every op of the path is per-utterance.  So a batch is run as two halves on two HIP streams: while one half
sits in a recurrence, the projections / convolutions / CTC of the other half use the rest of the chip, and two recurrences
run side by side: a half batch of at most 16 utterances still makes eight recurrences (4-row groups, one per XCD), the
forward kernel then asks for 40 KB of LDS and the backward one has 128 workgroups, so both launches fit on the chip at once
wherever the dispatcher puts them.  Gradients of both halves accumulate into the same flat buffer (atomics); the loss is
the mean of the half means weighted by their sizes, i.e. the batch mean of the reference.  Not for models with batch
statistics (nn.BatchNormalization would normalise each half by itself).
"""
import contextlib

import torch


class HalfBatches(object):
    def __init__(self, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.device = dev
        self.streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))

    def close(self):
        pass

    @contextlib.contextmanager
    def half(self, i):
        """ops issued inside run on stream i, after everything queued so far on the calling stream"""
        cur = torch.cuda.current_stream(self.device)
        st = self.streams[i]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            yield st

    def parts(self, batch_size):
        n0 = (batch_size + 1) // 2
        return [(i, sl, n) for i, (sl, n) in enumerate(((slice(0, n0), n0), (slice(n0, batch_size), batch_size - n0))) if n > 0]

    def loss(self, fn, batch_size):
        """fn(slice) -> mean loss of the utterances in that slice; returns the mean over the whole batch as ONE graph
        (an odd batch gives the first half one utterance more: weights n_i / B).  Simple, but the two backward passes
        then start together and stay in lockstep; `step` staggers them."""
        cur = torch.cuda.current_stream(self.device)
        total = None
        losses = []
        for i, sl, n in self.parts(batch_size):
            with self.half(i):
                losses.append((fn(sl), n, self.streams[i]))
        for l, n, st in losses:
            cur.wait_stream(st)
            term = l * (float(n) / batch_size)
            total = term if total is None else total + term
        return total

    def step(self, optimizer, fn, batch_size, stagger_us=0):
        """One train step: gradients of both halves, then optimizer.update().  Each half runs forward AND backward on its
        own stream with no dependency on the other; the second one starts later by the time the host needs to queue the
        first one's forward pass (about 1.6 ms for the BASELINE model) plus `stagger_us`, so that for the rest of the step the
        projections of one half fall into a recurrence of the other.  Returns the batch-mean loss (a detached tensor)."""
        from . import _lib
        optimizer.cleargrads()
        if optimizer.communicator is not None:
            optimizer.communicator.begin_backward(optimizer, passes=len(self.parts(batch_size)))
        cur = torch.cuda.current_stream(self.device)
        losses = []
        for i, sl, n in self.parts(batch_size):
            with self.half(i) as st:
                if i > 0:
                    _lib.check(_lib.lib().asr_stream_delay(st.cuda_stream, int(stagger_us)), "asr_stream_delay")
                losses.append((fn(sl) * (float(n) / batch_size), st))
        for k, (l, st) in enumerate(losses):       # (all forward work is queued before the first backward: no waiting)
            with torch.cuda.stream(st):
                l.backward()
            if optimizer.communicator is not None and k + 1 < len(losses):
                optimizer.communicator.end_pass()
        total = None
        for l, st in losses:
            cur.wait_stream(st)
            total = l.detach() if total is None else total + l.detach()
        optimizer.update()
        return total

    def join(self):
        """the calling stream waits for both halves (gradient kernels write the flat buffer behind autograd's back)"""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)
