"""Data parallelism over the GPUs of one node: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference is single-GPU (no collective anywhere); BASELINE.json adds data-parallel training.
Utterances shard over ranks (every op of the path is per-utterance, section 8e), the only exchange is the SUM of the
flat gradient buffer, issued in a few large slices on a side stream as soon as the backward pass has produced them
(the last layers finish first) and joined before the optimiser kernel, which applies 1/world and the global-norm clip.
"""
import os

import torch
import torch.distributed as dist


class Communicator(object):
    def __init__(self, backend=None, buckets=4):
        if not dist.is_initialized():
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group(backend=backend)
        self.backend = dist.get_backend()
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()
        self.buckets = max(1, int(buckets))
        self._stream = None
        self._pending = []
        self._plan = None

    # -- parameters -------------------------------------------------------------------------------
    def broadcast(self, flat):
        dist.broadcast(flat, src=0)

    # -- gradients --------------------------------------------------------------------------------
    def _make_plan(self, opt):
        """contiguous slices of the flat gradient buffer, split at parameter boundaries, last parameters first"""
        flat = opt._flat
        offs, sizes = flat["offsets"], flat["sizes"]
        total = offs[-1] + sizes[-1]
        target = (total + self.buckets - 1) // self.buckets
        plan, end = [], total
        start_idx = len(offs) - 1
        acc = 0
        for i in range(len(offs) - 1, -1, -1):
            acc += sizes[i]
            if acc >= target or i == 0:
                plan.append((offs[i], end, i, start_idx))       # [begin, end) covers params i..start_idx
                end = offs[i]
                start_idx = i - 1
                acc = 0
        return plan

    def begin_backward(self, opt, passes=1):
        """`passes` backward passes will write the gradient buffer one after the other in program order (asr/pipeline.py:
        two half batches, each on its own stream); a slice is reduced while the LAST pass walks past it, after the events
        the earlier passes left there."""
        opt._ensure_flat()
        self._plan = self._make_plan(opt)
        self._pending = []
        self._next = 0
        self._opt = opt
        self._passes, self._pass = max(1, int(passes)), 0
        self._events = [[] for _ in self._plan]
        if self.backend == "nccl" and self._stream is None:
            self._stream = torch.cuda.Stream()
        from . import link
        link._GRAD_LISTENER[0] = self._on_grad_buffer

    def _on_grad_buffer(self, param):
        """called when a backward kernel is about to write `param.grad`: everything behind it in the flat buffer
        (parameters registered later = layers closer to the loss) has already been enqueued."""
        if self._plan is None:
            return
        flat = self._opt._flat
        try:
            idx = flat["ids"].index(id(param))
        except ValueError:
            return
        while self._next < len(self._plan) and self._plan[self._next][2] > idx:
            self._passed(self._next)
            self._next += 1

    def _passed(self, k):
        """the current pass has queued every kernel that writes slice k"""
        if self._pass == self._passes - 1:
            self._launch(self._plan[k], self._events[k])
        elif self._stream is not None:
            from .functions import side_streams
            for st in [torch.cuda.current_stream()] + side_streams():
                ev = torch.cuda.Event()
                ev.record(st)
                self._events[k].append(ev)

    def end_pass(self):
        """between two backward passes: the pass just queued has written every slice it had not walked past yet"""
        if self._plan is None:
            return
        while self._next < len(self._plan):
            self._passed(self._next)
            self._next += 1
        self._pass += 1
        self._next = 0

    def _launch(self, item, events=()):
        begin, end = item[0], item[1]
        g = self._opt._flat["G"][begin:end]
        if self._stream is not None:
            for ev in events:
                self._stream.wait_event(ev)
            self._stream.wait_stream(torch.cuda.current_stream())
            from .functions import side_streams
            for st in side_streams():               # weight-gradient GEMMs run on side streams
                self._stream.wait_stream(st)
            with torch.cuda.stream(self._stream):
                self._pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self._pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))

    def finish_backward(self, opt):
        from . import link
        link._GRAD_LISTENER[0] = None
        if self._plan is None:          # update() without lossfun: reduce everything now
            opt._ensure_flat()
            dist.all_reduce(opt._flat["G"], op=dist.ReduceOp.SUM)
            return
        self._pass = self._passes - 1
        while self._next < len(self._plan):
            self._passed(self._next)
            self._next += 1
        for w in self._pending:
            w.wait()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        self._pending = []
        self._plan = None

    def allreduce_scalar_mean(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t / self.size

    def barrier(self):
        dist.barrier()
