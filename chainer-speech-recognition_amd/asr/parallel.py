"""Data parallelism over the GPUs of one node: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference is single-GPU (no collective anywhere); BASELINE.json adds data-parallel training.
Utterances shard over ranks (every op of the path is per-utterance, section 8e), the only exchange is the SUM of the
flat gradient buffer, issued in a few large contiguous slices and joined before the optimiser kernels, which apply
1/world and the global-norm clip to the REDUCED gradient (run/ctc/cnn/train.py:144-145 clips what the optimiser sees).

When a slice may go
--------------------
* A slice is *complete* once every parameter in it has been announced by ``link.grads_queued`` -- which the backward
  functions call AFTER they have queued the last kernel writing those gradients (main stream or side stream).  Round 1
  announced a parameter when its buffer was asked for, i.e. before the kernels; a slice starting at ``w_hh`` was then
  reduced before its weight-gradient GEMM had even been queued.
* Slices go in plan order (last layers first), the same order on every rank.
* A complete slice is not launched at once but at the next *recurrence boundary*: the persistent GRU kernels want one
  workgroup on every CU for the whole launch (csrc/gru.hip), so an RCCL kernel that is resident when such a launch starts
  -- possibly waiting for a slower peer -- would hold CUs the recurrence needs while its other workgroups spin.  Hence
  (``_ops.RECURRENCE_HOOKS``):  *before* a recurrence is queued the launch stream waits for every collective in flight;
  *right after* it is queued the complete slices are launched behind it on the communication stream.  A collective
  therefore runs in the gap between two recurrences, next to the projection and weight-gradient GEMMs of the NEXT layer
  (which it does not depend on), never beside a recurrence.  What is complete after the last recurrence (the first GRU
  layer, the convolutions) no longer has a recurrence to fear: behind the last recurrence of a pass (the count is the previous
  pass's) slices go the moment they are complete, beside the convolutions' backward pass; only the small slice at the front of
  the buffer (``make_plan``) is left for ``finish_backward``.  A backward pass without recurrences (the convolutional
  recipes) launches every slice the moment it is complete.
* ``beside_recurrences=True`` drops that rule: slices go the moment they are complete and no recurrence waits for them.  What a
  resident collective costs a recurrence was measured on one GPU with a stand-in (tools/gru_beside_collective.py, DESIGN.md section
  13.5: 8 / 16 / 32 workgroups streaming memory with 16 KB of LDS each, resident for the whole launch): forward 1.23 -> 1.50 / 1.56 /
  1.66 us per time step, backward 1.38 -> 1.57 / 1.60 / 1.74, no give-up, identical results; a stand-in whose LDS does not fit beside
  a recurrence workgroup (backward: 132 of 160 KB) simply makes the recurrence wait until it has left.  A 1.2 ms recurrence fully
  overlapped by collectives thus pays 0.2 - 0.4 ms -- about what the same collectives cost in the gaps -- so the rule stays the
  default and the switch is for interconnects slow enough that the gaps do not hold a slice.
"""
import os

import torch
import torch.distributed as dist


class Communicator(object):
    def __init__(self, backend=None, buckets=4, overlap=True, beside_recurrences=False):
        if not dist.is_initialized():
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group(backend=backend)
        self.backend = dist.get_backend()
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()
        self.buckets = max(1, int(buckets))
        self.overlap = bool(overlap)
        self.beside_recurrences = bool(beside_recurrences)
        self._stream = None
        self._pending = []
        self._plan = None
        self.launch_log = []            # (slice index, parameters announced so far) per launch of the last step: tests
        self._recurrent = False         # the previous backward pass met a recurrence: defer launches to the boundaries
        self.measure = False            # bench.py: time the launch stream's stalls on collectives (exposed_ms)
        self.exposed = []
        self._hold_all = False
        self._rec_count = self._rec_count_prev = 0      # recurrences met in this / the previous backward pass

    # -- parameters -------------------------------------------------------------------------------
    def broadcast(self, flat):
        dist.broadcast(flat, src=0)

    def bcast_data(self, model):
        """ChainerMN's ``comm.bcast_data(model)``: every initialised parameter and persistent buffer takes rank 0's value.
        Call it after the first forward pass (lazily sized parameters and the data-dependent weight-norm initialisation,
        asr/nn/convolution_2d.py:177-187, exist only then) and before the first update; ``Optimizer`` does the same for the
        flat parameter buffer whenever it (re)builds it."""
        from .link import bump_weight_epoch
        for t in list(model.parameters()) + list(model.buffers()):
            if t.numel() > 0:
                dist.broadcast(t.data, src=0)
        bump_weight_epoch()

    # -- gradients --------------------------------------------------------------------------------
    @staticmethod
    def make_plan(offsets, sizes, buckets):
        """contiguous slices [begin, end) of the flat gradient buffer, cut at parameter boundaries, last parameters first:
        (begin, end, first parameter index, last parameter index).  The last slice starts at element 0: it takes the reserved
        elements in front of the first parameter along (optimizers.RESERVED: the cross-rank "a recurrence gave up" mark)."""
        total = offsets[-1] + sizes[-1]
        # The slice at the FRONT of the buffer is summed last, after the whole backward pass has been queued, with nothing left to
        # hide it behind: keep it small -- the longest run of leading parameters within 1/32 of the buffer (the convolutions of the
        # BASELINE model: 0.25 M of 13.9 M parameters) -- and cut the rest into buckets - 1 slices of equal size.
        n = len(offsets)
        front = 0                                   # parameters 0 .. front form the last slice
        if buckets > 1 and n > 1:
            while front + 1 < n - 1 and offsets[front + 1] + sizes[front + 1] <= total // 32:
                front += 1
            rest = total - (offsets[front] + sizes[front])
            target = (rest + buckets - 2) // (buckets - 1)
        else:
            front = n - 1
            target = total
        plan, end = [], total
        last = n - 1
        acc = 0
        for i in range(n - 1, front, -1):
            acc += sizes[i]
            if acc >= target or i == front + 1:
                plan.append((offsets[i], end, i, last))
                end = offsets[i]
                last = i - 1
                acc = 0
        plan.append((0, end, 0, last))
        return plan

    def begin_backward(self, opt, passes=1):
        """`passes` backward passes will write the gradient buffer one after the other in program order (gradient accumulation
        over several minibatches before one update, end_pass() between them); a slice is reduced only after the LAST pass has
        announced it."""
        opt._ensure_flat()
        flat = opt._flat
        self._plan = self.make_plan(flat["offsets"], flat["sizes"], self.buckets)
        self._slice_of = {}
        for k, (_, _, lo, hi) in enumerate(self._plan):
            for i in range(lo, hi + 1):
                self._slice_of[flat["ids"][i]] = k
        self._pending = []
        self._next = 0
        self._opt = opt
        self._passes, self._pass = max(1, int(passes)), 0
        self._events = [[] for _ in self._plan]
        self._new_pass()
        self.launch_log = []
        self._defer, self._met_recurrence = self._recurrent and not self.beside_recurrences, False
        self._rec_count = 0
        # several passes: nothing is launched before finish_backward (the earlier passes' recurrences would otherwise meet the
        # collectives of a slice that is complete for THEIR pass only)
        self._hold_all = self._passes > 1
        self.exposed = []               # (event before, event after) around every join of the launch stream with the collectives
        if self.backend == "nccl" and self._stream is None:
            self._stream = torch.cuda.Stream()
        from . import link, _ops
        link._GRAD_LISTENER[0] = self._on_grads_queued
        _ops.RECURRENCE_HOOKS["before"] = self._before_recurrence
        _ops.RECURRENCE_HOOKS["after"] = self._after_recurrence

    def _new_pass(self):
        self._missing = [set(range(lo, hi + 1)) for (_, _, lo, hi) in self._plan]
        self._announced = 0
        self._index_of = {pid: i for i, pid in enumerate(self._opt._flat["ids"])}

    def _on_grads_queued(self, params):
        """every kernel that writes the gradients of `params` in this pass has been queued"""
        if self._plan is None:
            return
        for p in params:
            i = self._index_of.get(id(p))
            if i is None:
                continue
            k = self._slice_of[id(p)]
            if k < self._next and self._pass == self._passes - 1:
                raise RuntimeError("a gradient was written after its slice had been all-reduced (a parameter used by two "
                                   "layers?): build the Communicator with overlap=False")
            if i in self._missing[k]:
                self._missing[k].discard(i)
                self._announced += 1
        if self.overlap and not self._defer:
            self._launch_complete()

    def _complete(self, k):
        return not self._missing[k]

    # recurrence boundaries (see the module docstring)
    def _before_recurrence(self):
        self._met_recurrence = True
        self._rec_count += 1
        if self.beside_recurrences:
            return              # collectives in flight stay in flight; slices keep going the moment they are complete
        self._defer = True
        self._join_pending()

    def _after_recurrence(self):
        # Behind the LAST recurrence of the pass (as many as the previous pass had) nothing wants every CU any more: slices go the
        # moment they are complete again -- the first GRU layer's gradients are then summed beside the convolutions' backward pass
        # instead of after it.  Should more recurrences follow after all, each still waits for what is in flight (and defers again).
        if self._rec_count_prev and self._rec_count == self._rec_count_prev:
            self._defer = False
        if self.overlap:
            self._launch_complete()

    def _launch_complete(self):
        # the LAST slice (first parameters + the reserved mark) always waits for finish_backward: the optimiser plants the
        # "gave up" mark in it after the last recurrence has been queued and before it is summed
        if self._hold_all:
            return
        while self._next < len(self._plan) - 1 and self._complete(self._next):
            self._passed(self._next)
            self._next += 1

    def _passed(self, k):
        """the current pass has queued every kernel that writes slice k"""
        if self._pass == self._passes - 1:
            self._launch(k, self._events[k])
        elif self._stream is not None:
            from .functions import side_streams
            for st in [torch.cuda.current_stream()] + side_streams():
                ev = torch.cuda.Event()
                ev.record(st)
                self._events[k].append(ev)

    def end_pass(self):
        """between two backward passes: the pass just queued has written every slice"""
        if self._plan is None:
            return
        while self._next < len(self._plan):
            self._passed(self._next)
            self._next += 1
        self._pass += 1
        self._next = 0
        self._new_pass()

    def _launch(self, k, events=()):
        begin, end = self._plan[k][0], self._plan[k][1]
        g = self._opt._flat["G"][begin:end]
        self.launch_log.append((k, self._announced))
        from .functions import side_streams
        if self._stream is not None:
            for ev in events:
                self._stream.wait_event(ev)
            self._stream.wait_stream(torch.cuda.current_stream())
            for st in side_streams():               # weight-gradient GEMMs run on side streams
                self._stream.wait_stream(st)
            with torch.cuda.stream(self._stream):
                self._pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
        else:
            if g.is_cuda:                           # gloo on device tensors: it orders itself after the CURRENT stream only
                cur = torch.cuda.current_stream()
                for st in side_streams():
                    cur.wait_stream(st)
            self._pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))

    def _join_pending(self):
        """the current stream waits for every collective in flight (RCCL: a stream dependency, the host goes on)"""
        timed = self.measure and self._stream is not None and self._pending
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._pending:
            w.wait()
        if self._stream is not None and self._pending:
            torch.cuda.current_stream().wait_stream(self._stream)
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.exposed.append((e0, e1))
        self._pending = []

    def exposed_ms(self):
        """time the launch stream spent stalled on collectives during the last step (measure=True; synchronises)"""
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.exposed)

    def abort_backward(self):
        """the backward pass raised: take the listener and the recurrence hooks down, wait out what is in flight"""
        from . import link, _ops
        link._GRAD_LISTENER[0] = None
        _ops.RECURRENCE_HOOKS["before"] = _ops.RECURRENCE_HOOKS["after"] = None
        try:
            self._join_pending()
        finally:
            self._pending, self._plan = [], None

    def finish_backward(self, opt):
        from . import link, _ops
        link._GRAD_LISTENER[0] = None
        _ops.RECURRENCE_HOOKS["before"] = _ops.RECURRENCE_HOOKS["after"] = None
        if self._plan is None:          # update() without lossfun: reduce everything now
            opt._ensure_flat()
            dist.all_reduce(opt._flat["G"], op=dist.ReduceOp.SUM)
            return
        self._pass = self._passes - 1
        while self._next < len(self._plan):     # incl. slices with parameters that received no gradient in this step
            self._passed(self._next)
            self._next += 1
        self._join_pending()
        self._plan = None
        self._recurrent = self._met_recurrence
        self._rec_count_prev = self._rec_count

    def allreduce_scalar_mean(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t / self.size

    def barrier(self):
        dist.barrier()
