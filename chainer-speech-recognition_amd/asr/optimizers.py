"""Optimisers of the train step on one flat parameter buffer.

Mirrors asr/optimizers.py:1-75 (get/set learning rate and momentum, ``get_optimizer``, ``decay_learning_rate``)
and the slice of ``chainer.optimizer`` the training scripts use (run/ctc/cnn/train.py:142-147,200):
``setup(model)``, ``add_hook(GradientClipping(t))``, ``add_hook(WeightDecay(r))``, ``update(lossfun=lambda: loss)``.

All parameters live in ONE float32 buffer (their ``.data`` / ``.grad`` are views), so the step is two HIP launches
(squared norm for the global-norm clipping, fused clip + decay + update) and the data-parallel all-reduce is a few large
RCCL calls over contiguous slices.
"""
import torch

from . import _ops
from .link import bump_weight_epoch, refresh_compute_copies

F32 = torch.float32
RESERVED = 64       # floats in front of the first parameter in the flat buffers: element 0 carries the cross-rank "gave up" mark


class GradientClipping(object):
    """chainer.optimizer.GradientClipping(threshold): scale all gradients by threshold / ||g||_2 when that is < 1."""
    name = "GradientClipping"

    def __init__(self, threshold):
        self.threshold = threshold


class WeightDecay(object):
    """chainer.optimizer.WeightDecay(rate): g += rate * p."""
    name = "WeightDecay"

    def __init__(self, rate):
        self.rate = rate


class Optimizer(object):
    def __init__(self):
        self.target = None
        self.t = 0
        self._hooks = []
        self._flat = None
        self.communicator = None
        self._needs_broadcast = False
        self._gave_up = [None, None, 0]     # pinned host copy of ctl[6] (give-up drops so far), event of that copy, update it was queued behind
        self._gave_up_reported = 0.0
        self._incident_until = -1           # last update attributed to the incident reported last
        self._loss_scaling = None           # (interval, initial scale) once loss_scaling() was called
        self._ls = None                     # device float[4]: asr_step_control_scaled's loss_scale

    # -- Chainer surface ---------------------------------------------------------------------------
    def setup(self, link):
        self.target = link
        self.t = 0
        self._flat = None
        return self

    def add_hook(self, hook, name=None):
        self._hooks.append(hook)

    def set_communicator(self, comm):
        """Data parallelism: gradients are summed over ranks (RCCL) before the step and scaled by 1/world."""
        self.communicator = comm
        self._needs_broadcast = comm is not None    # the flat buffer (and the optimiser state with it) stays as it is
        if comm is not None and self._flat is not None:
            # a running optimiser: take rank 0's parameters and state NOW -- deferred to the next update the first data-parallel
            # forward pass would still run on the rank-local weights
            self._broadcast_state()

    def loss_scaling(self, interval=1000, scale=None):
        """chainer.Optimizer.loss_scaling: seed the backward pass with a loss scale S and take it out again in the update, so that
        small activation gradients survive a 16-bit format with a 5-bit exponent (the IEEE-half library, ASR_ACT=f16; harmless with
        bfloat16).  scale=None: dynamic -- S starts at 4096, is halved whenever a gradient overflows (that step is dropped, as every
        step with a non-finite gradient is) and doubled after `interval` applied steps in a row; a number: static.  S lives on the
        device and is changed there (asr_step_control_scaled): no host synchronisation, identical on every data-parallel rank."""
        self._loss_scaling = (int(interval) if scale is None else 0, 4096.0 if scale is None else float(scale))
        self._ls = None

    def loss_scale(self):
        """(S, overflows so far) -- synchronises: tests / logging only"""
        if self._ls is None:
            return (1.0 if self._loss_scaling is None else self._loss_scaling[1]), 0
        v = self._ls.cpu()
        return float(v[0]), int(v[3])

    def _loss_scale_buffer(self, dev):
        if self._loss_scaling is None:
            return None
        if self._ls is None or self._ls.device != dev:
            self._ls = torch.tensor([self._loss_scaling[1], 0.0, float(self._loss_scaling[0]), 0.0], dtype=F32, device=dev)
        return self._ls

    def update(self, lossfun=None, *args, **kwds):
        if lossfun is not None:
            loss = lossfun(*args, **kwds)
            self.cleargrads()
            if self.communicator is not None:
                self.communicator.begin_backward(self)
            try:
                ls = self._loss_scale_buffer(loss.device)
                if ls is None:
                    loss.backward()
                else:       # the seed is the device float itself: read when the loss head's backward kernel runs
                    loss.backward(gradient=ls[0].reshape(loss.shape).to(loss.dtype))
            except BaseException:
                if self.communicator is not None:
                    self.communicator.abort_backward()      # no listener / recurrence hooks left behind
                from .functions import release_side_keeps
                release_side_keeps(to_allocator=True)       # nothing stays pinned for a join that will not come
                raise
        self._ensure_flat()
        from .functions import join_side_stream
        join_side_stream()                  # weight-gradient GEMMs issued on the side stream
        P, G = self._flat["P"], self._flat["G"]
        # every abort word of this process ORed into one device word; data parallel: a raised word also plants a NaN in the
        # reserved element of the LOCAL gradient buffer before its slice is summed, so that EVERY rank drops this step
        any_abort = _ops.gather_abort(G.device, G[0:1] if self.communicator is not None else None)
        scale = 1.0
        if self.communicator is not None:
            self.communicator.finish_backward(self)
            scale = 1.0 / self.communicator.size
        self.t += 1                         # steps ATTEMPTED; the device counts the applied ones (flat["applied"])
        clip, decay = 0.0, 0.0
        for h in self._hooks:
            if isinstance(h, GradientClipping):
                clip = float(h.threshold)
            elif isinstance(h, WeightDecay):
                decay = float(h.rate)
        # the squared gradient norm is always taken: it drives the clipping and the device-side "drop this step" decision
        # (asr_step_control: non-finite norm = the reference's NaN check, run/ctc/cnn/train.py:193-197, or a persistent GRU
        # launch that gave up a wait and left garbage behind) -- without a host synchronisation
        alpha, beta1, beta2 = self._control_constants()
        _ops.step_control(G, self._flat["partials"], clip, scale, alpha, beta1, beta2, self._flat["applied"], self._flat["ctl"],
                          any_abort, 0, self._loss_scale_buffer(G.device))
        self._step(P, G, decay, self._flat["ctl"])
        bump_weight_epoch()
        refresh_compute_copies(self.target)     # every plain / transposed bf16 weight copy, one launch
        # The host learns of a given-up recurrence some steps late and without synchronising (a non-blocking query of an asynchronous
        # copy of ctl[5]).  WHICH updates are applied is decided on the device alone and identically on every rank: the abort word
        # is sticky, every step's gather_abort plants the NaN in front of the reduction, step_control drops the step from the REDUCED
        # element -- until the raising rank clears its word.  The report therefore comes LAST, behind this step's collectives (a rank
        # raising before finish_backward would leave its peers in an all-reduce) and behind this step's update kernels: a
        # free-running loop in which the ranks' hosts run ahead of their devices by different amounts raises at different updates on
        # different ranks, and none of them may skip queueing an update its peers apply.  The copy looked at is an EARLIER
        # update's (this update's is queued behind the look): "one update later" at the earliest, on every rank.
        try:
            self._raise_if_previous_step_gave_up()
        finally:
            self._watch_gave_up()

    def _watch_gave_up(self):
        """asynchronous copy of ctl[6] (steps dropped because a recurrence gave up, ever) whenever the previous copy has landed;
        st[2] = the update the copy in flight was queued behind"""
        st = self._gave_up
        if st[0] is None:
            st[0] = torch.zeros(1, dtype=F32).pin_memory()
        if st[1] is None or st[1].query():
            st[0].copy_(self._flat["ctl"][6:7], non_blocking=True)
            st[1] = torch.cuda.Event()
            st[1].record()
            st[2] = self.t

    def _raise_if_previous_step_gave_up(self):
        st = self._gave_up
        if st[1] is None or not st[1].query() or float(st[0][0]) <= self._gave_up_reported:
            return
        count, seen_at = float(st[0][0]), st[2]
        self._gave_up_reported = count
        if seen_at <= self._incident_until:
            return          # drops of the incident already reported: the updates queued before the abort words were cleared
        self._incident_until = self.t       # this update was queued in front of the clear below: it may be dropped as well
        st[1] = None
        _ops.clear_abort_words(self._flat["G"].device)       # sticky until somebody has been told: now
        _ops.reset_poll_status()
        from ._lib import AsrHipError
        raise AsrHipError("a persistent GRU kernel gave up an in-launch wait during an earlier step (on this rank or on a "
                          "data-parallel peer): every update from that step up to and including this one is dropped, on every "
                          "rank alike (%d so far); parameters and optimiser state are intact; training may continue" % int(count))

    def applied_steps(self):
        """number of update steps that were not dropped on the device (synchronises: tests / logging only)"""
        return 0 if self._flat is None else int(self._flat["applied"].cpu()[0])

    def _control_constants(self):
        return 0.0, 0.0, 0.0

    # -- flat buffers ------------------------------------------------------------------------------
    def _params(self):
        return [p for p in self.target.parameters() if p.numel() > 0]

    def _ensure_flat(self):
        params = self._params()
        if self._flat is not None and self._flat["ids"] == [id(p) for p in params] and \
                all(p.data_ptr() == a for p, a in zip(params, self._flat["ptrs"])):
            if self._needs_broadcast:
                self._broadcast_state()
            return
        if len(params) == 0:
            raise RuntimeError("optimizer.setup(model) was given a model without initialised parameters")
        dev = params[0].device
        sizes = [(p.numel() + 63) // 64 * 64 for p in params]           # 256-byte aligned slices
        total = RESERVED + sum(sizes)           # element 0 belongs to no parameter (asr_step_control's reserved_index)
        old = self._flat
        P = torch.empty(total, dtype=F32, device=dev)
        G = torch.empty(total, dtype=F32, device=dev)
        _ops.fill_(P, 0.0)
        _ops.fill_(G, 0.0)
        offs, o = [], RESERVED
        for p, n in zip(params, sizes):
            offs.append(o)
            view = P[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)                      # one-time host-driven move of the master weights
            if p.grad is not None:
                G[o:o + p.numel()].view(p.shape).copy_(p.grad)
            p.data = view
            p.grad = G[o:o + p.numel()].view(p.shape)
            o += n
        applied = old["applied"] if old is not None and old["applied"].device == dev else torch.zeros(1, dtype=torch.int32, device=dev)
        self._flat = dict(P=P, G=G, ids=[id(p) for p in params], ptrs=[p.data_ptr() for p in params], offsets=offs,
                          sizes=sizes, numels=[p.numel() for p in params], applied=applied,
                          ctl=(old["ctl"] if old is not None and old["ctl"].device == dev else torch.zeros(8, dtype=F32, device=dev)),
                          partials=torch.empty(_ops.sqnorm_partials_count(total), dtype=F32, device=dev))
        self._flat["sq"] = self._flat["ctl"][4:5]       # squared norm of the (summed) gradient of the last step
        self._init_state(total, dev, old)
        bump_weight_epoch()
        if self.communicator is not None:
            self._broadcast_state()

    def _state_buffers(self):
        return []

    def _broadcast_state(self):
        """every rank takes rank 0's parameters, optimiser state and applied-step count"""
        self._needs_broadcast = False
        ls = self._loss_scale_buffer(self._flat["P"].device)
        for t in [self._flat["P"], self._flat["applied"]] + ([] if ls is None else [ls]) + [b for b in self._state_buffers() if b is not None]:
            self.communicator.broadcast(t)
        bump_weight_epoch()

    def cleargrads(self):
        """model.cleargrads() of Chainer: here one fill of the flat gradient buffer."""
        self._ensure_flat()
        _ops.fill_(self._flat["G"], 0.0)
        for p, o in zip(self._params(), self._flat["offsets"]):
            if p.grad is None or p.grad.data_ptr() != self._flat["G"].data_ptr() + 4 * o:
                p.grad = self._flat["G"][o:o + p.numel()].view(p.shape)

    def flat_gradients(self):
        self._ensure_flat()
        return self._flat["G"]

    def flat_parameters(self):
        self._ensure_flat()
        return self._flat["P"]

    def _init_state(self, total, dev, old):
        pass

    def _carried(self, total, dev, old, previous):
        """a zeroed state buffer for the new flat layout with the slices of `previous` (the same state under the layout
        `old`) copied over for every parameter that is still there: a re-flatten -- a lazily sized parameter appeared, a
        communicator was attached after training had started -- must not forget the accumulated moments"""
        state = torch.empty(total, dtype=F32, device=dev)
        _ops.fill_(state, 0.0)
        if old is not None and previous is not None and previous.device == dev:
            where = {pid: (o, n) for pid, o, n in zip(old["ids"], old["offsets"], old["numels"])}
            for pid, o, n in zip(self._flat["ids"], self._flat["offsets"], self._flat["numels"]):
                src = where.get(pid)
                if src is not None and src[1] == n:
                    state[o:o + n].copy_(previous[src[0]:src[0] + n])
        return state

    def _step(self, P, G, decay, ctl):
        raise NotImplementedError


class Adam(Optimizer):
    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        super().__init__()
        self.alpha, self.beta1, self.beta2, self.eps = alpha, beta1, beta2, eps

        self.m = self.v = None

    def _init_state(self, total, dev, old):
        self.m = self._carried(total, dev, old, self.m)
        self.v = self._carried(total, dev, old, self.v)

    def _control_constants(self):
        return self.alpha, self.beta1, self.beta2

    def _state_buffers(self):
        return [self.m, self.v]

    def _step(self, P, G, decay, ctl):
        _ops.adam_ctl(P, G, self.m, self.v, self.beta1, self.beta2, self.eps, decay, ctl)


class SGD(Optimizer):
    kind = 0

    def __init__(self, lr=0.01):
        super().__init__()
        self.lr = lr
        self.momentum = 0.0
        self.vel = None

    def _init_state(self, total, dev, old):
        if self.kind:
            self.vel = self._carried(total, dev, old, self.vel)

    def _state_buffers(self):
        return [self.vel]

    def _step(self, P, G, decay, ctl):
        _ops.sgd_ctl(P, G, self.vel, self.kind, self.lr, self.momentum, decay, ctl)


class MomentumSGD(SGD):
    kind = 1

    def __init__(self, lr=0.01, momentum=0.9):
        super().__init__(lr)
        self.momentum = momentum


class NesterovAG(SGD):
    kind = 2

    def __init__(self, lr=0.01, momentum=0.9):
        super().__init__(lr)
        self.momentum = momentum


class optimizers(object):       # namespace mirroring ``chainer.optimizers``
    Adam, SGD, MomentumSGD, NesterovAG = Adam, SGD, MomentumSGD, NesterovAG


# ---- asr/optimizers.py:3-75 -----------------------------------------------------------------------
def get_learning_rate(opt):
    if isinstance(opt, Adam):
        return opt.alpha
    if isinstance(opt, SGD):
        return opt.lr
    raise NotImplementedError()


def set_learning_rate(opt, lr):
    if isinstance(opt, Adam):
        opt.alpha = lr
        return
    if isinstance(opt, SGD):
        opt.lr = lr
        return
    raise NotImplementedError()


def set_momentum(opt, momentum):
    if isinstance(opt, Adam):
        opt.beta1 = momentum
        return
    if isinstance(opt, (MomentumSGD, NesterovAG)):
        opt.momentum = momentum
        return
    if isinstance(opt, SGD):
        return
    raise NotImplementedError()


def get_optimizer(name, lr, momentum):
    if name == "sgd":
        return SGD(lr=lr)
    if name == "msgd":
        return MomentumSGD(lr=lr, momentum=momentum)
    if name == "nesterov":
        return NesterovAG(lr=lr, momentum=momentum)
    if name == "adam":
        return Adam(alpha=lr, beta1=momentum)
    raise NotImplementedError()


def decay_learning_rate(opt, factor, final_value):
    lr = get_learning_rate(opt)
    if lr <= final_value:
        return final_value
    set_learning_rate(opt, lr * factor)
