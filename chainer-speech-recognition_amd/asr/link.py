"""Link / Chain / Parameter: the slice of Chainer's object model the reference's `asr.nn` relies on.

The reference builds its models from ``chainer.Link`` / ``chainer.Chain`` (asr/nn/nn.py:240,296,330;
asr/nn/sru.py:441; asr/nn/convolution_2d.py:127).  Here they sit on ``torch.nn.Module`` (parameter
registry, device moves, state_dict); no arithmetic lives in this file.
"""
import contextlib
import math

import torch

from . import _lib

F32 = torch.float32
BF16 = _lib.act_dtype()      # the 16-bit activation format of the loaded library: bfloat16 (default build) or float16 (ASR_ACT=f16)

# bumped by the optimisers after every in-place parameter update done behind torch's back (HIP kernels)
_WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    _WEIGHT_EPOCH[0] += 1


class Parameter(torch.nn.Parameter):
    """``chainer.variable.Parameter``: ``.data`` and ``.grad`` as in Chainer; float32 master copy."""

    def __new__(cls, data=None):
        if data is None:
            data = torch.empty(0, dtype=F32)
        return super().__new__(cls, data, requires_grad=True)


class Link(torch.nn.Module):
    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_compute_cache", {})
        object.__setattr__(self, "_cast_registry", {})

    @contextlib.contextmanager
    def init_scope(self):
        yield

    # -- Chainer-style surface ---------------------------------------------------------------------
    @property
    def xp(self):
        return torch

    def params(self):
        return self.parameters()

    def namedparams(self):
        for name, p in self.named_parameters():
            yield "/" + name.replace(".", "/"), p

    def to_gpu(self, device=None):
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.to(dev)
        return self

    def to_cpu(self):
        self.to("cpu")
        return self

    def cleargrads(self):
        for p in self.parameters():
            if p.grad is not None:
                p.grad = None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """torch's load_state_dict on a freshly built model: a lazily sized parameter (still empty) takes the stored shape
        first, as Chainer's deserialiser initialises uninitialised parameters from the file."""
        for name, p in self._parameters.items():
            src = state_dict.get(prefix + name)
            if p is not None and p.numel() == 0 and src is not None and src.numel() > 0:
                p.data = torch.empty(src.shape, dtype=p.dtype, device=p.device)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        bump_weight_epoch()

    # -- compute copies ------------------------------------------------------------------------------
    def compute_copy(self, key, param, maker, layout=None):
        """bf16 (possibly re-laid-out) copy of a parameter, rebuilt only when the parameter changed.

        ``layout`` declares what ``maker`` does when that is a pure cast, so that ``refresh_compute_copies`` can redo
        every such copy of a model in one launch after an optimiser step: "plain" (same element order), "t_first" /
        "t_last" (the parameter seen as (shape[0], rest) / (rest, shape[-1]), transposed) or "t_each" (a stack of
        matrices, each transposed)."""
        stamp = (_WEIGHT_EPOCH[0], param._version, param.data_ptr())
        hit = self._compute_cache.get(key)
        if hit is not None and hit[0] == stamp:
            _await_copy(hit)
            return hit[1]
        with torch.no_grad():
            value = maker(param.detach())
        self._compute_cache[key] = _made_copy(stamp, value)
        if layout is not None and value.is_cuda:
            self._cast_registry[key] = (param, value, _cast_jobs(param, layout))
        return value


def _made_copy(stamp, value):
    """cache entry (stamp, value, origin, streams that have it): a copy made on one stream may be wanted on another one
    (asr/data/prefetch.py and evaluation passes launch from streams of their own).  No event is recorded when the copy is made -- every record is a
    marker packet the GPU spends 2-3 us on, forty of them behind the optimiser's refresh launch were a 115 us hole at the
    start of each step -- but only when a second stream first asks for the copy: an event recorded THEN on the stream of
    origin still lies behind the launch that made the copy."""
    if not value.is_cuda:
        return (stamp, value, None, None)
    cur = torch.cuda.current_stream()
    return (stamp, value, {"stream": cur, "event": None}, {cur.cuda_stream})


def _await_copy(entry):
    if entry[2] is None:
        return
    cur = torch.cuda.current_stream()
    if cur.cuda_stream not in entry[3]:
        origin = entry[2]
        if origin["event"] is None:
            origin["event"] = torch.cuda.Event()
            origin["event"].record(origin["stream"])
        cur.wait_event(origin["event"])
        entry[1].record_stream(cur)
        entry[3].add(cur.cuda_stream)


def _cast_jobs(param, layout):
    """(source offset, destination offset, rows, cols, transpose) in elements, one per matrix"""
    n, shape = param.numel(), tuple(param.shape)
    if layout == "plain":
        return ((0, 0, n // shape[-1], shape[-1], 0),)
    if layout == "t_first":
        return ((0, 0, shape[0], n // shape[0], 1),)
    if layout == "t_last":
        return ((0, 0, n // shape[-1], shape[-1], 1),)
    if layout == "t_each":
        each = shape[-2] * shape[-1]
        return tuple((i * each, i * each, shape[-2], shape[-1], 1) for i in range(n // each))
    raise ValueError(layout)


def refresh_compute_copies(model):
    """Redo, in place and with ONE launch, the registered bf16 copies of every parameter of ``model`` (the optimisers call
    this right after their update kernel; without it each copy is rebuilt by its own small launch in front of the GEMM
    that needs it).  Copies whose parameter moved since they were made are left to the lazy path."""
    from . import _lib
    todo, sig = [], []
    for m in model.modules():
        reg = getattr(m, "_cast_registry", None)
        if not reg:
            continue
        for key, (param, out, jobs) in reg.items():
            hit = m._compute_cache.get(key)
            if hit is None or hit[1] is not out or hit[0][2] != param.data_ptr() or not param.is_cuda:
                continue
            todo.append((m, key, param, out))
            sig.append((param.data_ptr(), out.data_ptr(), jobs))
    if not todo:
        return 0
    cached = getattr(model, "_cast_table", None)
    if cached is None or cached[0] != sig:
        rows, tile = [], 0
        for src, dst, jobs in sig:
            for so, do, r, c, t in jobs:
                rows.append([src + 4 * so, dst + 2 * do, r, c, t, tile])
                tile += ((r + 63) // 64) * ((c + 63) // 64)
        table = torch.tensor(rows, dtype=torch.int64).to(todo[0][2].device)
        cached = (sig, table, len(rows), tile)
        object.__setattr__(model, "_cast_table", cached)
    _lib.check(_lib.lib().asr_cast_bf16_many(_lib.stream(), _lib.ptr(cached[1]), cached[2], cached[3]), "asr_cast_bf16_many")
    for m, key, param, out in todo:
        m._compute_cache[key] = _made_copy((_WEIGHT_EPOCH[0], param._version, param.data_ptr()), out)
    return cached[2]


class Chain(Link):
    pass


_GRAD_LISTENER = [None]      # asr.parallel.Communicator hooks in here to overlap the all-reduce with backward


def grads_queued(*params):
    """A backward function calls this AFTER it has queued the last kernel (on the launch stream or a side stream) that
    writes the gradients of `params`: from here on a data-parallel all-reduce of them may be queued (asr/parallel.py)."""
    if _GRAD_LISTENER[0] is not None:
        _GRAD_LISTENER[0]([p for p in params if p is not None])


def grad_buffer(param):
    """The f32 buffer the backward kernels accumulate this parameter's gradient into (param.grad)."""
    if param.grad is None:
        from . import _ops
        g = torch.empty_like(param, memory_format=torch.contiguous_format)
        _ops.fill_(g.detach(), 0.0)
        param.grad = g
    return param.grad


# ---------------------------------------------------------------------------------------------- initialisers
# (parameter initialisation happens once on the host; plain torch RNG, no kernels of ours involved)
class Initializer(object):
    def __call__(self, shape):
        raise NotImplementedError


class Constant(Initializer):
    def __init__(self, value):
        self.value = value

    def __call__(self, shape):
        return torch.full(shape, float(self.value), dtype=F32)


class Normal(Initializer):
    """chainer.initializers.Normal(scale) -- used by run/ctc/cnn/model.py:145,162,170,182."""

    def __init__(self, scale=0.05):
        self.scale = scale

    def __call__(self, shape):
        return torch.randn(shape, dtype=F32) * self.scale


def _fans(shape):
    fan_out = shape[0]
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return fan_in, fan_out


class LeCunNormal(Initializer):
    """Chainer's default weight initialiser: N(0, scale * sqrt(1 / fan_in))."""

    def __init__(self, scale=1.0):
        self.scale = scale

    def __call__(self, shape):
        fan_in, _ = _fans(shape)
        return torch.randn(shape, dtype=F32) * (self.scale * math.sqrt(1.0 / fan_in))


class HeNormal(Initializer):
    """chainer.initializers.HeNormal(scale): N(0, scale * sqrt(2 / fan_in))  (asr/nn/nn.py:269-270)."""

    def __init__(self, scale=1.0):
        self.scale = scale

    def __call__(self, shape):
        fan_in, _ = _fans(shape)
        return torch.randn(shape, dtype=F32) * (self.scale * math.sqrt(2.0 / fan_in))


class Uniform(Initializer):
    def __init__(self, scale=0.05):
        self.scale = scale

    def __call__(self, shape):
        return (torch.rand(shape, dtype=F32) * 2 - 1) * self.scale


def get_initializer(init):
    """chainer.initializers._get_initializer: None -> LeCunNormal, number -> Constant, tensor -> copy."""
    if init is None:
        return LeCunNormal()
    if isinstance(init, Initializer):
        return init
    if isinstance(init, (int, float)):
        return Constant(init)
    if isinstance(init, torch.Tensor):
        value = init.detach().to(F32)
        return lambda shape: value.reshape(shape).clone()
    if callable(init):
        return init
    raise TypeError("unsupported initialiser %r" % (init,))


class initializers(object):     # namespace mirroring ``chainer.initializers``
    Normal = Normal
    HeNormal = HeNormal
    LeCunNormal = LeCunNormal
    Constant = Constant
    Uniform = Uniform
    _get_initializer = staticmethod(get_initializer)
