"""Data-parallel plumbing on CPU: two gloo ranks sum the flat gradient buffer in slices, in backward order."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeOpt(object):
    """the slice of asr.optimizers.Optimizer the Communicator touches, on CPU tensors"""

    def __init__(self, sizes, rank):
        torch.manual_seed(100 + rank)
        self.params = [torch.nn.Parameter(torch.randn(n)) for n in sizes]
        padded = [(n + 63) // 64 * 64 for n in sizes]
        offs, o = [], 0
        for n in padded:
            offs.append(o)
            o += n
        G = torch.zeros(o)
        P = torch.zeros(o)
        for p, off in zip(self.params, offs):
            P[off:off + p.numel()] = p.data
            p.data = P[off:off + p.numel()]
            p.grad = G[off:off + p.numel()]
        self._flat = dict(P=P, G=G, ids=[id(p) for p in self.params], offsets=offs, sizes=padded)

    def _ensure_flat(self):
        pass


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "chainer-speech-recognition_amd"))
    from asr import link
    from asr.parallel import Communicator
    comm = Communicator("gloo", buckets=3)
    sizes = [100, 3000, 64, 5000, 17, 4096]
    opt = _FakeOpt(sizes, rank)
    # parameters: rank 0's values win
    comm.broadcast(opt._flat["P"])
    p0 = opt._flat["P"].clone()
    # backward: gradients appear last parameter first, each announced through link.grad_buffer
    comm.begin_backward(opt)
    plan = list(comm._plan)
    launched = []
    for i in range(len(sizes) - 1, -1, -1):
        g = link.grad_buffer(opt.params[i])
        launched.append(comm._next)
        g.add_(torch.full_like(g, float(rank + 1) * (i + 1)))
    comm.finish_backward(opt)
    G = opt._flat["G"]
    ok = True
    for i, (p, off) in enumerate(zip(opt.params, opt._flat["offsets"])):
        expect = float(sum(r + 1 for r in range(world)) * (i + 1))
        ok = ok and bool(torch.all(G[off:off + p.numel()] == expect))
    # no slice may be reduced before every parameter inside it has been written
    covered = sorted((b, e) for b, e, _, _ in plan)
    contiguous = covered[0][0] == 0 and all(covered[k][1] == covered[k + 1][0] for k in range(len(covered) - 1)) \
        and covered[-1][1] == G.numel()
    # two backward passes (two half batches): nothing is reduced during the first one, everything once after the second
    opt._flat["G"].zero_()
    comm.begin_backward(opt, passes=2)
    during_first = 0
    for ps in range(2):
        for i in range(len(sizes) - 1, -1, -1):
            g = link.grad_buffer(opt.params[i])
            g.add_(torch.full_like(g, float(rank + 1) * (i + 1) * (ps + 1)))
        if ps == 0:
            during_first = len(comm._pending)
            comm.end_pass()
    comm.finish_backward(opt)
    for i, (p, off) in enumerate(zip(opt.params, opt._flat["offsets"])):
        expect = float(sum(r + 1 for r in range(world)) * (i + 1) * 3)
        ok = ok and bool(torch.all(G[off:off + p.numel()] == expect))
    ok = ok and during_first == 0
    q.put((rank, ok, contiguous, p0.sum().item(), launched, len(plan)))
    comm.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_allreduce_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1] for r in res), "summed gradients wrong"
    assert all(r[2] for r in res), "slices do not tile the flat buffer"
    assert res[0][3] == res[1][3], "broadcast did not equalise the parameters"
    assert res[0][5] >= 2                                   # really sliced
    # slices are launched progressively while 'backward' walks towards the first parameter
    assert res[0][4][0] == 0 and res[0][4][-1] >= 1
