"""Data-parallel plumbing on CPU: two gloo ranks sum the flat gradient buffer in slices, in backward order, and no slice
is ever reduced before every kernel that writes into it has been queued."""
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


def _env(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "chainer-speech-recognition_amd"))


def _worker(rank, world, port, q):
    _env(rank, world, port)
    from asr import link
    from asr.parallel import Communicator
    comm = Communicator("gloo", buckets=3)
    sizes = [100, 3000, 64, 5000, 17, 4096]
    opt = _FakeOpt(sizes, rank)
    # parameters: rank 0's values win
    comm.broadcast(opt._flat["P"])
    p0 = opt._flat["P"].clone()
    # backward: gradients appear last parameter first; each is announced AFTER it has been written
    comm.begin_backward(opt)
    plan = list(comm._plan)
    launched = []
    for i in range(len(sizes) - 1, -1, -1):
        g = link.grad_buffer(opt.params[i])
        g.add_(torch.full_like(g, float(rank + 1) * (i + 1)))
        link.grads_queued(opt.params[i])
        launched.append(comm._next)
    comm.finish_backward(opt)
    G = opt._flat["G"]
    ok = True
    for i, (p, off) in enumerate(zip(opt.params, opt._flat["offsets"])):
        expect = float(sum(r + 1 for r in range(world)) * (i + 1))
        ok = ok and bool(torch.all(G[off:off + p.numel()] == expect))
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
            link.grads_queued(opt.params[i])
        if ps == 0:
            during_first = len(comm._pending)
            comm.end_pass()
    comm.finish_backward(opt)
    for i, (p, off) in enumerate(zip(opt.params, opt._flat["offsets"])):
        expect = float(sum(r + 1 for r in range(world)) * (i + 1) * 3)
        ok = ok and bool(torch.all(G[off:off + p.numel()] == expect))
    ok = ok and during_first == 0
    # a parameter that gets no gradient at all (unused layer) does not hold its slice back for ever
    opt._flat["G"].zero_()
    comm.begin_backward(opt)
    for i in (5, 4, 2, 1, 0):
        g = link.grad_buffer(opt.params[i])
        g.add_(1.0)
        link.grads_queued(opt.params[i])
    comm.finish_backward(opt)
    ok = ok and bool(torch.all(opt.params[0].grad == world)) and bool(torch.all(opt.params[3].grad == 0))
    q.put((rank, ok, contiguous, p0.sum().item(), launched, len(plan)))
    comm.barrier()
    torch.distributed.destroy_process_group()


def _run(target, world, extra=()):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res.sort()
    return res


@pytest.mark.parametrize("world", [2, 8])
def test_two_rank_gradient_allreduce_gloo(world):
    """(the name is round 1's; world 8 = the width BASELINE configs[2] runs at: VERDICT r4 next 3a)"""
    res = _run(_worker, world)
    assert len(res) == world
    assert all(r[1] for r in res), "summed gradients wrong"
    assert all(r[2] for r in res), "slices do not tile the flat buffer"
    assert all(r[3] == res[0][3] for r in res), "broadcast did not equalise the parameters"
    assert res[0][5] >= 2                                   # really sliced
    # without recurrences, slices are launched progressively while 'backward' walks towards the first parameter
    assert res[0][4][0] >= 0 and res[0][4][-1] >= 1 and res[0][4][1] < res[0][4][-1] + 1


# ---------------------------------------------------------------------------------------------- the BASELINE plan
def _ds2_parameters():
    """(name, size) of asr.model.ds2.Model at BASELINE configs[1] in registration order (13.85 M parameters)"""
    out = [("conv0.W", 128 * 3 * 15), ("conv0.b", 128), ("conv1.W", 128 * 64 * 15), ("conv1.b", 128)]
    for layer in range(4):
        din = 384 if layer == 0 else 512
        out += [("gru%d.w_ih" % layer, 2 * 1536 * din), ("gru%d.w_hh" % layer, 2 * 1536 * 512),
                ("gru%d.b_ih" % layer, 2 * 1536), ("gru%d.b_hh" % layer, 2 * 1536)]
    out += [("dense0.W", 640 * 512), ("dense0.b", 640), ("dense1.W", 640 * 320), ("dense1.b", 640),
            ("dense2.W", 3000 * 320), ("dense2.b", 3000), ("ln.gamma", 3000), ("ln.beta", 3000)]
    return out


def _plan_worker(rank, world, port, q, buckets, beside=False):
    """replays the order in which asr/functions.py queues kernels and announces gradients during the backward pass of the
    BASELINE model -- including _GRU.backward's bias-gradients-first / w_ih-before-w_hh order and the recurrence hooks --
    and checks at every all-reduce launch that every writer of the slice has been queued"""
    _env(rank, world, port)
    from asr import link, _ops
    from asr.parallel import Communicator
    comm = Communicator("gloo", buckets=buckets, beside_recurrences=beside)
    names = _ds2_parameters()
    opt = _FakeOpt([n for _, n in names], rank)
    index = {name: i for i, (name, _) in enumerate(names)}
    written, violations, launches, in_recurrence = set(), [], [], [False]
    real_launch = comm._launch

    def checked_launch(k, events=()):
        lo, hi = comm._plan[k][2], comm._plan[k][3]
        missing = [names[i][0] for i in range(lo, hi + 1) if i not in written]
        if missing:
            violations.append((k, missing))
        launches.append((k, len(written), in_recurrence[0]))
        real_launch(k, events)

    comm._launch = checked_launch

    def write(*ps):                     # "queue the kernels that write these gradients"
        for n in ps:
            link.grad_buffer(opt.params[index[n]]).add_(1.0)
            written.add(index[n])

    def announce(*ps):
        link.grads_queued(*[opt.params[index[n]] for n in ps])

    for step in range(2):
        written.clear()
        opt._flat["G"].zero_()
        comm.begin_backward(opt)
        nslices = len(comm._plan)
        write("ln.gamma", "ln.beta"); announce("ln.gamma", "ln.beta")
        for d in (2, 1, 0):
            write("dense%d.W" % d, "dense%d.b" % d); announce("dense%d.W" % d, "dense%d.b" % d)
        for layer in (3, 2, 1, 0):
            g = "gru%d." % layer
            _ops._hook("before")
            in_recurrence[0] = True
            write(g + "b_ih", g + "b_hh")           # the recurrence kernel accumulates the bias gradients
            in_recurrence[0] = False
            _ops._hook("after")
            write(g + "w_ih"); write(g + "w_hh")    # side-stream GEMMs, queued after the hook
            announce(g + "w_ih", g + "w_hh", g + "b_ih", g + "b_hh")
        for c in (1, 0):
            write("conv%d.W" % c, "conv%d.b" % c); announce("conv%d.W" % c, "conv%d.b" % c)
        comm.finish_backward(opt)
        ok = bool(torch.all(opt._flat["G"][:16] == world))
    q.put((rank, violations, launches, nslices, ok))
    comm.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("buckets,world", [(1, 2), (4, 2), (7, 2), (40, 2), (4, 8)])
def test_baseline_plan_never_reduces_a_slice_before_its_writers(buckets, world):
    res = _run(_plan_worker, world, (buckets,))
    assert len(res) == world
    for rank, violations, launches, nslices, ok in res:
        assert violations == [], violations
        assert ok
        assert sorted(k for k, _, _ in launches[-nslices:]) == list(range(nslices))     # every slice exactly once per step
        assert not any(inside for _, _, inside in launches)
    assert all(r[2] == res[0][2] for r in res), "ranks issued their collectives in different orders"
    if buckets in (4, 7):
        # second step (the number of recurrences is known from the first): behind the last recurrence slices go the moment they are
        # complete -- everything but the small front slice (the convolutions) is on its way BEFORE the convolutions' gradients
        # are written, i.e. it is summed beside their backward pass and not after it
        _, _, launches, nslices, _ = res[0]
        second = launches[-nslices:]
        n_params = len(_ds2_parameters())
        assert [k for k, _, _ in second] == list(range(nslices))
        assert all(written <= n_params - 4 for k, written, _ in second[:-1]), second
        assert second[-1][1] == n_params
        first = launches[:nslices]                  # first step: nothing known yet, the first GRU layer's slice waits for the end
        assert first[-2][1] == n_params


def test_slices_may_go_beside_recurrences_on_request():
    """Communicator(beside_recurrences=True) (the measured alternative of DESIGN.md section 13.5): no launch waits for a recurrence boundary
    -- already in the FIRST step every slice but the front one is on its way before the convolutions' gradients are written --, the
    order is still the same on every rank and no slice goes before its writers."""
    res = _run(_plan_worker, 2, (4, True))
    for rank, violations, launches, nslices, ok in res:
        assert violations == [] and ok
        assert sorted(k for k, _, _ in launches[-nslices:]) == list(range(nslices))
    assert res[0][2] == res[1][2]
    _, _, launches, nslices, _ = res[0]
    n_params = len(_ds2_parameters())
    first = launches[:nslices]
    assert all(written <= n_params - 4 for k, written, _ in first[:-1]), first


def test_plan_keeps_the_last_slice_small():
    """the slice at the front of the buffer is summed after everything else has been queued: make_plan gives it the leading parameters
    within 1/32 of the buffer only (BASELINE model: the two convolutions), the slices tile the buffer and follow parameter bounds"""
    from asr.parallel import Communicator
    sizes = [n for _, n in _ds2_parameters()]
    offs, o = [], 64
    for n in sizes:
        offs.append(o)
        o += n
    for buckets in (1, 2, 3, 4, 7, 40):
        plan = Communicator.make_plan(offs, sizes, buckets)
        assert plan[-1][0] == 0 and plan[0][1] == o and plan[-1][2] == 0
        for a, c in zip(plan[:-1], plan[1:]):
            assert a[0] == c[1] and a[2] == c[3] + 1
        assert len(plan) <= max(1, buckets) + 1
        if buckets > 1:
            assert plan[-1][3] == 3 and plan[-1][1] - 64 == sum(sizes[:4])         # conv0.W, conv0.b, conv1.W, conv1.b
    assert Communicator.make_plan([64], [10], 4) == [(0, 74, 0, 0)]
    assert Communicator.make_plan([64, 74], [10, 10], 4) == [(74, 84, 1, 1), (0, 74, 0, 0)]


HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config")


@pytest.mark.parametrize("world", [2, 8])
def test_bench_starts_its_own_ranks(world, tmp_path):
    """VERDICT r2 (missing 2): `python bench.py --gpus N` from a bare shell (no WORLD_SIZE) starts N fresh rank processes through
    torch.distributed.run, relays rank 0's JSON line and exits with their status.  ASR_BENCH_REHEARSE=1 runs the whole multi-rank
    control flow -- rendezvous on 127.0.0.1, gloo, barrier + max-over-ranks timing, the line -- on CPU tensors without a GPU.
    VERDICT r4 next 1 / 3: the line is the LAST stdout line, under 4 KB, and names every rank that took part."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ASR_BENCH_REHEARSE"] = "1"
    env["ASR_BENCH_DETAIL"] = str(tmp_path / "bench_detail.json")
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[-1].startswith("{"), out.stdout
    assert len(lines[-1].encode()) < 4096
    line = json.loads(lines[-1])
    for key in HEADLINE_KEYS:
        assert key in line, key
    assert line["n_gpus"] == world and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 32 * world and line["config"]["parallelism"] == "dp%d" % world
    assert len(line["per_rank_ms_per_step"]) == world
    assert abs(line["ms_per_step"] - max(line["per_rank_ms_per_step"])) < 1e-6 * line["ms_per_step"] + 1e-9     # MAX over ranks
    ranks = line["ranks"]
    assert ranks["backend"] == "gloo" and ranks["world_size"] == world
    assert sorted(d["rank"] for d in ranks["devices"]) == list(range(world))
    assert len(set(d["pid"] for d in ranks["devices"])) == world                # really N processes
    detail = json.load(open(env["ASR_BENCH_DETAIL"]))
    assert detail["config"]["check"] == world * (world + 1) / 2.0               # 1 + 2 + ... : the ranks did exchange data


def test_headline_of_a_full_record_fits_the_driver():
    """VERDICT r4 (missing 1): the round-4 record was 23 KB and the driver could not parse it.  The round-4 record itself, pushed through
    bench.headline, gives a line under 4 KB that keeps roofline / cpu_baseline / parity and drops the tables."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(root, "profiles", "r04_bench_final.json")))
    assert len(json.dumps(full)) > 8354                                          # what broke the driver's capture
    text = bench.headline(full)
    assert len(text.encode()) < bench.HEADLINE_MAX_BYTES <= 4096 and "\n" not in text
    line = json.loads(text)
    for key in HEADLINE_KEYS + ("roofline", "roofline_gemm", "roofline_ctc_sweep", "cpu_baseline", "parity", "gpu_vs_cpu"):
        assert key in line, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    assert line["parity"]["pass"] is True and line["parity"]["loss_rel"] is not None
    assert "shapes" not in line["roofline_gemm"] and "extra_configs" not in line and "kernel_ms_per_step" not in line
    assert abs(line["value"] - full["value"]) < 1e-3 * full["value"]
    # a record padded far beyond anything real still yields a parseable line under the limit
    full["config"]["workload"] = full["config"]["workload"] * 40
    text = bench.headline(full)
    assert len(text.encode()) < 4096 and json.loads(text)["value"] == line["value"]


def test_bench_line_quotes_pmc_traffic_only_from_the_same_kernel_sources(tmp_path):
    """`roofline.traffic` comes from a committed rocprofv3 --pmc pass; a pass taken on other kernel sources must not end
    up in a fresh line -- bench.pmc_traffic_of_this_build compares the sha256 the pass recorded with the sources that are there now."""
    import hashlib
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_traffic", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    src = tmp_path / "chainer-speech-recognition_amd" / "csrc"
    src.mkdir(parents=True)
    (src / "gru.hip").write_text("kernel v1")
    (tmp_path / "profiles").mkdir()
    bench.ROOT = str(tmp_path)
    assert bench.pmc_traffic_of_this_build()[0] is None                                   # nothing there
    kernels = {"asr::gru::fwd_persistent_io_kernel<4>": {"dispatches": 4, "hbm_bytes_per_dispatch": 100.0},
               "asr::gru::bwd_ps_kernel<2>": {"dispatches": 4, "hbm_bytes_per_dispatch": 300.0}, "other": {"dispatches": 9, "hbm_bytes_per_dispatch": 1e9}}
    rec = {"kernels": kernels, "source_sha256": {"gru.hip": hashlib.sha256(b"kernel v1").hexdigest()}}
    (tmp_path / "profiles" / "r05_pmc_traffic.json").write_text(json.dumps(rec))
    traffic, source = bench.pmc_traffic_of_this_build()
    assert traffic == 200.0 and source == "profiles/r05_pmc_traffic.json"
    (src / "gru.hip").write_text("kernel v2")                                             # the kernel changed: the pass is stale
    traffic, source = bench.pmc_traffic_of_this_build()
    assert traffic is None and "stale" in source
    (tmp_path / "profiles" / "r04_pmc_traffic.json").write_text(json.dumps({"kernels": kernels}))      # an old pass without a hash: never quoted
    (src / "gru.hip").write_text("kernel v1")
    assert bench.pmc_traffic_of_this_build()[1] == "profiles/r05_pmc_traffic.json"
