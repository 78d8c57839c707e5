#!/usr/bin/env python3
"""Headline benchmark: utterances/sec of one CTC train step (T=1000, 40x3 features, |V|=3000) on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward (2 x conv + 4 x BiGRU-512 + dense + LayerNorm) + CTC loss + backward + gradient all-reduce (N > 1) +
GradientClipping(1) + WeightDecay(1e-5) + Adam -- BASELINE.json configs[1] (configs[2] per GPU when N > 1), synthetic
batch resident in HBM, random-init weights.  One JSON line on rank 0 (see the task contract): value = utterances of
all ranks / max-over-ranks time; `roofline` = the dominant kernel by time, measured with HIP events in this process;
`cpu_baseline` = the CPU oracle (oracle/model.py, torch-CPU fp32) timed on this host on a bounded sample (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PARITY_GATE_LOSS, PARITY_GATE_GRAD = 1e-4, 4e-2      # the line's own gate against the rounding-matched oracle (measured: 5e-6 / 1.1e-2; fp32 oracle: 3e-4 / 5.3e-2)
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


def log(msg):
    sys.stderr.write("[bench] %s\n" % msg)
    sys.stderr.flush()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU (BASELINE: 32)")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--vocab", type=int, default=None, help="default 3000 (ds2) / 119 (cnn: the reference's unigram inventory)")
    ap.add_argument("--config", choices=("ds2", "cnn"), default="ds2",
                    help="ds2: BASELINE configs[1] (the metric's configuration, default); cnn: BASELINE configs[4], the fully "
                         "convolutional `zhang+residual` recipe of run/ctc/cnn/model.py:142-204 (V=119 unless --vocab is given)")
    ap.add_argument("--num-conv-layers", type=int, default=4,
                    help="--config cnn: 4 = the reference's default depth (164 GFLOP/utterance); > 4 = the wide 'VGG-deep' "
                         "branch, always 8 conv layers (822 GFLOP/utterance): run/ctc/cnn/model.py:153-157,177-187")
    ap.add_argument("--cnn-extra", type=int, default=0, help="(used by the default run for its fp16 lines) time ONLY the configs[4] recipe with "
                    "this many conv layers in this process -- with the library ASR_ACT selects -- and print its extra_configs entry as JSON")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-census", action="store_true")
    ap.add_argument("--census", action="store_true", help="N > 1: run the single-stream kernel census on rank 0 after the timed region as the "
                                                         "N = 1 line does (default at N > 1: skipped, the line is the scaling point only)")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_configs (Gram-CTC at configs[3] size, the two configs[4] CNN steps, "
                                                           "the ragged-length step)")
    ap.add_argument("--ragged", action="store_true", help="time the step on the ragged batch of SURVEY 8d (x_length ~ U{600..T}, "
                                                          "length-exact recurrences) instead of full-length utterances")
    return ap.parse_args()


# the GEMM-class entries of the census (gemm_tn_acc_group: the grouped weight gradients of a GRU layer, one launch)
GEMM_OPS = ("gemm_nt", "gemm_tn_acc", "gemm_tn_acc_group", "conv_nt", "conv_tn_acc", "conv_mp_fwd", "conv_mp_bwd")


class Census(object):
    """Per-op-class device time of ONE extra step, HIP events recorded on the launch stream around every C-ABI call."""

    def __init__(self):
        self.events = []

    def wrap(self, ops):
        self._orig = {}
        for name in ("gemm_nt", "gemm_tn_acc", "gemm_tn_acc_group", "gru_fwd", "gru_bwd", "im2col", "col2im", "layernorm_fwd", "layernorm_bwd", "layernorm_ctc_bwd",
                     "maxout2_fwd", "maxout2_bwd", "maxpool_h_fwd", "maxpool_h_bwd", "colsum_acc", "adam_ctl",
                     "step_control", "fill_", "cast_bf16", "conv_weight_pack", "conv_weight_grad_unpack", "conv_nt", "pack_input_pad",
                     "conv_weight_pack_bwd", "conv_tn_acc", "maxout2_pool_fwd", "maxout2_pool_bwd", "conv_mp_fwd", "conv_mp_bwd"):
            fn = getattr(ops, name)
            self._orig[name] = fn
            setattr(ops, name, self._timed(name, fn))
        self._ops = ops

    def _timed(self, name, fn):
        def inner(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            self.events.append((name, e0, e1, gemm_shape(name, a, k, r) if name in GEMM_OPS else None))
            return r
        return inner

    def unwrap(self):
        for name, fn in self._orig.items():
            setattr(self._ops, name, fn)

    def totals(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, _ in self.events:
            ms, n = out.get(name, (0.0, 0))
            out[name] = (ms + e0.elapsed_time(e1), n + 1)
        return out

    def gemm_shapes(self):
        """one entry per distinct GEMM-class product of the step: ms per call against BOTH floors (half of these
        shapes are HBM-bound, a line priced against MFMA only cannot show it) -- flops / 2.5 PFLOP/s (dense bf16 MFMA peak) and
        algorithmic bytes (each operand read once, the result written once) / 6.3 TB/s (what a streaming copy reaches on MI355X);
        `frac` = the larger floor / the measured time = fraction of the roof that binds that shape"""
        torch.cuda.synchronize()
        agg = {}
        for name, e0, e1, sh in self.events:
            if sh is None:
                continue
            key = (name,) + sh["key"]
            ent = agg.setdefault(key, {"op": name, "form": sh["form"], "M": sh["M"], "N": sh["N"], "K": sh["K"], "out": sh["out"], "calls": 0,
                                       "ms": 0.0, "flops": sh["flops"], "bytes": sh["bytes"]})
            ent["calls"] += 1
            ent["ms"] += e0.elapsed_time(e1)
        out = []
        for ent in agg.values():
            ms = ent["ms"] / ent["calls"]
            f_ms = ent["flops"] / (MFMA_BF16_PEAK_TFLOPS * 1e12) * 1e3
            b_ms = ent["bytes"] / (HBM_ACHIEVABLE_GBPS * 1e9) * 1e3
            out.append({"op": ent["op"], "form": ent["form"], "M": ent["M"], "N": ent["N"], "K": ent["K"], "out": ent["out"], "calls": ent["calls"],
                        "ms_per_call": round(ms, 4), "mfma_floor_ms": round(f_ms, 4), "hbm_floor_ms": round(b_ms, 4),
                        "bound": "mfma" if f_ms >= b_ms else "hbm", "frac": round(max(f_ms, b_ms) / ms, 3),
                        "tflops": round(ent["flops"] / (ms * 1e-3) / 1e12, 1), "gbps": round(ent["bytes"] / (ms * 1e-3) / 1e9, 0)})
        out.sort(key=lambda e: -e["ms_per_call"] * e["calls"])
        return out


HBM_ACHIEVABLE_GBPS = 6300.0        # MI355X_MICROARCH.md: 6.29 TB/s measured float4 copy (79 % of the 8 TB/s specification)


def gemm_shape(name, a, k, result):
    """(M, N, K), flops and algorithmic bytes of one GEMM-class C-ABI call from the arguments of its asr._ops wrapper"""
    esz = lambda t: t.element_size()
    if name == "gemm_nt":
        A, Bm = a[0], a[1]
        M, K = A.shape
        N = Bm.shape[0]
        ob = esz(result)
        return {"key": (M, N, K, ob), "form": "NT", "M": M, "N": N, "K": K, "out": "f32" if ob == 4 else "bf16", "flops": 2.0 * M * N * K,
                "bytes": 2.0 * (M * K + N * K) + float(ob) * M * N}
    if name == "gemm_tn_acc":
        A, Bm = a[0], a[1]
        K, M = A.shape
        N = Bm.shape[1]
        return {"key": (M, N, K, 4), "form": "TN (split-K, f32 atomics)", "M": M, "N": N, "K": K, "out": "f32 +=", "flops": 2.0 * M * N * K,
                "bytes": 2.0 * K * (M + N) + 4.0 * M * N}
    if name == "gemm_tn_acc_group":
        prods = a[0]
        fl = sum(2.0 * p[0].shape[1] * p[1].shape[1] * p[0].shape[0] for p in prods)
        by = sum(2.0 * p[0].shape[0] * (p[0].shape[1] + p[1].shape[1]) + 4.0 * p[0].shape[1] * p[1].shape[1] for p in prods)
        desc = tuple((p[0].shape[1], p[1].shape[1], p[0].shape[0]) for p in prods)
        return {"key": desc, "form": "TN grouped: " + " + ".join("%dx%dx%d" % d for d in desc), "M": sum(d[0] for d in desc), "N": desc[0][1],
                "K": desc[0][2], "out": "f32 +=", "flops": fl, "bytes": by}
    if name == "conv_nt":
        x, W2, _, _, KH, KW, _, _, sgn, Tr, Hr = a[:11]
        Ts, Bn, Hs, Cs = x.shape
        M, N, K = Tr * Bn * Hr, W2.shape[0], KH * KW * Cs
        ob = esz(result)
        return {"key": (M, N, K, ob, int(sgn)), "form": "implicit conv NT (%s)" % ("forward" if sgn > 0 else "backward-data"), "M": M, "N": N, "K": K,
                "out": "f32" if ob == 4 else "bf16", "flops": 2.0 * M * N * K, "bytes": 2.0 * x.numel() + 2.0 * W2.numel() + float(ob) * M * N}
    if name == "conv_tn_acc":
        g2, x, scratch, KH, KW = a[:5]
        Ts, Bn, Hs, Cs = x.shape
        M, N, K = g2.shape[1], KH * KW * Cs, g2.shape[0]
        return {"key": (M, N, K, 4), "form": "implicit conv TN (weight gradient)", "M": M, "N": N, "K": K, "out": "f32 +=", "flops": 2.0 * M * N * K,
                "bytes": 2.0 * g2.numel() + 2.0 * x.numel() + 4.0 * M * N}
    if name == "conv_mp_fwd":           # first block in one pass: the convolution's product; only the pooled output and the winners are written
        x8, W2, _, KH, KW, _, _, Tout, Hout, _ = a[:10]
        M, N, K = Tout * x8.shape[1] * Hout, W2.shape[0], KH * KW * 8
        y, idx = result
        return {"key": (M, N, K, 2, 3), "form": "first block forward: conv + maxout + pooling, one pass", "M": M, "N": N, "K": K, "out": "bf16 pooled",
                "flops": 2.0 * M * N * K, "bytes": 2.0 * x8.numel() + 2.0 * W2.numel() + 2.0 * y.numel() + 1.0 * idx.numel()}
    if name == "conv_mp_bwd":
        gy, idx, x8, gW, _, KH, KW, _, _, Hout, _ = a[:11]
        M, N, K = gW.shape[0], KH * KW * 8, gy.shape[0] * gy.shape[1] * Hout
        return {"key": (M, N, K, 4, 3), "form": "first block backward: weight + bias gradient from the pooled gradient", "M": M, "N": N, "K": K,
                "out": "f32 +=", "flops": 2.0 * M * N * K, "bytes": 2.0 * gy.numel() + 1.0 * idx.numel() + 2.0 * x8.numel() + 4.0 * M * N}
    return None


def time_ctc(lib_mod, ops, T, B, V, L, x_len, l_len, labels, dev):
    """HIP-event time of the CTC kernels alone (forward: prep+rows+lattice, backward: grad)."""
    from asr import _lib
    lib = _lib.lib()
    xs = torch.randn(T, B, V, device=dev)
    n = lib.asr_ctc_workspace_bytes(T, B, V, L, 0)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    loss = torch.empty(B, device=dev)
    grad = torch.empty_like(xs)
    s = torch.cuda.current_stream()

    def fwd():
        assert lib.asr_ctc_forward(s.cuda_stream, xs.data_ptr(), labels.data_ptr(), None, x_len.data_ptr(), l_len.data_ptr(), T, B, V,
                                   L, 0, loss.data_ptr(), None, ws.data_ptr(), n) == 0

    def bwd():
        assert lib.asr_ctc_backward(s.cuda_stream, xs.data_ptr(), x_len.data_ptr(), T, B, V, L, 0, None, 0, 1.0 / B, grad.data_ptr(),
                                    ws.data_ptr(), n) == 0
    res = {}
    for name, fn in (("ctc_forward", fwd), ("ctc_grad", bwd)):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            fn()
        e1.record(s)
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 5
    return res


def cpu_baseline(cfg, T, V, dev=None, seconds_budget=25.0):
    """The CPU oracle (torch-CPU fp32 restatement, oracle/model.py) on a bounded sample of the same workload, and -- with
    `dev` -- the parity of the HIP path against it on that very sample: the SAME B=4 batch and the SAME initial parameters go
    through one forward + CTC + backward on both sides (the step of run/ctc/cnn/train.py:190-200 before the update)."""
    from asr.model import ds2
    from oracle import model as omodel
    torch.manual_seed(0)
    B = 4
    m = ds2.Model(cfg)
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, seed=0)
    # materialise the lazily-sized parameters with the shapes the HIP model would infer
    feat = cfg.ndim_conv * 6
    m.rnn_blocks.layers[0]._initialize_params(feat)
    for i in range(1, cfg.num_rnn_layers):
        m.rnn_blocks.layers[2 * i]._initialize_params(cfg.ndim_rnn)
    m.dense_blocks.layers[0]._initialize_params(cfg.ndim_rnn)
    m.dense_blocks.layers[7].norm._initialize_params(V)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ref = omodel.DS2Oracle(state, cfg.num_conv_layers, cfg.num_rnn_layers, cfg.bidirectional)
    mm = [torch.zeros_like(p) for p in ref.parameters()]
    vv = [torch.zeros_like(p) for p in ref.parameters()]
    # the GPU box shares its host: use the CPUs this process may run on, at most 16 (the 1-GPU share)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail, torch.get_num_threads()))
    torch.set_num_threads(cores)
    log("cpu_baseline: %d threads" % cores)
    parity = None
    t0 = time.time()
    if dev is not None:
        # warm-up pass of the oracle = the parity pass: forward + loss + backward, nothing updated yet
        logits_ref = ref(x)
        loss_ref = omodel.ctc_mean_loss(logits_ref, labels, x_len, l_len)
        loss_ref.backward()
        from asr.loss import connectionist_temporal_classification
        gpu = ds2.Model(cfg)
        gpu.load_state_dict(state)
        gpu.to_gpu(dev.index)
        ys = gpu(x.to(dev))
        loss = connectionist_temporal_classification(ys, labels.to(dev), 0, x_len.to(dev), l_len.to(dev))
        loss.backward()
        from asr.functions import join_side_stream
        join_side_stream()
        torch.cuda.synchronize()

        def cos(a, b):
            a, b = a.double().flatten(), b.double().flatten()
            return float(a @ b / (a.norm() * b.norm() + 1e-30))

        def rel(a, b):
            a, b = a.double().flatten(), b.double().flatten()
            return float((a - b).norm() / (b.norm() + 1e-30))
        logits = torch.stack(tuple(ys)).detach().float().cpu()
        worst, worst_name, norm_ratio, rel32 = 1.0, None, [], {}
        for name, p in gpu.named_parameters():
            gr = ref.g(name).grad
            c = cos(p.grad.detach().cpu(), gr)
            rel32[name] = rel(p.grad.detach().cpu(), gr)
            norm_ratio.append(float(p.grad.detach().cpu().double().norm() / (gr.double().norm() + 1e-30)))
            if c < worst:
                worst, worst_name = c, name
        # the gate: the SAME step on the rounding-matched oracle (oracle/bf16.py: bf16 roundings where the device rounds, float32
        # accumulation) -- what is left is summation order, the fast exp / rcp of the gate math and float32-vs-float64 CTC
        from asr import _ops as _o
        H = cfg.ndim_rnn
        refm = omodel.DS2Oracle(state, cfg.num_conv_layers, cfg.num_rnn_layers, cfg.bidirectional, matched=True,
                                gi_bf16=(_o.gru_gi_dtype(T, B, H, 2 if cfg.bidirectional else 1) == torch.bfloat16),
                                fused_logit_bias=(V % 4 == 0), gates_f16=_o.gru_gates_f16(T, B, H, 2 if cfg.bidirectional else 1))
        logits_m = refm(x)
        loss_m = omodel.ctc_mean_loss(logits_m, labels, x_len, l_len)
        loss_m.backward()
        relm = {name: rel(p.grad.detach().cpu(), refm.g(name).grad) for name, p in gpu.named_parameters()}
        wm = max(relm, key=relm.get)
        w32 = max(rel32, key=rel32.get)
        gate = {"loss_rel_max": PARITY_GATE_LOSS, "grad_rel_l2_max": PARITY_GATE_GRAD}
        loss_rel_m = abs(float(loss.item()) - float(loss_m.item())) / abs(float(loss_m.item()))
        parity = {"against": "oracle/model.py on the same B=%d batch and initial parameters, T=%d, V=%d: (matched) bf16 roundings where the "
                             "device rounds -- the gate; (fp32) plain float32 -- the price of bf16, reported" % (B, T, V),
                  "loss_gpu": float(loss.item()), "loss_oracle_matched": float(loss_m.item()), "loss_oracle_fp32": float(loss_ref.item()),
                  "matched": {"loss_rel": loss_rel_m, "logits_rel_l2": rel(logits, logits_m.detach()),
                              "worst_param_grad_rel_l2": relm[wm], "worst_param": wm},
                  "fp32": {"loss_rel": abs(float(loss.item()) - float(loss_ref.item())) / abs(float(loss_ref.item())),
                           "logits_cos": cos(logits, logits_ref.detach()), "worst_param_grad_cos": worst, "worst_param_cos": worst_name,
                           "worst_param_grad_rel_l2": rel32[w32], "worst_param": w32,
                           "grad_norm_ratio_min_max": [min(norm_ratio), max(norm_ratio)]},
                  "gate": gate, "pass": bool(loss_rel_m <= PARITY_GATE_LOSS and relm[wm] <= PARITY_GATE_GRAD),
                  "note": "the CTC kernel itself is held to 1e-4 on identical logits (tests/test_ctc_gpu.py)"}
        del refm
        del gpu, ys, loss
        for p in ref.parameters():
            p.grad = None
    omodel.train_step(ref, mm, vv, 1, x, labels, x_len, l_len)           # warm-up (allocations, oneDNN primitives)
    warm = time.time() - t0
    log("cpu_baseline: warm-up %.1f s" % warm)
    t0 = time.time()
    omodel.train_step(ref, mm, vv, 2, x, labels, x_len, l_len)
    one = time.time() - t0
    n = max(1, min(5, int(seconds_budget / max(one, 1e-3)) - 1))
    t0 = time.time()
    for s in range(n):
        omodel.train_step(ref, mm, vv, 3 + s, x, labels, x_len, l_len)
    dt = (time.time() - t0) / n
    out = {"value": B / dt, "unit": "utterances/s", "cores": cores, "kind": "port",
           "sample": "%d train steps of B=%d utterances (T=%d, V=%d, same model, fp32, torch-CPU oracle) after 2 warm-up; %.2f s/step"
                     % (n, B, T, V, dt)}
    # also n = 1 thread (one utterance, one timed step after one warm-up: ~20 s)
    torch.set_num_threads(1)
    x1, l1, xl1, ll1 = x[:1], labels[:1], x_len[:1], l_len[:1]
    omodel.train_step(ref, mm, vv, 9, x1, l1, xl1, ll1)
    t0 = time.time()
    omodel.train_step(ref, mm, vv, 10, x1, l1, xl1, ll1)
    one_thread = time.time() - t0
    torch.set_num_threads(cores)
    out["one_thread"] = {"value": 1.0 / one_thread, "unit": "utterances/s", "cores": 1,
                         "sample": "1 train step of B=1 utterance after 1 warm-up; %.2f s/step" % one_thread}
    # SURVEY 8d's C1 shape (BASELINE configs[0]: 2 x conv + ONE GRU layer, B=4, T=200, V=119) on all threads
    from asr.model import ds2 as _ds2
    c1 = _ds2.configure()
    c1.vocab_size, c1.num_rnn_layers, c1.bidirectional = 119, 1, False
    m1 = _ds2.Model(c1)
    m1.rnn_blocks.layers[0]._initialize_params(c1.ndim_conv * 6)
    m1.dense_blocks.layers[0]._initialize_params(c1.ndim_rnn)
    m1.dense_blocks.layers[7].norm._initialize_params(119)
    ref1 = omodel.DS2Oracle({k: v.detach().clone() for k, v in m1.state_dict().items()}, 2, 1, False)
    xc, lc, xlc, llc = omodel.synthetic_batch(4, 200, 119, Lmin=10, Lmax=30, seed=0)
    m1s, v1s = [torch.zeros_like(q) for q in ref1.parameters()], [torch.zeros_like(q) for q in ref1.parameters()]
    omodel.train_step(ref1, m1s, v1s, 1, xc, lc, xlc, llc)
    t0 = time.time()
    for k in range(3):
        omodel.train_step(ref1, m1s, v1s, 2 + k, xc, lc, xlc, llc)
    c1dt = (time.time() - t0) / 3
    out["c1"] = {"value": 4.0 / c1dt, "unit": "utterances/s", "cores": cores,
                 "sample": "3 train steps of BASELINE configs[0] (2 x conv + 1 GRU-512, B=4, T=200, V=119) after 1 warm-up; %.3f s/step" % c1dt}
    try:
        with open("/proc/cpuinfo") as f:
            names = [ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")]
        out["cpu_model"] = names[0] if names else None
        out["host_logical_cpus"] = len(names)
    except OSError:
        pass
    return out, parity


def cnn_config(args, V):
    from asr.model import cnn
    cfg = cnn.configure()
    cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense = V, 3, 128, 320      # run/ctc/cnn/args.py:28-30
    cfg.num_conv_layers, cfg.architecture = int(args.num_conv_layers), "zhang+residual"
    return cfg


def cnn_macs_per_frame(cfg):
    """multiply-accumulates of one forward pass of `zhang+residual` per frame, layer by layer from the recipe
    (run/ctc/cnn/model.py:142-204; SURVEY.md section 8d: 27.3 M at 4 conv layers, 137 M for the wide branch)"""
    import math
    h, dense, V, cin = cfg.ndim_h, cfg.ndim_dense, cfg.vocab_size, cfg.ndim_audio_features
    kh, kw = cfg.kernel_size
    taps = kh * kw
    height = cfg.num_mel_filters - (kh - 1)                       # first layer: pad_h = 0
    macs = height * (2 * h) * cin * taps
    height = int(math.ceil(height / 3.0))                         # MaxPooling2D((3, 1)), cover_all
    narrow, wide = min(cfg.num_conv_layers, 4), max(0, cfg.num_conv_layers - 4)
    for idx in range(narrow):
        co = 4 * h if (wide > 0 and idx == narrow - 1) else 2 * h
        macs += height * co * h * taps
    if wide > 0:
        macs += narrow * height * (4 * h) * (2 * h) * taps
    dense_in = 2 * h if wide > 0 else h
    kernel_height = int(math.ceil((cfg.num_mel_filters - 2) / 3))
    macs += (2 * dense) * dense_in * kernel_height + (2 * dense) * dense + V * dense
    return macs


def cpu_baseline_cnn(cfg, T, V, dev=None, seconds_budget=25.0):
    """--config cnn: oracle/cnn.py (torch-CPU fp32 restatement of the recipe) + the oracle's clip/decay/Adam step on a
    bounded sample, and the parity of the HIP path on that sample (same B=2 batch, same initial parameters)."""
    import numpy as np
    from asr.model.architectures import build_model
    from oracle import cnn as ocnn
    from oracle import model as omodel
    torch.manual_seed(0)
    B = 2
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, seed=0)
    gpu = build_model(cfg)
    if dev is None:
        raise RuntimeError("the CNN recipes size their layer norms lazily: a device is needed to materialise them")
    gpu.to_gpu(dev.index)
    with torch.no_grad():
        gpu(x.to(dev))
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in gpu.named_parameters()}
    plist = list(params.values())

    class Ref(object):
        def parameters(self):
            return plist

        def __call__(self, xx):
            return ocnn.logits_tbv(ocnn.forward(cfg.architecture, cfg, params, xx))
    ref = Ref()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail, torch.get_num_threads()))
    torch.set_num_threads(cores)
    log("cpu_baseline: %d threads" % cores)
    t0 = time.time()
    logits_ref = ref(x)
    loss_ref = omodel.ctc_mean_loss(logits_ref, labels, x_len, l_len)
    loss_ref.backward()
    from asr.loss import connectionist_temporal_classification
    ys = gpu(x.to(dev))
    loss = connectionist_temporal_classification(ys, labels.to(dev), 0, x_len.to(dev), l_len.to(dev))
    loss.backward()
    from asr.functions import join_side_stream
    join_side_stream()
    torch.cuda.synchronize()

    def cos(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return float(a @ b / (a.norm() * b.norm() + 1e-30))
    worst, worst_name = 1.0, None
    for name, p in gpu.named_parameters():
        c = cos(p.grad.detach().cpu(), params[name].grad)
        if c < worst:
            worst, worst_name = c, name
    parity = {"against": "oracle/cnn.py (torch-CPU fp32), same B=%d batch and initial parameters, T=%d, V=%d" % (B, T, V),
              "loss_gpu": float(loss.item()), "loss_oracle": float(loss_ref.item()),
              "loss_rel": abs(float(loss.item()) - float(loss_ref.item())) / abs(float(loss_ref.item())),
              "logits_cos": cos(torch.stack(tuple(ys)).detach().float().cpu(), logits_ref.detach()),
              "worst_param_grad_cos": worst, "worst_param": worst_name}
    del gpu, ys, loss
    for q in plist:
        q.grad = None
    mm = [torch.zeros_like(q) for q in plist]
    vv = [torch.zeros_like(q) for q in plist]
    omodel.train_step(ref, mm, vv, 1, x, labels, x_len, l_len)
    log("cpu_baseline: warm-up %.1f s" % (time.time() - t0))
    t0 = time.time()
    omodel.train_step(ref, mm, vv, 2, x, labels, x_len, l_len)
    one = time.time() - t0
    n = max(1, min(5, int(seconds_budget / max(one, 1e-3)) - 1))
    t0 = time.time()
    for k in range(n):
        omodel.train_step(ref, mm, vv, 3 + k, x, labels, x_len, l_len)
    dt = (time.time() - t0) / n
    return {"value": B / dt, "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": "%d train steps of B=%d utterances (T=%d, V=%d, same recipe, fp32, torch-CPU oracle) after 2 warm-up; %.2f s/step"
                      % (n, B, T, V, dt)}, parity


def time_gram_ctc(dev, T=1000, B=32, V=3000, L=120, iters=10):
    """BASELINE configs[3]: the Gram-CTC loss + gradient (asr_ctc_loss_grad, gram = 1) as a stand-alone operator at its own size:
    unigram ids U{1..118}, bigram ids U{119..V-1} with 30 % absent, label lengths U{40..120} -> N = 3L+1 <= 361 lattice nodes"""
    from asr import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(T, B, V, generator=g).to(dev)
    lab = torch.randint(1, 119, (B, L), generator=g, dtype=torch.int32).to(dev)
    big = torch.randint(119, V, (B, L), generator=g, dtype=torch.int32)
    big[torch.rand(B, L, generator=g) < 0.3] = -1
    big[:, 0] = -1
    big = big.to(dev)
    tl = torch.randint(40, L + 1, (B,), generator=g, dtype=torch.int32).to(dev)
    res = {}
    for gram in (1, 0):
        n = lib.asr_ctc_workspace_bytes(T, B, V, L, gram)
        ws = torch.empty(n, dtype=torch.uint8, device=dev)
        loss, lm, grad = torch.empty(B, device=dev), torch.empty((), device=dev), torch.empty_like(xs)
        s = torch.cuda.current_stream()

        def run():
            rc = lib.asr_ctc_loss_grad(s.cuda_stream, xs.data_ptr(), lab.data_ptr(), big.data_ptr() if gram else None, None, tl.data_ptr(),
                                       T, B, V, L, 0, 1.0 / B, loss.data_ptr(), lm.data_ptr(), grad.data_ptr(), ws.data_ptr(), n)
            assert rc == 0, rc
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(iters):
            run()
        e1.record(s)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        alg = 2.0 * T * B * V * 4
        res["gram_ctc" if gram else "ctc_same_batch"] = {"ms": ms, "algorithmic_bytes": alg, "achieved_GBps": alg / ms / 1e6,
                                                        "frac_of_hbm_peak": alg / ms / 1e6 / HBM_PEAK_GBPS, "loss_mean": float(lm.item())}
    res["workload"] = "BASELINE configs[3]: Gram-CTC loss + gradient, T=%d, B=%d, V=%d, L<=%d (N<=%d nodes, 7 diagonals), 30%% bigrams absent" \
                      % (T, B, V, L, 3 * L + 1)
    return res


def time_sru(dev, T=1000, B=32, D=512, iters=20):
    """the reference's only native kernel (asr/nn/sru.py:7-193) as an operator: forward / backward scans at T=1000, B=32, D=512 against
    the algorithmic bytes T B D (2 + 12 + 4 + 2) / T B D (2 + 12 + 4 + 2 + 6 + 2) and the ~6.3 TB/s a streaming kernel reaches"""
    from asr import _ops
    x = torch.randn(T, B, D, device=dev).to(torch.bfloat16)
    U = torch.randn(T * B, 3 * D, device=dev)
    bias = torch.randn(2 * D, device=dev) * 0.3
    c0 = torch.randn(B, D, device=dev)
    gH = torch.randn(T, B, D, device=dev).to(torch.bfloat16)
    gcT = torch.randn(B, D, device=dev)
    gb = torch.zeros(2 * D, device=dev)
    H, C, cT = _ops.sru_fwd(x, U, bias, c0, None, True)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    f = timed(lambda: _ops.sru_fwd(x, U, bias, c0, None, True))
    b = timed(lambda: _ops.sru_bwd(x, U, bias, C, c0, None, gH, gcT, gb, True))
    fb, bb = T * B * D * 20.0, T * B * D * 28.0
    return {"workload": "SRU scans (tanh), T=%d, B=%d, D=%d: asr_sru_fwd / asr_sru_bwd, time-chunked" % (T, B, D),
            "fwd_ms": f, "bwd_ms": b, "fwd_GBps": fb / f / 1e6, "bwd_GBps": bb / b / 1e6, "achievable_GBps": 6300.0,
            "fwd_frac": fb / f / 1e6 / 6300.0, "bwd_frac": bb / b / 1e6 / 6300.0}


def time_features(dev, B=32, N=160672, iters=50):
    """SURVEY 8(d) logfbank row (a1-a4 + a7): int16 signals resident in HBM -> (B, 3, 40, T) float32 normalised features, the two
    kernels of asr.fft.Processor.logfbank_batch (specgram + mel + log in one, deltas + normalisation + layout in the other).
    Algorithmic bytes: 2 N in + 3 * 40 * T * 4 out per utterance (0.8 MB at N = 160672 -> 1002 frames -> T = 1000); HIP events around
    `iters` back-to-back kernel pairs on the launch stream (outputs preallocated: what the kernels cost), and the wall time of the
    whole `logfbank_batch` call (host arithmetic, small H2D copies and allocations included: what a data loader pays)."""
    from asr import fft
    proc = fft.Processor(device=dev)
    g = torch.Generator().manual_seed(0)
    sig = torch.round(torch.randn(B, N, generator=g) * 3000).to(torch.int16).to(dev)
    lens = [N] * B
    mean = torch.zeros(3, 40)
    std = torch.ones(3, 40)
    x, xl = proc.logfbank_batch((sig, lens), mean.numpy(), std.numpy())
    T = int(x.shape[3])
    F = fft.num_frames(N, proc.frame_len, proc.frame_step)
    lengths = torch.tensor(lens, dtype=torch.int32, device=dev)
    nfr = torch.tensor([F] * B, dtype=torch.int32, device=dev)
    m, sd = mean.reshape(-1).to(dev), std.reshape(-1).to(dev)

    def pair():
        _, lm = fft._specgram(sig, lengths, nfr, F, proc.frame_len, proc.frame_step, proc.num_fft, 0.97, proc._window_d, proc._fbank_d, False, proc._bands_d)
        return fft._deltas(lm, nfr, T, m, sd)
    for _ in range(3):
        pair()
    torch.cuda.synchronize()
    # The host needs 40 - 150 us to queue a pair (two ctypes calls + two allocations; a busy box more): timed as they are queued, the
    # events would measure the host.  A one-wave delay kernel holds the stream while all `iters` pairs are queued behind it; the first
    # event sits behind the delay, so the interval is the kernels alone.
    from asr import _lib
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _lib.check(_lib.lib().asr_stream_delay(_lib.stream(), min(100000, 400 * iters)), "asr_stream_delay")
    e0.record()
    for _ in range(iters):
        pair()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    t0 = time.perf_counter()
    for _ in range(10):
        proc.logfbank_batch((sig, lens), mean.numpy(), std.numpy())
    torch.cuda.synchronize()
    wall_us = (time.perf_counter() - t0) / 10 * 1e6
    alg = B * (2.0 * N + 3 * 40 * T * 4)
    flops = B * F * 45e3
    return {"workload": "logfbank of %d x %d int16 samples -> (%d, 3, 40, %d) f32 (Hann-512, hop 160, 40 mels, deltas, normalised)" % (B, N, B, T),
            "kernels": "asr::fbank::specgram (+ mel + log) and asr::fbank::deltas (+ normalisation)", "bound": "hbm",
            "us_per_batch": us, "algorithmic_bytes": alg, "achieved": alg / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "utterances_per_s": B / (us * 1e-6),
            "gflops": flops / (us * 1e-6) / 1e9, "flops_note": "~45 kFLOP per frame (SURVEY 8d): the FFT, not the traffic, is what the kernel does",
            "call_wall_us": wall_us, "step_share": "one batch of features per train step: us_per_batch against ms_per_step"}


def settled_steps(step, steps, max_warmup=12):
    """`extra_configs` timing (one kept line had a 2.3 x outlier -- hipMalloc calls of the caching allocator inside a
    five-step region entered after two warm-ups, see asr/functions.py: _OnSide).  Warm-up runs until the allocator is quiescent -- two
    consecutive FREE-RUNNING steps (no synchronisation in between, as in the timed region) without a device allocation --, then `steps`
    free-running steps bracketed by synchronisations; an event per step boundary gives the per-step spread without touching the region.
    The line reports the device allocations that happened inside the region (0 is the healthy value)."""
    def allocs():
        return torch.cuda.memory_stats().get("num_device_alloc", 0)
    warm, quiet = 0, 0
    torch.cuda.synchronize()
    while warm < max_warmup and quiet < 2:
        a0 = allocs()
        step()
        warm += 1
        quiet = quiet + 1 if allocs() == a0 else 0
    quiesce_host_gc()
    torch.cuda.synchronize()
    a0 = allocs()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    last = None
    for i in range(steps):
        last = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    return {"ms_per_step": dt / steps * 1e3, "min_ms": per[0], "median_ms": per[len(per) // 2], "max_ms": per[-1], "warmup_steps": warm,
            "device_allocations_in_timed_region": allocs() - a0}, last


def time_cnn_config(args, nconv, dev, steps=10):
    """BASELINE configs[4] on ONE GPU: one train step of the `zhang+residual` recipe (4 conv layers / the wide 8-layer branch), B=32,
    T=1000, V=119: ms per step, utterances/s and the MFMA fraction of its GEMM-class time"""
    import copy
    from asr import _ops
    from asr import functions as asr_functions
    from asr.model.architectures import build_model
    from asr.loss import connectionist_temporal_classification
    from asr.optimizers import Adam, GradientClipping, WeightDecay
    from asr.data.synthetic import synthetic_batch
    a = copy.copy(args)
    a.num_conv_layers = nconv
    B, T, V = 32, args.frames, 119
    cfg = cnn_config(a, V)
    x, labels, x_len, l_len = (t.to(dev) for t in synthetic_batch(B, T, V, seed=0))
    half = _ops.BF16 is torch.float16          # the IEEE-half library (ASR_ACT=f16): dynamic loss scaling, as the reference's fp16 runs need

    def fresh(alpha):
        torch.manual_seed(0)
        model = build_model(cfg).to_gpu(dev.index)
        with torch.no_grad():
            model(x)
        opt = Adam(alpha=alpha, beta1=0.9)
        opt.setup(model)
        opt.add_hook(GradientClipping(1.0))
        opt.add_hook(WeightDecay(1e-5))
        if half:
            opt.loss_scaling()

        def step():
            loss = connectionist_temporal_classification(model(x), labels, 0, x_len, l_len)
            opt.update(lossfun=lambda: loss)
            return loss
        return model, opt, step
    alpha, note = 1e-3, None
    model, opt, step = fresh(alpha)
    spread, loss = settled_steps(step, steps)
    attempted, applied, scale = opt.t, opt.applied_steps(), opt.loss_scale()
    if half and (applied < attempted - 2 or not bool(torch.isfinite(loss))):
        # The un-normalised wide recipe grows its activations 3 x per residual block once Adam at 1e-3 has moved the random initial weights
        # twice: layer 16 passes 65504 in the FORWARD pass, the loss is NaN from then on and no loss scale helps (tools/debug_f16_wide8.py;
        # bfloat16 has the range).  A run whose updates are dropped is not a train step: the half line is measured at alpha = 1e-5, where
        # every step applies -- same kernels, same shapes, same work.
        note = ("at alpha = 1e-3 the forward pass leaves the half range after %d applied updates (%d of %d attempted steps dropped, loss %s): "
                "measured at alpha = 1e-5 instead" % (applied, attempted - applied, attempted, loss.item()))
        log("half build, %d conv layers: %s" % (nconv, note))
        del model, opt, step
        torch.cuda.empty_cache()
        alpha = 1e-5
        model, opt, step = fresh(alpha)
        spread, loss = settled_steps(step, steps)
        attempted, applied, scale = opt.t, opt.applied_steps(), opt.loss_scale()
    dt = spread["ms_per_step"] * steps * 1e-3
    census = Census()
    census.wrap(_ops)
    asr_functions._SIDE["enabled"] = False
    step()
    torch.cuda.synchronize()
    census.events = []
    step()
    tot = census.totals()
    shapes = census.gemm_shapes()
    asr_functions._SIDE["enabled"] = True
    census.unwrap()
    macs = cnn_macs_per_frame(cfg)
    gemm_ms = sum(tot.get(k, (0.0, 0))[0] for k in GEMM_OPS)
    flops = 3 * 2.0 * macs * T * B
    floors = sum(max(e["mfma_floor_ms"], e["hbm_floor_ms"]) * e["calls"] for e in shapes)
    times = sum(e["ms_per_call"] * e["calls"] for e in shapes)
    res = {"workload": "BASELINE configs[4] on one GPU: zhang+residual, %s, ndim_h 128, ndim_dense 320, B=%d, T=%d, V=%d, %s"
                       % ("4 conv layers" if nconv <= 4 else "wide branch (8 conv layers)", B, T, V,
                          "IEEE half MFMA operands / activations (libasr_hip_f16.so), dynamic loss scaling" if half else "bf16 (see dtype_note)"),
           "dtype": "fp16" if half else "bf16", "steps_attempted": attempted, "steps_applied": applied, "loss_scale": scale[0],
           "loss_scale_overflows": scale[1], "adam_alpha": alpha, "adam_alpha_requested": 1e-3,
           "comparable_to_baseline": alpha == 1e-3,      # False: the recipe does not train in IEEE half at the BASELINE learning rate
           "note": note,
           "ms_per_step": dt / steps * 1e3, "utterances_per_s": B * steps / dt, "steps": steps, "step_spread": spread,
           "final_loss": float(loss.item()),
           "gflop_per_utterance": 3 * 2.0 * macs * T / 1e9,
           "roofline": {"bound": "mfma", "achieved": flops / (gemm_ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (gemm_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "ms": gemm_ms, "traffic": None,
                        "frac_of_binding_roof": floors / times if times > 0 else None, "top_shapes": shapes[:6]}}
    del model, opt, step
    torch.cuda.empty_cache()
    return res


def cnn_extra_in_a_child(args, nconv):
    import subprocess
    env = dict(os.environ, ASR_ACT="f16")
    env.pop("ASR_HIP_LIB", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--cnn-extra", str(nconv), "--frames", str(args.frames)]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "child exited %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-400:])}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"error": "child timed out"}


def pmc_traffic_of_this_build():
    """`roofline.traffic`: HBM bytes per launch of the recurrence kernels from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    same command (a process cannot read its own PMC counters: tools/profile_round.sh runs the passes, tools/pmc_traffic.py condenses them
    into profiles/ and records the sha256 of the kernels' source).  A file taken on OTHER kernel sources is not quoted: the line then says
    `traffic: null` and why."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True)
    why = "no profiles/r*_pmc_traffic*.json"
    for path in files:
        try:
            rec = json.load(open(path))
        except ValueError:
            continue
        want = rec.get("source_sha256")
        if not want:
            why = "stale: %s carries no source hash (taken before round 5)" % os.path.basename(path)
            continue
        same = True
        for name, digest in want.items():
            with open(os.path.join(ROOT, "chainer-speech-recognition_amd", "csrc", name), "rb") as f:
                same = same and hashlib.sha256(f.read()).hexdigest() == digest
        if not same:
            why = "stale: %s was taken on other kernel sources (csrc/gru.hip changed since)" % os.path.basename(path)
            continue
        sel = [v for k, v in rec["kernels"].items() if "gru::fwd_persistent_io_kernel" in k or "gru::bwd_wide_kernel" in k or "gru::bwd_ps_kernel" in k]
        if sel:
            return sum(v["hbm_bytes_per_dispatch"] * v["dispatches"] for v in sel) / sum(v["dispatches"] for v in sel), "profiles/" + os.path.basename(path)
    return None, why


def sq_profile():
    """per-kernel SQ counter summary of a `bench.py` step (rocprofv3 --pmc passes condensed by tools/pmc_sq.py into profiles/): a
    process cannot read its own PMC counters, so the line quotes the committed pass of this same command"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.json")), reverse=True):
        try:
            return json.load(open(path)), "profiles/" + os.path.basename(path)
        except ValueError:
            pass
    return None, None


HEADLINE_MAX_BYTES = 4096       # the driver keeps ~8 KB of stdout: the line it parses must fit with room to spare


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _r(v, nd=4):
    """numbers of the headline at a readable precision (floats only; the detail file keeps every digit)"""
    if isinstance(v, float):
        return float("%.*g" % (nd + 2, v))
    if isinstance(v, str) and len(v) > 360:
        return v[:357] + "..."
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, list):
        return [_r(x, nd) for x in v]
    return v


def headline(out):
    """The ONE stdout line of the contract, cut down to what the driver and the judge read: metric / value / config, `roofline` of the
    dominant kernel, the two other roofline classes north_star names (GEMM blocks, CTC sweep), `cpu_baseline`, `parity`.  Tables
    (per-shape GEMM list, kernel census, extra configs, per-parameter parity) live in bench_detail.json."""
    h = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data"))
    h["config"] = _pick(out.get("config", {}), ("workload", "global_batch", "parallelism", "optimizer", "final_loss"))
    if "roofline" in out:
        h["roofline"] = _pick(out["roofline"], ("kernel", "bound", "regime", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                "algorithmic_bytes_per_launch", "ms_per_launch", "launches_per_step", "us_per_time_step",
                                                "fwd_us_per_time_step", "bwd_us_per_time_step", "timed_with"))
    if "roofline_gemm" in out:
        h["roofline_gemm"] = _pick(out["roofline_gemm"], ("bound", "frac", "achieved", "peak", "unit", "frac_of_mfma_peak", "ms"))
    if "roofline_ctc_sweep" in out:
        h["roofline_ctc_sweep"] = _pick(out["roofline_ctc_sweep"], ("bound", "frac", "achieved", "peak", "unit", "ms"))
    if "features" in out:
        h["features"] = _pick(out["features"], ("workload", "us_per_batch", "achieved", "peak", "unit", "frac", "utterances_per_s"))
    if out.get("cpu_baseline"):
        h["cpu_baseline"] = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "cpu_model", "sample"))
    if out.get("parity"):
        p = out["parity"]
        h["parity"] = {"pass": p.get("pass"), "loss_rel": p.get("matched", {}).get("loss_rel"),
                       "grad_rel": p.get("matched", {}).get("worst_param_grad_rel_l2"), "gate": p.get("gate"),
                       "loss_rel_fp32_oracle": p.get("fp32", {}).get("loss_rel")}
    for k in ("gpu_vs_cpu", "allreduce_exposed_ms_per_step", "ranks", "per_rank_ms_per_step", "device_allocations_in_timed_region", "detail"):
        if k in out:
            h[k] = out[k]
    h = _r(h)
    line = json.dumps(h, separators=(",", ":"))
    if len(line.encode()) >= HEADLINE_MAX_BYTES:         # never silently: drop the optional blocks, longest first, and say so
        for k in ("per_rank_ms_per_step", "features", "roofline_ctc_sweep", "roofline_gemm"):
            h.pop(k, None)
            h["dropped_for_size"] = h.get("dropped_for_size", []) + [k]
            line = json.dumps(h, separators=(",", ":"))
            if len(line.encode()) < HEADLINE_MAX_BYTES:
                break
    return line


def emit(out):
    """full record -> bench_detail.json (repo root, and gpurun_out/ when it exists) and stderr; compact line -> stdout, last"""
    paths = [os.environ.get("ASR_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = None
    for path in paths:
        try:
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
            written = written or os.path.relpath(path, ROOT)
        except OSError as e:
            log("could not write %s: %s" % (path, e))
    out["detail"] = written
    log("detail: " + json.dumps(out))
    print(headline(out))
    sys.stdout.flush()


def spawn_ranks(args):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE): start N fresh rank processes through torch.distributed.run
    BEFORE this process has touched the GPU (it never does), relay rank 0's JSON line, exit with the children's status.  Never an
    exec of a process that initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("no WORLD_SIZE in the environment: starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            log("[ranks] " + out)
    rc = proc.wait()
    if line is not None:
        print(line)
        sys.stdout.flush()
    if rc != 0 or line is None:
        raise SystemExit(rc if rc != 0 else 1)


REGION = {"device_allocations": None}       # hipMalloc calls of the caching allocator inside the last timed region (healthy: 0)


def quiesce_host_gc():
    """A full (generation-2) collection of this process takes ~35 ms (170 k tracked objects: tools/gc_probe.py) and CPython starts one
    when enough long-lived objects have piled up -- building a model does that.  Landing in the first steps of a timed region, where the
    host is not yet ahead of the device, it showed as one 50 - 90 ms step in a 100-step trace and in two bench lines of round 4.  Run it
    now, outside the region, and move the survivors to the permanent generation so that later collections only look at what the steps
    create (host hygiene of the measurement; the collector stays enabled)."""
    import gc
    gc.collect()
    gc.freeze()


def timed_region(step, steps, warmup, comm, sync, dev=None):
    """the contract's timed region: W untimed steps, barrier + synchronize, K steps, synchronize + barrier, MAX over ranks"""
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    quiesce_host_gc()
    sync()
    if comm is not None:
        comm.barrier()
    sync()
    a0 = torch.cuda.memory_stats().get("num_device_alloc", 0) if torch.cuda.is_available() else 0
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step()
    sync()
    if comm is not None:
        comm.barrier()
    sync()
    dt = time.perf_counter() - t0
    REGION["device_allocations"] = (torch.cuda.memory_stats().get("num_device_alloc", 0) - a0) if torch.cuda.is_available() else 0
    per_rank = [dt]
    if comm is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if comm.backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(t) for _ in range(comm.size)]
        dist.all_gather(gathered, t)
        per_rank = [float(g.item()) for g in gathered]
        dt = max(per_rank)
    return dt, per_rank, last


def rank_evidence(comm, world, rank, dev):
    """who took part: every rank reports (rank, local device index, device identity, host, pid), all-gathered, so that an N > 1 line
    shows N distinct devices and names the backend that carried the gradients ("nccl" is RCCL on ROCm).  Device identity = the
    uuid torch reports for the HIP device (falls back to the PCI bus id / the device name)."""
    import socket
    me = {"rank": rank, "host": socket.gethostname(), "pid": os.getpid(), "local_device": None, "device": None}
    if dev is not None:
        props = torch.cuda.get_device_properties(dev)
        ident = None
        for attr in ("uuid", "pci_bus_id"):
            try:
                v = getattr(props, attr, None)
                if v not in (None, ""):
                    ident = "%s:%s" % (attr, v)
                    break
            except Exception:       # noqa: BLE001  (an attribute torch-ROCm does not fill)
                pass
        me["local_device"] = dev.index
        me["device"] = ident or props.name
        me["device_name"] = props.name
    devices = [me]
    backend = None
    if comm is not None:
        backend = comm.backend
        gathered = [None] * comm.size
        torch.distributed.all_gather_object(gathered, me)
        devices = gathered
    distinct = len(set((d["host"], d["local_device"], d["device"]) for d in devices))
    return {"backend": backend or "none (one process)", "world_size": world, "devices": devices, "distinct_devices": distinct}


def rehearse(args, world, rank):
    """ASR_BENCH_REHEARSE=1: the multi-rank control flow of this file -- rendezvous, Communicator, sliced all-reduce joined before the
    'optimiser', barrier + max-over-ranks timing, the JSON line -- with gloo on CPU tensors and NO GPU: a stand-in step sums a flat
    buffer over the ranks.  tests/test_parallel_cpu.py runs `bench.py --gpus 2` this way; the numbers mean nothing."""
    from asr.parallel import Communicator
    comm = Communicator("gloo", buckets=3) if world > 1 else None
    flat = torch.full((4096,), float(rank + 1))

    def step():
        g = flat.clone()
        if comm is not None:
            torch.distributed.all_reduce(g)
        return float(g[0])
    dt, per_rank, last = timed_region(step, args.steps, args.warmup, comm, lambda: None)
    ranks = rank_evidence(comm, world, rank, None)
    if rank == 0:
        emit({"metric": "utterances/sec (T=1000, 40x3 feat, |V|~3000) CTC train step", "value": world * args.batch * args.steps / dt,
              "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "rehearsal", "data": "none",
              "config": {"workload": "REHEARSAL of the multi-rank control flow (gloo, CPU, no model)", "global_batch": world * args.batch,
                         "parallelism": "dp%d" % world, "check": last},
              "ranks": ranks, "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank]})
    if comm is not None:
        torch.distributed.destroy_process_group()


def main():
    args = parse()
    if args.vocab is None:
        args.vocab = 3000 if args.config == "ds2" else 119
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("ASR_BENCH_REHEARSE") == "1":
        return rehearse(args, world, rank)
    # ASR_BENCH_BACKEND=gloo + ASR_BENCH_ONE_DEVICE=1: rehearse the multi-rank control flow with several processes on
    # ONE GPU (no RCCL between ranks of one device); the numbers of such a run mean nothing
    if os.environ.get("ASR_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from asr import _lib, _ops
    from asr.loss import connectionist_temporal_classification
    from asr.model import ds2
    from asr.optimizers import Adam, GradientClipping, WeightDecay
    from asr.data.synthetic import synthetic_batch
    _lib.lib()      # fail loudly if the HIP library is missing
    if args.cnn_extra:
        print(json.dumps(time_cnn_config(args, args.cnn_extra, dev)))
        return

    comm = None
    if world > 1 or os.environ.get("ASR_BENCH_FORCE_COMM") == "1":      # the latter: exercise the RCCL path on one GPU
        if world == 1:          # a bare shell: the env:// rendezvous of a single rank
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(port))):
                os.environ.setdefault(k, v)
        from asr.parallel import Communicator
        comm = Communicator(os.environ.get("ASR_BENCH_BACKEND", "nccl"))
        comm.measure = True

    B, T, V = args.batch, args.frames, args.vocab
    torch.manual_seed(0)                         # identical initial weights on every rank (also broadcast below)
    if args.config == "cnn":
        from asr.model.architectures import build_model
        cfg = cnn_config(args, V)
        model = build_model(cfg).to_gpu(local_rank)
    else:
        cfg = ds2.configure()
        cfg.vocab_size = V
        model = ds2.Model(cfg).to_gpu(local_rank)
    x, labels, x_len, l_len = synthetic_batch(B, T, V, seed=rank, ragged=args.ragged)
    x, labels, x_len, l_len = x.to(dev), labels.to(dev), x_len.to(dev), l_len.to(dev)
    model_kw = {"x_length": x_len} if (args.ragged and args.config == "ds2") else {}
    if args.config == "cnn":
        with torch.no_grad():
            model(x)                             # the recipes size their layer norms lazily: materialise before opt.setup

    opt = Adam(alpha=1e-3, beta1=0.9)
    opt.setup(model)
    opt.add_hook(GradientClipping(1.0))
    opt.add_hook(WeightDecay(1e-5))
    if comm is not None:
        opt.set_communicator(comm)

    exposed = []

    def step():
        loss = connectionist_temporal_classification(model(x, **model_kw), labels, 0, x_len, l_len)
        opt.update(lossfun=lambda: loss)
        if comm is not None:
            exposed.append(list(comm.exposed))
        return loss

    log("rank %d/%d: model built, warm-up %d + timing %d steps" % (rank, world, args.warmup, args.steps))
    dt, per_rank, loss = timed_region(step, args.steps, args.warmup, comm, torch.cuda.synchronize, dev)
    loss_value = loss.item()
    region_allocs = REGION["device_allocations"]
    exposed_ms = None
    if comm is not None:
        # time the launch stream stood still waiting for collectives (events around every join), per step, this rank and max
        mine = sum(a.elapsed_time(b) for pairs in exposed[-args.steps:] for a, b in pairs) / args.steps
        t = torch.tensor([mine], device=dev if comm.backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        exposed_ms = {"rank0": mine, "max_over_ranks": float(t.item())}
    ranks = rank_evidence(comm, world, rank, dev)       # collective: every rank takes part before the others leave

    if rank != 0:
        return
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    out = {"metric": "utterances/sec (T=1000, 40x3 feat, |V|~3000) CTC train step", "value": value, "unit": "utterances/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "dtype_note": "bf16 MFMA operands / activations, float32 accumulation, master weights, optimiser state, statistics, logits and "
                         "CTC.  The fp16 of BASELINE configs[4] is the IEEE-half build of the same kernels (libasr_hip_f16.so, ASR_ACT=f16, "
                         "with chainer's loss_scaling on the device): extra_configs.cnn_*_fp16, measured in child processes (the wide 8-layer "
                         "recipe does NOT train in half at the BASELINE alpha = 1e-3 -- its entry says comparable_to_baseline: false and is "
                         "measured at 1e-5; bf16 is the supported format there); the recurrences (configs[1], this line) are bfloat16-only "
                         "(DESIGN.md 13.9)",
           "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
           "config": {"workload": ("BASELINE configs[1]: 2xconv + 4xBiGRU-512 + dense + LayerNorm + CTC train step, "
                                   "B=%d/GPU, T=%d, 3x40 features, V=%d, labels 40..120" % (B, T, V)) if args.config == "ds2" else
                                  ("BASELINE configs[4]: fully convolutional zhang+residual (run/ctc/cnn/model.py:142-204), ndim_h 128, "
                                   "ndim_dense 320, %s, + LayerNorm + CTC train step, B=%d/GPU, T=%d, 3x40 features, V=%d, labels 40..120; "
                                   "bf16 MFMA operands (the deliberate substitute for fp16: DESIGN.md section 4)"
                                   % ("4 conv layers" if args.num_conv_layers <= 4 else "wide branch: 8 conv layers", B, T, V)),
                      "global_batch": world * B, "parallelism": "dp%d" % world, "optimizer": "clip1+decay1e-5+adam",
                      "final_loss": loss_value,
                      "x_length": "U{600..%d}, length-exact recurrences (x_length passed to the model)" % T if args.ragged else "all %d" % T}}
    out["device_allocations_in_timed_region"] = region_allocs
    out["ranks"] = ranks
    if world > 1 and not args.census:
        # N > 1 lines are the scaling curve's points: the timed region and who took part, nothing else (the roofline, the census and
        # the CPU baseline belong to the N = 1 line, whose timed region runs through this very code path)
        args.no_census = args.no_extra = args.no_cpu_baseline = True
    if exposed_ms is not None:
        out["allreduce_exposed_ms_per_step"] = exposed_ms

    # which hand-off the persistent GRU kernels of the last launch agreed on (decided inside the launch: DESIGN.md section 5)
    if _ops.LAST_SYNC[0] is not None:
        words = _ops.LAST_SYNC[0][960:1008].cpu().tolist()
        nrec = sum(1 for v in words[32:48] if v > 0)
        out["config"]["gru_handoff"] = {"recurrences": nrec, "xcd_of_recurrence": [v - 1 for v in words[0:16] if v > 0],
                                        "split_placements": int(sum(words[16:32])),
                                        "form": "xcd-local" if nrec and sum(words[16:32]) == 0 else "placement-free"}
    log("%.2f ms/step, %.1f utt/s" % (ms_per_step, value))
    if not args.no_census:
        log("kernel census")
        if comm is not None:
            opt.set_communicator(None)      # the other ranks have left: the census steps must not wait for them
        from asr import functions as asr_functions
        census = Census()
        census.wrap(_ops)
        asr_functions._SIDE["enabled"] = False        # one stream for these extra steps: per-op times without overlap
        step()                                        # settles the caching allocator in the single-stream pattern
        torch.cuda.synchronize()
        census.events = []
        step()
        tot = census.totals()
        gemm_shape_table = census.gemm_shapes()
        asr_functions._SIDE["enabled"] = True
        census.unwrap()
        ctc = time_ctc(_lib, _ops, T, B, V, labels.shape[1], x_len, l_len, labels, dev)
        tot["ctc_forward"] = (ctc["ctc_forward"], 1)
        tot["ctc_grad"] = (ctc["ctc_grad"], 1)
        breakdown = {k: {"ms": round(ms, 3), "calls": n} for k, (ms, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])}
        out["kernel_ms_per_step"] = breakdown
        if args.config == "cnn":
            # dominant kernel class: the implicit-GEMM convolutions (forward, backward-data, weight gradient) and the 1x1 /
            # kernel_height "dense" convolutions -- MFMA-bound: 2 flops per multiply-accumulate, x3 for the train step
            macs = cnn_macs_per_frame(cfg)
            gemm_ms = sum(tot.get(k, (0.0, 0))[0] for k in GEMM_OPS)
            launches = sum(tot.get(k, (0.0, 0))[1] for k in GEMM_OPS)
            flops = 3 * 2.0 * macs * T * B
            out["roofline"] = {"bound": "mfma", "kernel": "asr::gemm implicit-GEMM convolutions (conv_nt / conv_tn_acc) + gemm_nt / gemm_tn",
                               "achieved": flops / (gemm_ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": flops / (gemm_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "traffic": None,
                               "ms": gemm_ms, "launches_per_step": launches, "gflop_per_utterance": 3 * 2.0 * macs * T / 1e9,
                               "note": "algorithmic flops of the recipe (SURVEY.md section 8d) / HIP-event time of the GEMM-class calls "
                                       "of one single-stream step; HBM-bound elementwise kernels (maxout, pooling, residual add, "
                                       "layer norm) and the CTC sweep are listed in kernel_ms_per_step"}
            ctc_bytes = 2.0 * T * B * V * 4
            ctc_ms = ctc["ctc_forward"] + ctc["ctc_grad"]
            out["roofline_ctc_sweep"] = {"bound": "hbm", "achieved": ctc_bytes / (ctc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                                         "unit": "GB/s", "frac": ctc_bytes / (ctc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "ms": ctc_ms,
                                         "algorithmic_bytes": ctc_bytes}
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"], out["parity"] = cpu_baseline_cnn(cfg, T, V, dev)
                out["gpu_vs_cpu"] = value / out["cpu_baseline"]["value"]
            emit(out)
            return
        H, nl = cfg.ndim_rnn, cfg.num_rnn_layers
        # dominant kernel class: the persistent GRU kernels.  One launch = one layer, both directions, all T steps.
        gru_ms = tot["gru_fwd"][0] + tot["gru_bwd"][0]
        launches = tot["gru_fwd"][1] + tot["gru_bwd"][1]
        per_launch_s = gru_ms * 1e-3 / launches
        # algorithmic HBM bytes of a launch, per (t, b) row and both directions:
        #   forward : gi bf16 in (6H*2), h bf16 out + in again by the next step (2H*2*2), f32 state out (2H*4), gates out (8H*4)
        #   backward: dy bf16 in (H*2), gates in (8H*4), f32 state in (2H*4), dgi out (6H*2), dgh out (6H*2)
        # plus the W_hh slice once per launch (2*3H*H*2)
        gi_bytes = 2 if _ops.gru_gi_dtype(T, B, H, 2) == torch.bfloat16 else 4
        gate_bytes = 2 if _ops.gru_gates_f16(T, B, H, 2) else 4          # saved gates: IEEE half where the default kernel pair serves
        fwd_bytes = T * B * (6 * H * gi_bytes + 2 * H * 2 * 2 + 2 * H * 4 + 8 * H * gate_bytes) + 2 * 3 * H * H * 2
        # (backward, partial-sum exchange: dgh is written once for the weight-gradient GEMM and not read back; the partial sums
        # travel through the L2 only)
        bwd_bytes = T * B * (H * 2 + 8 * H * gate_bytes + 2 * H * 4 + 6 * H * 2 + 6 * H * 2) + 2 * 3 * H * H * 2
        alg = 0.5 * (fwd_bytes + bwd_bytes)
        # measured HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, condensed
        # into profiles/ by tools/pmc_traffic.py (a process cannot read its own PMC counters)
        traffic, traffic_src = pmc_traffic_of_this_build()
        out["roofline"] = {"bound": "hbm", "regime": "latency chain (neither roof binds; priced against HBM as the contract asks)",
                           "kernel": "asr::gru::fwd_persistent_io_kernel / bwd_ps_kernel (one launch per layer)",
                           "hop_price_us": "0.8-1.0 (MI355X_MICROARCH.md: one producer -> consumer hop through the L2, <= 4 KB, idle chip)",
                           "fwd_us_per_time_step": tot["gru_fwd"][0] * 1e3 / tot["gru_fwd"][1] / T, "bwd_us_per_time_step": tot["gru_bwd"][0] * 1e3 / tot["gru_bwd"][1] / T,
                           "achieved": alg / per_launch_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": alg / per_launch_s / 1e9 / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                           "ms_per_launch": per_launch_s * 1e3, "launches_per_step": launches,
                           "us_per_time_step": per_launch_s * 1e6 / T, "algorithmic_bytes_per_launch": alg,
                           "note": "latency-bound recurrence (8000 dependent steps per train step): neither roofline binds -- a step is one "
                                   "cross-CU hand-off through the L2 (whole-line stores into an L2-resident ring, ~0.16 us to land, + one timed "
                                   "poll, ~0.38 us round trip) plus ~0.6 us of MFMA / LDS exchange / gate work; see DESIGN.md sections 5 and 11.4; "
                                   "HBM-bound CTC sweep and MFMA-bound GEMMs reported beside it"}
        ctc_bytes = 2.0 * T * B * V * 4
        ctc_ms = ctc["ctc_forward"] + ctc["ctc_grad"]
        out["roofline_ctc_sweep"] = {"bound": "hbm", "achieved": ctc_bytes / (ctc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                                     "unit": "GB/s", "frac": ctc_bytes / (ctc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                     "ms": ctc_ms, "algorithmic_bytes": ctc_bytes,
                                     "note": "the loss as a stand-alone operator (asr_ctc_forward + asr_ctc_backward): forward = prep + "
                                             "rows (one read of the logits) + lattice (T serial steps per utterance, latency-bound, no "
                                             "HBM stream); backward = grad (one read of the logits + one write of the gradient)",
                                     "grad_kernel": {"ms": ctc["ctc_grad"], "bytes": ctc_bytes,
                                                     "achieved": ctc_bytes / (ctc["ctc_grad"] * 1e-3) / 1e9,
                                                     "frac": ctc_bytes / (ctc["ctc_grad"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}}
        if "layernorm_ctc_bwd" in tot:
            # in the model the gradient of the logits is never written: LayerNormalization's backward forms it in registers
            # (csrc/ctc_ln.hip).  Algorithmic bytes of that sweep: x in (f32) + alpha/beta of the path + dx out (bf16)
            fb = T * B * (V * 4 + V * 2 + 2.0 * 241 * 8)
            fms = tot["layernorm_ctc_bwd"][0]
            out["roofline_ctc_sweep"]["in_model_backward"] = {
                "kernel": "asr::ctcln::bwd_kernel (CTC gradient + LayerNormalization backward, one sweep)", "ms": fms,
                "algorithmic_bytes": fb, "achieved": fb / (fms * 1e-3) / 1e9, "frac": fb / (fms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "replaces": "ctc::grad (%.3f ms alone) + ln::bwd_rows_f32: 1.73 GB of traffic -> 0.70 GB" % ctc["ctc_grad"]}
        frames = T * B
        macs_fwd = frames * 15.25e6
        # conv_nt / conv_tn_acc: the implicit-GEMM convolutions (forward, backward-data; weight gradient)
        gemm_ms = sum(tot.get(k, (0.0, 0))[0] for k in GEMM_OPS)
        rec_flops = 2 * (2 * nl * T) * (B * H * 3 * H * 2)      # recurrent MFMA work runs inside the GRU kernels
        gemm_flops = 3 * 2 * macs_fwd - rec_flops
        shapes = gemm_shape_table
        floors = sum(max(e["mfma_floor_ms"], e["hbm_floor_ms"]) * e["calls"] for e in shapes)
        times = sum(e["ms_per_call"] * e["calls"] for e in shapes)
        out["roofline_gemm"] = {"bound": "mfma|hbm per shape (bench_detail.json: shapes)", "achieved": gemm_flops / (gemm_ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": floors / times if times > 0 else None,
                                "frac_note": "time-weighted fraction of the roof that binds each shape: sum over the step's GEMM-class calls of "
                                             "max(flops / 2.5 PFLOP/s, algorithmic bytes / 6.3 TB/s), divided by their measured time",
                                "frac_of_mfma_peak": gemm_flops / (gemm_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                "ms": gemm_ms, "shapes": shapes}
        sq, sq_src = sq_profile()
        if sq is not None:
            # MFMA utilisation of the GEMM kernels that ship (north_star: "MFMA utilisation for the GEMM blocks against gfx950 peak"):
            # SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES per kernel over one bench step (rocprofv3 --pmc, tools/pmc_sq.py)
            out["roofline_gemm"]["mfma_busy_frac"] = {k: v.get("mfma_busy_frac") for k, v in sq.get("kernels", {}).items()
                                                      if "gemm" in k}
            out["roofline_gemm"]["mfma_busy_source"] = sq_src
    if world == 1 and not args.no_extra and args.config == "ds2" and not args.ragged:
        log("extra configs: ragged step, Gram-CTC at configs[3] size, configs[4] CNN steps")
        extra = {}
        # SURVEY 8d's ragged variant: the same model and optimiser on x_length ~ U{600..T}, recurrences length-exact
        from asr.data.synthetic import synthetic_batch as _sb
        xr, lr, xlr, llr = (t.to(dev) for t in _sb(B, T, V, seed=1, ragged=True))

        def ragged_step():
            loss_r = connectionist_temporal_classification(model(xr, x_length=xlr), lr, 0, xlr, llr)
            opt.update(lossfun=lambda: loss_r)
            return loss_r
        spr, lossr = settled_steps(ragged_step, 10)
        extra["ds2_ragged"] = {"workload": "configs[1] with x_length ~ U{%d..%d} (mean %.0f), length-exact BiGRU (reverse direction starts at each "
                                           "utterance's last frame)" % (int(0.6 * T), T, float(xlr.float().mean().item())),
                               "ms_per_step": spr["ms_per_step"], "utterances_per_s": B * 1e3 / spr["ms_per_step"], "step_spread": spr,
                               "final_loss": float(lossr.item())}
        out["features"] = time_features(dev, B)
        extra["gram_ctc"] = time_gram_ctc(dev, T, B, V, 120)
        extra["sru"] = time_sru(dev, T, B, 512)
        del model, opt
        torch.cuda.empty_cache()
        extra["cnn_4conv"] = time_cnn_config(args, 4, dev)
        extra["cnn_wide8"] = time_cnn_config(args, 8, dev)
        # configs[4] quotes "fp16 MFMA": the same two steps on the IEEE-half library, which is a per-process choice (ASR_ACT) -- child
        # processes, started the ordinary way (never an exec from this GPU-initialised one)
        for nconv, key in ((4, "cnn_4conv_fp16"), (8, "cnn_wide8_fp16")):
            extra[key] = cnn_extra_in_a_child(args, nconv)
        out["extra_configs"] = extra
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], out["parity"] = cpu_baseline(cfg, T, V, dev)
        out["gpu_vs_cpu"] = value / out["cpu_baseline"]["value"]
    emit(out)
    if out.get("parity") is not None and out["parity"].get("pass") is False:
        log("PARITY GATE FAILED: %s" % json.dumps(out["parity"]["matched"]))
        raise SystemExit(3)


if __name__ == "__main__":
    main()
