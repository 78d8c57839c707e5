"""the persistent form of the eight-wave NT kernel (ASR_DEBUG nt_8pp, read once per process) against the one-tile-per-workgroup form on the
model's short-K products: python tools/ab_nt8pp.py  -- runs itself once per setting and prints both"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(32000, 3072, 512, "bf16"), (32000, 3072, 384, "bf16"), (32000, 3000, 320, "f32"), (32000, 640, 512, "bf16"), (32000, 512, 640, "bf16"),
          (32000, 3072, 1024, "bf16"), (32000, 3072, 128, "bf16"), (32000, 3072, 256, "bf16"), (352000, 256, 960, "bf16"), (8192, 8192, 1024, "bf16"), (32000, 3072, 2048, "bf16"), (32000, 3000, 320, "bf16")]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
    import torch
    from asr import _ops
    dev = torch.device("cuda", 0)
    for M, N, K, od in SHAPES:
        a = torch.randn(M, K).to(dev).to(torch.bfloat16)
        b = torch.randn(N, K).to(dev).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        dt = torch.bfloat16 if od == "bf16" else torch.float32
        c = torch.empty(M, N, dtype=dt, device=dev)
        for _ in range(5):
            _ops.gemm_nt_8ph(a, b, bias, dt, c)
        torch.cuda.synchronize()
        best = 1e9
        for rnd in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _ops.gemm_nt_8ph(a, b, bias, dt, c)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        print("%d %d %d %s %.1f" % (M, N, K, od, best), flush=True)
    sys.exit(0)

res = {}
for rnd in range(2):
    for flag in ("nt_8pp=0", "nt_8pp=1,nt8pp_kmax=4096"):
        env = dict(os.environ, ASR_DEBUG=flag)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            print(out.stdout[-2000:], out.stderr[-2000:])
            sys.exit(1)
        for line in out.stdout.strip().splitlines():
            p = line.split()
            if len(p) == 5:
                res.setdefault((p[0], p[1], p[2], p[3]), {}).setdefault(flag, []).append(float(p[4]))
for k, v in res.items():
    one, pp = v.get("nt_8pp=0", []), v.get("nt_8pp=1,nt8pp_kmax=4096", [])
    fl = 2.0 * int(k[0]) * int(k[1]) * int(k[2])
    print("NT %6s x %5s x %5s %-4s  one tile per workgroup %s us   persistent %s us  (%.0f -> %.0f TFLOP/s)" % (
        k[0], k[1], k[2], k[3], " ".join("%.1f" % x for x in one), " ".join("%.1f" % x for x in pp), fl / min(one) / 1e6, fl / min(pp) / 1e6))
