"""Forward recurrence alone at the BASELINE size: the shipped 8-row form against ASR_DEBUG gru_fwd_rows=4 (two processes, the outputs
compared bit for bit through a file): python tools/time_gru_fwd_rows.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))


def child(tag):
    import torch
    from asr import _ops
    T, B, H, ndir = int(os.environ.get("GRU_T", "1000")), int(os.environ.get("GRU_B", "32")), 512, 2
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev).to(torch.bfloat16)
    whh16 = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev).to(torch.bfloat16).contiguous()
    bhh = torch.zeros(ndir * 3 * H, device=dev)
    fn = lambda: _ops.gru_fwd(gi.clone(), whh16, bhh, T, B, H, ndir)
    out = fn()
    torch.cuda.synchronize()
    try:
        _ops.gru_check_sync()
    except Exception as e:      # noqa: BLE001
        w = _ops.LAST_SYNC[0][960:1008].cpu().tolist()
        print(tag, "gave up:", e, "| xcd of recurrence", w[0:16], "| split", w[16:32], "| arrivals", w[32:48])
        buf = _ops.LAST_SYNC[0]
        words = buf.view(torch.int32) if buf.dtype != torch.int32 else buf
        off = (4096 + 16 * 8 * 128) // 4
        marks = words[1024:1024 + 512].cpu().tolist()
        print(" progress marks of workgroups 0, 16, 32, ...:", [hex(marks[i]) for i in range(0, 512, 16)])
        import collections
        print(" marks of ids < 256:", dict(collections.Counter(hex(m) for m in marks[:256])), "| ids >= 256:", dict(collections.Counter(hex(m) for m in marks[256:])))
        allr = words[off: off + 16 * 4 * 2048].cpu().reshape(16, 4, 32, 8, 8)
        for r in range(16):
            wr = [(sl, i) for sl in range(4) for i in range(32) if bool((allr[r, sl, i] != -1).any())]
            if wr:
                print(" rec %d: written (slot, wg block):" % r, wr[:70])
        for slot in range(0):
            ring = words[off + slot * 2048: off + (slot + 1) * 2048].cpu().reshape(32, 8, 8)       # [wg][row][8 dwords]
            sent = (ring == -1)
            print(" rec 0 slot %d: sentinel dwords per row (over 32 workgroups):" % slot, sent.sum(dim=(0, 2)).tolist(),
                  "| workgroups with a sentinel in rows 0-3:", [i for i in range(32) if bool(sent[i, :4].any())])
        return
    times = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 5 / T * 1e3)
    _ops.gru_check_sync()
    words = _ops.LAST_SYNC[0][960:1008].cpu().tolist() if _ops.LAST_SYNC[0] is not None else []
    print("%s: %s us per time step (incl. the clone of gi); recurrences %d, split placements %d" %
          (tag, " ".join("%.3f" % t for t in times), sum(1 for v in words[32:48] if v > 0), int(sum(words[16:32]))))
    torch.save([o.cpu() for o in out[:2]], "/tmp/gru_fwd_%s.pt" % tag)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for tag, dbg in (("rows8", ""), ("rows4", "gru_fwd_rows=4")):
        env = dict(os.environ, ASR_DEBUG=dbg)
        subprocess.run([sys.executable, os.path.abspath(__file__), tag], env=env, check=True)
    import torch
    a, b = torch.load("/tmp/gru_fwd_rows8.pt"), torch.load("/tmp/gru_fwd_rows4.pt")
    print("bit-identical:", all(torch.equal(x, y) for x, y in zip(a, b)))
