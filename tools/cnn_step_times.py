"""Where do slow `extra_configs` steps come from?  Replays bench.py's order -- the DS2 model first, then
del + empty_cache(), then the configs[4] recipes -- and prints, per step from the first one on: wall time with a synchronisation
on both sides, the caching allocator's device allocations / segment counts, the host time to queue the step."""
import argparse
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
sys.path.insert(0, ROOT)
import torch


def stats():
    s = torch.cuda.memory_stats()
    return dict(dev_alloc=s.get("num_device_alloc", 0), dev_free=s.get("num_device_free", 0), retries=s.get("num_alloc_retries", 0),
                segments=s.get("segment.all.current", 0), reserved_gb=round(s.get("reserved_bytes.all.current", 0) / 2 ** 30, 3),
                allocated_gb=round(s.get("allocated_bytes.all.current", 0) / 2 ** 30, 3))


def main():
    import bench
    from asr.loss import connectionist_temporal_classification
    from asr.optimizers import Adam, GradientClipping, WeightDecay
    from asr.data.synthetic import synthetic_batch
    from asr.model import ds2
    from asr.model.architectures import build_model
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=14)
    ap.add_argument("--skip-ds2", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    sys.argv = sys.argv[:1]
    args = bench.parse()
    B, T = 32, args.frames

    def run(label, model, V, steps):
        x, labels, x_len, l_len = (t.to(dev) for t in synthetic_batch(B, T, V, seed=0))
        with torch.no_grad():
            model(x)
        opt = Adam(alpha=1e-3, beta1=0.9)
        opt.setup(model)
        opt.add_hook(GradientClipping(1.0))
        opt.add_hook(WeightDecay(1e-5))
        torch.cuda.synchronize()
        prev = stats()
        for i in range(steps):
            t0 = time.perf_counter()
            loss = connectionist_temporal_classification(model(x), labels, 0, x_len, l_len)
            opt.update(lossfun=lambda: loss)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            cur = stats()
            print(json.dumps(dict(cfg=label, step=i, ms=round((t2 - t0) * 1e3, 3), host_queue_ms=round((t1 - t0) * 1e3, 3),
                                  new_dev_alloc=cur["dev_alloc"] - prev["dev_alloc"], new_dev_free=cur["dev_free"] - prev["dev_free"],
                                  segments=cur["segments"], reserved_gb=cur["reserved_gb"], allocated_gb=cur["allocated_gb"])))
            sys.stdout.flush()
            prev = cur
        # and free-running, as the bench times it
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            loss = connectionist_temporal_classification(model(x), labels, 0, x_len, l_len)
            opt.update(lossfun=lambda: loss)
        torch.cuda.synchronize()
        cur = stats()
        print(json.dumps(dict(cfg=label, free_running_ms_per_step=round((time.perf_counter() - t0) / 5 * 1e3, 3),
                              new_dev_alloc=cur["dev_alloc"] - prev["dev_alloc"], segments=cur["segments"], reserved_gb=cur["reserved_gb"])))
        del opt

    if not a.skip_ds2:
        torch.manual_seed(0)
        cfg = ds2.configure()
        cfg.vocab_size = 3000
        m = ds2.Model(cfg).to_gpu(0)
        run("ds2", m, 3000, 6)
        del m
    torch.cuda.empty_cache()
    print(json.dumps(dict(after_empty_cache=stats())))
    for nconv in (4, 8):
        import copy
        aa = copy.copy(args)
        aa.num_conv_layers = nconv
        torch.manual_seed(0)
        m = build_model(bench.cnn_config(aa, 119)).to_gpu(0)
        run("cnn%d" % nconv, m, 119, a.steps)
        del m
        torch.cuda.empty_cache()
        print(json.dumps(dict(after_empty_cache=stats())))


if __name__ == "__main__":
    main()
