import os, sys
sys.path.insert(0, "/root/repo/chainer-speech-recognition_amd"); sys.path.insert(0, "/root/repo")
import torch, json
import bench
from asr import functions as F
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda:0")
for side in (True, False, True, False):
    F._SIDE["enabled"] = side
    r = bench.time_cnn_config(args, 4, dev)
    print("side stream", side, r["ms_per_step"], r["step_spread"]["median_ms"], flush=True)
