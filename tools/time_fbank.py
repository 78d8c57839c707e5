"""feature kernels alone (for rocprofv3 --kernel-trace --stats): python tools/time_fbank.py [iters]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    print(json.dumps(bench.time_features(torch.device("cuda", 0), 32, 160672, it)))
