"""A tiny synthetic bucketed corpus in the reference's on-disk format (tools/preprocess/bucket.py:55-74): used by
make_golden.py (to run the reference's Reader on it) and by the tests (to run this repository's Reader on the same)."""
import os
import pickle

import numpy as np

LAYOUT = {0: [7, 4], 1: [12], 2: [5, 9, 3]}          # bucket -> utterances per piece


def build(root, sampling_rate=16000, bucket_split_sec=0.5):
    rs = np.random.RandomState(42)
    os.makedirs(os.path.join(root, "signal"), exist_ok=True)
    os.makedirs(os.path.join(root, "sentence"), exist_ok=True)
    kana = ["ア", "イ", "ウ", "カ", "キ"]
    for bucket, pieces in LAYOUT.items():
        for piece, count in enumerate(pieces):
            lo = int(sampling_rate * bucket_split_sec * bucket) + 2000
            signals = [(rs.randn(lo + rs.randint(0, 3000)) * 1000).astype(np.int16) for _ in range(count)]
            sentences = ["".join(kana[(bucket + piece + u + j) % 5] for j in range(1 + (u % 3))) for u in range(count)]
            name = "{}_{}_{}.bucket".format(bucket, piece, count)
            with open(os.path.join(root, "signal", name), "wb") as f:
                pickle.dump(signals, f)
            with open(os.path.join(root, "sentence", name), "wb") as f:
                pickle.dump(sentences, f)
    return root
