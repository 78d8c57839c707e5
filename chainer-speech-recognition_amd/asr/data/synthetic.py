"""Synthetic minibatches of the measured workload: what bench.py, the smoke test and the tests feed
the train step when there is no corpus.  Host tensors in the Loader's layout (asr/data/loaders/base.py:26-31):
x (B, 3, nmel, T) float32 ~ N(0, 1) (features after the Loader's normalisation are zero-mean / unit-variance per
(channel, mel)), labels (B, Lmax) int32 in 1..V-1 padded with the blank 0 (asr/data/processing.py:125), lengths int32."""
import torch


def synthetic_batch(B, T, V, Lmin=40, Lmax=120, seed=0, ragged=False, nmel=40):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, nmel, T, generator=g)
    l_len = torch.randint(Lmin, Lmax + 1, (B,), generator=g, dtype=torch.int32)
    labels = torch.randint(1, V, (B, Lmax), generator=g, dtype=torch.int32)
    for b in range(B):
        labels[b, l_len[b]:] = 0
    if ragged:
        x_len = torch.randint(int(0.6 * T), T + 1, (B,), generator=g, dtype=torch.int32)
    else:
        x_len = torch.full((B,), T, dtype=torch.int32)
    return x, labels, x_len, l_len


def synthetic_gram_labels(labels, l_len, V, first_bigram=119, p_missing=0.3, seed=0):
    """Gram-CTC labels for a synthetic batch: unigram ids stay, bigram ids U{first_bigram..V-1} with
    P(-1) = p_missing, bigram[:, 0] = -1 and -1 beyond the label length (asr/data/processing.py:139-147)."""
    g = torch.Generator().manual_seed(seed + 7)
    big = torch.randint(first_bigram, V, labels.shape, generator=g, dtype=torch.int32)
    drop = torch.rand(labels.shape, generator=g) < p_missing
    big[drop] = -1
    big[:, 0] = -1
    for b in range(labels.shape[0]):
        big[b, int(l_len[b]):] = -1
    return big
