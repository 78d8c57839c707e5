"""CPU oracle: standard CTC and Gram-CTC loss + gradient (float64 internals).  TEST INFRASTRUCTURE.

Restates, without the dense (B, N, N) connection matrices, what the reference computes in
``asr/loss/gram_ctc.py``:

* softmax / log                      asr/loss/gram_ctc.py:18-21, 48-57, 272-274
* path = [blank, u_i, b_i]* + blank  asr/loss/gram_ctc.py:24-32  (N = 3 L + 1)
* lattice edges                      asr/loss/gram_ctc.py:66-99 (forward), 103-140 (the same graph
                                     written on the reversed path)
* alpha / beta recursions            asr/loss/gram_ctc.py:142-178
* loss = -logsumexp(alpha_0+beta_0)  asr/loss/gram_ctc.py:279-281
* grad = (softmax - exp(label_prob - total)) * gy / B, zero for t >= input_length
                                     asr/loss/gram_ctc.py:284-297, 180-217

Standard CTC is Chainer's ``F.connectionist_temporal_classification`` (absent third-party; call
sites run/ctc/cnn/train.py:162,191).  It is the same computation on the path
[blank, u_i]* + blank (S = 2 L + 1) and equals Gram-CTC with ``label_bigram == -1``; the golden
cases ``ctc_*`` of tests/golden/gram_ctc.npz pin it through the reference's own file.

The edge set, derived from asr/loss/gram_ctc.py:80-98 (an edge goes from node s-k to node s):

    k = 0        always (self loop)
    k = 1, 2     iff the destination s is not a bigram node
    k = 3        iff s is a unigram node and u_i != u_{i-1}
    k = 5, 7     iff s is a bigram node
    k = 6        iff s is a bigram node and b_i != b_{i-2}
    and both end points lie inside the path (s < 3 len + 1) and are not bigram nodes with label -1.

Valid final nodes (asr/loss/gram_ctc.py:144-145,173 on the reversed path): the last blank, the last
unigram and the last bigram.
"""
import numpy as np

GRAM_KS = (0, 1, 2, 3, 5, 6, 7)
CTC_KS = (0, 1, 2)
NEG = -np.inf


def log_softmax(x, axis=-1):
    x = np.asarray(x, dtype=np.float64)
    m = x.max(axis=axis, keepdims=True)
    e = x - m
    return e - np.log(np.exp(e).sum(axis=axis, keepdims=True))


def _lse0(c):
    """logsumexp over axis 0 that maps an all -inf column to -inf."""
    m = c.max(axis=0)
    ms = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide="ignore"):
        return np.where(np.isfinite(m), ms + np.log(np.exp(c - ms).sum(axis=0)), NEG)


def gram_lattice(uni, big, length, blank=0):
    """labels (N,), alive (N,), allowed (len(GRAM_KS), N), final (N,) for one utterance."""
    L = int(length)
    N = 3 * L + 1
    uni = np.asarray(uni[:L], dtype=np.int64)
    big = np.asarray(big[:L], dtype=np.int64)
    s = np.arange(N)
    kind = s % 3
    labels = np.full(N, blank, dtype=np.int64)
    labels[1::3] = uni
    labels[2::3] = big
    alive = np.ones(N, dtype=bool)
    alive[2::3] = big != -1
    uni_ok = np.ones(L, dtype=bool)
    uni_ok[1:] = uni[1:] != uni[:-1]
    big_ok = np.ones(L, dtype=bool)
    big_ok[2:] = big[2:] != big[:-2]
    dst_uni_ok = np.zeros(N, dtype=bool)
    dst_uni_ok[1::3] = uni_ok
    dst_big_ok = np.zeros(N, dtype=bool)
    dst_big_ok[2::3] = big_ok
    allowed = np.zeros((len(GRAM_KS), N), dtype=bool)
    for j, k in enumerate(GRAM_KS):
        if k == 0:
            a = np.ones(N, dtype=bool)
        elif k in (1, 2):
            a = kind != 2
        elif k == 3:
            a = (kind == 1) & dst_uni_ok
        elif k in (5, 7):
            a = kind == 2
        else:  # k == 6
            a = (kind == 2) & dst_big_ok
        a = a & alive & (s - k >= 0)
        src_alive = np.zeros(N, dtype=bool)
        if k == 0:
            src_alive[:] = alive
        elif k < N:
            src_alive[k:] = alive[:N - k]
        allowed[j] = a & src_alive
    final = np.zeros(N, dtype=bool)
    final[N - 1] = True
    if N >= 3:
        final[N - 3] = True
        final[N - 2] = alive[N - 2]
    return labels, alive, allowed, final


def ctc_lattice(labels_b, length, blank=0):
    """Standard CTC path [blank, u_i]* + blank, S = 2 L + 1."""
    L = int(length)
    S = 2 * L + 1
    u = np.asarray(labels_b[:L], dtype=np.int64)
    labels = np.full(S, blank, dtype=np.int64)
    labels[1::2] = u
    s = np.arange(S)
    allowed = np.zeros((3, S), dtype=bool)
    allowed[0] = True
    allowed[1] = s >= 1
    skip = np.zeros(S, dtype=bool)
    ok = np.ones(L, dtype=bool)
    ok[1:] = u[1:] != u[:-1]
    ok[0] = False          # no node at s - 2 for the first label
    skip[1::2] = ok
    allowed[2] = skip
    final = np.zeros(S, dtype=bool)
    final[S - 1] = True
    if S >= 2:
        final[S - 2] = True
    return labels, np.ones(S, dtype=bool), allowed, final


def _alpha_beta(logy_path, allowed, final, ks):
    """logy_path (T, N) = log softmax gathered on the path (-inf on dead nodes)."""
    T, N = logy_path.shape
    alpha = np.full((T, N), NEG)
    beta = np.full((T, N), NEG)
    prev = np.full(N, NEG)
    prev[0] = 0.0
    cand = np.full((len(ks), N), NEG)
    for t in range(T):
        cand[:] = NEG
        for j, k in enumerate(ks):
            if k == 0:
                cand[j] = np.where(allowed[j], prev, NEG)
            elif k < N:
                cand[j, k:] = np.where(allowed[j, k:], prev[:N - k], NEG)
        prev = _lse0(cand) + logy_path[t]
        alpha[t] = prev
    nxt = np.where(final, 0.0, NEG)
    beta[T - 1] = nxt
    for t in range(T - 2, -1, -1):
        w = nxt + logy_path[t + 1]          # weight of standing on node s2 at time t+1
        cand[:] = NEG
        for j, k in enumerate(ks):
            if k == 0:
                cand[j] = np.where(allowed[j], w, NEG)
            elif k < N:
                cand[j, :N - k] = np.where(allowed[j, k:], w[k:], NEG)
        nxt = _lse0(cand)
        beta[t] = nxt
    return alpha, beta


def _loss_grad(xs, lattices, ks, input_length, reduce, gy, infeasible_loss=1e10):
    xs = np.asarray(xs)
    T, B, V = xs.shape
    logy = log_softmax(xs, axis=2)
    y = np.exp(logy)
    losses = np.zeros(B)
    grad = np.zeros((T, B, V))
    if input_length is None:
        input_length = np.full(B, T)
    gy = np.asarray(1.0 if gy is None else gy, dtype=np.float64)
    for b in range(B):
        labels, alive, allowed, final = lattices[b]
        xl = int(input_length[b])
        lab = np.where(alive, labels, 0)
        lp = logy[:xl, b][:, lab]
        lp = np.where(alive[None, :], lp, NEG)
        alpha, beta = _alpha_beta(lp, allowed, final, ks)
        ab = alpha + beta
        tot = _lse0(ab[0][:, None])[0]
        if not np.isfinite(tot):
            losses[b] = infeasible_loss
            occ = np.zeros((xl, V))
        else:
            losses[b] = -tot
            with np.errstate(under="ignore"):
                p = np.exp(ab - tot)
            p[~np.isfinite(ab)] = 0.0
            occ = np.zeros((xl, V))
            for s in np.nonzero(alive)[0]:
                occ[:, labels[s]] += p[:, s]
        scale = (gy / B) if reduce == "mean" else (gy[b] if gy.ndim else gy)
        grad[:xl, b] = (y[:xl, b] - occ) * scale
    loss = losses.mean() if reduce == "mean" else losses
    return loss, grad


def ctc_loss_grad(xs, labels, blank=0, input_length=None, label_length=None, reduce="mean", gy=None):
    """Chainer-convention CTC.  xs (T, B, V) pre-softmax activations; returns (loss, d loss / d xs)."""
    xs = np.asarray(xs)
    B = xs.shape[1]
    labels = np.asarray(labels)
    if label_length is None:
        label_length = np.full(B, labels.shape[1])
    lat = [ctc_lattice(labels[b], label_length[b], blank) for b in range(B)]
    return _loss_grad(xs, lat, CTC_KS, input_length, reduce, gy)


def gram_ctc_loss_grad(xs, label_unigram, label_bigram, blank=0, input_length=None, length_unigram=None,
                       reduce="mean", gy=None):
    """Reference-convention Gram-CTC (asr/loss/gram_ctc.py:300-315)."""
    xs = np.asarray(xs)
    B = xs.shape[1]
    label_unigram = np.asarray(label_unigram)
    label_bigram = np.asarray(label_bigram)
    if length_unigram is None:
        length_unigram = np.full(B, label_unigram.shape[1])
    lat = [gram_lattice(label_unigram[b], label_bigram[b], length_unigram[b], blank) for b in range(B)]
    return _loss_grad(xs, lat, GRAM_KS, input_length, reduce, gy)


def gram_connection_matrix(uni, big, length, max_nodes, blank=0, zero_padding=-1e10):
    """Dense (max_nodes, max_nodes) log connection matrix [dst, src] of one utterance, for comparing the
    band structure with the matrices the reference builds (asr/loss/gram_ctc.py:66-99)."""
    labels, alive, allowed, final = gram_lattice(uni, big, length, blank)
    N = labels.shape[0]
    m = np.full((max_nodes, max_nodes), zero_padding, dtype=np.float32)
    for j, k in enumerate(GRAM_KS):
        for s in range(N):
            if allowed[j, s]:
                m[s, s - k] = 0.0
    return m
