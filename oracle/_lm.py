"""ORACLE (test infrastructure only -- never imported by the product): back-off n-gram scoring.

Two independent CPU restatements of what the reference's ``LookupLanguageModel`` computes
(reference src/pydrobert/torch/_lm.py):

* :func:`backoff_log_probs` -- the defining recursion straight from the n-gram tables, no trie
  (the brute-force check the reference's own test uses, tests/test_lm.py:249-275):
  ``P(w | h) = table[h + w]`` if present, else ``backoff(h) + P(w | h[1:])``.
* :func:`trie_log_probs` -- a scalar walk of the flattened reverse trie
  (``_lookup_calc_idx_log_probs``, _lm.py:403-515; layout :609-677), float32 in the
  reference's order of operations, so it matches the reference bit for bit.

Pinned against the live reference by tests/golden/lm.npz (tests/test_oracle_golden.py).
"""
import math

import numpy as np

NINF = -math.inf


def context_of(hist, b, pos, order, sos):
    """The ``order - 1`` tokens before position ``pos`` of column ``b``, oldest first, padded
    with ``sos`` (_lm.py:452-461)."""
    ctx = []
    for n in range(order - 1, 0, -1):
        p = pos - n
        ctx.append(int(hist[p, b]) if p >= 0 else sos)
    return tuple(ctx)


def backoff_log_probs(prob_dicts, vocab_size, sos, hist, idx):
    """(B, V) float64 log-probabilities by direct recursion over the tables."""
    N = len(prob_dicts)
    hist = np.asarray(hist)
    B = hist.shape[1]
    idx = np.broadcast_to(np.asarray(idx), (B,))

    def entry(seq):
        d = prob_dicts[len(seq) - 1]
        val = d.get(seq[0] if len(seq) == 1 else seq, None)
        if val is None:
            return None
        if len(seq) == N:
            return float(val), 0.0
        return float(val[0]), float(val[1])

    def logp(seq):
        e = entry(seq)
        if e is not None and e[0] != NINF:
            return e[0]
        if len(seq) == 1:
            return NINF
        ctx = entry(seq[:-1])
        return (ctx[1] if ctx is not None else 0.0) + logp(seq[1:])

    out = np.empty((B, vocab_size))
    for b in range(B):
        ctx = context_of(hist, b, int(idx[b]), N, sos)
        for v in range(vocab_size):
            out[b, v] = logp(ctx + (v,))
    return out


def trie_log_probs(logps, logbs, ids, offsets, vocab_size, sos, order, hist, idx):
    """(B, V) float32 by walking the reverse trie like _lm.py:476-513, one (row, v) at a time."""
    V, N = vocab_size, order
    logps = np.asarray(logps, dtype=np.float32)
    hist = np.asarray(hist)
    B = hist.shape[1]
    idx = np.broadcast_to(np.asarray(idx), (B,))
    if N == 1:
        return np.broadcast_to(logps[:V], (B, V)).copy()
    logbs = np.asarray(logbs, dtype=np.float32)
    ids = np.asarray(ids).astype(np.int64)
    offsets = np.asarray(offsets).astype(np.int64)
    shift = 0 if 0 <= sos < V else 1
    U = V + shift + 1

    def child(node, tok):
        lo, hi = node + offsets[node], node + 1 + offsets[node + 1]
        for c in range(lo, hi):
            if ids[c - U] == tok:
                return c
        return -1

    out = np.empty((B, V), dtype=np.float32)
    zero = np.float32(0.0)
    for b in range(B):
        ctx = [V if (shift and t == sos) else t for t in context_of(hist, b, int(idx[b]), N, sos)]
        ctx = [t if 0 <= t < U - 1 else -1 for t in ctx][::-1]  # ctx[n - 1] = n-th most recent
        bo = [zero] * (N + 1)
        node = ctx[0]
        bo[1] = logbs[node] if node >= 0 else zero
        for n in range(2, N):
            if node >= 0:
                node = child(node, ctx[n - 1]) if ctx[n - 1] >= 0 else -1
            bo[n] = logbs[node] if node >= 0 else zero
        for v in range(V):
            lp, last_b, node = logps[v], bo[1], v
            for n in range(1, N):
                if node >= 0:
                    node = child(node, ctx[n - 1]) if ctx[n - 1] >= 0 else -1
                cur_b = zero if n == N - 1 else bo[n + 1]
                clobber = node >= 0 and np.isfinite(logps[node])
                if clobber:
                    lp, last_b = logps[node], cur_b
                else:
                    lp, last_b = np.float32(np.float32(lp + cur_b) + last_b), zero
            out[b, v] = lp
    return out
