"""ORACLE (test infrastructure only -- never imported by the product): the searches with a language
model in the loop, restated from the reference's Python loops in numpy float32.

* :func:`ctc_prefix_search_lm` -- ``CTCPrefixSearch.forward`` with shallow fusion / valid mixture
  (reference src/pydrobert/torch/_decoding.py:1064-1202; the mix :1113-1135, the state reorder
  :1154-1163, the freeze of utterances whose frames have run out :1165-1185) around the C
  restatement of the step (``oracle.ctc_prefix_search_advance``, :636-934).
* :func:`beam_search` -- ``BeamSearch.forward`` (:383-502, ``_to_width`` :352-372) around the C
  restatement of ``beam_search_advance`` (:41-155).

The language model is any object with the reference's interface in numpy
(``vocab_size``, ``update_input``, ``calc_idx_log_probs``, ``extract_by_src``, ``mix_by_mask``):
:class:`TableLM` is the stateless bigram-table model of tests/golden/make_golden.py,
:class:`NGramLM` the back-off n-gram model (``LookupLanguageModel``, _lm.py:403-515) scored by the
defining recursion over the n-gram tables -- one row per distinct context, cached -- and
:class:`CounterLM` a toy model WITH state (a per-row counter that feeds its scores), which makes the
extract / mix bookkeeping observable.

Pinned against the live reference by tests/golden/lm_search.npz (a LookupLanguageModel sweep: orders
2-4, both mixes, ragged lens, sos inside / outside the vocabulary; BeamSearch with eos /
finish_all_paths), tests/golden/search_sweep.npz and ctc_search.npz (bigram-table model):
tests/test_oracle_golden.py.
"""
import math

import numpy as np

NINF = -math.inf
f32 = np.float32


# ---------------------------------------------------------------------------------------------
# language models under the reference's interface, in numpy
# ---------------------------------------------------------------------------------------------
class TableLM:
    """Stateless bigram table: row = previous token, last row = start of sequence."""

    def __init__(self, table):
        self.table = np.asarray(table, dtype=f32)
        self.vocab_size = self.table.shape[1]

    def update_input(self, prev, hist):
        return prev

    def calc_idx_log_probs(self, hist, prev, idx):
        V, B = self.vocab_size, hist.shape[1]
        idx = np.broadcast_to(np.asarray(idx), (B,))
        tok = np.full((B,), V, dtype=np.int64)
        if hist.shape[0]:
            last = np.clip(hist[np.maximum(idx - 1, 0), np.arange(B)], 0, V - 1)
            tok = np.where(idx > 0, last, tok)
        return self.table[tok], prev

    def extract_by_src(self, prev, src):
        return prev

    def mix_by_mask(self, prev_true, prev_false, mask):
        return prev_true


class NGramLM:
    """Back-off n-gram scores by the defining recursion (tests/test_lm.py:249-275 of the reference):
    ``P(w | h) = table[h + w]`` if present and finite, else ``backoff(h) + P(w | h[1:])``; float64,
    rounded to float32 once.  One (V,) row per distinct context, cached."""

    def __init__(self, vocab_size, sos, prob_dicts):
        self.vocab_size, self.sos, self.order = int(vocab_size), int(sos), len(prob_dicts)
        V, N = self.vocab_size, self.order
        self._bo = [dict() for _ in range(N)]  # context tuple (length n) -> back-off weight
        self._ext = [dict() for _ in range(N)]  # context tuple (length n) -> (tokens, values) of explicit n+1-grams
        ext = [dict() for _ in range(N)]
        for n, d in enumerate(prob_dicts):
            for key, val in d.items():
                seq = (key,) if n == 0 else tuple(key)
                lp = float(val) if n == N - 1 else float(val[0])
                if n < N - 1:
                    self._bo[n + 1][seq] = float(val[1])
                if 0 <= seq[-1] < V and lp != NINF:
                    ext[n].setdefault(seq[:-1], []).append((seq[-1], lp))
        for n in range(N):
            for ctx, pairs in ext[n].items():
                w = np.array([p[0] for p in pairs], dtype=np.int64)
                self._ext[n][ctx] = (w, np.array([p[1] for p in pairs]))
        self._rows = {}

    def _row(self, ctx):
        row = self._rows.get(ctx)
        if row is None:
            if len(ctx) == 0:
                row = np.full((self.vocab_size,), NINF)
            else:
                row = self._bo[len(ctx)].get(ctx, 0.0) + self._row(ctx[1:])
            hit = self._ext[len(ctx)].get(ctx)
            if hit is not None:
                row = row.copy()
                row[hit[0]] = hit[1]
            self._rows[ctx] = row
        return row

    def update_input(self, prev, hist):
        return prev

    def calc_idx_log_probs(self, hist, prev, idx):
        B = hist.shape[1]
        idx = np.broadcast_to(np.asarray(idx), (B,))
        out = np.empty((B, self.vocab_size), dtype=f32)
        for b in range(B):
            ctx = []
            for n in range(self.order - 1, 0, -1):  # _lm.py:452-461: the tokens before idx, sos-padded
                p = int(idx[b]) - n
                ctx.append(int(hist[p, b]) if p >= 0 else self.sos)
            out[b] = self._row(tuple(ctx))
        return out, prev

    def extract_by_src(self, prev, src):
        return prev

    def mix_by_mask(self, prev_true, prev_false, mask):
        return prev_true


class CounterLM:
    """A model WITH state: every row carries a counter of how often it was extended by a real token
    (kept by extract_by_src / mix_by_mask exactly as the searches reorder it) and its scores depend on
    it: ``log_softmax(table[last token] * (1 + 0.1 * counter))``.  The torch twin is tests/_toy_lm.py."""

    def __init__(self, table):
        self.table = np.asarray(table, dtype=f32)
        self.vocab_size = self.table.shape[1]

    def update_input(self, prev, hist):
        if "count" not in prev:
            prev = {"count": np.zeros((hist.shape[1],), dtype=f32)}
        return prev

    def calc_idx_log_probs(self, hist, prev, idx):
        V, B = self.vocab_size, hist.shape[1]
        idx = np.broadcast_to(np.asarray(idx), (B,))
        tok = np.full((B,), V, dtype=np.int64)
        if hist.shape[0]:
            last = np.clip(hist[np.maximum(idx - 1, 0), np.arange(B)], 0, V - 1)
            tok = np.where(idx > 0, last, tok)
        x = self.table[tok] * (f32(1.0) + f32(0.1) * prev["count"])[:, None]
        return _log_softmax(x), {"count": prev["count"] + f32(1.0)}

    def extract_by_src(self, prev, src):
        return {"count": prev["count"][np.asarray(src)]}

    def mix_by_mask(self, prev_true, prev_false, mask):
        return {"count": np.where(np.asarray(mask), prev_true["count"], prev_false["count"])}


# ---------------------------------------------------------------------------------------------
# float32 pieces in the reference's order of operations
# ---------------------------------------------------------------------------------------------
def _softmax(x):
    x = np.asarray(x, dtype=f32)
    with np.errstate(invalid="ignore"):
        e = np.exp(x - x.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True, dtype=f32)).astype(f32)


def _log_softmax(x):
    x = np.asarray(x, dtype=f32)
    with np.errstate(invalid="ignore"):
        z = x - x.max(-1, keepdims=True)
        return (z - np.log(np.exp(z).sum(-1, keepdims=True, dtype=f32))).astype(f32)


# ---------------------------------------------------------------------------------------------
# CTCPrefixSearch.forward (_decoding.py:1064-1202)
# ---------------------------------------------------------------------------------------------
def ctc_prefix_search_lm(logits, width, lens=None, lm=None, beta=0.2, valid_mixture=False, initial_state=None):
    """-> (y (S, N, width) int64 zero past each length, y_lens (N, width), y_probs (N, width) float32)."""
    from . import ctc_prefix_search_advance

    logits = np.asarray(logits, dtype=f32)
    T, N, Vp1 = logits.shape
    V, W = Vp1 - 1, int(width)
    if lm is not None and lm.vocab_size != V:
        raise RuntimeError("Expected dim 2 of logits to be {}, got {}".format(lm.vocab_size + 1, Vp1))
    if lens is None:
        lens = np.full((N,), T, dtype=np.int64)
        len_min = len_max = T
    else:
        lens = np.asarray(lens, dtype=np.int64)
        len_min, len_max = (int(lens.min()), int(lens.max())) if N else (0, 0)
    probs = _softmax(logits)  # :1093
    nb = np.zeros((N, 1), dtype=f32)  # :1097-1105
    b = np.ones((N, 1), dtype=f32)
    y = np.zeros((0, N, 1), dtype=np.int64)
    y_lens = np.zeros((N, 1), dtype=np.int64)
    y_last = np.zeros((N, 1), dtype=np.int64)
    isp = np.ones((N, 1, 1), dtype=bool)
    prev = dict() if initial_state is None else initial_state
    fuse = lm is not None and beta != 0
    if lm is not None:
        prev = lm.update_input(prev, y)
    Kp = 1
    beta32 = f32(beta)
    for t in range(min(len_max, T)):
        valid = None if t < len_min else (t < lens)[:, None]  # (N, 1)
        nonext, blank = np.ascontiguousarray(probs[t, :, :V]), np.ascontiguousarray(probs[t, :, V])
        in_next = dict()
        if not fuse:
            ext = np.broadcast_to(nonext[:, None, :], (N, Kp, V))
        else:
            lm_lp, in_next = lm.calc_idx_log_probs(y.reshape(y.shape[0], N * Kp), prev, y_lens.reshape(-1))
            if valid_mixture:  # :1120-1128
                lm_p = beta32 * _softmax(lm_lp).reshape(N, Kp, V) * (f32(1) - blank.reshape(N, 1, 1))
                ext = f32(1.0 - beta) * nonext[:, None, :] + lm_p
            else:  # :1130-1135
                ext = np.exp(beta32 * _log_softmax(lm_lp)).reshape(N, Kp, V) * nonext[:, None, :]
        y_new, last_new, lens_new, (nb_new, b_new), isp_new, src, kept = ctc_prefix_search_advance(
            (np.ascontiguousarray(ext, dtype=f32), nonext, blank), W, (nb, b), y, y_last, y_lens, isp
        )
        if fuse:  # :1154-1163
            flat = (np.arange(0, Kp * N, Kp)[:, None] + src).reshape(-1)
            prev = lm.mix_by_mask(lm.extract_by_src(prev, flat), lm.extract_by_src(in_next, flat), kept.reshape(-1))
        if valid is None:
            y_lens, nb, b = lens_new, nb_new, b_new
        else:  # :1165-1181
            y_old = np.concatenate([np.broadcast_to(y, (y.shape[0], N, W)), np.zeros((1, N, W), np.int64)], 0)
            y_new = np.where(valid[None], y_new, y_old)
            y_lens = np.where(valid, lens_new, np.broadcast_to(y_lens, (N, W)))
            if Kp < W:
                pad = np.full((N, W - Kp), NINF, dtype=f32)
                nb, b = np.concatenate([nb, pad], 1), np.concatenate([b, pad], 1)
            nb, b = np.where(valid, nb_new, nb), np.where(valid, b_new, b)
        y, y_last, isp, Kp = y_new, last_new, isp_new, W  # (last / is-prefix keep spinning, :1183-1185)
    total = (nb + b).astype(f32)
    if Kp == 1 != W:  # :1190-1200
        y = np.repeat(y, W, 2)
        y_lens = np.repeat(y_lens, W, 1)
        total = np.concatenate([total, np.full((N, W - 1), NINF, dtype=f32)], 1)
    inside = np.arange(y.shape[0])[:, None, None] < y_lens[None]
    return np.where(inside, y, 0), y_lens, total


# ---------------------------------------------------------------------------------------------
# BeamSearch.forward (_decoding.py:352-502)
# ---------------------------------------------------------------------------------------------
def _to_width(y, lp, lens, W):
    S, N, Kp = y.shape
    if Kp < W:
        rem = W - Kp
        lp = np.concatenate([lp, np.full((N, rem), NINF, dtype=f32)], 1)
        y = np.concatenate([y, np.zeros((S, N, rem), np.int64)], 2)
        lens = np.concatenate([lens, np.zeros((N, rem), np.int64)], 1)
    elif Kp > W:  # (never taken by the loop: the step returns `width` paths)
        src = np.argsort(-lp, 1, kind="stable")[:, :W]
        lp = np.take_along_axis(lp, src, 1)
        y = np.take_along_axis(y, np.broadcast_to(src[None], (S, N, W)), 2)
        lens = np.take_along_axis(lens, src, 1)
    return y, lp, lens


def beam_search(lm, width, eos=None, finish_all_paths=False, pad_value=-100, initial_state=None, batch_size=None,
                max_iters=None):
    """-> (y (S, N, width) int64, y_lens (N, width), y_log_probs (N, width) float32); the batch
    dimension is dropped when ``batch_size`` is None, like the reference (:493-502)."""
    from . import beam_search_advance

    V, W = lm.vocab_size, int(width)
    if eos is not None:
        eos = (eos + V) % V  # the constructor's normalisation (:283-289)
    N = 1 if batch_size is None else int(batch_size)
    Kp = 1
    y = np.zeros((0, N), dtype=np.int64)
    prev = lm.update_input(dict() if initial_state is None else initial_state, y)
    y = y[:, :, None]
    lp = np.zeros((N, 1), dtype=f32)
    lens = np.zeros((N, 1), dtype=np.int64)
    if max_iters is None:
        if eos is None:
            raise RuntimeError("max_iters must be set when eos is unset")
        max_iters = 1073741824
    elif max_iters < 0:
        raise RuntimeError("max_iters must be non-negative, got {}".format(max_iters))
    pad_y = np.full((1, N, W), pad_value, dtype=np.int64)
    for t in range(max_iters):
        if eos is not None and t:  # :413-427
            last = np.take_along_axis(y.transpose(1, 2, 0), np.maximum(lens - 1, 0)[:, :, None], 2)[:, :, 0]
            eos_mask = (last == eos) & (lens > 0)
            done = eos_mask.all(1, keepdims=True) if finish_all_paths else eos_mask[:, :1]
            if done.all():
                break
        else:
            eos_mask = np.zeros((N, Kp), dtype=bool)
            done = eos_mask[:, :1]
        y_c = np.clip(y, 0, V - 1)  # :434
        lpt, in_next = lm.calc_idx_log_probs(y_c.reshape(y.shape[0], N * Kp), prev, np.asarray(t))
        lpt = _log_softmax(np.asarray(lpt, dtype=f32).reshape(N, Kp, V))  # :437-441
        if eos is not None:  # :448-458
            lpt = np.where(eos_mask[:, :, None], f32(NINF), lpt)
            onehot = np.arange(V) == eos
            lpt = np.where(eos_mask[:, :, None] & onehot[None, None], f32(0.0), lpt)
        y_new, lens_new, lp_new, src = beam_search_advance(lpt, W, lp, y_c, lens)  # :461-463
        if eos is not None:  # :465-468
            lens_new = lens_new - np.take_along_axis(eos_mask, src, 1).astype(np.int64)
        flat = (np.arange(0, Kp * N, Kp)[:, None] + src).reshape(-1)  # :471-477
        prev = lm.extract_by_src(in_next, flat)
        if eos is not None and done.any():  # :479-486
            y_p, lp_p, lens_p = _to_width(y, lp, lens, W)
            y_p = np.concatenate([y_p, pad_y], 0)
            if y_p.shape[0] != y_new.shape[0]:
                raise RuntimeError("history shapes differ: {} vs {}".format(y_p.shape, y_new.shape))
            y_new = np.where(done[None], y_p, y_new)
            lp_new = np.where(done, lp_p, lp_new)
            lens_new = np.where(done, lens_p, lens_new)
        y, lens, lp, Kp = y_new, lens_new, lp_new.astype(f32), W
    y, lp, lens = _to_width(y, lp, lens, W)
    if batch_size is None:
        y, lens, lp = y[:, 0], lens[0], lp[0]
    return y, lens, lp
