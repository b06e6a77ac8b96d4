"""ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy restatement of ctc_greedy_search (reference
_decoding.py:507-558), sequence_log_probs on tensors (_decoding.py:1516-1551) and
fill_after_eos (_string.py:30-42)."""
import numpy as np

__all__ = ["ctc_greedy_search", "fill_after_eos", "sequence_log_probs"]


def _np(x, dt=None):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    return x if dt is None else x.astype(dt)


def _log_softmax(x):
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


def ctc_greedy_search(logits, in_lens=None, blank_idx=-1, batch_first=False, is_probs=False):
    x = _np(logits, np.float64)
    V = x.shape[2]
    blank = (blank_idx + V) % V
    if not is_probs:
        x = _log_softmax(x)
    if not batch_first:
        x = x.transpose(1, 0, 2)
    N, T = x.shape[:2]
    mx, am = x.max(2), x.argmax(2)
    keep = am != blank
    keep[:, 1:] &= am[:, 1:] != am[:, :-1]
    if in_lens is not None:
        m = np.arange(T)[None] < _np(in_lens)[:, None]
        keep &= m
        mx = np.where(m, mx, 1.0 if is_probs else 0.0)
    out_lens = keep.sum(1)
    paths = am.copy()
    for n in range(N):
        paths[n, : out_lens[n]] = am[n][keep[n]]
    tot = mx.prod(1) if is_probs else mx.sum(1)
    if not batch_first:
        paths = paths.T
    return tot.astype(np.float32), paths.astype(np.int64), out_lens.astype(np.int64)


def sequence_log_probs(logits, hyp, dim=0, eos=None):
    x = _log_softmax(_np(logits, np.float64))
    h = _np(hyp).astype(np.int64)
    V = x.shape[-1]
    dim = dim % h.ndim
    mask = (h < 0) | (h >= V)
    if eos is not None:
        S = h.shape[dim]
        is_eos = h == eos
        first = np.where(is_eos.any(dim), is_eos.argmax(dim), S) + 1
        ar = np.arange(S).reshape([-1 if i == dim else 1 for i in range(h.ndim)])
        mask |= ar >= np.expand_dims(first, dim)
    idx = np.where(mask, 0, h)
    lp = np.take_along_axis(x, idx[..., None], -1)[..., 0]
    return np.where(mask, 0.0, lp).sum(dim).astype(np.float32)


def fill_after_eos(tokens, eos, dim=0, fill=None, value=None):
    """_string.py:30-42: positions strictly after the first eos along dim take the fill value
    (default: eos), converted to the dtype of the tensor being filled."""
    tok = _np(tokens)
    out = tok if value is None else _np(value)
    hit = tok == eos
    seen = np.cumsum(hit, axis=dim) - hit  # eos occurrences at earlier positions (dim: of the TOKENS)
    fill_ = float(eos) if fill is None else fill
    # masked_fill broadcasts the mask and the filled tensor against each other
    shape = np.broadcast_shapes(out.shape, seen.shape)
    out = np.broadcast_to(out, shape).copy()
    out[np.broadcast_to(seen > 0, shape)] = np.asarray(fill_).astype(out.dtype)
    return out
