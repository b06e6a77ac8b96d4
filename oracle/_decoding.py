"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes wrappers for pdt_oracle_decoding.c
(restating reference _decoding.py:41-155, :636-934, :1064-1202)."""
import ctypes

import numpy as np

__all__ = ["beam_search_advance", "ctc_prefix_search_advance", "ctc_prefix_search"]

_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_I = ctypes.c_int64


def declare(L):
    L.pdt_oracle_beam_search_advance.restype = _I
    L.pdt_oracle_beam_search_advance.argtypes = [
        _f32p, _I, _I, _I, _I, _f32p, _i64p, _I, _i64p, _i64p, _i64p, _f32p, _i64p,
    ]  # fmt: skip
    L.pdt_oracle_ctc_prefix_search_advance.restype = ctypes.c_int
    L.pdt_oracle_ctc_prefix_search_advance.argtypes = (
        [_f32p, _I, _I, _f32p, _f32p, _I, _I, _I, _I, _f32p, _f32p, _i64p, _I, _i64p, _i64p, _u8p]
        + [_i64p, _i64p, _i64p, _f32p, _f32p, _u8p, _i64p, _u8p]
    )
    L.pdt_oracle_ctc_prefix_search.restype = ctypes.c_int
    L.pdt_oracle_ctc_prefix_search.argtypes = [_f32p, _I, _I, _I, _i64p, _I, _i64p, _i64p, _f32p]


def _np(x, dtype):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def beam_search_advance(log_probs_t, width, log_probs_prev, y_prev, y_prev_lens=None):
    """_decoding.py:41-155 -> (y_next, y_next_lens, log_probs_next, next_src)."""
    from . import lib

    lpt = _np(log_probs_t, np.float32)
    lpp = _np(log_probs_prev, np.float32)
    yp = _np(y_prev, np.int64)
    ypl = None if y_prev_lens is None else _np(y_prev_lens, np.int64)
    N, Kp, V = lpt.shape
    S = yp.shape[0]
    L = lib()
    args = [_p(lpt, _f32p), N, Kp, V, int(width), _p(lpp, _f32p), _p(yp, _i64p), S, _p(ypl, _i64p)]
    S_out = L.pdt_oracle_beam_search_advance(*args, None, None, None, None)
    if S_out < 0:
        raise RuntimeError("Invalid lengths for t=0" if S_out == -2 else "invalid width")
    y_next = np.zeros((S_out, N, width), np.int64)
    lens = np.zeros((N, width), np.int64)
    lp = np.zeros((N, width), np.float32)
    src = np.zeros((N, width), np.int64)
    rc = L.pdt_oracle_beam_search_advance(
        *args, _p(y_next, _i64p), _p(lens, _i64p), _p(lp, _f32p), _p(src, _i64p)
    )
    assert rc == 0
    return y_next, lens, lp, src


def ctc_prefix_search_advance(probs_t, width, probs_prev, y_prev, y_prev_last, y_prev_lens,
                              prev_is_prefix):  # fmt: skip
    """_decoding.py:636-934.  Returns the reference's 7-tuple (probabilities as a pair)."""
    from . import lib

    ext, nonext, blank = (_np(x, np.float32) for x in probs_t)
    nb, b = (_np(x, np.float32) for x in probs_prev)
    yp = _np(y_prev, np.int64)
    last = _np(y_prev_last, np.int64)
    lens = _np(y_prev_lens, np.int64)
    isp = _np(prev_is_prefix, np.uint8)
    N, Kp, V = ext.shape
    S = yp.shape[0]
    W = int(width)
    y_next = np.zeros((S + 1, N, W), np.int64)
    o_last, o_lens, o_src = (np.zeros((N, W), np.int64) for _ in range(3))
    o_nb, o_b = (np.zeros((N, W), np.float32) for _ in range(2))
    o_isp = np.zeros((N, W, W), np.uint8)
    o_non = np.zeros((N, W), np.uint8)
    rc = lib().pdt_oracle_ctc_prefix_search_advance(
        _p(ext, _f32p), Kp * V, V, _p(nonext, _f32p), _p(blank, _f32p), N, Kp, V, W,
        _p(nb, _f32p), _p(b, _f32p), _p(yp, _i64p), S, _p(last, _i64p), _p(lens, _i64p),
        _p(isp, _u8p), _p(y_next, _i64p), _p(o_last, _i64p), _p(o_lens, _i64p), _p(o_nb, _f32p),
        _p(o_b, _f32p), _p(o_isp, _u8p), _p(o_src, _i64p), _p(o_non, _u8p),
    )  # fmt: skip
    if rc:
        raise RuntimeError("oracle ctc_prefix_search_advance failed")
    return y_next, o_last, o_lens, (o_nb, o_b), o_isp.astype(bool), o_src, o_non.astype(bool)


def ctc_prefix_search(logits, width, lens=None):
    """CTCPrefixSearch(width)(logits, lens) without LM (_decoding.py:1064-1202)."""
    from . import lib

    lg = _np(logits, np.float32)
    T, N, Vp1 = lg.shape
    ln = None if lens is None else _np(lens, np.int64)
    S = T if ln is None else (int(ln.max()) if N else 0)
    W = int(width)
    y = np.zeros((S, N, W), np.int64)
    y_lens = np.zeros((N, W), np.int64)
    y_probs = np.zeros((N, W), np.float32)
    rc = lib().pdt_oracle_ctc_prefix_search(
        _p(lg, _f32p), T, N, Vp1, _p(ln, _i64p), W, _p(y, _i64p), _p(y_lens, _i64p),
        _p(y_probs, _f32p),
    )  # fmt: skip
    if rc:
        raise RuntimeError("oracle ctc_prefix_search failed")
    return y, y_lens, y_probs
