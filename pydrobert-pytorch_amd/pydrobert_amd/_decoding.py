"""Beam-search decoding on MI355X.

Host-side mirror of the reference's ``_decoding.py`` for the operators on the hot path:
``CTCPrefixSearch`` / ``ctc_prefix_search_advance`` and ``BeamSearch`` /
``beam_search_advance``.  The step functions and the fused CTC search run in
``csrc/beam_advance.hip`` and ``csrc/ctc_search.hip`` through the C ABI
(``include/pdt_amd.h``); the Modules keep the reference's control flow around a
user-supplied language model.
"""
import math
import weakref
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch.library import custom_op, register_autograd

from . import _cabi, argcheck, config, switches
from ._lm import ExtractableSequentialLanguageModel, LookupLanguageModel, MixableSequentialLanguageModel

__all__ = [
    "BeamSearch",
    "CTCGreedySearch",
    "CTCPrefixSearch",
    "RandomWalk",
    "SequenceLogProbabilities",
    "beam_search_advance",
    "ctc_greedy_search",
    "ctc_prefix_search",
    "ctc_prefix_search_advance",
    "random_walk_advance",
    "sequence_log_probs",
]

MAX_CTC_WIDTH = 32


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.requires_grad:
        t = t.detach()
    return t if t.dtype == torch.float else t.float()


def _i64(t: torch.Tensor) -> torch.Tensor:
    if t.requires_grad:
        t = t.detach()
    return t if t.dtype == torch.long else t.long()


def _beam_search_advance_impl(
    log_probs_t: torch.Tensor,
    width: int,
    log_probs_prev: torch.Tensor,
    y_prev: torch.Tensor,
    y_prev_lens: Optional[torch.Tensor],
    grows: Optional[bool] = None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    # `grows`: the caller already knows whether some path is as long as the history (y_next
    # then has one row more, reference :133-135), which saves the read-back of max(y_prev_lens);
    # None = find out here
    if log_probs_t.dim() != 3:
        raise RuntimeError("log_probs_t must be 3 dimensional")
    N, Kp, V = log_probs_t.shape
    if width < 1:
        raise RuntimeError("Expected width to be >= 1, got {}".format(width))
    if log_probs_prev.shape != (N, Kp):
        raise RuntimeError(
            "Expected log_probs_prev to be of shape {}, got {}".format((N, Kp), tuple(log_probs_prev.shape))
        )
    if y_prev.dim() != 3:
        raise RuntimeError("y_prev must be 3 dimensional")
    if y_prev.shape[1:] != (N, Kp):
        raise RuntimeError(
            "Expected the last two dimensions of y_prev to be {}, got {}".format(
                (N, Kp), tuple(y_prev.shape[1:])
            )
        )
    S = y_prev.size(0)
    if y_prev_lens is not None and y_prev_lens.shape != (N, Kp):
        raise RuntimeError(
            "Expected y_prev_lens to have shape {}, got {}".format((N, Kp), tuple(y_prev_lens.shape))
        )
    device = _cabi.require_hip(log_probs_t, log_probs_prev, y_prev, y_prev_lens)
    lpt, lpp, yp = _f32(log_probs_t), _f32(log_probs_prev), _i64(y_prev)
    ypl = None if y_prev_lens is None else _i64(y_prev_lens)
    grow = True
    if grows is not None:
        grow = grows
    elif ypl is not None and N * Kp:
        # :133-135 don't make y bigger unless we have to; :139-140 -- the reference's own host read
        # (`y_prev_lens.max()`), as one small kernel that raises a word in pinned host memory
        flag = _cabi.host_flag()
        with _cabi.on_device(device):
            rc = _cabi.lib().pdt_lens_reach(
                _cabi.ptr(ypl), ypl.stride(0), ypl.stride(1), N, Kp, S, flag.ptr, _cabi.stream_ptr(device)
            )
            _cabi.check(rc, "pdt_lens_reach")
        seen = _cabi.wait_flag(flag, device)
        if S:
            grow = bool(seen & 1)
        elif seen & 2:
            raise RuntimeError("Invalid lengths for t=0")
    S_out = S + (1 if grow else 0)
    with _cabi.on_device(device):
        y_next = torch.empty((S_out, N, width), device=device, dtype=torch.long)
        y_next_lens = torch.empty((N, width), device=device, dtype=torch.long)
        next_src = torch.empty((N, width), device=device, dtype=torch.long)
        lp_next = torch.empty((N, width), device=device, dtype=torch.float)
        if N and V:
            rc = _cabi.lib().pdt_beam_search_advance(
                _cabi.ptr(lpt), lpt.stride(0), lpt.stride(1), lpt.stride(2), N, Kp, V, int(width),
                _cabi.ptr(lpp), lpp.stride(0), lpp.stride(1),
                _cabi.ptr(yp), S, yp.stride(0), yp.stride(1), yp.stride(2),
                _cabi.ptr(ypl), 0 if ypl is None else ypl.stride(0), 0 if ypl is None else ypl.stride(1),
                S_out, _cabi.ptr(y_next), _cabi.ptr(y_next_lens), _cabi.ptr(lp_next),
                _cabi.ptr(next_src), _cabi.stream_ptr(device),
            )  # fmt: skip
            _cabi.check(rc, "pdt_beam_search_advance")
    return y_next, y_next_lens, lp_next.to(log_probs_t.dtype), next_src


@custom_op("pydrobert_amd::beam_search_advance", mutates_args=())
def _beam_search_advance_op(
    log_probs_t: torch.Tensor,
    width: int,
    log_probs_prev: torch.Tensor,
    y_prev: torch.Tensor,
    y_prev_lens: Optional[torch.Tensor],
    grows: Optional[bool] = None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    return _beam_search_advance_impl(log_probs_t, width, log_probs_prev, y_prev, y_prev_lens, grows)


@_beam_search_advance_op.register_fake
def _(log_probs_t, width, log_probs_prev, y_prev, y_prev_lens, grows=None):
    N = log_probs_t.shape[0]
    S = y_prev.shape[0]
    if grows is not None:
        S_out = S + (1 if grows else 0)
    elif y_prev_lens is not None:  # data dependent: S or S + 1 (:133-135)
        S_out = torch.library.get_ctx().new_dynamic_size()
    else:
        S_out = S + 1
    return (
        y_prev.new_empty((S_out, N, width), dtype=torch.long),
        y_prev.new_empty((N, width), dtype=torch.long),
        log_probs_t.new_empty((N, width)),
        y_prev.new_empty((N, width), dtype=torch.long),
    )


def _beam_search_advance_setup(ctx, inputs, output):
    log_probs_t, _, log_probs_prev = inputs[:3]
    y_next, y_next_lens, lp_next, next_src = output
    # the token a new path ends in sits at its last position
    tok = y_next.gather(0, (y_next_lens - 1).clamp(min=0).unsqueeze(0)).squeeze(0)
    ctx.save_for_backward(next_src, tok, torch.isfinite(lp_next))
    ctx.shape_t, ctx.dtype_t, ctx.dtype_prev = log_probs_t.shape, log_probs_t.dtype, log_probs_prev.dtype


def _beam_search_advance_backward(ctx, g_y, g_lens, g_lp, g_src):
    # log_probs_next[n, k] = log_probs_prev[n, src] + log_probs_t[n, src, tok] (reference
    # _decoding.py:121-131: the top-k VALUES stay in the graph), so the gradient of an entry goes
    # to exactly those two addends; padded (-inf) entries carry none
    src, tok, valid = ctx.saved_tensors
    N, Kp, V = ctx.shape_t
    g = torch.where(valid, g_lp, torch.zeros_like(g_lp)).float()
    g_prev = g.new_zeros((N, Kp)).scatter_add_(1, src, g)
    g_t = g.new_zeros((N, Kp * V)).scatter_add_(1, src * V + tok.clamp(0, V - 1), g).view(N, Kp, V)
    return g_t.to(ctx.dtype_t), None, g_prev.to(ctx.dtype_prev), None, None, None


register_autograd(
    "pydrobert_amd::beam_search_advance", _beam_search_advance_backward, setup_context=_beam_search_advance_setup
)


def beam_search_advance(
    log_probs_t: torch.Tensor,
    width: int,
    log_probs_prev: torch.Tensor,
    y_prev: torch.Tensor,
    y_prev_lens: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Beam search step function (reference _decoding.py:41-155).

    Returns ``(y_next, y_next_lens, log_probs_next, next_src)``.
    """
    if not torch.jit.is_scripting():
        # nothing to trace, transform or differentiate: the implementation behind the operator, directly
        if _cabi.plain_call(log_probs_t, log_probs_prev, y_prev, y_prev_lens):
            return _beam_search_advance_impl(log_probs_t, width, log_probs_prev, y_prev, y_prev_lens)
    return torch.ops.pydrobert_amd.beam_search_advance(
        log_probs_t, width, log_probs_prev, y_prev, y_prev_lens
    )


def _ctc_prefix_search_advance_impl(
    ext: torch.Tensor,
    nonext: torch.Tensor,
    blank: torch.Tensor,
    width: int,
    nb: torch.Tensor,
    b: torch.Tensor,
    y_prev: torch.Tensor,
    y_prev_last: torch.Tensor,
    y_prev_lens: torch.Tensor,
    prev_is_prefix: torch.Tensor,
    lm_mix: Optional[Tuple[float, bool]] = None,
) -> Optional[Tuple[
    torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor,
    torch.Tensor, torch.Tensor,
]]:  # fmt: skip
    # `lm_mix` = (beta, valid_mixture): `ext` then holds the language model's scores (N, K', V) and the
    # kernel mixes them with the frame's probabilities itself (pdt_ctc_prefix_search_advance_lm); None is
    # returned when that entry point does not take the shapes (the caller makes the two calls)
    if width < 1:
        raise RuntimeError("width must be positive")
    if ext.dim() != 3:
        raise RuntimeError("ext_probs_t must be 3 dimensional")
    N, Kp, V = ext.shape
    if nonext.shape != (N, V):
        raise RuntimeError(
            "expected nonext_probs_t to have shape {}, got {}".format((N, V), tuple(nonext.shape))
        )
    if blank.shape != (N,):
        raise RuntimeError(
            "expected blank_probs_t to have shape {}, got {}".format((N,), tuple(blank.shape))
        )
    if nb.shape != (N, Kp):
        raise RuntimeError(
            "expected nb_probs_prev to have shape {}, got {}".format((N, Kp), tuple(nb.shape))
        )
    if b.shape != (N, Kp):
        raise RuntimeError(
            "expected b_probs_prev to have shape {}, got {}".format((N, Kp), tuple(b.shape))
        )
    if y_prev.dim() != 3:
        raise RuntimeError("y_prev must be 3 dimensional")
    if y_prev.shape[1:] != (N, Kp):
        raise RuntimeError(
            "expected last two dimensions of y_prev to be {}, got {}".format(
                (N, Kp), tuple(y_prev.shape[1:])
            )
        )
    S = y_prev.size(0)
    if y_prev_last.shape != (N, Kp):
        raise RuntimeError(
            "expected y_prev_last to have shape {}, got {}".format((N, Kp), tuple(y_prev_last.shape))
        )
    if y_prev_lens.shape != (N, Kp):
        raise RuntimeError(
            "expected y_prev_lens to have shape {}, got {}".format((N, Kp), tuple(y_prev_lens.shape))
        )
    if prev_is_prefix.shape != (N, Kp, Kp):
        raise RuntimeError(
            "expected prev_is_prefix to have shape {}, got {}".format(
                (N, Kp, Kp), tuple(prev_is_prefix.shape)
            )
        )
    device = _cabi.require_hip(ext, nonext, blank, nb, b, y_prev, y_prev_last, y_prev_lens,
                               prev_is_prefix)  # fmt: skip
    dtype = ext.dtype if lm_mix is None else nonext.dtype
    ext, nonext, blank, nb, b = (_f32(x) for x in (ext, nonext, blank, nb, b))
    yp, last, lens = _i64(y_prev), _i64(y_prev_last), _i64(y_prev_lens)
    isp = prev_is_prefix.detach() if prev_is_prefix.requires_grad else prev_is_prefix
    if isp.dtype != torch.bool:
        isp = isp.bool()
    W = int(width)
    with _cabi.on_device(device):
        y_next = torch.empty((S + 1, N, W), device=device, dtype=torch.long)
        o_last = torch.empty((N, W), device=device, dtype=torch.long)
        o_lens = torch.empty((N, W), device=device, dtype=torch.long)
        o_src = torch.empty((N, W), device=device, dtype=torch.long)
        o_nb = torch.empty((N, W), device=device, dtype=torch.float)
        o_b = torch.empty((N, W), device=device, dtype=torch.float)
        o_isp = torch.empty((N, W, W), device=device, dtype=torch.bool)
        o_non = torch.empty((N, W), device=device, dtype=torch.bool)
        if N and lm_mix is not None:
            ext = ext.contiguous()
            rc = _cabi.lib().pdt_ctc_prefix_search_advance_lm(
                _cabi.ptr(ext), float(lm_mix[0]), int(lm_mix[1]),
                _cabi.ptr(nonext), nonext.stride(0), nonext.stride(1),
                _cabi.ptr(blank), blank.stride(0), N, Kp, V, W,
                _cabi.ptr(nb), nb.stride(0), nb.stride(1), _cabi.ptr(b), b.stride(0), b.stride(1),
                _cabi.ptr(yp), S, yp.stride(0), yp.stride(1), yp.stride(2),
                _cabi.ptr(last), last.stride(0), last.stride(1),
                _cabi.ptr(lens), lens.stride(0), lens.stride(1),
                _cabi.ptr(isp), isp.stride(0), isp.stride(1), isp.stride(2),
                _cabi.ptr(y_next), _cabi.ptr(o_last), _cabi.ptr(o_lens), _cabi.ptr(o_nb),
                _cabi.ptr(o_b), _cabi.ptr(o_isp), _cabi.ptr(o_src), _cabi.ptr(o_non),
                _cabi.stream_ptr(device),
            )  # fmt: skip
            if rc == _cabi.PDT_E_UNSUPPORTED:
                return None
            _cabi.check(rc, "pdt_ctc_prefix_search_advance_lm")
        elif N:
            rc = _cabi.lib().pdt_ctc_prefix_search_advance(
                _cabi.ptr(ext), ext.stride(0), ext.stride(1), ext.stride(2),
                _cabi.ptr(nonext), nonext.stride(0), nonext.stride(1),
                _cabi.ptr(blank), blank.stride(0), N, Kp, V, W,
                _cabi.ptr(nb), nb.stride(0), nb.stride(1), _cabi.ptr(b), b.stride(0), b.stride(1),
                _cabi.ptr(yp), S, yp.stride(0), yp.stride(1), yp.stride(2),
                _cabi.ptr(last), last.stride(0), last.stride(1),
                _cabi.ptr(lens), lens.stride(0), lens.stride(1),
                _cabi.ptr(isp), isp.stride(0), isp.stride(1), isp.stride(2),
                _cabi.ptr(y_next), _cabi.ptr(o_last), _cabi.ptr(o_lens), _cabi.ptr(o_nb),
                _cabi.ptr(o_b), _cabi.ptr(o_isp), _cabi.ptr(o_src), _cabi.ptr(o_non),
                _cabi.stream_ptr(device),
            )  # fmt: skip
            _cabi.check(rc, "pdt_ctc_prefix_search_advance")
    if dtype != torch.float:
        o_nb, o_b = o_nb.to(dtype), o_b.to(dtype)
    return y_next, o_last, o_lens, o_nb, o_b, o_isp, o_src, o_non


@custom_op("pydrobert_amd::ctc_prefix_search_advance", mutates_args=())
def _ctc_prefix_search_advance_op(
    ext: torch.Tensor,
    nonext: torch.Tensor,
    blank: torch.Tensor,
    width: int,
    nb: torch.Tensor,
    b: torch.Tensor,
    y_prev: torch.Tensor,
    y_prev_last: torch.Tensor,
    y_prev_lens: torch.Tensor,
    prev_is_prefix: torch.Tensor,
) -> Tuple[
    torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor,
    torch.Tensor, torch.Tensor,
]:  # fmt: skip
    return _ctc_prefix_search_advance_impl(
        ext, nonext, blank, width, nb, b, y_prev, y_prev_last, y_prev_lens, prev_is_prefix
    )


def _ctc_prefix_search_advance_lm_impl(
    lm_log_probs: torch.Tensor,
    beta: float,
    valid_mixture: bool,
    nonext: torch.Tensor,
    blank: torch.Tensor,
    width: int,
    nb: torch.Tensor,
    b: torch.Tensor,
    y_prev: torch.Tensor,
    y_prev_last: torch.Tensor,
    y_prev_lens: torch.Tensor,
    prev_is_prefix: torch.Tensor,
) -> Tuple[
    torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor,
    torch.Tensor, torch.Tensor,
]:  # fmt: skip
    """``fusion_ext`` + ``ctc_prefix_search_advance`` as ONE kernel: the extension probabilities (reference
    _decoding.py:1110-1135) are formed inside the step and never written.  ``lm_log_probs`` is ``(N, K', V)``.
    No gradient; shapes the kernel does not take (V > 1024, beams above 32) make the two calls here."""
    out = _ctc_prefix_search_advance_impl(
        lm_log_probs, nonext, blank, width, nb, b, y_prev, y_prev_last, y_prev_lens, prev_is_prefix,
        (beta, valid_mixture),
    )  # fmt: skip
    if out is None:
        N, Kp, V = lm_log_probs.shape
        ext = _fusion_ext_impl(lm_log_probs.reshape(N * Kp, V), nonext, blank, beta, valid_mixture)
        out = _ctc_prefix_search_advance_impl(
            ext, nonext, blank, width, nb, b, y_prev, y_prev_last, y_prev_lens, prev_is_prefix
        )
    return out


@custom_op("pydrobert_amd::ctc_prefix_search_advance_lm", mutates_args=())
def _ctc_prefix_search_advance_lm_op(
    lm_log_probs: torch.Tensor,
    beta: float,
    valid_mixture: bool,
    nonext: torch.Tensor,
    blank: torch.Tensor,
    width: int,
    nb: torch.Tensor,
    b: torch.Tensor,
    y_prev: torch.Tensor,
    y_prev_last: torch.Tensor,
    y_prev_lens: torch.Tensor,
    prev_is_prefix: torch.Tensor,
) -> Tuple[
    torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor,
    torch.Tensor, torch.Tensor,
]:  # fmt: skip
    return _ctc_prefix_search_advance_lm_impl(
        lm_log_probs, beta, valid_mixture, nonext, blank, width, nb, b, y_prev, y_prev_last, y_prev_lens,
        prev_is_prefix,
    )  # fmt: skip


@_ctc_prefix_search_advance_lm_op.register_fake
def _(lm_log_probs, beta, valid_mixture, nonext, blank, width, nb, b, y_prev, y_prev_last, y_prev_lens, prev_is_prefix):
    N, W, S = lm_log_probs.shape[0], width, y_prev.shape[0]
    i64 = lambda *s: y_prev.new_empty(s, dtype=torch.long)  # noqa: E731
    return (
        i64(S + 1, N, W), i64(N, W), i64(N, W), nonext.new_empty((N, W)), nonext.new_empty((N, W)),
        nonext.new_empty((N, W, W), dtype=torch.bool), i64(N, W), nonext.new_empty((N, W), dtype=torch.bool),
    )  # fmt: skip


def _ctc_step_with_lm_scores(lm_log_probs, beta, valid_mixture, nonext, blank, width, nb, b, y_prev, y_prev_last,
                             y_prev_lens, prev_is_prefix):
    """The operator above, or -- nothing tracing, transforming or differentiating -- what is behind it."""
    args = (lm_log_probs, beta, valid_mixture, nonext, blank, width, nb, b, y_prev, y_prev_last, y_prev_lens,
            prev_is_prefix)  # fmt: skip
    if _cabi.plain_call(lm_log_probs, nonext, blank, nb, b, y_prev, y_prev_last, y_prev_lens, prev_is_prefix):
        return _ctc_prefix_search_advance_lm_impl(*args)
    return torch.ops.pydrobert_amd.ctc_prefix_search_advance_lm(*args)


@_ctc_prefix_search_advance_op.register_fake
def _(ext, nonext, blank, width, nb, b, y_prev, y_prev_last, y_prev_lens, prev_is_prefix):
    N, W, S = ext.shape[0], width, y_prev.shape[0]
    i64 = lambda *s: y_prev.new_empty(s, dtype=torch.long)  # noqa: E731
    return (
        i64(S + 1, N, W), i64(N, W), i64(N, W), ext.new_empty((N, W)), ext.new_empty((N, W)),
        ext.new_empty((N, W, W), dtype=torch.bool), i64(N, W), ext.new_empty((N, W), dtype=torch.bool),
    )  # fmt: skip


def _ctc_advance_setup(ctx, inputs, output):
    ext, nonext, blank, _, nb, b, y_prev, y_prev_last, y_prev_lens, prev_is_prefix = inputs
    _, o_last, _, o_nb, _, _, o_src, o_non = output
    ctx.save_for_backward(ext, nonext, blank, nb, b, y_prev, y_prev_last, y_prev_lens, prev_is_prefix,
                          o_last, o_nb, o_src, o_non)  # fmt: skip


def _ctc_advance_backward(ctx, g_y, g_last, g_lens, g_nb, g_b, g_isp, g_src, g_non):
    """Adjoint of the masses of one CTC prefix-search step (reference _decoding.py:777-880; the
    selection itself is piecewise constant).  A new entry i with source s = next_src[i] holds
      extension by v:   nb' = w(s, v) * ext[s, v],  b' = 0,   w(s, v) = (nb[s] if v != last[s] else 0) + b[s]
      non-extension:    nb' = nb[s] * nonext[last[s]] + sum over prefixes k that BECOME s when extended
                              by need(k, s) of w(k, need) * ext[k, need],
                        b'  = (nb[s] + b[s]) * blank
    The dense work is (N, K', K') -- nothing of size V besides the scatter into grad ext."""
    (ext, nonext, blank, nb, b, y_prev, last, lens, is_prefix, o_last, o_nb, src, non) = ctx.saved_tensors
    N, Kp, V = ext.shape
    S = y_prev.shape[0]
    f = torch.float
    ext_, nonext_, blank_, nb_, b_ = ext.to(f), nonext.to(f), blank.to(f), nb.to(f), b.to(f)
    valid = torch.isfinite(o_nb)
    zero = torch.zeros((), device=ext.device, dtype=f)
    # absent (padded) prefixes hold -inf masses: they are the source of nothing valid
    nb_, b_ = torch.where(torch.isfinite(nb_), nb_, zero), torch.where(torch.isfinite(b_), b_, zero)
    gnb = torch.where(valid, g_nb.to(f), zero)
    gb = torch.where(valid & non, g_b.to(f), zero)
    lastc = last.clamp(0, V - 1)
    # --- extension entries
    is_ext = valid & ~non
    tok = o_last.clamp(0, V - 1)
    last_s = lastc.gather(1, src)
    w = torch.where(tok != last_s, nb_.gather(1, src), zero) + b_.gather(1, src)
    e = ext_.reshape(N, Kp * V).gather(1, src * V + tok)
    ge = torch.where(is_ext, gnb, zero)
    g_ext = ge.new_zeros((N, Kp * V)).scatter_add_(1, src * V + tok, ge * w)
    g_nb_prev = ge.new_zeros((N, Kp)).scatter_add_(1, src, torch.where(tok != last_s, ge * e, zero))
    g_b_prev = ge.new_zeros((N, Kp)).scatter_add_(1, src, ge * e)
    # --- non-extension entries: gradient of stay_nb[s] / stay_b[s], summed over the entries that kept s
    gs_nb = ge.new_zeros((N, Kp)).scatter_add_(1, src, torch.where(non, gnb, zero))
    gs_b = ge.new_zeros((N, Kp)).scatter_add_(1, src, gb)
    p_last = nonext_.gather(1, lastc)
    g_nonext = ge.new_zeros((N, V)).scatter_add_(1, lastc, gs_nb * nb_)
    g_nb_prev = g_nb_prev + gs_nb * p_last + gs_b * blank_.unsqueeze(1)
    g_b_prev = g_b_prev + gs_b * blank_.unsqueeze(1)
    g_blank = (gs_b * (nb_ + b_)).sum(1)
    # merged extensions: prefix k + need(k, s) == prefix s
    if S:
        at = lens.clamp(max=S - 1).unsqueeze(2).expand(N, Kp, Kp).transpose(0, 1)
        need = y_prev.gather(0, at).transpose(0, 1).clamp(0, V - 1)  # (N, k, s)
    else:
        need = torch.zeros((N, Kp, Kp), dtype=torch.long, device=ext.device)
    becomes = ((lens + 1).unsqueeze(2) == lens.unsqueeze(1)) & is_prefix.bool()
    gm = torch.where(becomes, gs_nb.unsqueeze(1).expand(N, Kp, Kp), zero)  # d stay_nb[s] / d term(k, s)
    differs = need != lastc.unsqueeze(2)
    wk = torch.where(differs, nb_.unsqueeze(2), zero) + b_.unsqueeze(2)
    ek = ext_.gather(2, need)
    k_idx = torch.arange(Kp, device=ext.device).view(1, Kp, 1)
    g_ext.scatter_add_(1, (k_idx * V + need).reshape(N, Kp * Kp), (gm * wk).reshape(N, Kp * Kp))
    g_nb_prev = g_nb_prev + torch.where(differs, gm * ek, zero).sum(2)
    g_b_prev = g_b_prev + (gm * ek).sum(2)
    return (g_ext.view(N, Kp, V).to(ext.dtype), g_nonext.to(nonext.dtype), g_blank.to(blank.dtype), None,
            g_nb_prev.to(nb.dtype), g_b_prev.to(b.dtype), None, None, None, None)  # fmt: skip


register_autograd(
    "pydrobert_amd::ctc_prefix_search_advance", _ctc_advance_backward, setup_context=_ctc_advance_setup
)


def ctc_prefix_search_advance(
    probs_t: Tuple[torch.Tensor, torch.Tensor, torch.Tensor],
    width: int,
    probs_prev: Tuple[torch.Tensor, torch.Tensor],
    y_prev: torch.Tensor,
    y_prev_last: torch.Tensor,
    y_prev_lens: torch.Tensor,
    prev_is_prefix: torch.Tensor,
) -> Tuple[
    torch.Tensor, torch.Tensor, torch.Tensor, Tuple[torch.Tensor, torch.Tensor], torch.Tensor,
    torch.Tensor, torch.Tensor,
]:  # fmt: skip
    """CTC prefix search step function (reference _decoding.py:636-934).

    Returns ``(y_next, y_next_last, y_next_lens, (nb_probs_next, b_probs_next),
    next_is_prefix, next_src, next_is_nonext)``.
    """
    if not torch.jit.is_scripting():
        # nothing to trace, transform or differentiate: the implementation behind the operator, directly
        if _cabi.plain_call(probs_t[0], probs_t[1], probs_t[2], probs_prev[0], probs_prev[1], y_prev,
                            y_prev_last, y_prev_lens, prev_is_prefix):  # fmt: skip
            y_next, last, lens, nb, b, isp, src, non = _ctc_prefix_search_advance_impl(
                probs_t[0], probs_t[1], probs_t[2], width, probs_prev[0], probs_prev[1], y_prev,
                y_prev_last, y_prev_lens, prev_is_prefix,
            )  # fmt: skip
            return y_next, last, lens, (nb, b), isp, src, non
    y_next, last, lens, nb, b, isp, src, non = torch.ops.pydrobert_amd.ctc_prefix_search_advance(
        probs_t[0], probs_t[1], probs_t[2], width, probs_prev[0], probs_prev[1], y_prev,
        y_prev_last, y_prev_lens, prev_is_prefix,
    )  # fmt: skip
    return y_next, last, lens, (nb, b), isp, src, non


@custom_op("pydrobert_amd::ctc_prefix_search", mutates_args=())
def _ctc_prefix_search_op(
    logits: torch.Tensor, width: int, lens: Optional[torch.Tensor]
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    if logits.dim() != 3:
        raise RuntimeError("logits must be 3 dimensional")  # :1073-1074
    device = _cabi.require_hip(logits, lens)
    T, N, Vp1 = logits.shape
    V = Vp1 - 1
    if V < 1:
        raise RuntimeError("logits must have at least one non-blank class")
    if width < 1:
        raise RuntimeError("width must be positive")
    if width > MAX_CTC_WIDTH:  # (CTCPrefixSearch / ctc_prefix_search run such beams frame by frame)
        raise RuntimeError("the one-kernel search holds at most {} prefixes".format(MAX_CTC_WIDTH))
    dtype = logits.dtype
    logits = _f32(logits)
    if lens is None:
        S = T
    elif lens.dim() != 1:
        raise RuntimeError("lens must be 1 dimensional")  # :1084-1085
    elif lens.size(0) != N:
        raise RuntimeError("expected dim 0 of lens to be {}, got {}".format(N, lens.size(0)))
    else:
        lens = _i64(lens).contiguous()
        S = int(lens.max().item()) if N else 0  # the reference's len_max host read (:1089)
        S = max(0, min(S, T))
    L = _cabi.lib()
    with torch.cuda.device(device):
        y = torch.empty((S, N, width), device=device, dtype=torch.long)
        y_lens = torch.empty((N, width), device=device, dtype=torch.long)
        y_probs = torch.empty((N, width), device=device, dtype=torch.float)
        ws = torch.empty(
            (int(L.pdt_ctc_prefix_search_workspace_bytes(T, N, V, width)),),
            device=device, dtype=torch.uint8,
        )  # fmt: skip
        rc = L.pdt_ctc_prefix_search(
            _cabi.ptr(logits), T, N, V, logits.stride(0), logits.stride(1), logits.stride(2),
            _cabi.ptr(lens), int(width), S, _cabi.ptr(y), _cabi.ptr(y_lens), _cabi.ptr(y_probs),
            _cabi.ptr(ws), _cabi.stream_ptr(device),
        )  # fmt: skip
    _cabi.check(rc, "pdt_ctc_prefix_search")
    return y, y_lens, y_probs.to(dtype)


@_ctc_prefix_search_op.register_fake
def _(logits, width, lens):
    T, N = logits.shape[0], logits.shape[1]
    S = T if lens is None else torch.library.get_ctx().new_dynamic_size()  # lens.max() (:1089)
    return (
        logits.new_empty((S, N, width), dtype=torch.long),
        logits.new_empty((N, width), dtype=torch.long),
        logits.new_empty((N, width)),
    )


def ctc_prefix_search(
    logits: torch.Tensor, width: int, lens: Optional[torch.Tensor] = None
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``CTCPrefixSearch(width)(logits, lens)`` without a language model, as ONE kernel.

    Reference: ``CTCPrefixSearch.forward`` (_decoding.py:1064-1202) with ``lm=None``.
    Returns ``(y (S, N, width) int64, y_lens (N, width) int64, y_probs (N, width))``; rows of
    ``y`` beyond ``y_lens`` are zero (the reference leaves them undefined).  The fused kernel's
    probabilities are cut off from the graph; logits that require grad take the frame-by-frame
    route of :class:`CTCPrefixSearch` instead (same beams, differentiable probabilities).
    """
    if torch.jit.is_scripting() or not (
        width > 32 or (torch.is_grad_enabled() and logits.requires_grad)  # (32: MAX_CTC_WIDTH)
    ):
        return torch.ops.pydrobert_amd.ctc_prefix_search(logits, width, lens)
    return CTCPrefixSearch(width)(logits, lens)


def _fusion_ext_impl(
    lm_log_probs: torch.Tensor, nonext: torch.Tensor, blank: torch.Tensor, beta: float, valid_mixture: bool
) -> torch.Tensor:
    """Extension probabilities ``(N, K', V)`` of one frame from the LM scores ``(N * K', V)`` and
    the frame's CTC probabilities, in one pass (``csrc/fusion_ext.hip``; reference
    _decoding.py:1110-1135).  No gradient: ``CTCPrefixSearch`` composes torch ops instead when
    one is wanted."""
    N, V = nonext.shape
    if lm_log_probs.dim() != 2 or lm_log_probs.size(1) != V or (N and lm_log_probs.size(0) % N):
        raise RuntimeError("lm_log_probs must be of shape (N * K', V)")
    Kp = lm_log_probs.size(0) // N if N else 1
    device = _cabi.require_hip(lm_log_probs, nonext, blank)
    lm, ne, bl = _f32(lm_log_probs).contiguous(), _f32(nonext), _f32(blank)
    with _cabi.on_device(device):
        out = torch.empty((N, Kp, V), device=device, dtype=torch.float)
        if N and V:
            rc = _cabi.lib().pdt_fusion_ext(
                _cabi.ptr(lm), N, Kp, V, _cabi.ptr(ne), ne.stride(0), ne.stride(1), _cabi.ptr(bl),
                bl.stride(0), float(beta), int(valid_mixture), _cabi.ptr(out), _cabi.stream_ptr(device),
            )  # fmt: skip
            _cabi.check(rc, "pdt_fusion_ext")
    return out.to(nonext.dtype)


@custom_op("pydrobert_amd::fusion_ext", mutates_args=())
def _fusion_ext_op(
    lm_log_probs: torch.Tensor, nonext: torch.Tensor, blank: torch.Tensor, beta: float, valid_mixture: bool
) -> torch.Tensor:
    return _fusion_ext_impl(lm_log_probs, nonext, blank, beta, valid_mixture)


def _fusion_ext(lm_log_probs, nonext, blank, beta, valid_mixture):
    """The operator, or -- nothing tracing, transforming or differentiating -- what is behind it."""
    if _cabi.plain_call(lm_log_probs, nonext, blank):
        return _fusion_ext_impl(lm_log_probs, nonext, blank, beta, valid_mixture)
    return torch.ops.pydrobert_amd.fusion_ext(lm_log_probs, nonext, blank, beta, valid_mixture)


@_fusion_ext_op.register_fake
def _(lm_log_probs, nonext, blank, beta, valid_mixture):
    N, V = nonext.shape
    return nonext.new_empty((N, lm_log_probs.shape[0] // max(N, 1), V))


class CTCPrefixSearch(torch.nn.Module):
    """Beam search over CTC prefixes, optionally with shallow fusion (reference
    _decoding.py:937-1204).

    Without a language model (``lm=None`` or ``beta == 0``) the whole search is one fused
    kernel.  With one, the reference's per-frame loop is kept -- the LM forward is the
    user's PyTorch code -- and each frame's prefix bookkeeping is one kernel
    (``ctc_prefix_search_advance``).
    """

    __constants__ = ["width", "beta", "valid_mixture"]

    def __init__(
        self,
        width: int,
        beta: float = 0.2,
        lm: Optional[MixableSequentialLanguageModel] = None,
        valid_mixture: bool = False,
    ):
        width = argcheck.is_posi(width, name="width")
        beta = argcheck.is_closed01(beta, name="beta")
        valid_mixture = argcheck.is_bool(valid_mixture, "valid_mixture")
        super().__init__()
        self.width, self.beta, self.valid_mixture = width, beta, valid_mixture
        if lm is None:
            self.add_module("lm", None)
        else:
            self.lm = lm

    def reset_parameters(self) -> None:
        if self.lm is not None and hasattr(self.lm, "reset_parameters"):
            self.lm.reset_parameters()

    def extra_repr(self) -> str:
        return ", ".join("{}={}".format(x, getattr(self, x)) for x in self.__constants__)

    def forward(
        self,
        logits: torch.Tensor,
        lens: Optional[torch.Tensor] = None,
        prev_: Optional[Dict[str, torch.Tensor]] = None,
        initial_state: Optional[Dict[str, torch.Tensor]] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        # (``prev_`` is the reference's runtime keyword, _decoding.py:1064-1068; ``initial_state`` the
        # name its documented call signature uses, :1053-1060 -- both are accepted)
        if initial_state is None:
            initial_state = prev_
        if logits.dim() != 3:
            raise RuntimeError("logits must be 3 dimensional")
        # The one-kernel search returns probabilities that are cut off from the graph.  When the
        # caller wants gradients with respect to the logits (the reference's probabilities are
        # differentiable, _decoding.py:1093, :1188), the search runs frame by frame instead: every
        # frame is one kernel whose masses carry an autograd formula.
        # Beams wider than the one-kernel search holds (32 prefixes) also go frame by frame, on the
        # plain step kernel (csrc/advance_wide.hip).
        stepwise = (torch.is_grad_enabled() and logits.requires_grad) or self.width > 32  # (MAX_CTC_WIDTH; TorchScript takes no globals)
        prev: Dict[str, torch.Tensor] = dict()
        if initial_state is not None:
            prev = initial_state
        if self.lm is None:
            if stepwise:
                return self._frame_by_frame(logits, lens, prev)
            return ctc_prefix_search(logits, self.width, lens)
        else:
            if self.lm.vocab_size != logits.size(2) - 1:
                raise RuntimeError(
                    "Expected dim 2 of logits to be {}, got {}".format(self.lm.vocab_size + 1, logits.size(2))
                )
            if self.beta == 0.0 and not stepwise:
                return ctc_prefix_search(logits, self.width, lens)
            return self._frame_by_frame(logits, lens, prev)

    @torch.jit.unused
    def _fuses_lookup_lm(self, logits: torch.Tensor) -> bool:
        """Whether a frame with the language model in the loop can run as ONE kernel
        (csrc/ctc_lm_step.hip): the model is this package's n-gram LookupLanguageModel with its own
        scoring methods (a subclass that overrides them must be called), of order two or more with its
        forward index built, the beam fits the frame routine and nothing wants gradients."""
        lm = self.lm
        if type(lm) is not LookupLanguageModel or not switches.get("PDT_CTC_LM_FUSED"):
            return False
        if lm.max_ngram < 2 or self.width > 32 or self.beta == 0.0 or logits.device.type != "cuda":
            return False
        shift = 0 if (0 <= lm.sos < lm.vocab_size) else 1
        if lm.succ_start.numel() != lm.vocab_size + shift + 2 or lm.logps.device != logits.device:
            return False
        return not (torch.is_grad_enabled() and logits.requires_grad)

    @torch.jit.unused
    def _lookup_lm_frame(
        self, nonext: torch.Tensor, blank: torch.Tensor, nb: torch.Tensor, b: torch.Tensor, y: torch.Tensor,
        y_last: torch.Tensor, y_lens: torch.Tensor, is_prefix: torch.Tensor, y_next: torch.Tensor,
        lens: Optional[torch.Tensor], t: int,
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """Scores of the n-gram model, the mix with the frame's probabilities and the prefix step in one
        launch (include/pdt_amd.h: pdt_ctc_lookup_lm_advance).  ``y_next`` (t + 1, N, W) is written in
        place (the caller alternates between two buffers); utterances with ``lens <= t`` keep their beam."""
        lm, W = self.lm, self.width
        N, V = nonext.shape
        Kp, S = nb.size(1), y.size(0)
        device = _cabi.require_hip(nonext, blank, nb, b, y, y_last, y_lens, is_prefix, y_next, lens, *_lm_buffers(lm))
        nonext, blank, nb, b = (_f32(x) for x in (nonext, blank, nb, b))
        shift = 0 if (0 <= lm.sos < V) else 1
        with torch.cuda.device(device):
            o_last = torch.empty((N, W), device=device, dtype=torch.long)
            o_lens = torch.empty((N, W), device=device, dtype=torch.long)
            o_src = torch.empty((N, W), device=device, dtype=torch.long)
            o_nb = torch.empty((N, W), device=device, dtype=torch.float)
            o_b = torch.empty((N, W), device=device, dtype=torch.float)
            o_isp = torch.empty((N, W, W), device=device, dtype=torch.bool)
            o_non = torch.empty((N, W), device=device, dtype=torch.bool)
            if N:
                rc = _cabi.lib().pdt_ctc_lookup_lm_advance(
                    _cabi.ptr(nonext), nonext.stride(0), nonext.stride(1), _cabi.ptr(blank), blank.stride(0),
                    N, Kp, V, W, _cabi.ptr(nb), nb.stride(0), nb.stride(1), _cabi.ptr(b), b.stride(0), b.stride(1),
                    _cabi.ptr(y), S, y.stride(0), y.stride(1), y.stride(2),
                    _cabi.ptr(y_last), y_last.stride(0), y_last.stride(1),
                    _cabi.ptr(y_lens), y_lens.stride(0), y_lens.stride(1),
                    _cabi.ptr(is_prefix), is_prefix.stride(0), is_prefix.stride(1), is_prefix.stride(2),
                    _cabi.ptr(lm.logps), _cabi.ptr(lm.logbs), _cabi.ptr(lm.child_start), _cabi.ptr(lm.ids_wide),
                    _cabi.ptr(lm.succ_start), _cabi.ptr(lm.succ_tok), _cabi.ptr(lm.succ_node),
                    lm.max_ngram, V + shift + 1, lm.sos, float(self.beta), int(self.valid_mixture),
                    _cabi.ptr(y_next), _cabi.ptr(o_last), _cabi.ptr(o_lens), _cabi.ptr(o_nb), _cabi.ptr(o_b),
                    _cabi.ptr(o_isp), _cabi.ptr(o_src), _cabi.ptr(o_non), y.element_size(),
                    _cabi.ptr(lens), t, y_next.stride(0), y_next.stride(1), y_next.stride(2),
                    _cabi.stream_ptr(device),
                )  # fmt: skip
                _cabi.check(rc, "pdt_ctc_lookup_lm_advance")
        return o_last, o_lens, o_nb, o_b, o_isp

    @torch.jit.unused
    def _searches_in_one_call(self) -> bool:
        """PDT_CTC_LM_SEARCH=0 keeps the host's frame loop around the one-kernel frames (comparisons)."""
        return switches.get("PDT_CTC_LM_SEARCH") != 0

    @torch.jit.unused
    def _lookup_lm_search(
        self, probs: torch.Tensor, lens: Optional[torch.Tensor], n_frames: int
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Every frame of the search with the n-gram model in the loop from ONE call of the library
        (include/pdt_amd.h: pdt_ctc_lookup_lm_search): the frame kernel of :meth:`_lookup_lm_frame`
        launched ``n_frames`` times from C, the beam's state and histories in a workspace in between."""
        lm, W = self.lm, self.width
        T, N, V = probs.size(0), probs.size(1), probs.size(2) - 1
        device = _cabi.require_hip(probs, lens, *_lm_buffers(lm))
        probs = _f32(probs)
        shift = 0 if (0 <= lm.sos < V) else 1
        L = _cabi.lib()
        with torch.cuda.device(device):
            y = torch.empty((n_frames, N, W), device=device, dtype=torch.long)
            y_lens = torch.empty((N, W), device=device, dtype=torch.long)
            nb = torch.empty((N, W), device=device, dtype=torch.float)
            b = torch.empty((N, W), device=device, dtype=torch.float)
            if N:
                ws_bytes = int(L.pdt_ctc_lookup_lm_search_workspace_bytes(n_frames, N, V, W, lm.max_ngram, V + shift + 1))
                ws = torch.empty(ws_bytes, device=device, dtype=torch.uint8)
                lens_dev = None if lens is None else _i64(lens).contiguous()
                rc = L.pdt_ctc_lookup_lm_search(
                    _cabi.ptr(probs), probs.stride(0), probs.stride(1), probs.stride(2), _cabi.ptr(lens_dev),
                    n_frames, N, V, W, _cabi.ptr(lm.logps), _cabi.ptr(lm.logbs), _cabi.ptr(lm.child_start),
                    _cabi.ptr(lm.ids_wide), _cabi.ptr(lm.succ_start), _cabi.ptr(lm.succ_tok),
                    _cabi.ptr(lm.succ_node), lm.max_ngram, V + shift + 1, lm.sos, float(self.beta),
                    int(self.valid_mixture), _cabi.ptr(y), _cabi.ptr(y_lens), _cabi.ptr(nb), _cabi.ptr(b),
                    _cabi.ptr(ws), ws_bytes, _cabi.stream_ptr(device),
                )  # fmt: skip
                _cabi.check(rc, "pdt_ctc_lookup_lm_search")
        return y, y_lens, nb + b

    @torch.jit.unused
    def _searches_through_a_factor_table(self, logits: torch.Tensor) -> bool:
        """A bigram LookupLanguageModel whose (contexts, V) factor table stays in the Infinity Cache:
        the search of csrc/ctc_lm_table.hip (PDT_CTC_LM_TABLE=0: the other routes, for comparisons)."""
        lm = self.lm
        if not switches.get("PDT_CTC_LM_TABLE") or not self._fuses_lookup_lm(logits) or lm.max_ngram < 2:
            return False
        V = lm.vocab_size
        # (a row per context: U^(order - 1) of them -- 4 MB for a bigram model over 1000 tokens, 4 GB for a
        # trigram model: the card has 288)
        rows = (V + 1) ** (lm.max_ngram - 1)
        return V + 1 <= 80 * 64 and rows < (1 << 30) and rows * V * 4 <= _FACTOR_TABLE_MAX_BYTES and logits.dtype == torch.float

    @torch.jit.unused
    def _lm_table_search(
        self, logits: torch.Tensor, lens: Optional[torch.Tensor], n_frames: int
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """The whole search in one launch, the softmax of the logits included, the model's factor rows
        read from a table (include/pdt_amd.h: pdt_ctc_lm_table_search)."""
        lm, W = self.lm, self.width
        T, N, V = logits.size(0), logits.size(1), logits.size(2) - 1
        device = _cabi.require_hip(logits, lens, *_lm_buffers(lm))
        logits = _f32(logits)
        factors, fmax, sos_row, ctx_base = _factor_table(lm, self.beta, self.valid_mixture, device)
        L = _cabi.lib()
        with torch.cuda.device(device):
            y = torch.empty((n_frames, N, W), device=device, dtype=torch.long)
            y_lens = torch.empty((N, W), device=device, dtype=torch.long)
            y_probs = torch.empty((N, W), device=device, dtype=torch.float)
            if N:
                ws = torch.empty(int(L.pdt_ctc_lm_table_search_workspace_bytes(n_frames, N, V, W)), device=device,
                                 dtype=torch.uint8)
                lens_dev = None if lens is None else _i64(lens).contiguous()
                rc = L.pdt_ctc_lm_table_search(
                    _cabi.ptr(logits), n_frames, N, V, logits.stride(0), logits.stride(1), logits.stride(2),
                    _cabi.ptr(lens_dev), W, n_frames, _cabi.ptr(factors), _cabi.ptr(fmax), factors.size(0),
                    factors.stride(0), sos_row, ctx_base, factors.size(0) // ctx_base,
                    float(self.beta), int(self.valid_mixture), _cabi.ptr(y), _cabi.ptr(y_lens), _cabi.ptr(y_probs),
                    _cabi.ptr(ws), _cabi.stream_ptr(device),
                )  # fmt: skip
                _cabi.check(rc, "pdt_ctc_lm_table_search")
        return y, y_lens, y_probs

    def _frame_by_frame(
        self, logits: torch.Tensor, lens: Optional[torch.Tensor], state: Dict[str, torch.Tensor]
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """One ``ctc_prefix_search_advance`` kernel per frame, around the user's language model
        when there is one (the semantics of reference _decoding.py:1083-1202).  Utterances whose
        frames have run out are frozen on the device (a ``where`` per state tensor); the only
        host read is the number of frames to run."""
        T, N, V, W = logits.size(0), logits.size(1), logits.size(2) - 1, self.width
        device, dtype = logits.device, logits.dtype
        n_frames = T
        if lens is not None:
            if lens.dim() != 1:
                raise RuntimeError("lens must be 1 dimensional")
            if lens.size(0) != N:
                raise RuntimeError("expected dim 0 of lens to be {}, got {}".format(N, lens.size(0)))
            n_frames = min(T, int(lens.max().item())) if N else 0
        if not torch.jit.is_scripting():
            if self.lm is not None and self.beta != 0.0 and n_frames > 0 and self._searches_through_a_factor_table(logits):
                return self._lm_table_search(logits, lens, n_frames)
        probs = logits.softmax(2)
        # beam state: one empty prefix per utterance, all of its mass on "ends in blank"
        nb = torch.zeros((N, 1), device=device, dtype=dtype)
        b = torch.ones((N, 1), device=device, dtype=dtype)
        y = torch.empty((0, N, 1), dtype=torch.long, device=device)
        y_lens = torch.zeros((N, 1), dtype=torch.long, device=device)
        y_last = y_lens
        is_prefix = torch.ones((N, 1, 1), device=device, dtype=torch.bool)
        fuse = self.beta != 0.0
        one_kernel = False  # the n-gram model scored inside the step kernel (csrc/ctc_lm_step.hip)
        hist_pair: List[torch.Tensor] = []
        lens_dev: Optional[torch.Tensor] = None
        if self.lm is not None:
            if fuse:
                if not torch.jit.is_scripting():
                    one_kernel = self._fuses_lookup_lm(logits) and dtype == torch.float
                if one_kernel and n_frames > 0:
                    if self._searches_in_one_call():
                        return self._lookup_lm_search(probs, lens, n_frames)
                state = self.lm.update_input(state, y)
                if one_kernel:
                    # the history as 16-bit tokens between the frames (copying the (t, N, K) tensor is what a
                    # long search pays per frame), in two buffers of the final size used in turn
                    # ... each history token-contiguous -- (N, W, S) storage seen as (S, N, W) -- so that a
                    # column of the new beam is a plain 16-byte-at-a-time copy of its source's
                    y = y.to(torch.int16 if V <= 32767 else torch.long)
                    s_max = (n_frames + 7) // 8 * 8
                    hist_pair = [torch.empty((N, W, s_max), dtype=y.dtype, device=device).permute(2, 0, 1)
                                 for _ in range(2)]
                    lens_dev = None if lens is None else _i64(lens).contiguous()
        Kp = 1
        # row of batch element n's first prefix in the flattened (N * K') LM state, before and after
        # the beam has its full width
        first_rows = torch.arange(0, N, 1, device=device).unsqueeze(1)
        beam_rows = torch.arange(0, W * N, W, device=device).unsqueeze(1)
        for t in range(n_frames):
            nonext_t, blank_t = probs[t, :, :V], probs[t, :, V]
            ext_t = nonext_t.unsqueeze(1).expand(N, Kp, V)
            state_next: Dict[str, torch.Tensor] = dict()
            mix_in_step = False
            lm_lp = nonext_t
            if one_kernel:
                # (the model keeps no state between frames: nothing to extract or mix afterwards; the
                # history alternates between two buffers of the final size: no allocation per frame)
                y_new = hist_pair[t % 2][: t + 1]
                y_last, y_lens, nb, b, is_prefix = self._lookup_lm_frame(
                    nonext_t, blank_t, nb, b, y, y_last, y_lens, is_prefix, y_new, lens_dev, t
                )
                y, Kp = y_new, W
                continue
            if self.lm is not None:
                if fuse:
                    lm_lp, state_next = self.lm.calc_idx_log_probs(y.flatten(1), state, y_lens.flatten())
                    if not (torch.is_grad_enabled() and (lm_lp.requires_grad or probs.requires_grad)):
                        # one pass over the LM scores instead of four (csrc/fusion_ext.hip) -- scripted;
                        # else inside the step kernel itself, below
                        if torch.jit.is_scripting():
                            ext_t = torch.ops.pydrobert_amd.fusion_ext(
                                lm_lp.reshape(N * Kp, V), nonext_t, blank_t, self.beta, self.valid_mixture
                            )
                        elif switches.get("PDT_CTC_STEP_MIX"):
                            mix_in_step = True
                        else:
                            ext_t = _fusion_ext(
                                lm_lp.reshape(N * Kp, V), nonext_t, blank_t, self.beta, self.valid_mixture
                            )
                    elif self.valid_mixture:  # convex combination that still sums to 1 - blank (:1120-1128)
                        lm_p = lm_lp.softmax(-1).view(N, Kp, V) * (1 - blank_t.view(N, 1, 1))
                        ext_t = (1.0 - self.beta) * ext_t + self.beta * lm_p
                    else:  # shallow fusion: p_ctc * p_lm ** beta (:1130-1135)
                        ext_t = ext_t * (self.beta * lm_lp.log_softmax(-1)).exp().view(N, Kp, V)
            if torch.jit.is_scripting():
                y_new, last_new, lens_new, masses, is_prefix, src, kept = ctc_prefix_search_advance(
                    (ext_t, nonext_t, blank_t), W, (nb, b), y, y_last, y_lens, is_prefix
                )
                nb_new, b_new = masses
            else:
                if mix_in_step:
                    y_new, last_new, lens_new, nb_new, b_new, is_prefix, src, kept = _ctc_step_with_lm_scores(
                        lm_lp.reshape(N, Kp, V), self.beta, self.valid_mixture, nonext_t, blank_t, W, nb, b, y,
                        y_last, y_lens, is_prefix,
                    )  # fmt: skip
                else:
                    y_new, last_new, lens_new, masses, is_prefix, src, kept = ctc_prefix_search_advance(
                        (ext_t, nonext_t, blank_t), W, (nb, b), y, y_last, y_lens, is_prefix
                    )
                    nb_new, b_new = masses
            if self.lm is not None:
                if fuse:
                    rows = (src + (first_rows if Kp == 1 else beam_rows)).flatten()
                    state = self.lm.mix_by_mask(
                        self.lm.extract_by_src(state, rows), self.lm.extract_by_src(state_next, rows), kept.flatten()
                    )  # :1154-1163
            if lens is not None:
                live = (lens > t).unsqueeze(1)  # (N, 1): utterances that still have this frame
                if Kp < W:  # the first frame: widen the old state with absent entries
                    absent = nb.new_full((N, W - Kp), -float("inf"))
                    nb, b = torch.cat([nb, absent], 1), torch.cat([b, absent], 1)
                    y, y_lens = y.expand(-1, -1, W), y_lens.expand(-1, W)
                y_old = torch.cat([y, y.new_zeros((1, N, W))], 0)
                y_new = torch.where(live.unsqueeze(0), y_new, y_old)
                lens_new = torch.where(live, lens_new, y_lens)
                nb_new, b_new = torch.where(live, nb_new, nb), torch.where(live, b_new, b)
            y, y_last, y_lens, nb, b, Kp = y_new, last_new, lens_new, nb_new, b_new, W
        if y.dtype != torch.long or not y.is_contiguous():
            y = y.long().contiguous()
        total = nb + b
        if Kp < W:  # no frame at all: fill the beam with absent entries (:1190-1200)
            y, y_lens = y.repeat(1, 1, W), y_lens.repeat(1, W)
            total = torch.cat([total, total.new_full((N, W - Kp), -float("inf"))], 1)
        return y, y_lens, total


def _lm_buffers(lm: "LookupLanguageModel"):
    """Every trie buffer of an n-gram model the kernels read through raw pointers."""
    return (lm.logps, lm.logbs, lm.child_start, lm.ids_wide, lm.succ_start, lm.succ_tok, lm.succ_node)


def _identity_of(*tensors):
    """(address, version) of every tensor, or None when one carries no version counter (inference
    tensors): then there is nothing to recognise an unchanged tensor by and the caller rebuilds."""
    try:
        return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)
    except RuntimeError:
        return None


# dense tables of bigram LookupLanguageModels, per model object (BeamSearch._bigram_table)
_BIGRAM_TABLES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
# ... and the factor tables of the mixes they have been searched with (_factor_table)
_FACTOR_TABLES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_FACTOR_TABLE_MAX_BYTES = 6 << 30  # (a trigram model over 1000 tokens: 4 GB of this card's 288)


def _bigram_scores(lm: "LookupLanguageModel", device: torch.device):
    """``(scores (U, V) float32, sos_row)`` of a bigram model: row ``c`` = its scores after context
    token ``c`` (the start-of-sequence token's row last when it lies outside the vocabulary) -- one
    call of the model's own scoring kernel over every context."""
    scores, sos_row, _ = _context_scores(lm, device, 2)
    return scores, sos_row


def _context_scores(lm: "LookupLanguageModel", device: torch.device, order: Optional[int] = None):
    """``(scores (U^(order-1), V) float32, sos_row, U)``: the model's scores after EVERY context of
    ``order - 1`` symbols -- row ``r`` = the context whose symbols are the digits of ``r`` in base ``U``,
    oldest first; the symbols are the V tokens, then the start-of-sequence token when it lies outside the
    vocabulary (``U = V + 1``); ``sos_row`` = the all-sos context of the empty prefix (the reference pads
    short histories with sos, _lm.py:403-515).  One call of the model's own scoring kernel."""
    V = lm.vocab_size
    order = lm.max_ngram if order is None else order
    shift = 0 if (0 <= lm.sos < V) else 1
    U, n1 = V + shift, order - 1
    sos_sym = lm.sos if shift == 0 else V
    with torch.no_grad():
        rows = torch.arange(U**n1, device=device)
        digits = [(rows // (U ** (n1 - 1 - i))) % U for i in range(n1)]
        hist = torch.stack(digits)  # (order - 1, rows), oldest symbol first
        if shift:
            hist = torch.where(hist == V, torch.full_like(hist, lm.sos), hist)
        table, _ = lm.calc_idx_log_probs(hist, dict(), torch.tensor(n1, device=device))
    sos_row = sum(sos_sym * U**i for i in range(n1))
    return _f32(table).contiguous(), sos_row, U


def _factor_table(lm: "LookupLanguageModel", beta: float, valid_mixture: bool, device: torch.device):
    """``(factors (rows, V), row maxima (rows,), sos_row, U)``: the model's factor of CTCPrefixSearch's mix for
    every context (include/pdt_amd.h: pdt_lm_factor_table), built once per model, mix and device and kept
    while every trie buffer of the model is the same tensor at the same version (see
    BeamSearch._bigram_table for what that does not see).  The factors overwrite the scores they are
    formed from (a trigram model's table is gigabytes)."""
    ident = _identity_of(*_lm_buffers(lm))
    key = None if ident is None else (ident, str(device), float(beta), bool(valid_mixture))
    ent = _FACTOR_TABLES.get(lm)
    if key is not None and ent is not None and ent[0] == key:
        return ent[1], ent[2], ent[3], ent[4]
    factors, sos_row, U = _context_scores(lm, device)
    rows, V = factors.shape
    with torch.cuda.device(device):
        rc = _cabi.lib().pdt_lm_factor_table(
            _cabi.ptr(factors), rows, V, float(beta), int(valid_mixture), _cabi.ptr(factors), factors.stride(0),
            _cabi.stream_ptr(device),
        )
    _cabi.check(rc, "pdt_lm_factor_table")
    fmax = factors.max(1)[0].contiguous()
    if key is not None:
        _FACTOR_TABLES[lm] = (key, factors, fmax, sos_row, U)
    return factors, fmax, sos_row, U


class BeamSearch(torch.nn.Module):
    """Beam search driven by an :class:`ExtractableSequentialLanguageModel` (reference
    _decoding.py:158-504).  Each iteration is the user's LM forward, a ``log_softmax``, the
    overridable :meth:`update_log_probs_for_step` hook and ONE ``beam_search_advance`` kernel.
    """

    __constants__ = ["width", "eos", "finish_all_paths", "pad_value"]
    # iterations between host reads of the termination count in the fused loop (None: 8 for this
    # package's LookupLanguageModel, 1 -- the reference's behaviour -- for any other model)
    host_check_interval: Optional[int] = None

    def __init__(
        self,
        lm: ExtractableSequentialLanguageModel,
        width: int,
        eos: Optional[int] = None,
        finish_all_paths: bool = False,
        pad_value: int = config.INDEX_PAD_VALUE,
    ):
        width = argcheck.is_posi(width, "width")
        eos = argcheck.is_int(eos, "eos", True)
        finish_all_paths = argcheck.is_bool(finish_all_paths, "finish_all_paths")
        pad_value = argcheck.is_int(pad_value, "pad_value")
        super().__init__()
        if eos is not None:
            if eos < -lm.vocab_size or eos > lm.vocab_size - 1:
                raise ValueError(
                    "Expected eos to be in the range [{}, {}], got {}".format(
                        -lm.vocab_size, lm.vocab_size - 1, eos
                    )
                )
            eos = (eos + lm.vocab_size) % lm.vocab_size
        self.lm, self.width, self.eos = lm, width, eos
        self.finish_all_paths, self.pad_value = finish_all_paths, pad_value
        try:
            device = next(iter(lm.parameters())).device
        except StopIteration:
            device = torch.device("cpu")
        self.register_buffer("device_buffer", torch.empty(0, device=device))

    def reset_parameters(self) -> None:
        if hasattr(self.lm, "reset_parameters"):
            self.lm.reset_parameters()

    def extra_repr(self) -> str:
        return ", ".join("{}={}".format(x, getattr(self, x)) for x in self.__constants__)

    def update_log_probs_for_step(
        self,
        log_probs_prev: torch.Tensor,
        log_probs_t: torch.Tensor,
        y_prev: torch.Tensor,
        y_prev_lens: torch.Tensor,
        eos_mask: torch.Tensor,
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Hook: subclasses may rescore paths ``(N, K)`` and extensions ``(N, K, V)`` at every
        step (reference _decoding.py:306-350).  The default is the identity."""
        return log_probs_prev, log_probs_t

    def _to_width(self, y, log_probs, lens):
        # reference _decoding.py:352-372
        S, N, Kp = y.shape
        if Kp < self.width:
            rem = self.width - Kp
            log_probs = torch.cat([log_probs, log_probs.new_full((N, rem), -float("inf"))], 1)
            y = torch.cat([y, y.new_zeros(S, N, rem)], 2)
            lens = torch.cat([lens, lens.new_zeros(N, rem)], 1)
        elif Kp > self.width:
            log_probs, src = log_probs.topk(self.width, 1)
            y = y.gather(2, src.unsqueeze(0).expand(S, N, self.width))
            lens = lens.gather(1, src)
        return y, log_probs, lens

    @torch.jit.unused
    def _check_growth(self, lens: torch.Tensor, hist: torch.Tensor) -> None:
        if switches.get("PDT_CHECK_INVARIANTS") == 1 and lens.numel():
            if int(lens.max()) < hist.size(0):
                raise RuntimeError("BeamSearch: no path is as long as the history ({} < {}): the step must not "
                                   "grow y".format(int(lens.max()), hist.size(0)))

    @torch.jit.unused
    def _bigram_table(self, device: torch.device):
        """``(table (U, V), stats (U, 2), sos_row)`` for a bigram :class:`LookupLanguageModel` whose dense
        table stays below 64 MiB, else ``None``: row ``c`` holds the model's scores after context token
        ``c`` (one call of its own scoring kernel over every context, made once per model and device and
        kept while the model's buffers are unchanged), ``stats`` every row's maximum and log-sum-exp.
        An iteration of the search then reads its prefixes' rows straight from the table
        (``pdt_beam_search_step_table``) instead of having the model write ``(N K, V)`` scores first."""
        lm = self.lm
        if type(lm) is not LookupLanguageModel or lm.max_ngram != 2 or not switches.get("PDT_BEAM_TABLE"):
            return None
        V = lm.vocab_size
        shift = 0 if (0 <= lm.sos < V) else 1
        U = V + shift
        if U * V * 4 > (64 << 20) or lm.logps.device != device:
            return None
        # (kept while EVERY trie buffer is the same tensor at the same version; models whose buffers carry
        # no version counter -- built under inference_mode -- get a fresh table per search.  Writes the
        # counter does not see (`.data`, raw pointers) are the caller's to announce: `del
        # _BIGRAM_TABLES[lm]` or PDT_BEAM_TABLE=0)
        ident = _identity_of(*_lm_buffers(lm))
        key = None if ident is None else (ident, str(device))
        ent = _BIGRAM_TABLES.get(lm)
        if key is not None and ent is not None and ent[0] == key:
            return ent[1], ent[2], ent[3]
        table, _ = _bigram_scores(lm, device)
        with torch.no_grad():
            stats = torch.empty((U, 2), device=device, dtype=torch.float)
            with torch.cuda.device(device):
                rc = _cabi.lib().pdt_row_log_softmax_stats(
                    _cabi.ptr(table), table.stride(0), table.stride(1), U, V, _cabi.ptr(stats),
                    _cabi.stream_ptr(device),
                )
            _cabi.check(rc, "pdt_row_log_softmax_stats")
        sos_row = lm.sos if shift == 0 else V
        if key is not None:
            _BIGRAM_TABLES[lm] = (key, table, stats, sos_row)
        return table, stats, sos_row

    @torch.jit.unused
    def _table_search(self, dense, N: int, max_iters: int, squeeze: bool):
        """``pdt_beam_search_table`` + ``pdt_beam_search_table_paths``: the whole search in one launch, the
        paths written once at the end (one host read in between: the number of rows ``y`` has).  ``None``
        when the kernel does not take the shape."""
        table, stats, sos_row = dense
        device, W, V = table.device, self.width, self.lm.vocab_size
        L = _cabi.lib()
        with _cabi.on_device(device):
            trie = torch.empty((N, max_iters, W), dtype=torch.int32, device=device)
            lp = torch.empty((N, W), device=device)
            lens = torch.empty((N, W), dtype=torch.long, device=device)
            finish = torch.empty((N,), dtype=torch.int32, device=device)
            t_stop = torch.zeros((1,), dtype=torch.int32, device=device)
            stream = _cabi.stream_ptr(device)
            rc = L.pdt_beam_search_table(
                _cabi.ptr(table), table.stride(0), table.stride(1), table.size(0), _cabi.ptr(stats), int(sos_row),
                N, V, W, max_iters, int(self.eos is not None), int(self.eos or 0), int(self.finish_all_paths),
                _cabi.ptr(trie), _cabi.ptr(lp), _cabi.ptr(lens), _cabi.ptr(finish), _cabi.ptr(t_stop), stream,
            )  # fmt: skip
            if rc == _cabi.PDT_E_UNSUPPORTED:
                return None
            _cabi.check(rc, "pdt_beam_search_table")
            T = int(t_stop.item())  # the reference leaves its loop at the iteration that finds every element finished
            y = torch.empty((T, N, W), dtype=torch.long, device=device)
            rc = L.pdt_beam_search_table_paths(
                _cabi.ptr(trie), _cabi.ptr(finish), N, max_iters, W, T, int(self.pad_value), _cabi.ptr(y), stream
            )
            _cabi.check(rc, "pdt_beam_search_table_paths")
        if squeeze:
            y, lens, lp = y.squeeze(1), lens.squeeze(0), lp.squeeze(0)
        return y, lens, lp

    @torch.jit.unused
    def _forward_fused(
        self, prev: Dict[str, torch.Tensor], batch_size: Optional[int], max_iters: Optional[int]
    ) -> Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """The search with every iteration's bookkeeping in ONE kernel (csrc/beam_step.hip) around the
        language model's forward: no ``log_softmax`` / ``masked_fill`` / ``where`` passes over
        ``(N, K, V)``, no per-iteration clamp of the history, and the number of unfinished batch
        elements is read from the device every eighth iteration instead of every one (iterations run
        past the end only append padding, which is cut off again).  Taken when the step hook is the
        default one (a subclass that overrides ``update_log_probs_for_step`` must see the tensors),
        nothing wants gradients and the beam fits a wave; returns ``None`` otherwise."""
        if type(self).update_log_probs_for_step is not BeamSearch.update_log_probs_for_step:
            return None
        if self.width > 64 or not switches.get("PDT_BEAM_FUSED"):
            return None
        if torch.is_grad_enabled() and (
            any(p.requires_grad for p in self.lm.parameters())
            or any(torch.is_tensor(v) and v.requires_grad for v in prev.values())
        ):
            return None
        device = self.device_buffer.device
        if device.type != "cuda":
            return None
        N = 1 if batch_size is None else batch_size
        V, W = self.lm.vocab_size, self.width
        if max_iters is None:
            if self.eos is None:
                raise RuntimeError("max_iters must be set when eos is unset")
            max_iters = 1073741824
        elif max_iters < 0:
            raise RuntimeError("max_iters must be non-negative, got {}".format(max_iters))
        L = _cabi.lib()
        has_eos = self.eos is not None
        y = torch.empty((0, N), dtype=torch.long, device=device)
        prev = self.lm.update_input(prev, y)
        y = y.unsqueeze(2)
        log_probs = torch.zeros((N, 1), device=device)
        lens = torch.zeros((N, 1), dtype=torch.long, device=device)
        # Iterations between host reads of the "everything finished" count.  Between reads the search may
        # call the language model up to `check_every - 1` times past the reference's stopping point (their
        # output is cut off again) -- invisible for a stateless, deterministic model like this package's
        # LookupLanguageModel, but a stochastic or stateful user model would see extra calls: those are
        # checked every iteration unless the caller says otherwise through `host_check_interval`.  Even
        # then the count of iteration i is read AFTER the model's call of iteration i has been issued, so
        # such a model sees AT MOST ONE call more than under the reference, which breaks before calling
        # it (_decoding.py:426); the call's output is discarded.
        check_every = self.host_check_interval
        if check_every is None:
            check_every = 8 if type(self.lm) is LookupLanguageModel else 1
        check_every = max(1, min(int(check_every), 1024))
        counts = torch.zeros((check_every,), dtype=torch.int32, device=device)
        pad_from = torch.full((N,), 2147483647, dtype=torch.int32, device=device)
        steps = torch.arange(0, 1024, device=device)
        row_base = {1: torch.zeros((N, 1), dtype=torch.long, device=device),
                    W: torch.arange(0, W * N, W, device=device).unsqueeze(1)}
        Kp, t, t_stop = 1, 0, -1
        out_dtype = torch.float
        dense = self._bigram_table(device)
        if (dense is not None and 0 < max_iters <= 4096 and N > 0 and 4 * N * max_iters * W <= (1 << 30)
                and switches.get("PDT_BEAM_SEARCH")):  # (the trie: a word per beam entry and iteration, 1 GiB at most)
            # a bigram table model, a bounded search: every iteration in ONE launch, no history copies
            done = self._table_search(dense, N, max_iters, batch_size is None)
            if done is not None:
                return done
        first_rows = None if dense is None else torch.full((N, 1), dense[2], dtype=torch.long, device=device)
        if dense is not None:  # (what does not change from one iteration to the next, once)
            table, stats, _ = dense
            table_args = (_cabi.ptr(table), table.stride(0), table.stride(1), table.size(0), _cabi.ptr(stats))
            eos_args = (int(has_eos), int(self.eos or 0), int(self.finish_all_paths), int(self.pad_value))
            counts_ptr, pad_from_ptr, stream = counts.data_ptr(), _cabi.ptr(pad_from), _cabi.stream_ptr(device)
            step_table = L.pdt_beam_search_step_table
        while t < max_iters:
            if dense is not None:
                # a bigram table model: the prefixes' rows of its dense table, by their last tokens
                rows = first_rows if t == 0 else y[t - 1]
                with _cabi.on_device(device):
                    y_new = torch.empty((t + 1, N, W), dtype=torch.long, device=device)
                    lens_new = torch.empty((N, W), dtype=torch.long, device=device)
                    lp_new = torch.empty((N, W), device=device)
                    src = torch.empty((N, W), dtype=torch.long, device=device)
                    rc = step_table(
                        *table_args, rows.data_ptr(), N, Kp, V, W,
                        log_probs.data_ptr(), log_probs.stride(0), log_probs.stride(1),
                        y.data_ptr(), t, y.stride(0), y.stride(1), y.stride(2),
                        lens.data_ptr(), lens.stride(0), lens.stride(1), *eos_args,
                        y_new.data_ptr(), lens_new.data_ptr(), lp_new.data_ptr(), src.data_ptr(),
                        counts_ptr + 4 * (t % check_every), pad_from_ptr, stream,
                    )  # fmt: skip
                if rc:
                    _cabi.check(rc, "pdt_beam_search_step_table")
                y, lens, log_probs, Kp = y_new, lens_new, lp_new, W
                t += 1
                if has_eos and (t % check_every == 0 or t == max_iters):
                    seen = counts.tolist()
                    lo = t - ((t - 1) % check_every + 1)
                    for i in range(lo, t):
                        if i > 0 and seen[i % check_every] == 0:
                            t_stop = i
                            break
                    if t_stop >= 0:
                        break
                    counts.zero_()
                continue
            if t and t % 1024 == 0:
                steps = torch.arange(t, t + 1024, device=device)
            scores, state_next = self.lm.calc_idx_log_probs(y.flatten(1), prev, steps[t % 1024])
            if torch.is_grad_enabled() and scores.requires_grad:
                if t == 0:  # (a model whose weights are no registered parameters: the differentiable loop)
                    return None
                raise RuntimeError("BeamSearch: the language model's output wants gradients inside a search "
                                   "that was started without any (set PDT_BEAM_FUSED=0)")
            out_dtype = scores.dtype
            scores = _f32(scores).reshape(N, Kp, V)
            with torch.cuda.device(device):
                y_new = torch.empty((t + 1, N, W), dtype=torch.long, device=device)
                lens_new = torch.empty((N, W), dtype=torch.long, device=device)
                lp_new = torch.empty((N, W), device=device)
                src = torch.empty((N, W), dtype=torch.long, device=device)
                rc = L.pdt_beam_search_step(
                    _cabi.ptr(scores), scores.stride(0), scores.stride(1), scores.stride(2), N, Kp, V, W,
                    _cabi.ptr(log_probs), log_probs.stride(0), log_probs.stride(1),
                    _cabi.ptr(y), t, y.stride(0), y.stride(1), y.stride(2),
                    _cabi.ptr(lens), lens.stride(0), lens.stride(1), int(has_eos), int(self.eos or 0),
                    int(self.finish_all_paths), int(self.pad_value), _cabi.ptr(y_new), _cabi.ptr(lens_new),
                    _cabi.ptr(lp_new), _cabi.ptr(src), counts.data_ptr() + 4 * (t % check_every),
                    _cabi.ptr(pad_from), _cabi.stream_ptr(device),
                )  # fmt: skip
            _cabi.check(rc, "pdt_beam_search_step")
            prev = self.lm.extract_by_src(state_next, (src + row_base[Kp]).flatten())
            y, lens, log_probs, Kp = y_new, lens_new, lp_new, W
            t += 1
            if has_eos and (t % check_every == 0 or t == max_iters):
                # the one host read per `check_every` iterations: an iteration that started with every
                # batch element finished is where the reference leaves its loop (:426)
                seen = counts.tolist()
                lo = t - ((t - 1) % check_every + 1)
                for i in range(lo, t):
                    if i > 0 and seen[i % check_every] == 0:
                        t_stop = i
                        break
                if t_stop >= 0:
                    break
                counts.zero_()
        if t_stop >= 0:  # the iterations from t_stop on only appended padding
            y = y[:t_stop]
        if has_eos and y.size(0):  # finished elements: pad_value from their first padding row on
            rows = torch.arange(y.size(0), device=device).view(-1, 1, 1)
            y = torch.where(rows >= pad_from.view(1, N, 1), y.new_full((), self.pad_value), y)
        y, log_probs, lens = self._to_width(y, log_probs.to(out_dtype), lens)
        if batch_size is None:
            y, lens, log_probs = y.squeeze(1), lens.squeeze(0), log_probs.squeeze(0)
        return y, lens, log_probs

    def forward(
        self,
        initial_state_: Optional[Dict[str, torch.Tensor]] = None,
        batch_size: Optional[int] = None,
        max_iters: Optional[int] = None,
        initial_state: Optional[Dict[str, torch.Tensor]] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        # (``initial_state_`` is the reference's runtime keyword, _decoding.py:383-386;
        # ``initial_state`` the name of its documented call signature -- both are accepted)
        if initial_state is None:
            initial_state = initial_state_
        prev = dict() if initial_state is None else initial_state
        device = self.device_buffer.device
        N = 1 if batch_size is None else batch_size
        V, W = self.lm.vocab_size, self.width
        if not torch.jit.is_scripting():
            fused = self._forward_fused(prev, batch_size, max_iters)
            if fused is not None:
                return fused
        Kp = 1
        y = torch.empty((0, N), dtype=torch.long, device=device)
        prev = self.lm.update_input(prev, y)
        y = y.unsqueeze(2)
        log_probs = torch.full((N, Kp), -math.log(Kp), device=device)
        lens = torch.zeros((N, Kp), dtype=torch.long, device=device)
        if max_iters is None:
            if self.eos is None:
                raise RuntimeError("max_iters must be set when eos is unset")
            max_iters = 1073741824
        elif max_iters < 0:
            raise RuntimeError("max_iters must be non-negative, got {}".format(max_iters))
        pad_row = torch.full((1, N, W), self.pad_value, device=device, dtype=torch.long)
        track_eos = self.eos is not None
        for t in range(max_iters):
            step = torch.tensor(t, device=device)
            ended = torch.zeros((N, Kp), device=device, dtype=torch.bool)
            frozen = ended[:, :1]  # batch elements whose search is over (:413-427)
            any_frozen = False
            if track_eos and t:
                tail = y.permute(1, 2, 0).gather(2, (lens - 1).clamp(min=0).unsqueeze(2)).squeeze(2)
                ended = (tail == self.eos) & (lens > 0)
                frozen = ended.all(1, keepdim=True) if self.finish_all_paths else ended[:, :1]
                # the one host read of the step: (everything is over, something is over)
                code = int((frozen.all().long() * 2 + frozen.any().long()).item())
                any_frozen = code > 0
                if code > 1:
                    break
            hist = y.clamp(0, V - 1)
            lp_t, state_next = self.lm.calc_idx_log_probs(hist.flatten(1), prev, step)
            lp_t = lp_t.reshape(N, Kp, V).log_softmax(-1)
            log_probs, lp_t = self.update_log_probs_for_step(log_probs, lp_t, hist, lens, ended)
            if track_eos:  # a path that has ended repeats eos at no cost and emits nothing else (:448-458)
                only_eos = torch.full_like(lp_t, -float("inf"))
                only_eos[..., self.eos] = 0.0
                lp_t = torch.where(ended.unsqueeze(2), only_eos, lp_t)
            # some path is as long as the history whenever the loop gets here (a live element has
            # a live path of t tokens), so y grows by a row: no read-back of the lengths
            # (PDT_CHECK_INVARIANTS=1 reads them back and checks)
            if not torch.jit.is_scripting():
                self._check_growth(lens, hist)
            y_new, lens_new, lp_new, src = torch.ops.pydrobert_amd.beam_search_advance(
                lp_t, W, log_probs, hist, lens, True
            )
            if track_eos:  # ended sources stay as long as they were (:465-468)
                lens_new = lens_new - ended.gather(1, src).to(lens_new)
            rows = (src + torch.arange(0, Kp * N, Kp, device=device).unsqueeze(1)).flatten()
            prev = self.lm.extract_by_src(state_next, rows)
            if any_frozen:  # finished batch elements keep what they had (:479-486)
                y, log_probs, lens = self._to_width(y, log_probs, lens)
                grown = torch.cat([y, pad_row.expand(y_new.size(0) - y.size(0), -1, -1)], 0)
                y_new = torch.where(frozen.unsqueeze(0), grown, y_new)
                lp_new = torch.where(frozen, log_probs, lp_new)
                lens_new = torch.where(frozen, lens, lens_new)
            y, lens, log_probs, Kp = y_new, lens_new, lp_new, W
        y, log_probs, lens = self._to_width(y, log_probs, lens)
        if batch_size is None:
            y, lens, log_probs = y.squeeze(1), lens.squeeze(0), log_probs.squeeze(0)
        return y, lens, log_probs


# ---------------------------------------------------------------------------------------
# greedy CTC search, sequence log-probabilities, random walk
# ---------------------------------------------------------------------------------------
@custom_op("pydrobert_amd::ctc_greedy_search", mutates_args=())
def _ctc_greedy_search_op(
    logits: torch.Tensor,
    in_lens: Optional[torch.Tensor],
    blank_idx: int,
    batch_first: bool,
    is_probs: bool,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    if logits.dim() != 3:
        raise RuntimeError("logits must be 3-dimensional")
    V = logits.size(2)
    if blank_idx < -V or blank_idx > (V - 1):
        raise RuntimeError(
            "Blank index out of range (expected to be in the range of [-{},{}], but got {})".format(
                V, V - 1, blank_idx
            )
        )
    blank_idx = (blank_idx + V) % V
    device = _cabi.require_hip(logits, in_lens)
    x = _f32(logits)
    if batch_first:
        N, T = x.shape[:2]
        st, sn = x.stride(1), x.stride(0)
    else:
        T, N = x.shape[:2]
        st, sn = x.stride(0), x.stride(1)
    lens = None if in_lens is None else _i64(in_lens).contiguous()
    with torch.cuda.device(device):
        max_ = torch.empty((N,), device=device, dtype=torch.float)
        paths = torch.empty((N, T) if batch_first else (T, N), device=device, dtype=torch.long)
        out_lens = torch.empty((N,), device=device, dtype=torch.long)
        pst, psn = (paths.stride(1), paths.stride(0)) if batch_first else (paths.stride(0), paths.stride(1))
        rc = _cabi.lib().pdt_ctc_greedy_search(
            _cabi.ptr(x), T, N, V, st, sn, x.stride(2), _cabi.ptr(lens), blank_idx, int(is_probs),
            _cabi.ptr(max_), _cabi.ptr(paths), pst, psn, _cabi.ptr(out_lens), _cabi.stream_ptr(device),
        )  # fmt: skip
    _cabi.check(rc, "pdt_ctc_greedy_search")
    return max_.to(logits.dtype), paths, out_lens


@_ctc_greedy_search_op.register_fake
def _(logits, in_lens, blank_idx, batch_first, is_probs):
    N = logits.shape[0] if batch_first else logits.shape[1]
    return (
        logits.new_empty((N,)),
        logits.new_empty(logits.shape[:2], dtype=torch.long),
        logits.new_empty((N,), dtype=torch.long),
    )


def _ctc_greedy_setup(ctx, inputs, output):
    logits, in_lens, _, batch_first, is_probs = inputs
    ctx.save_for_backward(logits, in_lens)
    ctx.cfg = (batch_first, is_probs)


def _ctc_greedy_backward(ctx, g_max, g_paths, g_lens):
    """``max_`` is the sum (product) over the valid frames of the best class's log-probability
    (probability): differentiable in the reference (_decoding.py:526-553).  The frames' maxima
    are recomputed with device ops and differentiated by autograd."""
    logits, in_lens = ctx.saved_tensors
    batch_first, is_probs = ctx.cfg
    with torch.enable_grad():
        x = logits.detach().requires_grad_(True)
        y = x if is_probs else x.log_softmax(2)
        if not batch_first:
            y = y.transpose(0, 1)
        best = y.max(2)[0]  # (N, T)
        if in_lens is not None:
            valid = torch.arange(best.size(1), device=best.device).unsqueeze(0) < in_lens.unsqueeze(1)
            best = best.masked_fill(~valid, 1.0 if is_probs else 0.0)
        total = best.prod(1) if is_probs else best.sum(1)
        (g,) = torch.autograd.grad(total, x, g_max.to(total.dtype))
    return g, None, None, None, None


register_autograd("pydrobert_amd::ctc_greedy_search", _ctc_greedy_backward, setup_context=_ctc_greedy_setup)


def ctc_greedy_search(
    logits: torch.Tensor,
    in_lens: Optional[torch.Tensor] = None,
    blank_idx: int = -1,
    batch_first: bool = False,
    is_probs: bool = False,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Functional version of :class:`CTCGreedySearch` (reference _decoding.py:507-558):
    returns ``(max_, paths, out_lens)``.  One pass over the logits."""
    return torch.ops.pydrobert_amd.ctc_greedy_search(logits, in_lens, blank_idx, batch_first, is_probs)


def _slp_dims(hyp: torch.Tensor, dim: int) -> Tuple[int, int, int, int]:
    hyp_dim = hyp.dim()
    if dim < -hyp_dim or dim > hyp_dim - 1:
        raise RuntimeError(
            "Dimension out of range (expected to be in range of [{}, {}], but got {})".format(
                -hyp_dim, hyp_dim - 1, dim
            )
        )
    dim = (hyp_dim + dim) % hyp_dim
    shape = tuple(hyp.shape)
    return dim, int(math.prod(shape[:dim])), shape[dim], int(math.prod(shape[dim + 1 :]))


@custom_op("pydrobert_amd::sequence_log_probs", mutates_args=())
def _sequence_log_probs_op(
    logits: torch.Tensor, hyp: torch.Tensor, dim: int, eos: Optional[int]
) -> torch.Tensor:
    """Fused log-softmax + gather + masked sum over ``dim`` (csrc/seq_ops.hip)."""
    dim, A, S, B = _slp_dims(hyp, dim)
    if logits.shape[:-1] != hyp.shape:
        raise RuntimeError("logits must have shape hyp.shape + (num_classes,)")
    device = _cabi.require_hip(logits, hyp)
    x = _f32(logits).contiguous()
    h = _i64(hyp).contiguous()
    with torch.cuda.device(device):
        out = torch.empty((A, B), device=device, dtype=torch.float)
        rc = _cabi.lib().pdt_sequence_log_probs_forward(
            _cabi.ptr(x), _cabi.ptr(h), A, S, B, x.shape[-1], int(eos is not None),
            int(eos) if eos is not None else 0, _cabi.ptr(out), _cabi.stream_ptr(device),
        )  # fmt: skip
    _cabi.check(rc, "pdt_sequence_log_probs_forward")
    shape = tuple(hyp.shape)
    return out.view(shape[:dim] + shape[dim + 1 :]).to(logits.dtype)


@_sequence_log_probs_op.register_fake
def _(logits, hyp, dim, eos):
    d = dim % hyp.dim()
    return logits.new_empty(tuple(hyp.shape[:d]) + tuple(hyp.shape[d + 1 :]))


@custom_op("pydrobert_amd::sequence_log_probs_backward", mutates_args=())
def _sequence_log_probs_backward_op(
    logits: torch.Tensor, hyp: torch.Tensor, dim: int, eos: Optional[int], grad_out: torch.Tensor
) -> torch.Tensor:
    dim, A, S, B = _slp_dims(hyp, dim)
    device = logits.device
    x = _f32(logits).contiguous()
    h = _i64(hyp).contiguous()
    g = grad_out.detach().float().contiguous()
    with torch.cuda.device(device):
        grad = torch.empty_like(x)
        rc = _cabi.lib().pdt_sequence_log_probs_backward(
            _cabi.ptr(x), _cabi.ptr(h), A, S, B, x.shape[-1], int(eos is not None),
            int(eos) if eos is not None else 0, _cabi.ptr(g), _cabi.ptr(grad),
            _cabi.stream_ptr(device),
        )  # fmt: skip
    _cabi.check(rc, "pdt_sequence_log_probs_backward")
    return grad.view(logits.shape).to(logits.dtype)


@_sequence_log_probs_backward_op.register_fake
def _(logits, hyp, dim, eos, grad_out):
    return torch.empty_like(logits)


def _slp_setup_context(ctx, inputs, output):
    logits, hyp, dim, eos = inputs
    ctx.save_for_backward(logits, hyp)
    ctx.cfg = (dim, eos)


def _slp_backward(ctx, grad_out):
    logits, hyp = ctx.saved_tensors
    dim, eos = ctx.cfg
    grad = torch.ops.pydrobert_amd.sequence_log_probs_backward(logits, hyp, dim, eos, grad_out)
    return grad, None, None, None


register_autograd(
    "pydrobert_amd::sequence_log_probs", _slp_backward, setup_context=_slp_setup_context
)


def _sequence_log_probs_ps(
    logits: Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]],
    hyp: torch.Tensor,
    dim: int,
) -> torch.Tensor:
    # padded view + out-of-range tokens beyond each length: same kernel, same masking rule
    if dim < -2 or dim > 1:
        raise RuntimeError(
            "Dimension out of range (expected to be in range of [-2, 1], but got {})".format(dim)
        )
    data, batch_sizes, unsorted = logits[0], logits[1], logits[3]
    S = batch_sizes.size(0)
    padded, lens = torch._pad_packed_sequence(data, batch_sizes, False, 0.0, S)  # (S, N, V)
    if unsorted is not None:
        padded, lens = padded.index_select(1, unsorted), lens.index_select(0, unsorted.cpu())
    h = hyp if dim % 2 == 0 else hyp.t()
    h = h[:S]
    beyond = torch.arange(S, device=h.device).unsqueeze(1) >= lens.to(h.device).unsqueeze(0)
    return torch.ops.pydrobert_amd.sequence_log_probs(padded, h.masked_fill(beyond, -1), 0, None)


def sequence_log_probs(
    logits: Any, hyp: torch.Tensor, dim: int = 0, eos: Optional[int] = None
) -> torch.Tensor:
    """Functional version of :class:`SequenceLogProbabilities` (reference
    _decoding.py:1516-1633): joint log-probability of the token sequences ``hyp`` under
    ``logits`` (a tensor of shape ``hyp.shape + (V,)`` or a ``PackedSequence``).  Fused
    log-softmax + gather + masked sum; differentiable w.r.t. ``logits``."""
    if isinstance(logits, torch.Tensor):
        return torch.ops.pydrobert_amd.sequence_log_probs(logits, hyp, dim, eos)
    elif torch.jit.isinstance(
        logits, Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]
    ):
        return _sequence_log_probs_ps(logits, hyp, dim)
    raise RuntimeError("logits must be either a Tensor or PackedSequence")


def random_walk_advance(
    log_probs_t: torch.Tensor,
    log_probs_prev: torch.Tensor,
    y_prev: torch.Tensor,
    y_prev_lens: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Random walk step function (reference _decoding.py:1207-1283).  Sampling uses torch's
    generator on the tensors' device, as in the reference; a handful of small launches."""
    if log_probs_t.dim() != 2:
        raise RuntimeError("log_probs_t must be 2-dimensional")
    N = log_probs_t.size(0)
    if log_probs_prev.dim() != 1 or log_probs_prev.size(0) != N:
        raise RuntimeError(
            "Expected log_probs_prev to be of shape ({},), got {}".format(N, log_probs_prev.shape)
        )
    if y_prev.dim() != 2:
        raise RuntimeError("y_prev must be 2-dimensional")
    if y_prev.size(1) != N:
        raise RuntimeError("Expected dim 1 of y_prev to be {}, got {}".format(N, y_prev.size(-1)))
    S = y_prev.size(0)
    if y_prev_lens is not None and (y_prev_lens.dim() != 1 or y_prev_lens.size(0) != N):
        raise RuntimeError(
            "Expected y_prev_lens to have shape ({},), got {}".format(N, y_prev_lens.shape)
        )
    y_t = torch.multinomial(log_probs_t.exp(), 1, True)  # (N, 1)
    log_probs_next = log_probs_prev + log_probs_t.gather(1, y_t).squeeze(1)
    y_t = y_t.T
    if S:
        if y_prev_lens is None:
            y_next = torch.cat([y_prev, y_t], 0)
        else:
            y_next = torch.cat([y_prev, y_t], 0) if int(y_prev_lens.max().item()) >= S else y_prev
            y_next = y_next.scatter(0, y_prev_lens.unsqueeze(0), y_t)
    else:
        y_next = y_t
    return y_next, log_probs_next


class CTCGreedySearch(torch.nn.Module):
    """CTC greedy search (reference _decoding.py:561-635)."""

    __constants__ = "blank_idx", "batch_first", "is_probs"

    def __init__(self, blank_idx: int = -1, batch_first: bool = False, is_probs: bool = False):
        blank_idx = argcheck.is_int(blank_idx, "blank_idx")
        batch_first = argcheck.is_bool(batch_first, "batch_first")
        is_probs = argcheck.is_bool(is_probs, "is_probs")
        super().__init__()
        self.blank_idx, self.batch_first, self.is_probs = blank_idx, batch_first, is_probs

    def extra_repr(self) -> str:
        return ", ".join("{}={}".format(x, getattr(self, x)) for x in self.__constants__)

    def forward(
        self, logits: torch.Tensor, in_lens: Optional[torch.Tensor] = None
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return ctc_greedy_search(logits, in_lens, self.blank_idx, self.batch_first, self.is_probs)


class SequenceLogProbabilities(torch.nn.Module):
    """Calculate joint log probability of sequences (reference _decoding.py:1636-1720)."""

    __constants__ = "dim", "eos"

    def __init__(self, dim: int = 0, eos: Optional[int] = None):
        dim = argcheck.is_int(dim, "dim")
        eos = argcheck.is_int(eos, "eos", True)
        super().__init__()
        self.dim, self.eos = dim, eos

    def extra_repr(self) -> str:
        s = "dim={}".format(self.dim)
        if self.eos is not None:
            s += ", eos={}".format(self.eos)
        return s

    def forward(self, logits: Any, hyp: torch.Tensor) -> torch.Tensor:
        return sequence_log_probs(logits, hyp, self.dim, self.eos)


class RandomWalk(torch.nn.Module):
    """Perform a random walk on the outputs of a language model (reference
    _decoding.py:1286-1513)."""

    __constants__ = ["eos"]

    def __init__(self, lm, eos: Optional[int] = None):
        eos = argcheck.is_int(eos, "eos", True)
        super().__init__()
        if eos is not None:
            if eos < -lm.vocab_size or eos > lm.vocab_size - 1:
                raise ValueError(
                    "Expected eos to be in the range [{}, {}], got {}".format(
                        -lm.vocab_size, lm.vocab_size - 1, eos
                    )
                )
            eos = (eos + lm.vocab_size) % lm.vocab_size
        self.lm, self.eos = lm, eos
        try:
            device = next(iter(lm.parameters())).device
        except StopIteration:
            device = torch.device("cpu")
        self.register_buffer("device_buffer", torch.empty(0, device=device))

    def reset_parameters(self) -> None:
        if hasattr(self.lm, "reset_parameters"):
            self.lm.reset_parameters()

    def extra_repr(self) -> str:
        return "eos={}".format(self.eos)

    def update_log_probs_for_step(self, log_probs_prev, log_probs_t, y_prev, y_prev_lens, eos_mask):
        """Hook (reference _decoding.py:1393-1436); identity by default."""
        return log_probs_prev, log_probs_t

    def forward(
        self,
        prev_: Optional[Dict[str, torch.Tensor]] = None,
        batch_size: Optional[int] = None,
        max_iters: Optional[int] = None,
        initial_state: Optional[Dict[str, torch.Tensor]] = None,
    ):
        # (``prev_``: the reference's runtime keyword, _decoding.py:1446-1449; ``initial_state``: its
        # documented call signature -- both are accepted)
        if initial_state is None:
            initial_state = prev_
        prev = dict() if initial_state is None else initial_state
        device = self.device_buffer.device
        N = 1 if batch_size is None else batch_size
        if max_iters is None:
            if self.eos is None:
                raise RuntimeError("max_iters must be set when eos is unset")
            max_iters = 1073741824
        elif max_iters < 0:
            raise RuntimeError("max_iters must be non-negative, got {}".format(max_iters))
        y = torch.empty((0, N), device=device, dtype=torch.long)
        prev = self.lm.update_input(prev, y)
        y_lens = torch.zeros(N, dtype=torch.long, device=device)
        eos_mask = torch.zeros(N, device=device, dtype=torch.bool)
        log_probs = torch.zeros(N, device=device)
        for t in range(max_iters):
            if bool(eos_mask.all()):
                break
            t_ = torch.tensor(t, device=device)
            lp_t, prev = self.lm.calc_idx_log_probs(y[:t], prev, t_)
            lp_t = lp_t.log_softmax(-1)
            log_probs, lp_t = self.update_log_probs_for_step(log_probs, lp_t, y[:t], y_lens, eos_mask)
            if self.eos is not None:  # ended paths emit eos for free (:1483-1492)
                lp_t = lp_t.masked_fill(eos_mask.unsqueeze(1), -float("inf"))
                lp_t[:, self.eos] = lp_t[:, self.eos].masked_fill(eos_mask, 0.0)
            y, log_probs = random_walk_advance(lp_t, log_probs, y, y_lens)
            if self.eos is not None:
                y_lens = y_lens + (~eos_mask).long()
                eos_mask = y.gather(0, y_lens.unsqueeze(0) - 1).squeeze(0) == self.eos
            else:
                y_lens = y_lens + 1
        if batch_size is None:
            y, y_lens, log_probs = y.squeeze(1), y_lens.squeeze(0), log_probs.squeeze(0)
        return y, y_lens, log_probs
