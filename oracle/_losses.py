"""ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy restatement of the two sequence losses
(reference _string.py:1188-1251 and :1400-1472) on top of the C string oracle."""
import numpy as np

__all__ = ["hard_optimal_completion_distillation_loss", "minimum_error_rate_loss"]


def _np(x, dt=None):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    return x if dt is None else x.astype(dt)


def _log_softmax(x):
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


def hard_optimal_completion_distillation_loss(logits, ref, hyp, eos=None, include_eos=True,
                                              batch_first=False, ins_cost=1.0, del_cost=1.0,
                                              sub_cost=1.0, weight=None, reduction="mean",
                                              ignore_index=-2):  # fmt: skip
    from . import optimal_completion

    logits = _np(logits, np.float64)
    opt = optimal_completion(ref, hyp, eos=eos, include_eos=include_eos, batch_first=batch_first,
                             ins_cost=ins_cost, del_cost=del_cost, sub_cost=sub_cost,
                             padding=ignore_index, exclude_last=True)  # fmt: skip
    lsm = _log_softmax(logits)  # (.., .., V)
    pad = opt == ignore_index
    idx = np.where(pad, 0, opt)
    lp = np.take_along_axis(lsm, idx, 2)  # (.., .., C)
    w = np.ones(logits.shape[-1]) if weight is None else _np(weight, np.float64)
    loss = np.where(pad, 0.0, -lp * w[idx]).sum(2)
    loss = loss / np.maximum((~pad).sum(2), 1)
    if reduction == "mean":
        sd = 1 if batch_first else 0
        loss = (loss.sum(sd) / np.maximum((~pad).any(2).sum(sd), 1)).mean()
    elif reduction == "sum":
        loss = loss.sum()
    return np.asarray(loss, np.float32)


def minimum_error_rate_loss(log_probs, ref, hyp, eos=None, include_eos=True, sub_avg=True,
                            batch_first=False, norm=True, ins_cost=1.0, del_cost=1.0,
                            sub_cost=1.0, reduction="mean"):  # fmt: skip
    from . import error_rate

    lp = _np(log_probs, np.float64)
    ref, hyp = _np(ref), _np(hyp)
    if batch_first:
        B, S, H = hyp.shape
        if ref.ndim == 2:
            ref = np.repeat(ref[:, None], S, 1)
        ref, hyp = ref.reshape(B * S, -1), hyp.reshape(B * S, H)
    else:
        H, B, S = hyp.shape
        if ref.ndim == 2:
            ref = np.repeat(ref[..., None], S, 2)
        ref, hyp = ref.reshape(ref.shape[0], -1), hyp.reshape(H, -1)
    er = error_rate(ref, hyp, eos=eos, include_eos=include_eos, norm=norm, batch_first=batch_first,
                    ins_cost=ins_cost, del_cost=del_cost, sub_cost=sub_cost).reshape(B, S).astype(np.float64)  # fmt: skip
    if sub_avg:
        er = er - er.mean(1, keepdims=True)
    p = np.exp(_log_softmax(lp))
    loss = er * p
    if reduction == "mean":
        loss = loss.mean()
    elif reduction == "sum":
        loss = loss.sum()
    return np.asarray(loss, np.float32)
