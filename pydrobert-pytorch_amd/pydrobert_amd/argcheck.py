"""The handful of constructor validators the hot-path Modules use.

Same contract as the reference's ``argcheck.is_*`` (argcheck.py:202-221): return the
value cast to the canonical type or raise ``ValueError("<name> (<val>) is not a ...")``.
"""

import numpy as np
import torch


def _nv(name, val):
    if isinstance(val, str):
        val = "'{}'".format(val)
    return "{}".format(val) if name is None else "{} ({})".format(name, val)


def _check(t, ts, val, name, allow_none):
    if val is None and allow_none:
        return None
    # bool is an int subclass; the reference accepts it wherever an int is accepted
    if isinstance(val, ts):
        return val if type(val) is t else t(val)
    tname = t.__name__
    article = "an" if tname[0] in "aeiou" else "a"
    raise ValueError("{} is not {} {}".format(_nv(name, val), article, tname))


def is_int(val, name=None, allow_none=False):
    return _check(int, (int, np.integer), val, name, allow_none)


def is_bool(val, name=None, allow_none=False):
    return _check(bool, (bool,), val, name, allow_none)


def is_float(val, name=None, allow_none=False):
    return _check(float, (float, int, np.integer, np.floating), val, name, allow_none)


def is_tensor(val, name=None, allow_none=False):
    return _check(torch.Tensor, (torch.Tensor,), val, name, allow_none)


def is_in(val, collection, name=None, allow_none=False):
    if val is None and allow_none:
        return None
    if val not in collection:
        raise ValueError("{} is not one of {}".format(_nv(name, val), list(collection)))
    return val


def _num(val, name, allow_none, pred, what):
    if val is None and allow_none:
        return None
    if not isinstance(val, (int, float, np.integer, np.floating)) or isinstance(val, bool):
        raise ValueError("{} is not num-like".format(_nv(name, val)))
    if not pred(val):
        raise ValueError("{} is not {}".format(_nv(name, val), what))
    return val


def is_posi(val, name=None, allow_none=False):
    val = is_int(val, name, allow_none)
    return _num(val, name, allow_none, lambda v: v > 0, "positive")


def is_nonnegi(val, name=None, allow_none=False):
    val = is_int(val, name, allow_none)
    return _num(val, name, allow_none, lambda v: v >= 0, "non-negative")


def is_nonnegf(val, name=None, allow_none=False):
    val = is_float(val, name, allow_none)
    return _num(val, name, allow_none, lambda v: v >= 0, "non-negative")


def is_posf(val, name=None, allow_none=False):
    val = is_float(val, name, allow_none)
    return _num(val, name, allow_none, lambda v: v > 0, "positive")


def is_closed01(val, name=None, allow_none=False):
    val = is_float(val, name, allow_none)
    return _num(val, name, allow_none, lambda v: 0.0 <= v <= 1.0, "in [0, 1]")
