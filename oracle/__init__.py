"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain C via ctypes, numpy for small float pieces) of the reference's
sequence-level hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package, and only as the checker:
the product package (``pydrobert-pytorch_amd/``) never imports it and has no CPU
fallback.

Parity status: PINNED.  Every function here is checked bit-for-bit (integers, float32
DP values) or to the stated tolerance (float probabilities, warped features) against
the live reference in the build container -- see ``tests/golden/make_golden.py`` -- and
against the reference's own known-answer tests restated under ``tests/``.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB_PATH = os.path.join(_BUILD, "libpdt_oracle.so")
_SOURCES = ["pdt_oracle_string.c", "pdt_oracle_decoding.c"]

MODE_FINAL, MODE_PREFIX, MODE_MASK = 0, 1, 2
WARN_REF_NO_EOS, WARN_HYP_NO_EOS, WARN_EMPTY_REF = 1, 2, 4

_lib = None


_SAN_LIB_PATH = os.path.join(_BUILD, "libpdt_oracle_san.so")


def build(force: bool = False, sanitize: bool = False) -> str:
    """Compile the C restatement with gcc (no reference sources involved).

    ``sanitize``: the AddressSanitizer + UndefinedBehaviorSanitizer build (CPU only; a process that
    loads it must have been started with ``LD_PRELOAD=$(gcc -print-file-name=libasan.so)``:
    ``tests/test_oracle_sanitizers.py`` does that around ``tests/fuzz/oracle_sweep.py``)."""
    out = _SAN_LIB_PATH if sanitize else _LIB_PATH
    srcs = [os.path.join(_HERE, s) for s in _SOURCES if os.path.exists(os.path.join(_HERE, s))]
    deps = srcs + [os.path.join(_HERE, "pdt_oracle.h")]
    if (
        not force
        and os.path.exists(out)
        and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps)
    ):
        return out
    os.makedirs(_BUILD, exist_ok=True)
    flags = ["-O2"]
    if sanitize:
        flags = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                 "-fno-sanitize-recover=all"]  # fmt: skip
    cmd = (
        ["gcc"] + flags + ["-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"]
        + ["-Wall", "-Wextra", "-o", out]
        + srcs
        + ["-lm"]
    )
    subprocess.run(cmd, check=True)
    return out


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        # PDT_ORACLE_SANITIZE=1: the sanitizer build (see build()); a test switch of the CHECKER,
        # read where the checker is loaded -- the product never comes here
        _lib = ctypes.CDLL(build(sanitize=os.environ.get("PDT_ORACLE_SANITIZE", "") == "1"))
        _declare(_lib)
    return _lib


_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int)


def _declare(L):
    L.pdt_oracle_string_matching.restype = ctypes.c_int
    L.pdt_oracle_string_matching.argtypes = (
        [_i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
        + [_i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
        + [ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int]
        + [ctypes.c_float] * 3
        + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int]
        + [_f32p, _u8p, _i64p, _i64p, _i32p]
    )
    L.pdt_oracle_optimal_completion_from_mask.restype = ctypes.c_int64
    L.pdt_oracle_optimal_completion_from_mask.argtypes = [
        _u8p,
        _i64p,
        ctypes.c_int64,
        ctypes.c_int64,
        ctypes.c_int64,
        ctypes.c_int64,
        ctypes.c_int64,
        ctypes.c_int64,
        ctypes.c_int64,
        _i64p,
    ]
    L.pdt_oracle_lens_from_eos.restype = None
    L.pdt_oracle_lens_from_eos.argtypes = [
        _i64p,
        ctypes.c_int64,
        ctypes.c_int64,
        ctypes.c_int64,
        ctypes.c_int64,
        ctypes.c_int64,
        _i64p,
    ]
    from . import _decoding

    _decoding.declare(L)


def _as_i64(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=np.int64)


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


def string_matching(
    ref,
    hyp,
    eos=None,
    include_eos=False,
    batch_first=False,
    ins_cost=1.0,
    del_cost=1.0,
    sub_cost=1.0,
    norm=False,
    mode=MODE_FINAL,
    exclude_last=False,
    padding=-100,
    return_mistakes=False,
    faithful=True,
):
    """Restates ``_string_matching`` (_string.py:146-406).

    Returns ``(result, ref_lens, hyp_lens, warn_flags)``; ``result`` is float32 ``(N,)``,
    ``(Hout, N)`` (``(N, Hout)`` if batch_first) or a bool mask ``(Hout, R, N)``.
    """
    ref, hyp = _as_i64(ref), _as_i64(hyp)
    if ref.ndim != 2 or hyp.ndim != 2:
        raise RuntimeError("ref and hyp must be 2 dimensional")
    if batch_first:
        (N, R), (N2, H) = ref.shape, hyp.shape
        rst, rsn, hst, hsn = 1, R, 1, H
    else:
        (R, N), (H, N2) = ref.shape, hyp.shape
        rst, rsn, hst, hsn = N, 1, N2, 1
    if N != N2:
        raise RuntimeError("ref has batch size {}, but hyp has {}".format(N, N2))
    Hout = H + (0 if exclude_last else 1)
    out = mask = None
    if mode == MODE_FINAL:
        out = np.empty((N,), np.float32)
    elif mode == MODE_PREFIX:
        out = np.empty((Hout, N), np.float32)
    else:
        # the initial row mask is appended BEFORE the loop (_string.py:271-278), so there is one row
        # even for H == 0 with exclude_last
        Hout = max(1, Hout)
        mask = np.zeros((Hout, R, N), np.uint8)
    rl, hl = np.empty((N,), np.int64), np.empty((N,), np.int64)
    flags = ctypes.c_int(0)
    rc = lib().pdt_oracle_string_matching(
        _ptr(ref, _i64p), R, rst, rsn, _ptr(hyp, _i64p), H, hst, hsn, N,
        int(eos is not None), int(eos if eos is not None else 0), int(include_eos),
        float(ins_cost), float(del_cost), float(sub_cost), int(norm), int(mode),
        int(exclude_last), float(padding), int(return_mistakes), int(faithful),
        _ptr(out, _f32p), _ptr(mask, _u8p), _ptr(rl, _i64p), _ptr(hl, _i64p),
        ctypes.byref(flags),
    )  # fmt: skip
    if rc == -2:  # row 0 of an empty mask / prefix buffer (_string.py:275, :285)
        raise IndexError("index 0 is out of bounds for dimension 0 with size 0")
    if rc != 0:
        raise RuntimeError("oracle string_matching failed: {}".format(rc))
    if mode == MODE_MASK:
        res = mask.astype(bool)
    elif mode == MODE_PREFIX and batch_first:
        res = np.ascontiguousarray(out.T)
    else:
        res = out
    return res, rl, hl, flags.value


def error_rate(ref, hyp, eos=None, include_eos=False, norm=True, batch_first=False,
               ins_cost=1.0, del_cost=1.0, sub_cost=1.0, faithful=True):  # fmt: skip
    """_string.py:409-434"""
    return string_matching(ref, hyp, eos, include_eos, batch_first, ins_cost, del_cost,
                           sub_cost, norm=norm, return_mistakes=True,
                           faithful=faithful)[0]  # fmt: skip


def edit_distance(ref, hyp, eos=None, include_eos=False, norm=False, batch_first=False,
                  ins_cost=1.0, del_cost=1.0, sub_cost=1.0, faithful=True):  # fmt: skip
    """_string.py:437-461"""
    return string_matching(ref, hyp, eos, include_eos, batch_first, ins_cost, del_cost,
                           sub_cost, norm=norm, faithful=faithful)[0]  # fmt: skip


def prefix_error_rates(ref, hyp, eos=None, include_eos=True, norm=True, batch_first=False,
                       ins_cost=1.0, del_cost=1.0, sub_cost=1.0, padding=-100,
                       exclude_last=False, faithful=True):  # fmt: skip
    """_string.py:520-550"""
    return string_matching(ref, hyp, eos, include_eos, batch_first, ins_cost, del_cost,
                           sub_cost, norm=norm, mode=MODE_PREFIX, exclude_last=exclude_last,
                           padding=padding, return_mistakes=True, faithful=faithful)[0]  # fmt: skip


def prefix_edit_distances(ref, hyp, eos=None, include_eos=True, norm=False,
                          batch_first=False, ins_cost=1.0, del_cost=1.0, sub_cost=1.0,
                          padding=-100, exclude_last=False, faithful=True):  # fmt: skip
    """_string.py:553-583"""
    return string_matching(ref, hyp, eos, include_eos, batch_first, ins_cost, del_cost,
                           sub_cost, norm=norm, mode=MODE_PREFIX, exclude_last=exclude_last,
                           padding=padding, return_mistakes=False, faithful=faithful)[0]  # fmt: skip


def optimal_completion(ref, hyp, eos=None, include_eos=True, batch_first=False,
                       ins_cost=1.0, del_cost=1.0, sub_cost=1.0, padding=-100,
                       exclude_last=False, faithful=True):  # fmt: skip
    """_string.py:464-517.  Returns int64 ``(Hout, N, C)`` (``(N, Hout, C)`` if batch_first)."""
    ref = _as_i64(ref)
    mask = string_matching(ref, hyp, eos, include_eos, batch_first, ins_cost, del_cost,
                           sub_cost, mode=MODE_MASK, exclude_last=exclude_last,
                           faithful=faithful)[0]  # fmt: skip
    Hout, R, N = mask.shape
    rst, rsn = (1, R) if batch_first else (N, 1)
    m8 = np.ascontiguousarray(mask.astype(np.uint8))
    L = lib()
    C = L.pdt_oracle_optimal_completion_from_mask(
        _ptr(m8, _u8p), _ptr(ref, _i64p), R, rst, rsn, Hout, N, int(padding), 0, None
    )
    tgt = np.full((Hout, N, C), padding, np.int64)
    if C > 0:
        L.pdt_oracle_optimal_completion_from_mask(
            _ptr(m8, _u8p), _ptr(ref, _i64p), R, rst, rsn, Hout, N, int(padding), C,
            _ptr(tgt, _i64p),
        )  # fmt: skip
    if batch_first:
        tgt = np.ascontiguousarray(tgt.transpose(1, 0, 2))
    return tgt


from ._decoding import *  # noqa: E402,F401,F403
from ._img import *  # noqa: E402,F401,F403
from ._losses import *  # noqa: E402,F401,F403
from ._seqops import *  # noqa: E402,F401,F403
from ._lm import backoff_log_probs, trie_log_probs  # noqa: E402,F401
from ._search import CounterLM, NGramLM, TableLM, beam_search, ctc_prefix_search_lm  # noqa: E402,F401
