"""Run-time switches of the package, read from the environment ONCE (at import).

Every ``PDT_*`` variable that selects between routes of an operator is listed here with its
default; nothing on an operator's call path reads ``os.environ``.  The host-side ones live in this
module's table; the kernel-selection ones live in the native library (``csrc/switches.hpp``,
``pdt_amd_set_switch``) and are forwarded.  ``set`` / ``override`` change one afterwards -- the tests
run two routes in one process that way.  The table for maintainers is in INTEGRATION.md.
"""

import contextlib
import ctypes
import os

from . import _cabi

# host-side switches: name -> default
_HOST_DEFAULTS = {
    # reuse the (lengths, classes, match tables) of the previous string operator on the same
    # (ref, hyp) pair.  OFF by default: a hit is decided by tensor identity, which in-place writes
    # that bypass the version counter (`.data`, raw pointers, graph replays) do not change.
    "PDT_LEV_CACHE": 0,
    "PDT_CTC_LM_FUSED": 1,  # CTCPrefixSearch + LookupLanguageModel: a frame in one kernel
    "PDT_CTC_LM_SEARCH": 1,  # ... and every frame from one call of the library
    "PDT_CTC_LM_TABLE": 1,  # ... a bigram model's factor rows from a table built once per model (csrc/ctc_lm_table.hip)
    "PDT_CTC_STEP_MIX": 1,  # CTCPrefixSearch + any other LM: the step kernel forms the extension probabilities itself (0: fusion_ext first)
    "PDT_BEAM_FUSED": 1,  # BeamSearch: one kernel per iteration
    "PDT_BEAM_SEARCH": 1,  # ... and every iteration from ONE launch, the paths read off a trie at the end (csrc/beam_step.hip)
    "PDT_BEAM_TABLE": 1,  # BeamSearch over a bigram LookupLanguageModel reads its dense table
    "PDT_CHECK_INVARIANTS": 0,  # BeamSearch's loop checks that the history grows (a host read per iteration)
}
# switches of the native library (csrc/switches.hpp); the library reads the environment itself
_NATIVE = (
    "PDT_LEV_BITPAR", "PDT_OC_BITPAR", "PDT_OC_WAVES", "PDT_CTC_EXACT_DIV", "PDT_CTC_ROWREG",
    "PDT_STEP_WIDE", "PDT_LM_CACHE", "PDT_LM_PERSISTENT", "PDT_LM_STEP_WAVES", "PDT_WARP_BANDS", "PDT_CTC_LEAN_EXTRA",
    "PDT_CTC_PAIR", "PDT_STEP_FLAT",
)  # fmt: skip


def _from_env(name, default):
    raw = os.environ.get(name, "")
    try:
        return int(raw) if raw != "" else default
    except ValueError:
        return default


_values = {name: _from_env(name, deft) for name, deft in _HOST_DEFAULTS.items()}


def names():
    """Every switch the package knows."""
    return tuple(_HOST_DEFAULTS) + _NATIVE


def get(name: str) -> int:
    if name in _values:
        return _values[name]
    if name in _NATIVE:
        out = ctypes.c_int(0)
        _cabi.check(_cabi.lib().pdt_amd_get_switch(name.encode(), ctypes.addressof(out)), "pdt_amd_get_switch")
        return out.value
    raise KeyError(name)


def set(name: str, value: int) -> None:  # noqa: A001 (mirrors `get`)
    if name in _values:
        _values[name] = int(value)
    elif name in _NATIVE:
        _cabi.check(_cabi.lib().pdt_amd_set_switch(name.encode(), int(value)), "pdt_amd_set_switch")
    else:
        raise KeyError(name)


@contextlib.contextmanager
def override(**kw):
    """``with switches.override(PDT_BEAM_FUSED=0): ...`` -- restores the previous values on exit."""
    old = {k: get(k) for k in kw}
    try:
        for k, v in kw.items():
            set(k, v)
        yield
    finally:
        for k, v in old.items():
            set(k, v)
