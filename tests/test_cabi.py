"""CPU: the C-ABI library loads and exports exactly what include/pdt_amd.h declares; the
ctypes table mirrors the header; argument validation happens before anything touches a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pdt_amd.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    from pydrobert_amd import _cabi

    if not os.path.exists(_cabi.LIB_PATH):
        g.build()
    return _cabi.lib()


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pdt_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libpdt_amd.so does not export " + n


def test_ctypes_table_matches_header():
    from pydrobert_amd import _cabi

    assert sorted(_cabi.SIGNATURES) == _declared()
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _cabi.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), (name, len(params), len(args))
        for p, a in zip(params, args):
            if "const char *" in p:
                assert a is ctypes.c_char_p, (name, p)
            elif "*" in p:
                assert a is ctypes.c_void_p, (name, p)
            elif "float" in p:
                assert a is ctypes.c_float, (name, p)
            elif "int64_t" in p:
                assert a is ctypes.c_int64, (name, p)
            else:
                assert a is ctypes.c_int, (name, p)


def test_abi_version_and_arg_validation_without_gpu(lib):
    """Entry points validate arguments and return early for empty batches before any launch,
    so these calls are safe on a machine without a GPU."""
    from pydrobert_amd import _cabi

    assert lib.pdt_amd_abi_version() >= 1
    assert lib.pdt_oc_mask_words(512) == 16 and lib.pdt_oc_mask_words(513) == 17
    assert lib.pdt_ctc_prefix_search_workspace_bytes(10, 4, 30, 16) >= 10 * 4 * 16 * 8
    assert lib.pdt_spline_workspace_bytes(2, 3, 1, 1) > 0
    # the bit-parallel Levenshtein kernels: hypotheses of at most 1024 tokens, any reference
    assert lib.pdt_lev_workspace_bytes(512, 512, 4096) > 0 and lib.pdt_lev_workspace_bytes(9, 1025, 4) == 0
    assert lib.pdt_lev_workspace_bytes(512, 512, 8192) > lib.pdt_lev_workspace_bytes(512, 512, 4096)
    z = [0] * 32
    # N == 0: OK, nothing to do
    assert lib.pdt_lev(0, 5, 1, 1, 0, 5, 1, 1, 0, 0, 0, 0, 1.0, 1.0, 1.0, 0, 0, 0, 0.0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0) == _cabi.PDT_OK
    # bad mode / negative sizes
    assert lib.pdt_lev(0, 5, 1, 1, 0, 5, 1, 1, 0, 0, 0, 0, 1.0, 1.0, 1.0, 0, 7, 0, 0.0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0) == _cabi.PDT_E_ARG
    assert lib.pdt_lev(0, -1, 1, 1, 0, 5, 1, 1, 4, 0, 0, 0, 1.0, 1.0, 1.0, 0, 0, 0, 0.0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0) == _cabi.PDT_E_ARG
    # null pointers with a non-empty batch
    assert lib.pdt_lev(0, 5, 1, 1, 0, 5, 1, 1, 4, 0, 0, 0, 1.0, 1.0, 1.0, 0, 0, 0, 0.0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0) == _cabi.PDT_E_ARG
    assert lib.pdt_ctc_prefix_search(0, 5, 2, 3, 1, 1, 1, 0, 64, 5, 0, 0, 0, 0, 0) != _cabi.PDT_OK
    assert lib.pdt_ctc_prefix_search(0, 5, 0, 3, 1, 1, 1, 0, 4, 5, 0, 0, 0, 0, 0) == _cabi.PDT_OK
    assert lib.pdt_spec_augment_apply(0, 0, 5, 5, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) == _cabi.PDT_OK
    assert lib.pdt_dense_image_warp(0, 0, 1, 1, 4, 4, 0, 5, 0, 0, 0) == _cabi.PDT_E_ARG
    # later rows: the same contract
    assert lib.pdt_pad_variable(0, 0, 4, 3, 4, 0, 0, 0, 0, 0, 0, 0) == _cabi.PDT_OK  # N == 0
    assert lib.pdt_pad_variable(0, 2, 4, 3, 3, 0, 0, 0, 0, 6, 0, 0) == _cabi.PDT_E_ARG  # null pointers
    assert lib.pdt_pad_variable(0, 2, 4, 3, 4, 0, 0, 9, 0, 6, 0, 0) == _cabi.PDT_E_ARG  # bad mode
    assert lib.pdt_pad_variable_backward(0, 0, 4, 3, 0, 0, 1, 6, 0, 0) == _cabi.PDT_OK
    assert lib.pdt_lookup_lm_log_probs(0, 3, 2, 2, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 10, 3, 11, 0, 0, 0, 0) == _cabi.PDT_OK  # rows == 0
    assert lib.pdt_lookup_lm_log_probs(0, 3, 2, 2, 1, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 10, 1, 11, 0, 0, 0, 0) == _cabi.PDT_E_ARG  # order < 2
    assert lib.pdt_lookup_lm_log_probs(0, 3, 2, 2, 1, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 10, 40, 11, 0, 0, 0, 0) == _cabi.PDT_E_TOO_LONG
    assert lib.pdt_spec_augment_apply_backward(0, 0, 5, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) == _cabi.PDT_OK
    assert lib.pdt_dense_image_warp_backward(0, 0, 1, 1, 4, 4, 0, 0, 0, 0, 0) == _cabi.PDT_E_ARG
    del z


def test_check_maps_status_to_runtime_error():
    from pydrobert_amd import _cabi

    _cabi.check(0, "x")
    for rc in (_cabi.PDT_E_ARG, _cabi.PDT_E_TOO_LONG, 700):
        with pytest.raises(RuntimeError):
            _cabi.check(rc, "x")


def test_switches_read_once_and_settable(lib):
    """Every PDT_* switch is listed in one table per side (csrc/switches.hpp, pydrobert_amd/switches.py),
    initialised from the environment once, and can be changed and read back through the C ABI."""
    from pydrobert_amd import _cabi, switches

    assert lib.pdt_amd_set_switch(b"PDT_NO_SUCH_SWITCH", 1) == _cabi.PDT_E_ARG
    for name in switches.names():
        old = switches.get(name)
        with switches.override(**{name: old + 1}):
            assert switches.get(name) == old + 1
        assert switches.get(name) == old
    with pytest.raises(KeyError):
        switches.get("PDT_NO_SUCH_SWITCH")
    # nothing on an operator's call path reads the environment
    pkg = os.path.join(ROOT, "pydrobert-pytorch_amd")
    for sub, pat in (("pydrobert_amd", r"os\.environ|getenv"), ("csrc", r"getenv")):
        for fn in sorted(os.listdir(os.path.join(pkg, sub))):
            if not fn.endswith((".py", ".hip", ".hpp")) or fn in ("switches.py", "_cabi.py"):
                continue
            text = open(os.path.join(pkg, sub, fn)).read()
            hits = [l for l in text.splitlines() if re.search(pat, l) and not l.lstrip().startswith(("//", "#"))]
            if fn == "pdt_api.hip":  # the one reader (pdt::switches)
                assert len(hits) == 1, hits
            else:
                assert not hits, (fn, hits)
