"""CPU: host-side behaviour of the drop-in boundary -- constructor validation, shape errors,
the LM interface, and the absence of any CPU compute path."""
import numpy as np
import pytest
import torch

from pydrobert_amd import argcheck, config
from pydrobert_amd import functional as F
from pydrobert_amd import modules as M

from _toy_lm import BigramLM


def test_public_surface_matches_reference_names():
    # reference functional.py:24-58 / modules.py:72-124, restricted to the hot path
    for name in ("error_rate", "edit_distance", "prefix_error_rates", "prefix_edit_distances",
                 "optimal_completion", "fill_after_eos", "beam_search_advance",
                 "ctc_prefix_search_advance", "polyharmonic_spline", "warp_1d_grid",
                 "dense_image_warp", "sparse_image_warp", "spec_augment",
                 "spec_augment_draw_parameters", "spec_augment_apply_parameters"):  # fmt: skip
        assert callable(getattr(F, name)), name
    for name in ("ErrorRate", "EditDistance", "PrefixErrorRates", "PrefixEditDistances",
                 "OptimalCompletion", "FillAfterEndOfSequence", "BeamSearch", "CTCPrefixSearch",
                 "PolyharmonicSpline", "Warp1DGrid", "DenseImageWarp", "SparseImageWarp",
                 "SpecAugment", "SequentialLanguageModel", "ExtractableSequentialLanguageModel",
                 "MixableSequentialLanguageModel"):  # fmt: skip
        assert isinstance(getattr(M, name), type), name
    assert config.INDEX_PAD_VALUE == -100 and config.DEFT_INS_COST == 1.0


def test_defaults_match_reference_signatures():
    import inspect

    def defaults(f):
        return {k: v.default for k, v in inspect.signature(f).parameters.items() if v.default is not inspect._empty}

    assert defaults(F.error_rate) == dict(eos=None, include_eos=False, norm=True, batch_first=False,
                                          ins_cost=1.0, del_cost=1.0, sub_cost=1.0, warn=True)  # fmt: skip
    assert defaults(F.edit_distance)["norm"] is False
    d = defaults(F.prefix_error_rates)
    assert d["include_eos"] is True and d["norm"] is True and d["padding"] == -100 and d["exclude_last"] is False
    assert defaults(F.prefix_edit_distances)["norm"] is False
    assert defaults(F.optimal_completion)["include_eos"] is True
    d = defaults(F.sparse_image_warp)
    assert d["indexing"] == "hw" and d["field_interpolation_order"] == 2 and d["include_flow"] is True
    assert list(inspect.signature(F.ctc_prefix_search_advance).parameters) == [
        "probs_t", "width", "probs_prev", "y_prev", "y_prev_last", "y_prev_lens", "prev_is_prefix"]  # fmt: skip
    assert list(inspect.signature(F.beam_search_advance).parameters) == [
        "log_probs_t", "width", "log_probs_prev", "y_prev", "y_prev_lens"]  # fmt: skip


def test_calls_written_against_the_reference_bind():
    """tests/golden/signatures.json (made from the live reference by make_signatures.py) lists the
    parameters of every mirrored function, Module constructor and ``forward``: each must exist
    here under the same name, in the same positional order, with a default where the reference
    has one -- so positional AND keyword calls written against the reference bind unchanged
    (e.g. ``BeamSearch.forward(initial_state_=...)``, ``CTCPrefixSearch.forward(prev_=...)``)."""
    import inspect
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "signatures.json")) as f:
        sig = json.load(f)

    def check(what, fn, ref_params):
        fn = getattr(fn, "__wrapped__", fn)
        mine = [p for p in inspect.signature(fn).parameters.values() if p.name != "self"]
        names = [p.name for p in mine]
        assert names[: len(ref_params)] == [r[0] for r in ref_params], (what, names, ref_params)
        for p, (name, has_default, kind) in zip(mine, ref_params):
            assert p.kind.name == kind, (what, name)
            if has_default:
                assert p.default is not inspect.Parameter.empty, (what, name)
        for p in mine[len(ref_params):]:  # extras must not be required
            assert p.default is not inspect.Parameter.empty or p.kind.name.startswith("VAR"), (what, p.name)

    assert len(sig["functional"]) >= 22 and len(sig["modules"]) >= 27
    for name, ref_params in sig["functional"].items():
        check(name, getattr(F, name), ref_params)
    for name, d in sig["modules"].items():
        cls = getattr(M, name)
        check(name + ".__init__", cls.__init__, d["__init__"])
        check(name + ".forward", cls.forward, d["forward"])


def test_argcheck_contract():
    assert argcheck.is_int(3, "x") == 3 and argcheck.is_float(2, "x") == 2.0
    assert argcheck.is_int(None, "x", True) is None
    for fn, bad in ((argcheck.is_int, 1.5), (argcheck.is_bool, 1), (argcheck.is_float, "a"),
                    (argcheck.is_posi, 0), (argcheck.is_nonnegi, -1), (argcheck.is_closed01, 1.5),
                    (argcheck.is_nonnegf, -0.1)):  # fmt: skip
        with pytest.raises(ValueError):
            fn(bad, "x")
    with pytest.raises(ValueError, match="is not one of"):
        argcheck.is_in("q", ("a", "b"), "x")


def test_module_constructors_validate():
    M.ErrorRate(eos=3, norm=False)
    with pytest.raises(ValueError):
        M.ErrorRate(eos="a")
    with pytest.raises(ValueError):
        M.PrefixErrorRates(padding=1.5)
    with pytest.raises(ValueError):
        M.CTCPrefixSearch(4, beta=2.0)
    with pytest.raises(ValueError):
        M.SpecAugment(max_time_warp=-1)
    with pytest.raises(ValueError):
        M.DenseImageWarp(mode="cubic")
    with pytest.raises(ValueError):
        M.SparseImageWarp(indexing="xy")
    lm = BigramLM(torch.zeros(5, 4))
    with pytest.raises(ValueError):
        M.BeamSearch(lm, 0)
    with pytest.raises(ValueError):
        M.BeamSearch(lm, 2, eos=7)
    assert M.BeamSearch(lm, 2, eos=-1).eos == 3
    r = repr(M.OptimalCompletion(eos=1, padding=-3))
    assert "eos=1" in r and "padding=-3" in r
    assert "warp_t=80.0" in repr(M.SpecAugment())


def test_no_cpu_compute_path():
    """CPU tensors are refused: the package has no fallback implementation."""
    ref = torch.zeros((3, 2), dtype=torch.long)
    for fn in (F.error_rate, F.edit_distance, F.prefix_error_rates, F.optimal_completion):
        with pytest.raises(RuntimeError, match="ROCm"):
            fn(ref, ref)
    with pytest.raises(RuntimeError, match="ROCm"):
        F.ctc_prefix_search(torch.zeros(2, 1, 3), 2)
    with pytest.raises(RuntimeError, match="ROCm"):
        F.spec_augment_apply_parameters(torch.zeros(1, 4, 3), (torch.ones(1), torch.ones(1)) + (torch.empty(0),) * 6, 1)
    with pytest.raises(RuntimeError, match="ROCm"):
        F.polyharmonic_spline(torch.zeros(1, 3, 1), torch.zeros(1, 3, 1), torch.zeros(1, 2, 1), 1)


def test_shape_errors_precede_device_checks():
    with pytest.raises(RuntimeError, match="2 dimensional"):
        F.error_rate(torch.zeros(3, dtype=torch.long), torch.zeros((3, 1), dtype=torch.long))
    with pytest.raises(RuntimeError, match="batch size"):
        F.error_rate(torch.zeros((3, 2), dtype=torch.long), torch.zeros((3, 1), dtype=torch.long))
    with pytest.raises(RuntimeError, match="3 dimensional"):
        F.beam_search_advance(torch.zeros(2, 3), 1, torch.zeros(2, 3), torch.zeros(0, 2, 3))
    with pytest.raises(RuntimeError, match="width must be positive"):
        F.ctc_prefix_search_advance((torch.zeros(1, 1, 2), torch.zeros(1, 2), torch.zeros(1)), 0,
                                    (torch.zeros(1, 1),) * 2, torch.zeros(0, 1, 1), torch.zeros(1, 1),
                                    torch.zeros(1, 1), torch.ones(1, 1, 1, dtype=torch.bool))  # fmt: skip
    with pytest.raises(RuntimeError, match="three dimensions"):
        F.spec_augment(torch.zeros(4, 3), 1.0, 0.0, 1, 1, 0.1, 1, 0.1, 1, 1)
    with pytest.raises(RuntimeError, match="values of lengths"):
        F.spec_augment(torch.zeros(2, 4, 3), 1.0, 0.0, 1, 1, 0.1, 1, 0.1, 1, 1, torch.tensor([0, 4]))


def test_fill_after_eos_needs_a_device():
    """No CPU path in the product (the semantics are checked on the GPU, tests/test_string_gpu.py)."""
    tok = torch.tensor([[1, 2], [0, 3], [4, 0], [0, 5]])
    with pytest.raises(RuntimeError, match="ROCm"):
        F.fill_after_eos(tok, 0, 0, -1)
    with pytest.raises(RuntimeError, match="ROCm"):
        M.FillAfterEndOfSequence(0)(tok)


def test_lm_interface_forward_idx_normalisation():
    table = torch.randn(5, 4).log_softmax(-1)
    lm = BigramLM(table)
    hist = torch.tensor([[0, 1], [2, 3], [1, 1]])
    full = lm(hist)
    assert full.shape == (4, 2, 4)
    assert torch.equal(full[0], table[4].expand(2, 4))
    lp, _ = lm(hist, idx=-1)
    assert torch.equal(lp, full[3])
    lp, _ = lm(hist, idx=torch.tensor([1, 3]))
    assert torch.equal(lp[0], full[1, 0]) and torch.equal(lp[1], full[3, 1])
    with pytest.raises(RuntimeError, match="2 dimensional"):
        lm(hist[0])
    with pytest.raises(RuntimeError, match="between"):
        lm(hist, idx=9)
    with pytest.raises(TypeError):
        M.SequentialLanguageModel(3)  # abstract


def test_spec_augment_draw_parameters_refuses_cpu_tensors():
    """The draw is one HIP kernel (csrc/img_warp.hip spec_augment_draw_kernel): like every operator of
    the package it refuses CPU tensors loudly instead of falling back (the ranges of the drawn
    parameters -- reference tests/test_img.py:226-281 -- are checked on the GPU, tests/test_img_gpu.py);
    a call that draws nothing needs no kernel and returns the reference's zero-size pairs."""
    feats = torch.zeros(5, 12, 6)
    with pytest.raises(RuntimeError, match="ROCm"):
        F.spec_augment_draw_parameters(feats, 8.0, 2.0, 12, 4, 0.1, 3, 0.03, 2, torch.full((5,), 12))
    out = F.spec_augment_draw_parameters(feats, 0.0, 0.0, 0, 0, 0.1, 3, 0.03, 0)  # (round-4 regression: raised)
    assert len(out) == 8 and all(o.shape == (0,) and o.dtype == torch.float and o.device.type == "cpu" for o in out)
    sa = M.SpecAugment(0.0, 0.0, 0, 0).train()  # every group disabled: features pass through, on any device
    assert sa(feats) is feats


# ---- language model host logic ---------------------------------------
def test_trie_builder_matches_reference_buffers():
    """The array-based builder reproduces the reference's flattened trie bit for bit, dtypes
    included (buffers captured from the live reference in tests/golden/lm.npz)."""
    import torch
    from _lm_fixtures import dicts_from_golden, golden
    from pydrobert_amd.modules import LookupLanguageModel

    g = golden()
    for tag in ("A", "B", "C", "U"):
        V, sos, N, dicts = dicts_from_golden(g, tag)
        lm = LookupLanguageModel(V, sos, dicts)
        assert lm.max_ngram == N
        assert lm.max_ngram_nodes == int(g[tag + "_cfg"][3])
        assert lm.max_direct_descendants == int(g[tag + "_cfg"][4])
        for name in ("logps", "logbs", "ids", "offsets"):
            exp, act = g[tag + "_" + name], getattr(lm, name).numpy()
            assert exp.dtype == act.dtype and exp.shape == act.shape, (tag, name, exp.dtype, act.dtype)
            assert np.array_equal(exp, act, equal_nan=exp.dtype.kind == "f"), (tag, name)
        # destructive=False left the caller's tables alone
        assert dicts_from_golden(g, tag)[3] == dicts
    uniform = LookupLanguageModel(7, 2)
    assert torch.allclose(uniform.logps, torch.full((7,), -np.log(7.0)).float())
    assert uniform.offsets.numel() == uniform.ids.numel() == uniform.logbs.numel() == 0


@pytest.mark.parametrize("which", ["unigram", "bigram", "trigram"])
def test_trie_known_answers(which):
    """The reference's hand-written tables (facts from tests/test_lm.py:117-139): walking the
    built trie from the last token backwards finds every entry with its values."""
    from pydrobert_amd.modules import LookupLanguageModel

    prob_dicts = {
        "unigram": [{0: 0.0, 1: 1.0, 4: 4.0}],
        "bigram": [
            {1: (1.0, -1.0), 2: (2.0, -2.0), 3: (3.0, -3)},
            {(1, 0): 1.1, (1, 1): 11.1, (3, 2): 23.1},
        ],
        "trigram": [
            {1: (1.0, -1.0), 2: (2.0, -2.0), 3: (3.0, -3.0), 4: (4.0, -4.0)},
            {(1, 1): (11.1, -11.1), (2, 3): (32.1, -32.1), (2, 4): (42.1, -42.1), (4, 1): (14.1, -14.1)},
            {(0, 0, 1): 1.2, (0, 0, 2): 2.2, (4, 1, 4): 414.2, (3, 4, 1): 143.2},
        ],
    }[which]
    V, N = 5, len(prob_dicts)
    lm = LookupLanguageModel(V, 0, prob_dicts)
    off, ids = lm.offsets.long().numpy(), lm.ids.long().numpy()
    logps, logbs = lm.logps.numpy(), lm.logbs.numpy()
    U = V + 1
    S = 0
    for n, d in enumerate(prob_dicts):
        for key, exp in d.items():
            key = (key,) if n == 0 else key
            node = key[-1]
            for tok in key[-2::-1]:
                lo, hi = node + off[node], node + 1 + off[node + 1]
                S = max(S, hi - lo)
                hits = [c for c in range(lo, hi) if ids[c - U] == tok]
                assert len(hits) == 1, (key, tok)
                node = hits[0]
            if n == N - 1:
                assert np.isclose(logps[node], exp), key
            else:
                assert np.isclose(logps[node], exp[0]) and np.isclose(logbs[node], exp[1]), key
    assert lm.max_direct_descendants >= S
    if N > 1:  # (0, 0) is only a suffix of (0, 0, x): added with probability 0, no penalty
        assert lm.max_ngram_nodes == len(prob_dicts[-1])


def test_trie_builder_rejects_bad_tables():
    from pydrobert_amd.modules import LookupLanguageModel

    with pytest.raises(ValueError, match="at least unigrams"):
        LookupLanguageModel(3, 0, [])
    with pytest.raises(ValueError, match="Unexpected unigrams"):
        LookupLanguageModel(3, 0, [{7: 0.0}])
    with pytest.raises(ValueError, match="must not be empty"):
        LookupLanguageModel(3, 0, [{0: (0.0, 0.0)}, {}])
    with pytest.raises(ValueError, match="not a sequence of length"):
        LookupLanguageModel(3, 0, [{0: (0.0, 0.0)}, {(0, 1, 2): 0.0}])
    with pytest.raises(ValueError, match="Unexpected tokens"):
        LookupLanguageModel(3, 0, [{0: (0.0, 0.0)}, {(0, 5): 0.0}])


def test_lookup_lm_state_dict_resizes():
    """load_state_dict accepts tables of another size and re-derives the order (reference
    tests/test_lm.py:422-489)."""
    from _lm_fixtures import dicts_from_golden, golden
    from pydrobert_amd.modules import LookupLanguageModel

    g = golden()
    V, sos, N, dicts = dicts_from_golden(g, "B")
    full = LookupLanguageModel(V, sos, dicts)
    blank = LookupLanguageModel(V, sos)
    assert blank.max_ngram == 1
    blank.load_state_dict(full.state_dict())
    assert blank.max_ngram == N and blank.max_ngram_nodes == full.max_ngram_nodes
    assert blank.max_direct_descendants == full.max_direct_descendants
    for name in ("logps", "logbs", "ids", "offsets"):
        assert np.array_equal(getattr(blank, name).numpy(), getattr(full, name).numpy(), equal_nan=True)
    # back down to a unigram table
    uni = LookupLanguageModel(V, sos, [{k: v[0] for k, v in dicts[0].items()}])
    blank.load_state_dict(uni.state_dict())
    assert blank.max_ngram == 1 and blank.max_ngram_nodes == V + blank.shift
    # wrong vocabulary size is detected
    wrong = LookupLanguageModel(V + 4, sos)
    with pytest.raises(RuntimeError):
        wrong.load_state_dict(full.state_dict())
    sd = full.state_dict()
    del sd["ids"]
    with pytest.raises(RuntimeError, match="Missing key"):
        blank.load_state_dict(sd)


def test_shallow_fusion_dict_plumbing():
    from pydrobert_amd.modules import LookupLanguageModel, MixableShallowFusionLanguageModel, ShallowFusionLanguageModel

    a, b = LookupLanguageModel(4, 0), LookupLanguageModel(4, 0)
    with pytest.raises(ValueError, match="vocab_size"):
        ShallowFusionLanguageModel(a, LookupLanguageModel(5, 0))
    with pytest.raises(ValueError, match="matches second_prefix"):
        ShallowFusionLanguageModel(a, b, 0.1, "x.", "x.")
    with pytest.raises(ValueError, match="cannot be empty"):
        ShallowFusionLanguageModel(a, b, 0.1, "", "y.")
    lm = MixableShallowFusionLanguageModel(a, b, 0.5, "one.", "two.")
    import torch

    prev = {"one.h": torch.zeros(2), "two.c": torch.ones(3)}
    first, second = lm.split_dicts(prev)
    assert list(first) == ["h"] and list(second) == ["c"]
    assert set(lm.merge_dicts(first, second)) == set(prev)
    with pytest.raises(RuntimeError, match="does not start with"):
        lm.split_dicts({"three.x": torch.zeros(1)})


def test_direct_calls_only_when_the_dispatcher_has_nothing_to_do():
    """`_cabi.plain_call`: the step functions' wrappers call the implementation behind their operator only in
    plain eager mode -- not for inputs autograd would follow, tensor subclasses (fake tensors), under a
    dispatch or function mode, or inside a functorch transform."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from torch.overrides import TorchFunctionMode
    from torch.utils._python_dispatch import TorchDispatchMode

    from pydrobert_amd import _cabi

    a, b = torch.zeros(3), torch.zeros(3, dtype=torch.long)
    assert _cabi.plain_call(a, b, None)
    g = torch.zeros(3, requires_grad=True)
    assert not _cabi.plain_call(a, g)
    with torch.no_grad():
        assert _cabi.plain_call(a, g)
    with FakeTensorMode() as mode:
        assert not _cabi.plain_call(a)
        assert not _cabi.plain_call(mode.from_tensor(a))
    assert not _cabi.plain_call(FakeTensorMode().from_tensor(a))

    class Quiet(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            return func(*args, **(kwargs or {}))

    with Quiet():
        assert not _cabi.plain_call(a)

    class QuietF(TorchFunctionMode):
        def __torch_function__(self, func, types, args=(), kwargs=None):
            return func(*args, **(kwargs or {}))

    with QuietF():
        assert not _cabi.plain_call(a)
    seen = []
    torch.vmap(lambda x: (seen.append(_cabi.plain_call(x)), x)[1])(torch.zeros(2, 3))
    assert seen == [False]
    assert _cabi.plain_call(a)
