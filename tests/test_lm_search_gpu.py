"""The searches with a language model in the loop on the GPU, every route of the package against
(1) the live reference's own outputs (tests/golden/lm_search.npz: CTCPrefixSearch / BeamSearch driving
the reference's LookupLanguageModel, orders 2-4, both mixes, ragged lens, sos in / out of the
vocabulary, eos / finish_all_paths; and a model WITH state) and (2) the oracle's restatement of the two
loops (oracle/_search.py, itself pinned to those fixtures on the CPU) at the C3 sample size -- so a
mistake shared by the host loop and the fused kernels cannot pass: neither is the other's reference.

Routes of CTCPrefixSearch + LookupLanguageModel: "table" = the whole search in one launch with a
bigram model's factor rows read from a table (pdt_ctc_lm_table_search; the default for order two --
higher orders take "search"), "search" = every frame from one library call
(pdt_ctc_lookup_lm_search), "frame" = the host's loop around the one-kernel frame
(pdt_ctc_lookup_lm_advance), "three" = lookup scores -> fusion_ext -> ctc_prefix_search_advance.
Routes of BeamSearch: "table" = fused iterations reading a bigram model's dense table, "fused" = fused
iterations around the model's forward, "loop" = the reference-shaped loop around beam_search_advance."""
import numpy as np
import pytest
import torch

import _drift
import oracle
from pydrobert_amd import modules as M

from _lm_fixtures import lm_search_golden, same_paths
from _toy_lm import CounterLM

pytestmark = pytest.mark.gpu
DEV = "cuda"
CTC_ROUTES = {"table": dict(PDT_CTC_LM_TABLE=1, PDT_CTC_LM_FUSED=1, PDT_CTC_LM_SEARCH=1),
              "search": dict(PDT_CTC_LM_TABLE=0, PDT_CTC_LM_FUSED=1, PDT_CTC_LM_SEARCH=1),
              "frame": dict(PDT_CTC_LM_TABLE=0, PDT_CTC_LM_FUSED=1, PDT_CTC_LM_SEARCH=0),
              "three": dict(PDT_CTC_LM_TABLE=0, PDT_CTC_LM_FUSED=0)}
# (search: every iteration of a bigram-table model from ONE launch, the paths read off a trie at the end -- round 5)
BEAM_ROUTES = {"search": dict(PDT_BEAM_FUSED=1, PDT_BEAM_TABLE=1, PDT_BEAM_SEARCH=1),
               "table": dict(PDT_BEAM_FUSED=1, PDT_BEAM_TABLE=1, PDT_BEAM_SEARCH=0),
               "fused": dict(PDT_BEAM_FUSED=1, PDT_BEAM_TABLE=0), "loop": dict(PDT_BEAM_FUSED=0)}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _np(out):
    return tuple(x.cpu().numpy() for x in out)


def _masked(y, yl):
    inside = np.arange(y.shape[0])[:, None, None] < yl[None]
    return np.where(inside, y, 0)


@pytest.mark.parametrize("route", list(CTC_ROUTES))
def test_ctc_prefix_search_lookup_lm_sweep_of_the_reference(route, switch):
    g, tags, models = lm_search_golden()
    lms = {tag: M.LookupLanguageModel(V, sos, [d.copy() for d in dicts]).to(DEV) for tag, (V, sos, N, dicts) in models.items()}
    for name, value in CTC_ROUTES[route].items():
        switch(name, value)
    for i in range(30):
        t = "ctc%d_" % i
        K, beta, vm, ti = g[t + "cfg"]
        lens = None if g[t + "lens"][0] < 0 else _t(g[t + "lens"])
        search = M.CTCPrefixSearch(int(K), float(beta), lms[tags[int(ti)]], valid_mixture=bool(vm))
        y, yl, yp = _np(search(_t(g[t + "logits"]), lens))
        what = (i, tags[int(ti)], route)
        assert np.array_equal(yl, g[t + "y_lens"]) and np.array_equal(_masked(y, yl), g[t + "y"]), what
        assert np.allclose(yp, g[t + "y_probs"], rtol=1e-5, atol=0), what


@pytest.mark.parametrize("route", list(BEAM_ROUTES))
def test_beam_search_lookup_lm_sweep_of_the_reference(route, switch):
    g, tags, models = lm_search_golden()
    lms = {tag: M.LookupLanguageModel(V, sos, [d.copy() for d in dicts]).to(DEV) for tag, (V, sos, N, dicts) in models.items()}
    for name, value in BEAM_ROUTES[route].items():
        switch(name, value)
    for i in range(20):
        t = "beam%d_" % i
        K, eos, fin, N, iters, ti = (int(x) for x in g[t + "cfg"])
        bs = M.BeamSearch(lms[tags[ti]], K, None if eos == -1000 else eos, bool(fin), -7).to(DEV)
        y, yl, lp = _np(bs(dict(), None if N < 0 else N, iters))
        what = (i, tags[ti], route)
        assert same_paths(y, yl, g[t + "y"], g[t + "y_lens"]), what
        assert np.allclose(lp, g[t + "lp"], rtol=1e-5, atol=1e-6), what


@pytest.mark.parametrize("beam_route", ["fused", "loop"])
def test_a_model_with_state_in_both_searches(beam_route, switch):
    """tests/golden/lm_search.npz, the CounterLM cases: the package's loops reorder a user model's state
    (extract_by_src / mix_by_mask) as the reference does -- CTCPrefixSearch's frame loop around any
    model, BeamSearch's fused iterations (host read every iteration for models that are not this
    package's n-gram model) and its step-by-step loop."""
    g, _, _ = lm_search_golden()
    for name, value in BEAM_ROUTES[beam_route].items():
        switch(name, value)
    for i in range(12):
        t = "cnt%d_" % i
        K, beta, vm, eos, iters = g[t + "cfg"]
        lens = None if g[t + "lens"][0] < 0 else _t(g[t + "lens"])
        lm = CounterLM(_t(g[t + "table"]))
        y, yl, yp = _np(M.CTCPrefixSearch(int(K), float(beta), lm, valid_mixture=bool(vm))(_t(g[t + "logits"]), lens))
        assert np.array_equal(yl, g[t + "y_lens"]) and np.array_equal(_masked(y, yl), g[t + "y"]), i
        assert np.allclose(yp, g[t + "y_probs"], rtol=1e-5, atol=0), i
        bs = M.BeamSearch(lm, int(K), None if eos == -1000 else int(eos), bool(i % 2), -9).to(DEV)
        by, byl, blp = _np(bs(dict(), g[t + "logits"].shape[1], int(iters)))
        assert same_paths(by, byl, g[t + "by"], g[t + "by_lens"]), (i, beam_route)
        assert np.allclose(blp, g[t + "blp"], rtol=1e-5, atol=1e-6), (i, beam_route)


@pytest.mark.parametrize("valid_mixture", [False, True])
def test_c3_sample_ctc_search_with_a_trigram_model_against_the_oracle(valid_mixture, switch):
    """The same at order THREE (round 5): the factor-table search over the model's (V + 1)^2 contexts -- a
    4 GB table built by the model's own scoring kernel -- and the frame-kernel routes, against
    oracle.ctc_prefix_search_lm around oracle.NGramLM (the defining back-off recursion) at T=1000, V=1000."""
    import bench

    T, N, V, K, beta = 1000, 5, 1000, 16, 0.2
    dicts = bench.synthetic_trigram_dicts(V)
    lm = M.LookupLanguageModel(V, V, [d.copy() for d in dicts]).to(DEV)
    olm = oracle.NGramLM(V, V, dicts)
    lg = bench.speechlike_logits(T, N, V, torch.device(DEV), 0x5EED0043 + int(valid_mixture), dicts)
    lens = torch.tensor([T, T - 1, 640, T, 3], device=DEV)
    ey, eyl, eyp = oracle.ctc_prefix_search_lm(lg.cpu().numpy(), K, lens.cpu().numpy(), olm, beta, valid_mixture)
    assert eyl.max() > 40 and np.isfinite(eyp).all() and (eyp > 0).all(), (eyl.max(), eyp.min())
    search = M.CTCPrefixSearch(K, beta, lm, valid_mixture=valid_mixture)
    for route in ("table", "search"):
        for name, value in CTC_ROUTES[route].items():
            switch(name, value)
        with torch.no_grad():
            y, yl, yp = _np(search(lg, lens))
        assert np.array_equal(yl, eyl), (route, np.argwhere(yl != eyl)[:5])
        assert np.array_equal(_masked(y, yl)[: ey.shape[0]], ey), (route, np.argwhere(_masked(y, yl)[: ey.shape[0]] != ey)[:5])
        _drift.check_log_probs(yp, eyp, "C3 + trigram LM vs oracle, T=1000 V=1000 K=16, route {}, valid_mixture={}".format(route, valid_mixture))
    from pydrobert_amd import _decoding as _dec

    _dec._FACTOR_TABLES.clear()  # (4 GB a model: not kept for the rest of the suite)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("valid_mixture", [False, True])
def test_c3_sample_ctc_search_with_the_bigram_model_against_the_oracle(valid_mixture, switch):
    """C3's shape per utterance (T=1000, V=1000, K=16; BASELINE config 3 with the shipped n-gram model
    in the loop) for a sample of utterances: every route of the package against oracle.ctc_prefix_search_lm
    around oracle.NGramLM built from the same n-gram tables.  Tokens and lengths exact; a probability
    after 1000 frames is a product of 1000 factors each within an ulp or two, so the tolerance is on the
    log-probabilities (as for the search without a model, tests/test_full_size_gpu.py)."""
    import bench

    T, N, V, K, beta = 1000, 6, 1000, 16, 0.2
    dicts = bench.synthetic_bigram_dicts(V)
    lm = M.LookupLanguageModel(V, V, [d.copy() for d in dicts]).to(DEV)
    olm = oracle.NGramLM(V, V, dicts)
    # (blank-dominated frames, tokens drawn along the model's bigrams: the masses survive 1000 frames)
    lg = bench.speechlike_logits(T, N, V, torch.device(DEV), 0x5EED0033 + int(valid_mixture), dicts)
    lens = torch.tensor([T, T - 1, 700, 513, T, 2], device=DEV)
    ey, eyl, eyp = oracle.ctc_prefix_search_lm(lg.cpu().numpy(), K, lens.cpu().numpy(), olm, beta, valid_mixture)
    assert eyl.max() > 40 and np.isfinite(eyp).all() and (eyp > 0).all(), (eyl.max(), eyp.min())
    search = M.CTCPrefixSearch(K, beta, lm, valid_mixture=valid_mixture)
    for route, sw in CTC_ROUTES.items():
        for name, value in sw.items():
            switch(name, value)
        with torch.no_grad():
            y, yl, yp = _np(search(lg, lens))
        assert np.array_equal(yl, eyl), (route, np.argwhere(yl != eyl)[:5])
        assert np.array_equal(_masked(y, yl)[: ey.shape[0]], ey), (route, np.argwhere(_masked(y, yl)[: ey.shape[0]] != ey)[:5])
        _drift.check_log_probs(yp, eyp, "C3 + bigram LM vs oracle, T=1000 V=1000 K=16, route {}, valid_mixture={}".format(route, valid_mixture))


def test_c3_sample_beam_search_with_the_bigram_model_against_the_oracle(switch):
    """BeamSearch(width 16, eos) over the bigram model, V = 1000, 100 iterations, a sample of batch
    elements: every route against oracle.beam_search around oracle.NGramLM."""
    import bench

    V, K, N, iters = 1000, 16, 5, 100
    dicts = bench.synthetic_bigram_dicts(V)
    lm = M.LookupLanguageModel(V, V, [d.copy() for d in dicts]).to(DEV)
    olm = oracle.NGramLM(V, V, dicts)
    for eos, fin in ((0, False), (None, False), (3, True)):
        ey, eyl, elp = oracle.beam_search(olm, K, eos, fin, -100, None, N, iters)
        bs = M.BeamSearch(lm, K, eos, fin).to(DEV)
        for route, sw in BEAM_ROUTES.items():
            for name, value in sw.items():
                switch(name, value)
            with torch.no_grad():
                y, yl, lp = _np(bs(dict(), N, iters))
            assert same_paths(y, yl, ey, eyl), (eos, fin, route)
            assert np.allclose(lp, elp, rtol=1e-5, atol=1e-5), (eos, fin, route, np.abs(lp - elp).max())
