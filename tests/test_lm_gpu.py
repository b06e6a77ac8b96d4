"""LookupLanguageModel on the GPU: the lookup kernel against the
live reference's scores (tests/golden/lm.npz, bit-exact), against the oracle's two CPU
restatements on tables of every size class, and inside the searches."""
import numpy as np
import pytest
import torch

import oracle
from pydrobert_amd import modules as M

from _lm_fixtures import dicts_from_golden, golden, random_dicts
from _toy_lm import BigramLM

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("tag", ["A", "B", "C", "U"])
def test_scores_match_reference_bit_for_bit(tag):
    g = golden()
    V, sos, N, dicts = dicts_from_golden(g, tag)
    lm = M.LookupLanguageModel(V, sos, dicts).to(DEV)
    hist, idx = _t(g[tag + "_hist"]), _t(g[tag + "_idx"])
    assert np.array_equal(g[tag + "_full"], lm(hist).cpu().numpy(), equal_nan=True)
    assert np.array_equal(g[tag + "_at_idx"], lm(hist, None, idx)[0].cpu().numpy(), equal_nan=True)
    assert np.array_equal(g[tag + "_at_4"], lm(hist, None, 4)[0].cpu().numpy(), equal_nan=True)
    assert np.array_equal(g[tag + "_at_0"], lm(hist[:0], None, 0)[0].cpu().numpy(), equal_nan=True)
    # negative positions count from the end; chunked form is the same computation
    assert torch.equal(lm(hist, None, -1)[0], lm(hist, None, hist.shape[0])[0])
    assert torch.equal(lm.calc_full_log_probs_chunked(hist, dict(), 3), lm(hist))
    # a transposed (strided) history
    ht = hist.t().contiguous().t()
    assert torch.equal(lm(ht), lm(hist))


def test_default_model_is_uniform():
    g = golden()
    lm = M.LookupLanguageModel(7, 2).to(DEV)
    act = lm(torch.zeros((3, 2), dtype=torch.long, device=DEV))
    assert np.allclose(g["D_full"], act.cpu().numpy())


@pytest.mark.parametrize("N", [1, 2, 3, 5])
@pytest.mark.parametrize("sos", [-1, 0])
def test_scores_match_oracle_small_tables(N, sos):
    """Dense random tables in the uint8-offset regime (where the reference's own builder fails
    under NumPy 2): every possible history of every length, as tests/test_lm.py:218-306."""
    rng = np.random.default_rng(100 + N)
    V = 5
    dicts = random_dicts(rng, V, N, 0.5, sos if sos < 0 else None)
    lm = M.LookupLanguageModel(V, sos, [d.copy() for d in dicts]).to(DEV)
    assert N not in (2, 3) or lm.offsets.dtype == torch.uint8
    for n in range(0, N + 1):
        hist = np.array(np.meshgrid(*[np.arange(V)] * n, indexing="ij")).reshape(n, -1) if n else np.zeros((0, 1), int)
        act = lm(_t(hist.astype(np.int64)), None, -1)[0].cpu().numpy()
        brute = oracle.backoff_log_probs(dicts, V, sos, hist, n)
        assert np.allclose(brute, act, atol=1e-5, equal_nan=True), (N, n)
        walk = oracle.trie_log_probs(
            lm.logps.cpu().numpy(), lm.logbs.cpu().numpy(), lm.ids.cpu().numpy(),
            lm.offsets.cpu().numpy(), V, sos, N, hist, n,
        )  # fmt: skip
        assert np.array_equal(walk, act, equal_nan=True), (N, n)


def test_sos_context_known_answer():
    """Facts from the reference's tests/test_lm.py:348-364 (0 = sos)."""
    prob_dicts = [
        {0: (-99, 0.0), 1: (0.1, -0.1), 2: (0.2, -0.2), 3: (0.3, -0.3)},
        {(0, 1): (0.01, -0.01), (0, 2): (0.02, -0.02)},
        {(0, 0, 1): 0.001},
    ]
    lm = M.LookupLanguageModel(4, 0, prob_dicts, destructive=True).to(DEV)
    act = lm(torch.empty((0, 1), device=DEV, dtype=torch.long))
    exp = torch.tensor([[[-99.0, 0.001, 0.02, 0.3]]], device=DEV)
    assert torch.allclose(exp, act, atol=1e-5)


def test_nonuniform_idx_matches_full():
    """tests/test_lm.py:309-345: per-row positions select rows of the full table."""
    rng = np.random.default_rng(7)
    S, N, B, V, sos = 20, 5, 4, 10, -1
    dicts = random_dicts(rng, V, N, 0.02 if N > 3 else 0.5, sos)
    lm = M.LookupLanguageModel(V, sos, dicts, destructive=True).to(DEV)
    hist = torch.randint(0, V, (S, B), device=DEV)
    full = lm(hist)
    assert not torch.isnan(full).any()
    idx = torch.randint(0, S + 1, (B,), device=DEV)
    exp = full.gather(0, idx.view(1, B, 1).expand(1, B, V)).squeeze(0)
    assert torch.equal(exp, lm(hist, idx=idx)[0])
    with pytest.raises(RuntimeError):
        lm(hist, idx=torch.tensor([S + 1] * B, device=DEV))


def test_large_vocabulary_and_wide_types():
    """int32 offsets / int16 ids: a sparse trigram table over 3000 tokens."""
    rng = np.random.default_rng(8)
    V, sos = 3000, 3000
    uni = {v: (float(rng.normal()), float(rng.normal())) for v in range(V)}
    bi = {tuple(int(x) for x in rng.integers(0, V, 2)): (float(rng.normal()), float(rng.normal())) for _ in range(70000)}
    tri = {tuple(int(x) for x in rng.integers(0, V, 3)): float(rng.normal()) for _ in range(5000)}
    for k in list(bi)[:3000]:  # trigrams that extend real bigrams
        tri[(int(rng.integers(0, V)),) + k] = float(rng.normal())
    dicts = [uni, bi, tri]
    lm = M.LookupLanguageModel(V, sos, dicts).to(DEV)
    assert lm.ids.dtype == torch.int16 and lm.offsets.dtype in (torch.int32, torch.int64)
    keys = list(tri)[-40:] + list(bi)[:40]
    hist = np.array([list(k[:2]) if len(k) == 3 else [k[0], k[0]] for k in keys]).T
    act = lm(_t(hist.astype(np.int64)), None, -1)[0].cpu().numpy()
    brute = oracle.backoff_log_probs(dicts, V, sos, hist, 2)
    assert np.allclose(brute, act, atol=1e-4, equal_nan=True)


def test_state_dict_round_trip_on_device():
    g = golden()
    V, sos, N, dicts = dicts_from_golden(g, "C")
    full = M.LookupLanguageModel(V, sos, dicts).to(DEV)
    hist = _t(g["C_hist"])
    exp = full(hist)
    blank = M.LookupLanguageModel(V, sos).to(DEV)
    assert blank(hist).shape == exp.shape  # uniform before loading
    blank.load_state_dict(full.state_dict())
    assert torch.equal(exp, blank(hist))


def test_ctc_prefix_search_with_lookup_lm_fusion():
    """CTCPrefixSearch(width, beta, lm=LookupLanguageModel) against the live reference."""
    g = golden()
    V, sos, N, dicts = dicts_from_golden(g, "B")
    lm = M.LookupLanguageModel(V, sos, dicts).to(DEV)
    search = M.CTCPrefixSearch(int(g["ctc_width"]), float(g["ctc_beta"]), lm)
    y, yl, yp = search(_t(g["ctc_logits"]), _t(g["ctc_lens"]))
    assert torch.equal(yl.cpu(), torch.from_numpy(g["ctc_y_lens"]))
    mask = torch.arange(y.shape[0], device=DEV).view(-1, 1, 1) < yl.unsqueeze(0)
    assert torch.equal(torch.where(mask, y, torch.zeros_like(y)).cpu(), torch.from_numpy(g["ctc_y"]))
    assert np.allclose(g["ctc_y_probs"], yp.cpu().numpy(), rtol=1e-5, atol=1e-30)


@pytest.mark.parametrize("order", [2, 3, 4])
def test_one_kernel_frames_equal_the_three_kernel_route(order, switch):
    """A frame of CTCPrefixSearch with the n-gram model in the loop as ONE kernel
    (csrc/ctc_lm_step.hip: scores, mix, lists, prefix step) against the route through
    lookup_lm_log_probs -> fusion_ext -> ctc_prefix_search_advance (PDT_CTC_LM_FUSED=0): the same
    arithmetic in the same order, so the same bits -- shallow fusion and valid mixture, ragged lens,
    sos in and outside the vocabulary, widths below and above the number of tokens."""
    rng = np.random.default_rng(700 + order)
    big = 40 if order == 2 else 12
    for V, W, T, N, sos, vm, beta in [(9, 4, 14, 5, -1, False, 0.3), (6, 8, 9, 3, 2, True, 0.6),
                                      (big, 16, 25, 4, -1, False, 0.2), (10, 5, 12, 6, 0, True, 0.25)]:
        dicts = random_dicts(rng, V, order, 0.5 if V ** order < 3000 else 0.1, sos if sos < 0 else None)
        for v in range(V):  # (every token a unigram: a vocabulary entry without one scores -inf everywhere)
            dicts[0].setdefault(v, (float(rng.normal()), float(rng.normal())) if order > 1 else float(rng.normal()))
        lm = M.LookupLanguageModel(V, sos, dicts, destructive=True).to(DEV)
        lg = rng.normal(size=(T, N, V + 1)).astype(np.float32)
        np.put_along_axis(lg, rng.integers(0, V + 1, (T, N, 1)), 5.0, 2)
        lens = rng.integers(0, T + 1, N)
        search = M.CTCPrefixSearch(W, beta, lm, valid_mixture=vm)
        switch("PDT_CTC_LM_TABLE", "0")  # (order two would otherwise take the factor-table search for both)
        for ln in (None, torch.from_numpy(lens).to(DEV)):
            switch("PDT_CTC_LM_FUSED", "0")
            ey, eyl, eyp = search(_t(lg), ln)
            switch("PDT_CTC_LM_FUSED", "1")
            assert search._fuses_lookup_lm(_t(lg))
            # one kernel per frame launched by the host loop, then every frame from one call of the
            # library (histories in slots instead of copied from frame to frame)
            for whole in ("0", "1"):
                switch("PDT_CTC_LM_SEARCH", whole)
                y, yl, yp = search(_t(lg), ln)
                what = (order, V, W, vm, ln is None, whole)
                assert torch.isfinite(eyp[:, 0]).all(), what
                assert y.shape == ey.shape and torch.equal(yl, eyl) and torch.equal(yp, eyp), what
                mask = torch.arange(y.shape[0], device=DEV).view(-1, 1, 1) < yl.unsqueeze(0)
                assert torch.equal(torch.where(mask, y, ey), ey), what


def test_beam_search_with_lookup_lm():
    g = golden()
    V, sos, N, dicts = dicts_from_golden(g, "C")
    lm = M.LookupLanguageModel(V, sos, dicts).to(DEV)
    y, yl, lp = M.BeamSearch(lm, 4, eos=0).to(DEV)(dict(), batch_size=3, max_iters=10)
    assert torch.equal(yl.cpu(), torch.from_numpy(g["beam_lens"]))
    assert np.allclose(g["beam_lp"], lp.cpu().numpy(), rtol=1e-5, atol=1e-6)
    S = y.shape[0]
    mask = torch.arange(S, device=DEV).view(-1, 1, 1) < yl.unsqueeze(0)
    exp = torch.from_numpy(g["beam_y"]).to(DEV)[:S]
    assert torch.equal(torch.where(mask, y, exp), exp)


@pytest.mark.parametrize("sos", [-1, 0, 3])
def test_beam_search_reads_a_bigram_models_table(sos, switch):
    """BeamSearch over a bigram LookupLanguageModel reads its prefixes' scores (and their log-softmax
    statistics) from the model's dense (context, token) table, built once, instead of having the model
    write (N K, V) scores every iteration: the same paths, lengths and log-probabilities -- to the bit --
    as the fused iterations around the model's own forward (PDT_BEAM_TABLE=0) and as the step-by-step
    loop; a changed model gets a new table."""
    rng = np.random.default_rng(4100 + sos)
    # (vocabularies above 64 tokens: the flat selection of beam_step_flat_kernel, round 5 -- also against the
    # sorted-list-per-prefix form of the same route, PDT_STEP_FLAT=0)
    for V, W, N, eos, iters in [(7, 3, 4, 0, 12), (12, 8, 3, None, 9), (40, 16, 5, 5, 20), (5, 5, 2, 1, 6),
                                (130, 16, 4, 3, 25), (70, 6, 3, None, 10), (300, 40, 3, 7, 30)]:
        dicts = random_dicts(rng, V, 2, 0.5, sos if sos < 0 else None)
        for v in range(V):
            dicts[0].setdefault(v, (float(rng.normal()), float(rng.normal())))
        lm = M.LookupLanguageModel(V, sos if sos < V else 0, dicts, destructive=True).to(DEV)
        search = M.BeamSearch(lm, W, eos=eos).to(DEV)
        outs = []
        switch("PDT_BEAM_SEARCH", 0)
        for fused, table in (("1", "1"), ("1", "0"), ("0", "0")):
            switch("PDT_BEAM_FUSED", fused)
            switch("PDT_BEAM_TABLE", table)
            outs.append(search(dict(), batch_size=N, max_iters=iters))
        (y, yl, lp), (y1, yl1, lp1), (y2, yl2, lp2) = outs
        what = (sos, V, W, N, eos)
        # every iteration from one launch, the paths off a trie (PDT_BEAM_SEARCH, round 5; rows of more than 64
        # tokens -- shorter ones keep the iterations): the same tensors, rows beyond the lengths included
        switch("PDT_BEAM_FUSED", "1")
        switch("PDT_BEAM_TABLE", "1")
        switch("PDT_BEAM_SEARCH", 1)
        for bs_, it_ in ((N, iters), (N, 1), (None, iters), (N, 3 * iters)):
            ys, yls, lps = search(dict(), batch_size=bs_, max_iters=it_)
            switch("PDT_BEAM_SEARCH", 0)
            yt, ylt, lpt = search(dict(), batch_size=bs_, max_iters=it_)
            switch("PDT_BEAM_SEARCH", 1)
            assert ys.shape == yt.shape and torch.equal(ys, yt) and torch.equal(yls, ylt) and torch.equal(lps, lpt), (what, bs_, it_)
        switch("PDT_BEAM_SEARCH", 0)
        if V > 64:
            switch("PDT_BEAM_FUSED", "1")
            switch("PDT_BEAM_TABLE", "1")
            switch("PDT_STEP_FLAT", 0)
            y3, yl3, lp3 = search(dict(), batch_size=N, max_iters=iters)
            switch("PDT_STEP_FLAT", 1)
            assert torch.equal(y, y3) and torch.equal(yl, yl3) and torch.equal(lp, lp3), what
        assert y.shape == y1.shape and torch.equal(y, y1) and torch.equal(yl, yl1) and torch.equal(lp, lp1), what
        assert torch.equal(yl, yl2) and torch.allclose(lp, lp2, rtol=1e-5, atol=1e-6), what
        switch("PDT_BEAM_FUSED", "1")
        switch("PDT_BEAM_TABLE", "1")
        assert search._bigram_table(search.device_buffer.device) is not None
        with torch.no_grad():
            lm.logps.add_(0.25 * torch.randn_like(lm.logps))  # (version counter moves: the table is rebuilt)
        ya, yla, lpa = search(dict(), batch_size=N, max_iters=iters)
        switch("PDT_BEAM_TABLE", "0")
        yb, ylb, lpb = search(dict(), batch_size=N, max_iters=iters)
        assert torch.equal(ya, yb) and torch.equal(yla, ylb) and torch.equal(lpa, lpb), what
        switch("PDT_BEAM_SEARCH", 1)


def test_beam_search_every_iteration_in_one_launch(switch):
    """`pdt_beam_search_table` (round 5): a bigram-table model's whole search from ONE launch -- the beam in
    LDS, a (source, token) word per entry and iteration in a trie, the paths read off it at the end -- against
    the iteration-per-launch table route: the same `y` (rows beyond the lengths and the padding included),
    lengths and log-probabilities, to the bit.  Rows of 65 .. 1024 tokens with beams up to 16 take the
    rows-per-wave form, wider beams and longer rows the chunk-numbered one; eos biased so that batch elements
    finish at different iterations (and some never do)."""
    rng = np.random.default_rng(5150)
    switch("PDT_BEAM_FUSED", "1")
    switch("PDT_BEAM_TABLE", "1")
    for case in range(24):
        V = int(rng.choice([65, 66, 100, 128, 129, 300, 640, 1000, 1024, 1025, 1500]))
        w_max = min(64, 256 // ((V + 63) // 64))  # (beyond: the iterations, by PDT_E_UNSUPPORTED)
        W = int(rng.integers(1, min(16, w_max) + 1)) if case % 3 or w_max < 17 else int(rng.integers(17, w_max + 1))
        N, iters = int(rng.integers(1, 7)), int(rng.integers(1, 40))
        eos = None if case % 5 == 0 else int(rng.integers(0, V))
        fin = bool(case % 2)
        sos = -1 if case % 4 == 0 else int(rng.integers(0, V))
        dicts = random_dicts(rng, V, 2, 4.0 / V, sos if sos < 0 else None)
        for v in range(V):
            dicts[0].setdefault(v, (float(rng.normal()), float(rng.normal())))
        if eos is not None:  # a likely eos: paths end at different iterations
            lp_, bo_ = dicts[0][eos]
            dicts[0][eos] = (lp_ + float(rng.uniform(2.0, 6.0)), bo_)
        lm = M.LookupLanguageModel(V, sos, dicts, destructive=True).to(DEV)
        search = M.BeamSearch(lm, W, eos=eos, finish_all_paths=fin, pad_value=int(rng.integers(-9, 3))).to(DEV)
        outs = []
        for one_launch in (1, 0):
            switch("PDT_BEAM_SEARCH", one_launch)
            outs.append(search(dict(), batch_size=N, max_iters=iters))
        (y, yl, lp), (y0, yl0, lp0) = outs
        what = (case, V, W, N, iters, eos, fin, sos)
        assert y.shape == y0.shape, (what, y.shape, y0.shape)
        assert torch.equal(y, y0) and torch.equal(yl, yl0) and torch.equal(lp, lp0), what
    switch("PDT_BEAM_SEARCH", 1)


def test_shallow_fusion_of_two_models():
    """tests/test_lm.py:512-560: fused scores = first + beta * second, states kept apart."""
    g = golden()
    V, sos, N, dicts = dicts_from_golden(g, "B")
    first = M.LookupLanguageModel(V, sos, dicts).to(DEV)
    table = torch.randn((V + 1, V), device=DEV, generator=torch.Generator(DEV).manual_seed(3)).log_softmax(-1)
    second = BigramLM(table)
    beta = 0.7
    lm = M.MixableShallowFusionLanguageModel(first, second, beta)
    hist = _t(g["B_hist"]).clamp(0, V - 1)
    exp = first(hist) + beta * second(hist)
    assert torch.allclose(exp, lm(hist))
    idx = torch.tensor(5, device=DEV)
    lp, state = lm(hist, None, idx)
    assert torch.allclose(exp[5], lp)
    assert lm.extract_by_src(state, torch.arange(hist.shape[1], device=DEV)) == state
    # drives a search like any other LM
    y, yl, yp = M.CTCPrefixSearch(3, 0.2, lm)(torch.randn((6, 2, V + 1), device=DEV))
    assert y.shape == (6, 2, 3) and torch.isfinite(yp).all()


def test_scripted_lookup_lm():
    """"JIT scripting is possible with this module, but not tracing" (reference _lm.py:568)."""
    g = golden()
    V, sos, N, dicts = dicts_from_golden(g, "C")
    lm = M.LookupLanguageModel(V, sos, dicts).to(DEV)
    scripted = torch.jit.script(lm)
    hist, idx = _t(g["C_hist"]), _t(g["C_idx"])
    assert torch.equal(lm(hist), scripted(hist))
    assert torch.equal(lm(hist, None, idx)[0], scripted(hist, None, idx)[0])
    search = M.CTCPrefixSearch(4, 0.3, lm)
    logits = torch.randn((7, 3, V + 1), device=DEV, generator=torch.Generator(DEV).manual_seed(5))
    # (eager: the factor-table search with its fused softmax, for every order whose contexts fit a table since
    # round 5; scripted: the frame loop on torch's softmax -- the same beams, probabilities to the last ulps)
    exp, act = search(logits), torch.jit.script(search)(logits)
    assert torch.equal(exp[0], act[0]) and torch.equal(exp[1], act[1])
    assert torch.allclose(exp[2], act[2], rtol=1e-5, atol=0)


@pytest.mark.parametrize("V", [7, 512, 513, 1024, 1500])
def test_fusion_ext_matches_the_torch_composition(V):
    """pdt_fusion_ext against the expressions CTCPrefixSearch composes when a gradient is wanted
    (reference _decoding.py:1120-1135): rows in 8 / 16 registers per lane and streamed rows, a
    strided view of the frame's probabilities."""
    g = torch.Generator(device=DEV).manual_seed(V)
    N, Kp = 5, 3
    probs = torch.randn((N, V + 1), device=DEV, generator=g).softmax(1)
    nonext, blank = probs[:, :V], probs[:, V]
    lm_lp = torch.randn((N * Kp, V), device=DEV, generator=g) * 4
    beta = 0.37
    ext = nonext.unsqueeze(1).expand(N, Kp, V)
    exp_shallow = ext * (beta * lm_lp.log_softmax(-1)).exp().view(N, Kp, V)
    exp_mix = (1.0 - beta) * ext + beta * (lm_lp.softmax(-1).view(N, Kp, V) * (1 - blank.view(N, 1, 1)))
    act_shallow = torch.ops.pydrobert_amd.fusion_ext(lm_lp, nonext, blank, beta, False)
    act_mix = torch.ops.pydrobert_amd.fusion_ext(lm_lp, nonext, blank, beta, True)
    assert torch.allclose(act_shallow, exp_shallow, rtol=2e-5, atol=1e-12)
    assert torch.allclose(act_mix, exp_mix, rtol=2e-5, atol=1e-12)
