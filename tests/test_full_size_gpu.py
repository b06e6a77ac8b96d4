"""Every BASELINE.json config at its real shape on the GPU (C2 ... C5).

The oracle cannot run these sizes whole in seconds, so each test compares a SAMPLE of
utterances with the oracle (utterances are independent in every operator; the C oracle releases
the GIL, so the sample is spread over the host's cores) and checks size-independent properties
on ALL of them.  Bit-exact for tokens / lengths / indices / DP values, 1e-5 relative for
probabilities, 1e-4 absolute for warped features.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

import _drift
import oracle
from pydrobert_amd import functional as F

pytestmark = pytest.mark.gpu
RTOL = 1e-5
WORKERS = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4))


def _peaky(T, N, V, device, seed, scale=12.0, chunk=64):
    """N(0,1) + 12 on one class per frame (blank included)."""
    g = torch.Generator(device=device).manual_seed(seed)
    lg = torch.empty((T, N, V + 1), device=device)
    for t0 in range(0, T, chunk):
        part = lg[t0 : t0 + chunk]
        part.normal_(generator=g)
        peak = torch.randint(0, V + 1, (part.shape[0], N, 1), device=device, generator=g)
        part.scatter_add_(2, peak, torch.full((part.shape[0], N, 1), scale, device=device))
    return lg


def _oracle_search(lg, K, lens=None):
    """oracle.ctc_prefix_search over the utterances of lg (T, n, V + 1), one thread each."""
    n = lg.shape[1]

    def one(i):
        return oracle.ctc_prefix_search(
            np.ascontiguousarray(lg[:, i : i + 1]), K, None if lens is None else lens[i : i + 1]
        )

    with ThreadPoolExecutor(WORKERS) as ex:
        parts = list(ex.map(one, range(n)))
    S = max(p[0].shape[0] for p in parts)
    y = np.zeros((S, n, K), np.int64)
    for i, p in enumerate(parts):
        y[: p[0].shape[0], i] = p[0][:, 0]
    return y, np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])


def _check_search_sample(y, yl, yp, exp, config):
    """Tokens and lengths exact.  A prefix probability at these sizes is a product over 500-1000
    frames, every factor within an ulp or two of the reference's (reciprocal of the normaliser,
    exp); the north star states the tolerance on LOG-probabilities: asserted as an ABSOLUTE bound
    (``_drift.LOG_ATOL`` = 3e-5 on values of magnitude 20-90, i.e. < 1e-6 relative), and the measured
    maximum of every configuration is recorded (tests/_drift.py -> profiles/r05_logprob_drift.json)."""
    ey, eyl, eyp = exp
    assert np.array_equal(yl, eyl), np.argwhere(yl != eyl)[:5]
    assert np.array_equal(y[: ey.shape[0]], ey), np.argwhere(y[: ey.shape[0]] != ey)[:5]
    _drift.check_log_probs(yp, eyp, config)


def _check_search_properties(y, yl, yp, T, V, lens=None):
    """On every utterance: lengths within the frames seen, probabilities finite, positive and
    descending, prefixes of one beam pairwise distinct, tokens in range, zeros past the length."""
    S, N, K = y.shape
    assert yl.shape == (N, K) and yp.shape == (N, K)
    cap = torch.full((N, 1), T, device=y.device) if lens is None else lens.clamp(max=T).unsqueeze(1)
    assert bool((yl >= 0).all()) and bool((yl <= cap).all())
    assert bool(torch.isfinite(yp).all()) and bool((yp > 0).all()) and bool((yp <= 1.0 + 1e-5).all())
    assert bool((yp[:, :-1] >= yp[:, 1:]).all())
    pos = torch.arange(S, device=y.device).view(S, 1, 1)
    inside = pos < yl.unsqueeze(0)
    assert bool((y[~inside] == 0).all())
    assert bool(((y >= 0) & (y < V))[inside].all())
    # distinct prefixes: a polynomial hash of the valid tokens plus the length must differ
    # pairwise inside a beam (collisions of a 61-bit hash on 16 entries are out of the question)
    mult = torch.randint(1, 2**31 - 1, (S, 1, 1), device=y.device, dtype=torch.long)
    h = ((y + 1) * inside * mult).sum(0) % ((1 << 61) - 1) + (yl << 40)
    hs = h.sort(1)[0]
    assert bool((hs[:, 1:] != hs[:, :-1]).all())


# ------------------------------------------------------------------------------------------
# C2: optimal_completion over all 4096 utterances
# ------------------------------------------------------------------------------------------
def test_c2_optimal_completion_all_utterances(device):
    N, T, V = 4096, 512, 256
    g = torch.Generator(device="cpu").manual_seed(0x5EED0002)
    ref = torch.randint(0, V, (T, N), generator=g).to(device)
    hyp = torch.randint(0, V, (T, N), generator=g).to(device)
    oc = F.optimal_completion(ref, hyp, warn=False)
    H1, N_, C = oc.shape
    assert (H1, N_) == (T + 1, N) and C >= 1
    valid = oc != -100
    # every set is non-empty (there is always a way to go on), ascending, then padding only
    assert bool(valid[:, :, 0].all())
    assert bool((valid[:, :, 1:] <= valid[:, :, :-1]).all())
    asc = (oc[:, :, 1:] > oc[:, :, :-1]) | ~valid[:, :, 1:]
    assert bool(asc.all())
    # the empty prefix completes with the first reference token only
    assert torch.equal(oc[0, :, 0], ref[0]) and bool((oc[0, :, 1:] == -100).all())
    # the width is the largest set
    assert int(valid.sum(2).max()) == C
    # every target occurs in its reference (checked through a per-utterance token table)
    table = torch.zeros((N, V), dtype=torch.bool, device=device)
    table.scatter_(1, ref.t(), True)
    flat = oc.clamp(min=0)
    seen = table.unsqueeze(0).expand(H1, N, V).gather(2, flat)
    assert bool((seen | ~valid).all())
    # 64 utterances spread over the batch against the oracle
    idx = torch.arange(0, N, N // 64)
    exp = oracle.optimal_completion(ref[:, idx].cpu().numpy(), hyp[:, idx].cpu().numpy(), faithful=False)
    got = oc[:, idx].cpu().numpy()
    Ce = exp.shape[2]
    assert np.array_equal(exp, got[:, :, :Ce]) and (got[:, :, Ce:] == -100).all()


# ------------------------------------------------------------------------------------------
# C3: the fused search and both step functions at N=1024, V=1000, K=16
# ------------------------------------------------------------------------------------------
def test_c3_ctc_prefix_search(device):
    T, N, V, K = 1000, 1024, 1000, 16
    lg = _peaky(T, N, V, device, 0x5EED0003)
    y, yl, yp = F.ctc_prefix_search(lg, K)
    assert y.shape == (T, N, K)
    _check_search_properties(y, yl, yp, T, V)
    idx = torch.arange(0, N, N // 16)
    exp = _oracle_search(lg[:, idx].cpu().numpy(), K)
    _check_search_sample(y[:, idx].cpu().numpy(), yl[idx].cpu().numpy(), yp[idx].cpu().numpy(), exp, "C3_search T=1000 V=1000 K=16 (16 utterances)")
    # ragged lengths: the same logits, every utterance cut somewhere
    g = torch.Generator(device=device).manual_seed(5)
    lens = torch.randint(T // 2, T + 1, (N,), device=device, generator=g)
    lens[0], lens[1] = T, 0
    y2, yl2, yp2 = F.ctc_prefix_search(lg, K, lens)
    assert y2.shape[0] == int(lens.max())
    live = lens > 0
    _check_search_properties(y2[:, live], yl2[live], yp2[live], T, V, lens[live])
    full = lens == T
    assert torch.equal(y2[:, full], y[: y2.shape[0], full]) and torch.equal(yp2[full], yp[full])
    # the empty utterance: one empty prefix of probability 1, the rest of the beam invalid
    assert bool((yl2[1] == 0).all()) and float(yp2[1, 0]) == 1.0 and bool(torch.isinf(yp2[1, 1:]).all())
    idx = torch.arange(2, N, N // 8)
    exp = _oracle_search(lg[:, idx].cpu().numpy(), K, lens[idx].cpu().numpy())
    _check_search_sample(y2[:, idx].cpu().numpy(), yl2[idx].cpu().numpy(), yp2[idx].cpu().numpy(), exp, "C3_search ragged lens (8 utterances)")


def _cmp_ctc_step(act, exp, what):
    y, last, lens, (nb, b), isp, src, non = [
        tuple(z.cpu().numpy() for z in x) if isinstance(x, tuple) else x.cpu().numpy() for x in act
    ]
    ey, elast, elens, (enb, eb), eisp, esrc, enon = exp
    assert y.shape == ey.shape, what
    assert np.array_equal(src, esrc), (what, np.argwhere(src != esrc)[:5])
    assert np.array_equal(lens, elens) and np.array_equal(non, enon) and np.array_equal(last, elast), what
    assert np.array_equal(isp, eisp), (what, np.argwhere(isp != eisp)[:5])
    ok = np.isfinite(enb)
    assert np.array_equal(np.isfinite(nb), ok), what
    assert np.allclose(nb[ok], enb[ok], rtol=RTOL, atol=0) and np.allclose(b[ok], eb[ok], rtol=RTOL, atol=0), what
    inside = np.arange(ey.shape[0])[:, None, None] < elens[None]
    assert np.array_equal(np.where(inside, y, 0), np.where(inside, ey, 0)), what


def test_c3_ctc_prefix_search_advance(device):
    """The step function at N=1024, K=16, V=1000 driven for S=100 frames of history (the shape
    SURVEY section 8(a) times); at a few frames -- the last one included -- all seven outputs of
    a slice of 32 utterances are compared with the oracle fed with the same state."""
    N, V, K, S = 1024, 1000, 16, 100
    lg = _peaky(S + 1, N, V, device, 0x5EED0013)
    nb, b = torch.zeros((N, 1), device=device), torch.ones((N, 1), device=device)
    y = torch.zeros((0, N, 1), dtype=torch.long, device=device)
    last = lens = torch.zeros((N, 1), dtype=torch.long, device=device)
    isp = torch.ones((N, 1, 1), dtype=torch.bool, device=device)
    sl = slice(500, 532)
    for t in range(S + 1):
        p = lg[t].softmax(1)
        nonext, blank = p[:, :V].contiguous(), p[:, V].contiguous()
        ext = nonext.unsqueeze(1).expand(N, nb.shape[1], V)
        act = F.ctc_prefix_search_advance((ext, nonext, blank), K, (nb, b), y, last, lens, isp)
        if t in (0, 1, 2, 37, S - 1, S):
            c = lambda x: x.cpu().numpy()  # noqa: E731
            exp = oracle.ctc_prefix_search_advance(
                (c(ext[sl].contiguous()), c(nonext[sl]), c(blank[sl])), K, (c(nb[sl]), c(b[sl])),
                c(y[:, sl]), c(last[sl]), c(lens[sl]), c(isp[sl]),
            )
            got = (act[0][:, sl], act[1][sl], act[2][sl], (act[3][0][sl], act[3][1][sl]), act[4][sl], act[5][sl],
                   act[6][sl])
            _cmp_ctc_step(got, exp, ("frame", t))
        y, last, lens, (nb, b), isp = act[0], act[1], act[2], act[3], act[4]
    assert y.shape == (S + 1, N, K)
    assert bool(torch.isfinite(nb + b).all()) and bool((lens <= S + 1).all())
    # the step-by-step search and the fused kernel agree on the final beam
    yf, ylf, ypf = F.ctc_prefix_search(lg, K)
    assert torch.equal(ylf, lens)
    assert torch.allclose(ypf, nb + b, rtol=RTOL, atol=0)
    inside = torch.arange(S + 1, device=device).view(-1, 1, 1) < lens.unsqueeze(0)
    assert torch.equal(yf * inside, y * inside)


def test_c3_beam_search_advance(device):
    N, V, K, S = 1024, 1000, 16, 100
    g = torch.Generator(device=device).manual_seed(0x5EED0023)
    lpt = torch.randn((N, K, V), device=device, generator=g).log_softmax(-1)
    lpp = torch.randn((N, K), device=device, generator=g)
    yp = torch.randint(0, V, (S, N, K), device=device, generator=g)
    for ypl in (None, torch.full((N, K), S, device=device), torch.randint(0, S, (N, K), device=device, generator=g)):
        act = F.beam_search_advance(lpt, K, lpp, yp, ypl)
        exp = oracle.beam_search_advance(lpt.cpu().numpy(), K, lpp.cpu().numpy(), yp.cpu().numpy(),
                                         None if ypl is None else ypl.cpu().numpy())  # fmt: skip
        y, yl, lp, src = (x.cpu().numpy() for x in act)
        assert y.shape == exp[0].shape
        assert np.array_equal(src, exp[3]) and np.array_equal(yl, exp[1])
        assert np.array_equal(lp, exp[2])  # one float32 add per candidate: bit-identical
        inside = np.arange(y.shape[0])[:, None, None] < exp[1][None]
        assert np.array_equal(np.where(inside, y, 0), np.where(inside, exp[0], 0))


# ------------------------------------------------------------------------------------------
# C4: SpecAugment and sparse_image_warp on 2048 x 1000 x 80
# ------------------------------------------------------------------------------------------
def test_c3_searches_with_the_bigram_model_in_the_loop(device, switch):
    """C3 shapes (N=1024, T=1000, V=1000, K=16) with a bigram LookupLanguageModel, speech-like logits
    (every prefix mass stays positive).  The search with the model's factor table (one launch, fused
    softmax: the default) against the search from one library call that takes torch's softmax (history
    slots, factor rows scored in the kernel): the same prefixes and lengths, probabilities to 1e-5 on
    the logarithm, except where two masses of a beam sit within 1e-5 of each other (the two softmaxes
    differ in the last ulp) -- at most a handful of utterances in 1024.  That second route equals the
    host's frame loop around the same frame kernel to the bit -- ragged lengths, both mixes -- and
    BeamSearch reading the model's dense table equals BeamSearch with the model scoring every prefix."""
    import bench
    from pydrobert_amd import modules as M

    T, N, V, K = 1000, 1024, 1000, 16
    dicts = bench.synthetic_bigram_dicts(V)
    lg = bench.speechlike_logits(T, N, V, device, 0x5EED0003, dicts)
    lm = M.LookupLanguageModel(V, V, [d.copy() for d in dicts]).to(device)
    lens = torch.randint(T // 2, T + 1, (N,), device=device, generator=torch.Generator(device=device).manual_seed(9))
    with torch.no_grad():
        for vm, ln in ((False, lens), (True, None)):
            search = M.CTCPrefixSearch(K, 0.2, lm, valid_mixture=vm)
            switch("PDT_CTC_LM_TABLE", "1")
            ty, tyl, typ = search(lg, ln)
            switch("PDT_CTC_LM_TABLE", "0")
            switch("PDT_CTC_LM_SEARCH", "1")
            y, yl, yp = search(lg, ln)
            switch("PDT_CTC_LM_SEARCH", "0")
            ey, eyl, eyp = search(lg, ln)
            mask = torch.arange(y.shape[0], device=device).view(-1, 1, 1) < yl.unsqueeze(0)
            assert torch.equal(yl, eyl) and torch.equal(yp, eyp) and torch.equal(torch.where(mask, y, ey), ey), vm
            assert bool((yp > 0).all()) and bool((typ > 0).all())
            tmask = torch.arange(ty.shape[0], device=device).view(-1, 1, 1) < tyl.unsqueeze(0)
            same = (tyl == yl).all(1) & (torch.where(tmask, ty, 0) == torch.where(mask, y, 0)).all(0).all(1)
            assert int((~same).sum()) <= 8, (vm, int((~same).sum()))
            _drift.check_log_probs(typ[same].cpu().numpy(), yp[same].cpu().numpy(),
                                   "C3 + bigram LM, table route vs step route, valid_mixture={} (all 1024 utterances)".format(vm))
            # where the beams differ they hold near-tied masses: the sorted masses still agree
            sa, sb = typ[~same].sort(1)[0].double().log(), yp[~same].sort(1)[0].double().log()
            assert bool(((sa - sb).abs() <= 1e-4 * sb.abs().clamp(min=1.0)).all()), vm
        bs = M.BeamSearch(lm, K, eos=0).to(device)
        switch("PDT_BEAM_TABLE", "1")
        a = bs(dict(), batch_size=N, max_iters=100)
        switch("PDT_BEAM_TABLE", "0")
        b = bs(dict(), batch_size=N, max_iters=100)
        assert all(torch.equal(x, z) for x, z in zip(a, b))


def test_c4_spec_augment(device):
    from pydrobert_amd import modules as M

    N, T, Fq = 2048, 1000, 80
    g = torch.Generator(device=device).manual_seed(0x5EED0004)
    feats = torch.randn((N, T, Fq), device=device, generator=g)
    lens = torch.randint(500, T + 1, (N,), device=device, generator=g)
    sa = M.SpecAugment(max_time_warp=80.0, max_freq_warp=0.0, max_time_mask=100, max_freq_mask=27,
                       max_time_mask_proportion=0.04, num_time_mask=2, num_time_mask_proportion=1.0,
                       num_freq_mask=2, interpolation_order=1)  # fmt: skip
    torch.manual_seed(4)
    params = sa.draw_parameters(feats, lens)
    w_0, w, v_0, v, t_0, t, f_0, f = params
    assert t.shape == (N, 2) and f.shape == (N, 2)
    assert bool((t <= (lens.float() * 0.04).floor().unsqueeze(1)).all()) and bool((f <= 27).all())
    out = sa.apply_parameters(feats, params, lens)
    assert out.shape == feats.shape and bool(torch.isfinite(out).all())
    # masked bands are exactly zero on every utterance
    tt = torch.arange(T, device=device).view(1, T, 1)
    ff = torch.arange(Fq, device=device).view(1, Fq, 1)
    tmask = ((tt >= t_0.unsqueeze(1)) & (tt < (t_0 + t).unsqueeze(1))).any(2)  # (N, T)
    fmask = ((ff >= f_0.unsqueeze(1)) & (ff < (f_0 + f).unsqueeze(1))).any(2)  # (N, F)
    band = (tmask.unsqueeze(2) | fmask.unsqueeze(1)) & (tt < lens.view(N, 1, 1))
    assert bool((out[band] == 0).all())
    # a time warp interpolates between two frames: every value stays inside the column's range
    lo, hi = feats.amin(1, keepdim=True), feats.amax(1, keepdim=True)
    inside = tt < lens.view(N, 1, 1)
    assert bool((((out >= lo - 1e-5) & (out <= hi + 1e-5)) | ~inside).all())
    # zero warp and no masks is the identity on the valid frames
    z = torch.zeros_like(w)
    e = torch.zeros((N, 0), dtype=torch.long, device=device)
    same = F.spec_augment_apply_parameters(feats, (w_0, z, v_0, v, e, e, e, e), 1, lens)
    # (the identity grid comes out of a float32 spline: positions are exact to ~1e-4 frames)
    assert float(((same - feats) * inside).abs().max()) < 2e-3
    # sampled utterances against the float64 oracle
    idx = torch.arange(0, N, N // 16)
    c = lambda x: x[idx].cpu().numpy()  # noqa: E731
    exp = oracle.spec_augment_apply_parameters(c(feats), tuple(c(p) if p.numel() else p.cpu().numpy() for p in params),
                                               1, c(lens))  # fmt: skip
    valid = np.arange(T)[None, :, None] < c(lens)[:, None, None]
    # (float32 warp grid at T = 1000: positions are exact to ~1e-4 frames, values to ~5e-4;
    # the reference's own float32 graph sits 2e-2 from the float64 oracle here, see
    # test_c4_reference_sample)
    assert (np.abs(exp - c(out)) * valid).max() < 1e-3


def test_c4_sparse_image_warp(device):
    N, T, Fq = 2048, 1000, 80
    g = torch.Generator(device=device).manual_seed(0x5EED0014)
    img = torch.randn((N, 1, T, Fq), device=device, generator=g)
    src = torch.rand((N, 3, 2), device=device, generator=g) * torch.tensor([T - 1.0, Fq - 1.0], device=device)
    dst = src + torch.randn((N, 3, 2), device=device, generator=g)
    out = F.sparse_image_warp(img, src, dst, pinned_boundary_points=1, field_interpolation_order=2, include_flow=False)
    assert out.shape == img.shape and bool(torch.isfinite(out).all())
    # bilinear gather with border padding: values stay within the image's range
    assert float(out.amax()) <= float(img.amax()) + 1e-5 and float(out.amin()) >= float(img.amin()) - 1e-5
    # control points that do not move: identity
    # (positions come out of a float32 spline over coordinates up to 1000: ~3e-3 pixels off)
    same = F.sparse_image_warp(img, src, src, pinned_boundary_points=1, field_interpolation_order=2, include_flow=False)
    assert float((same - img).abs().max()) < 5e-2
    idx = torch.arange(0, N, N // 4)
    c = lambda x: x[idx].cpu().numpy()  # noqa: E731
    exp = oracle.sparse_image_warp(c(img), c(src), c(dst), "hw", 2, pinned_boundary_points=1, include_flow=False)
    # (float32 spline over coordinates up to 1000, like the reference's: sampling positions are
    # good to ~2e-3 pixels, values of N(0, 1) data to ~1e-2; see test_c4_reference_sample)
    assert np.abs(exp - c(out)).max() < 2e-2


def test_c4_reference_sample(device):
    """Two utterances of C4's frame count through the LIVE reference (tests/golden/c4_sample.npz,
    made by tests/golden/make_golden.py).  At T = 1000 the reference's float32 spline graph
    (mm-based cdist, float32 solve) is itself 2e-2 (SpecAugment) / 7e-4 (sparse warp) away from
    the float64 oracle; the kernels evaluate the grid in float64 and sample in float32, so they
    must agree with the reference within ITS error and with the oracle far more closely."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c4_sample.npz"))
    feats = g["feats"].astype(np.float32)
    lens, T = g["lens"], feats.shape[1]
    params = tuple(g["p%d" % i] for i in range(8))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    act = F.spec_augment_apply_parameters(t(feats), tuple(t(p) for p in params), 1, t(lens)).cpu().numpy()
    valid = np.arange(T)[None, :, None] < lens[:, None, None]
    exact = oracle.spec_augment_apply_parameters(feats, params, 1, lens)
    err_ref = (np.abs(act - g["sa_o1"]) * valid).max()
    err_exact = (np.abs(act - exact) * valid).max()
    ref_exact = (np.abs(g["sa_o1"] - exact) * valid).max()
    assert err_exact < 1e-3 and err_ref < 3e-2 and err_exact < ref_exact, (err_exact, err_ref, ref_exact)
    img = feats[:, None]
    act = F.sparse_image_warp(t(img), t(g["siw_src"]), t(g["siw_dst"]), pinned_boundary_points=1,
                              field_interpolation_order=2, include_flow=False).cpu().numpy()  # fmt: skip
    exact = oracle.sparse_image_warp(img, g["siw_src"], g["siw_dst"], "hw", 2, pinned_boundary_points=1,
                                     include_flow=False)  # fmt: skip
    err_ref, err_exact = np.abs(act - g["siw"]).max(), np.abs(act - exact).max()
    assert err_exact < 2e-3 and err_ref < 2e-3, (err_exact, err_ref)


# ------------------------------------------------------------------------------------------
# C5: one GPU's shard -- N=4096, T=512, V=5000: error_rate + decode
# ------------------------------------------------------------------------------------------
def test_c5_shard(device):
    T, N, V, K = 512, 4096, 5000, 16
    rng = np.random.default_rng(0x5EED0005)
    ref = torch.from_numpy(rng.integers(0, V, (T, N))).to(device)
    hyp = torch.from_numpy(rng.integers(0, V, (T, N))).to(device)
    er = F.error_rate(ref, hyp, warn=False)
    idx = torch.arange(0, N, 8)
    exp = oracle.error_rate(ref[:, idx].cpu().numpy(), hyp[:, idx].cpu().numpy(), faithful=False)
    assert np.array_equal(exp, er[idx].cpu().numpy())
    assert torch.equal(F.error_rate(hyp, ref, warn=False), er) and bool((er <= 1.0).all()) and bool((er > 0.5).all())
    lg = _peaky(T, N, V, device, 0x5EED0006)  # 42 GB, generated on the device
    y, yl, yp = F.ctc_prefix_search(lg, K)
    assert y.shape == (T, N, K)
    _check_search_properties(y, yl, yp, T, V)
    idx = torch.arange(0, N, N // 64)
    exp = _oracle_search(lg[:, idx].cpu().numpy(), K)
    _check_search_sample(y[:, idx].cpu().numpy(), yl[idx].cpu().numpy(), yp[idx].cpu().numpy(), exp, "C5_shard decode T=512 V=5000 K=16 (64 utterances)")
    # the shard is independent of its neighbours: decoding half of it gives the same beams
    y2, yl2, yp2 = F.ctc_prefix_search(lg[:, : N // 2], K)
    assert torch.equal(y2, y[:, : N // 2]) and torch.equal(yl2, yl[: N // 2]) and torch.equal(yp2, yp[: N // 2])
