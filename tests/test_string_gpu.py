"""GPU parity of the edit-distance family: HIP kernels (through the C ABI) vs the oracle.

Bit-exact comparison (float32 ``==``) on seeded random inputs at sizes the oracle
finishes in seconds; the structure follows the reference's tests/test_string.py.
"""
import contextlib
import warnings
import zlib

import numpy as np
import pytest
import torch

import oracle
from pydrobert_amd import functional as F
from pydrobert_amd import modules as M

pytestmark = pytest.mark.gpu

COSTS = [
    (1.0, 1.0, 1.0),
    (2.0, 2.0, 2.0),
    (3.0, 3.0, 4.0),  # NIST
    (2.0, 0.5, 1.0),
    (0.5, 1.0, 0.5),
    (1.0, 2.0, 3.0),
    (0.1, 0.7, 1.3),  # not exactly representable: exercises the exact-unroll path
    (1.1, 1.1, 1.1),
]

NAMES = ["error_rate", "edit_distance", "prefix_error_rates", "prefix_edit_distances",
         "optimal_completion"]  # fmt: skip


def _call_both(name, ref, hyp, device, **kw):
    exp = getattr(oracle, name)(ref, hyp, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        act = getattr(F, name)(
            torch.from_numpy(ref).to(device), torch.from_numpy(hyp).to(device), warn=False, **kw
        )
    torch.cuda.synchronize()
    return exp, act.cpu().numpy()


def _assert_same(exp, act, what):
    assert exp.shape == act.shape, (what, exp.shape, act.shape)
    assert exp.dtype == act.dtype, (what, exp.dtype, act.dtype)
    assert np.array_equal(exp, act), (what, np.argwhere(exp != act)[:5])


@pytest.mark.parametrize("name", NAMES)
def test_random_small_all_options(device, name):
    rng = np.random.default_rng(zlib.adler32(name.encode()))
    for it in range(60):
        N = int(rng.integers(1, 9))
        R = int(rng.integers(1, 14))
        H = int(rng.integers(1, 14))
        V = int(rng.integers(2, 5))
        eos = None if rng.random() < 0.3 else int(rng.integers(0, V))
        bf = bool(rng.integers(0, 2))
        shape_r, shape_h = ((N, R), (N, H)) if bf else ((R, N), (H, N))
        ref = rng.integers(0, V, shape_r)
        hyp = rng.integers(0, V, shape_h)
        c = COSTS[int(rng.integers(0, len(COSTS)))]
        kw = dict(eos=eos, include_eos=bool(rng.integers(0, 2)), batch_first=bf,
                  ins_cost=c[0], del_cost=c[1], sub_cost=c[2])  # fmt: skip
        if name in ("error_rate", "edit_distance"):
            kw["norm"] = bool(rng.integers(0, 2))
        elif name.startswith("prefix"):
            kw["norm"] = bool(rng.integers(0, 2))
            kw["exclude_last"] = bool(rng.integers(0, 2))
            kw["padding"] = int(rng.integers(-5, 0))
        else:
            kw["exclude_last"] = bool(rng.integers(0, 2))
            kw["padding"] = int(rng.integers(-5, 0))
        exp, act = _call_both(name, ref, hyp, device, **kw)
        _assert_same(exp, act, (name, it, kw))


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("costs", COSTS, ids=lambda c: "c{}_{}_{}".format(*c))
def test_medium_ragged(device, name, costs):
    """Ragged lengths around the 64-lane strip boundaries, tie-rich alphabet."""
    rng = np.random.default_rng(1234)
    N, R, H, V = 24, 150, 137, 5
    if costs == (0.1, 0.7, 1.3) or name == "optimal_completion":
        N = 8
    eos = V
    ref = rng.integers(0, V, (R, N))
    hyp = rng.integers(0, V, (H, N))
    rl = rng.integers(0, R + 1, N)
    hl = rng.integers(0, H + 1, N)
    rl[:4] = [0, 64, 65, 128]
    hl[:4] = [5, 0, 63, 64]
    for n in range(N):
        if rl[n] < R:
            ref[rl[n], n] = eos
        if hl[n] < H:
            hyp[hl[n], n] = eos
    kw = dict(eos=eos, ins_cost=costs[0], del_cost=costs[1], sub_cost=costs[2])
    exp, act = _call_both(name, ref, hyp, device, **kw)
    _assert_same(exp, act, (name, costs))


@pytest.mark.parametrize("name", ["error_rate", "edit_distance", "prefix_error_rates"])
def test_long_reference_chunks(device, name):
    """R > 512 makes the skewed kernel sweep several 512-column chunks."""
    rng = np.random.default_rng(7)
    N, R, H, V = 6, 1100, 300, 6
    ref = rng.integers(0, V, (R, N))
    hyp = rng.integers(0, V, (H, N))
    ref[700, 1] = V
    ref[512, 2] = V
    ref[513, 3] = V
    hyp[100, 4] = V
    for costs in [(1.0, 1.0, 1.0), (3.0, 3.0, 4.0)]:
        kw = dict(eos=V, ins_cost=costs[0], del_cost=costs[1], sub_cost=costs[2])
        exp = getattr(oracle, name)(ref, hyp, faithful=False, **kw)
        act = getattr(F, name)(torch.from_numpy(ref).to(device), torch.from_numpy(hyp).to(device),
                               warn=False, **kw).cpu().numpy()  # fmt: skip
        _assert_same(exp, act, (name, costs))


def test_wide_tokens_and_negative_tokens(device):
    """int64 tokens outside int32 take the first-occurrence remap path; negatives are fine."""
    rng = np.random.default_rng(11)
    N, R, H = 5, 70, 66
    base = np.array([-(2**40), -2, -1, 0, 2**31, 2**33 + 7, 2**62], dtype=np.int64)
    ref = base[rng.integers(0, len(base), (R, N))]
    hyp = base[rng.integers(0, len(base), (H, N))]
    for name in NAMES:
        exp, act = _call_both(name, ref, hyp, device, eos=-1, ins_cost=3.0, del_cost=3.0, sub_cost=4.0)
        _assert_same(exp, act, name)


def test_config1_shape_bitexact(device):
    """BASELINE config 1: N=64, T=128, V=32, unit costs, norm=True."""
    rng = np.random.default_rng(0x5EED0001)
    ref = rng.integers(0, 32, (128, 64))
    hyp = rng.integers(0, 32, (128, 64))
    for name in NAMES:
        exp, act = _call_both(name, ref, hyp, device)
        _assert_same(exp, act, name)


def test_known_answers(device):
    """The reference's hand-computed pairs (tests/test_string.py:171-221), restated."""
    eos = 0
    pairs = (
        ((1, 2, 3), (1, 2, 3), 0), ((2, 3), (1, 2, 3), 1), ((1, 3), (1, 2, 3), 1),
        ((3,), (1, 2, 3), 2), ((1, 2, 3), (1, 3), 1), ((1, 2, 3), (1, 2), 1),
        ((1, 2, 3), (1,), 2), ((1, 3, 1, 2, 3), (1, 2, 3), 2), ((1, 2, 3), (4, 5, 6), 3),
        ((2, 2, 2), (2,), 2), (tuple(), (1,), 1), (tuple(), tuple(), 0),
    )  # fmt: skip
    for include_eos in (0, 1):
        for batch_first in (False, True):
            seqs_r = [torch.tensor(x[0] + (eos,) * include_eos, dtype=torch.long) for x in pairs]
            seqs_h = [torch.tensor(x[1] + (eos,) * include_eos, dtype=torch.long) for x in pairs]
            ref = torch.nn.utils.rnn.pad_sequence(seqs_r, padding_value=eos, batch_first=batch_first)
            hyp = torch.nn.utils.rnn.pad_sequence(seqs_h, padding_value=eos, batch_first=batch_first)
            exp = torch.tensor([float(x[2]) for x in pairs])
            ref_lens = torch.tensor([len(x[0]) + include_eos for x in pairs])
            hyp_lens = torch.tensor([len(x[1]) + include_eos for x in pairs])
            exp_n = torch.where(ref_lens == 0, hyp_lens.ne(0).float(), exp / ref_lens.float())
            for cls in (M.EditDistance, M.ErrorRate):
                for norm in (False, True):
                    mod = cls(eos=eos, warn=False, norm=norm, include_eos=bool(include_eos),
                              batch_first=batch_first)  # fmt: skip
                    act = mod(ref.to(device), hyp.to(device)).cpu()
                    assert torch.equal(exp_n if norm else exp, act)


def test_optimal_completion_strings(device):
    """OCD known answers (reference tests/test_string.py:115-168), restated."""
    eos, padding = ord("#"), -1
    triplets = (
        ("sunday#", "saturday#", ["s", "u", "un", "und", "n", "nd", "a", "y", "#", ""]),
        ("sunday#", "satrapy#", ["s", "u", "un", "und", "unda", "y", "y#", "#", ""]),
        ("abc#", "abc#", ["a", "b", "c", "#", ""]),
        ("foot#", "bot#", ["f", "fo", "o", "ot#", ""]),
        ("abc#", "def#", ["a", "ab", "abc", "abc#", ""]),
    )
    for include_eos in (True, False):
        for batch_first in (True, False):
            for exclude_last in (True, False):
                ref = torch.nn.utils.rnn.pad_sequence(
                    [torch.tensor([ord(c) for c in w]) for (w, _, _) in triplets],
                    batch_first=batch_first, padding_value=padding).to(device)  # fmt: skip
                hyp = torch.nn.utils.rnn.pad_sequence(
                    [torch.tensor([ord(c) for c in w]) for (_, w, _) in triplets],
                    batch_first=batch_first, padding_value=eos).to(device)  # fmt: skip
                oc = M.OptimalCompletion(eos=eos, padding=padding, batch_first=batch_first,
                                         exclude_last=exclude_last, include_eos=include_eos)  # fmt: skip
                act = oc(ref, hyp).cpu()
                if not batch_first:
                    act = act.transpose(0, 1)
                assert act.shape[0] == len(triplets)
                for act_bt, (_, _, exp_bt) in zip(act, triplets):
                    if not include_eos:
                        exp_bt = [nexts.replace("#", "") for nexts in exp_bt[:-1]]
                    if exclude_last:
                        exp_bt = exp_bt[:-1]
                    assert act_bt.shape[0] >= len(exp_bt)
                    assert torch.all(act_bt[len(exp_bt):].eq(padding))
                    for a, e in zip(act_bt, exp_bt):
                        a = a.masked_select(a.ne(padding))
                        assert sorted(e) == sorted(chr(i) for i in a.tolist())


def test_warnings_and_errors(device):
    ref = torch.tensor([[1, 2, 3], [1, 2, 0]], device=device).t().contiguous()
    hyp = torch.tensor([[1, 2, 3], [1, 0, 0]], device=device).t().contiguous()
    with pytest.warns(UserWarning, match="did not contain the eos"):
        F.error_rate(ref, hyp, eos=0, include_eos=True)
    with pytest.warns(UserWarning, match="non-uniform"):
        F.error_rate(ref, hyp, ins_cost=2.0)
    empty = torch.zeros((3, 2), dtype=torch.long, device=device)
    with pytest.warns(UserWarning, match="empty transcripts"):
        F.error_rate(empty, hyp, eos=0)
    with pytest.raises(RuntimeError, match="2 dimensional"):
        F.error_rate(ref[0], hyp)
    with pytest.raises(RuntimeError, match="batch size"):
        F.error_rate(ref, hyp[:, :1])
    with pytest.raises(RuntimeError, match="ROCm"):
        F.error_rate(ref.cpu(), hyp.cpu())
    with pytest.raises(ValueError):
        M.ErrorRate(eos="a")


def test_full_size_properties(device):
    """BASELINE config 2 size (N=4096, T=512, V=256): size-independent properties.

    * prefix row h equals error_rate of the hypothesis truncated to h tokens (spot rows);
    * the last valid prefix entry equals error_rate;
    * d(x, x) = 0; 0 <= d <= max(R, H); |d(ref,hyp) - d(hyp,ref)| = 0 for unit costs;
    * every optimal-completion set at h=0 is {ref[0]}.
    """
    N, T, V = 4096, 512, 256
    g = torch.Generator(device="cpu").manual_seed(0x5EED0002)
    ref = torch.randint(0, V, (T, N), generator=g).to(device)
    hyp = torch.randint(0, V, (T, N), generator=g).to(device)
    d = F.edit_distance(ref, hyp, warn=False)
    assert torch.equal(F.edit_distance(hyp, ref, warn=False), d)
    assert torch.equal(F.edit_distance(ref, ref, warn=False), torch.zeros_like(d))
    assert d.min() >= 0 and d.max() <= T
    er = F.error_rate(ref, hyp, warn=False)
    assert torch.equal(er, d / T)
    pre = F.prefix_error_rates(ref, hyp, warn=False)
    assert pre.shape == (T + 1, N)
    assert torch.equal(pre[-1], er)
    assert torch.equal(pre[0], torch.ones_like(er))
    for h in (1, 63, 64, 65, 300):
        assert torch.equal(F.error_rate(ref, hyp[:h], warn=False), pre[h])
    # a sample of utterances against the oracle
    idx = torch.arange(0, N, 512)
    exp = oracle.prefix_error_rates(ref[:, idx].cpu().numpy(), hyp[:, idx].cpu().numpy(), faithful=False)
    assert np.array_equal(exp, pre[:, idx].cpu().numpy())
    oc = F.optimal_completion(ref[:, :256], hyp[:, :256], warn=False)
    assert oc.shape[:2] == (T + 1, 256)
    assert torch.equal(oc[0, :, 0], ref[0, :256])
    assert (oc[0, :, 1:] == -100).all()
    exp = oracle.optimal_completion(ref[:, :4].cpu().numpy(), hyp[:, :4].cpu().numpy(), faithful=False)
    got = oc[:, :4].cpu().numpy()
    C = exp.shape[2]
    assert np.array_equal(exp, got[:, :, :C]) and (got[:, :, C:] == -100).all()


def test_optimal_completion_long_reference(device):
    """R > 512: the row-synchronous kernel's BIG instantiation (more than 8 columns per lane);
    R = 2048 is the last reference it holds in registers."""
    rng = np.random.default_rng(13)
    N, R, H, V = 3, 700, 90, 9
    ref = rng.integers(0, V, (R, N))
    hyp = rng.integers(0, V, (H, N))
    ref[650, 1] = V
    hyp[40, 2] = V
    for kw in (dict(eos=V), dict(eos=V, include_eos=False, exclude_last=True), dict()):
        exp = oracle.optimal_completion(ref, hyp, faithful=False, **kw)
        act = F.optimal_completion(torch.from_numpy(ref).to(device), torch.from_numpy(hyp).to(device),
                                   warn=False, **kw).cpu().numpy()  # fmt: skip
        assert exp.shape == act.shape and np.array_equal(exp, act), kw
    ref = torch.from_numpy(rng.integers(0, V, (2048, 2))).to(device)
    hyp2 = torch.from_numpy(rng.integers(0, V, (5, 2))).to(device)
    oc = F.optimal_completion(ref, hyp2, warn=False)
    exp = oracle.optimal_completion(ref.cpu().numpy(), hyp2.cpu().numpy(), faithful=False)
    assert np.array_equal(exp, oc.cpu().numpy())


@pytest.mark.parametrize("R,V", [(2049, 5), (3000, 40), (5000, 3000)])
def test_optimal_completion_beyond_the_register_rows(device, R, V):
    """References longer than 2048 tokens run the plain one-workgroup-per-utterance form
    (csrc/lev_generic.hip): class bitmasks wider than a wave, the generic expansion; ragged
    lengths, batch-first layout, costs that are exact in float32 and (up to R = 3000: the replay is
    O(R^2) per row in the oracle too) costs that are not."""
    rng = np.random.default_rng(R)
    N, H = 3, 37
    ref = rng.integers(0, V, (R, N))
    hyp = rng.integers(0, V, (H, N))
    ref[R - 300, 1] = V
    hyp[20, 2] = V
    for kw in (dict(eos=V), dict(eos=V, include_eos=False, exclude_last=True, ins_cost=3.0, del_cost=3.0, sub_cost=4.0),
               dict(batch_first=True)):  # fmt: skip
        r, h = (ref.T.copy(), hyp.T.copy()) if kw.get("batch_first") else (ref, hyp)
        exp = oracle.optimal_completion(r, h, faithful=False, **kw)
        act = F.optimal_completion(torch.from_numpy(r).to(device), torch.from_numpy(h).to(device),
                                   warn=False, **kw).cpu().numpy()  # fmt: skip
        assert exp.shape == act.shape and np.array_equal(exp, act), kw
    if R <= 3000:  # costs that are NOT exact in float32: the reference's unrolled deletion, term by term
        kw = dict(eos=V, ins_cost=0.1, del_cost=0.7, sub_cost=1.3)
        exp = oracle.optimal_completion(ref, hyp, **kw)
        act = F.optimal_completion(torch.from_numpy(ref).to(device), torch.from_numpy(hyp).to(device), warn=False, **kw)
        assert exp.shape == tuple(act.shape) and np.array_equal(exp, act.cpu().numpy())


@pytest.mark.parametrize("name", ["edit_distance", "prefix_edit_distances", "error_rate"])
def test_inexact_costs_beyond_the_register_rows(device, name):
    """Cost-mode distances with costs that are inexact in float32 and a reference of more than 2048
    tokens: the plain workgroup kernel replays the reference's arithmetic (csrc/lev_generic.hip).
    (error_rate counts mistakes for such costs: the cell-by-cell kernel, at any length.)"""
    rng = np.random.default_rng(len(name))
    R, H, N, V = 2300, 29, 3, 6
    ref, hyp = rng.integers(0, V, (R, N)), rng.integers(0, V, (H, N))
    ref[R - 200, 0] = V
    hyp[17, 1] = V
    for kw in (dict(eos=V, ins_cost=0.1, del_cost=0.7, sub_cost=1.3),
               dict(eos=V, include_eos=False, ins_cost=0.3, del_cost=0.1, sub_cost=0.2, norm=True),
               dict(ins_cost=0.1, del_cost=0.7, sub_cost=1.3, batch_first=True)):  # fmt: skip
        if name == "prefix_edit_distances":
            kw = dict(kw, exclude_last=bool(kw.get("norm", False)))
        r, h = (ref.T.copy(), hyp.T.copy()) if kw.get("batch_first") else (ref, hyp)
        exp = getattr(oracle, name)(r, h, **kw)
        act = getattr(F, name)(torch.from_numpy(r).to(device), torch.from_numpy(h).to(device), warn=False, **kw)
        assert exp.shape == tuple(act.shape) and np.array_equal(exp, act.cpu().numpy(), equal_nan=True), (name, kw)


def test_fill_after_eos(device):
    """a7: the reference's own outputs (tests/golden/fill.npz), the oracle on random shapes, every
    dtype class of the value tensor, strided views, the module, and the gradient w.r.t. value."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fill.npz"))
    tok, eos = torch.from_numpy(g["tokens"]).to(device), int(g["eos"])
    val, valb = torch.from_numpy(g["value"]).to(device), torch.from_numpy(g["value_bool"]).to(device)
    for dim in (0, 1, 2, -1):
        assert np.array_equal(F.fill_after_eos(tok, eos, dim).cpu().numpy(), g["default_d%d" % dim])
        assert np.array_equal(F.fill_after_eos(tok, eos, dim, -7.0).cpu().numpy(), g["fill_d%d" % dim])
        assert np.array_equal(F.fill_after_eos(tok, eos, dim, 0.5, val).cpu().numpy(), g["value_d%d" % dim])
        assert np.array_equal(F.fill_after_eos(tok, eos, dim, 1.0, valb).cpu().numpy(), g["bool_d%d" % dim])
    assert np.array_equal(M.FillAfterEndOfSequence(eos, 1, 9.0)(tok).cpu().numpy(), g["module"])
    # hand case of the reference's docstring form
    t2 = torch.tensor([[1, 2], [0, 3], [4, 0], [0, 5]], device=device)
    assert F.fill_after_eos(t2, 0, 0, -1).t().tolist() == [[1, 0, -1, -1], [2, 3, 0, -1]]
    assert M.FillAfterEndOfSequence(0)(t2).t().tolist() == [[1, 0, 0, 0], [2, 3, 0, 0]]
    rng = np.random.default_rng(17)
    for shape in [(1,), (130,), (70, 3), (3, 70), (5, 129, 4), (2, 3, 4, 5), (0, 4), (64, 64)]:
        tk = rng.integers(0, 5, shape)
        for dim in range(-1, len(shape)):
            for dt in (np.float32, np.float64, np.int64, np.int32, np.int16, np.uint8, np.float16):
                v = (rng.normal(size=shape) * 20).astype(dt)
                exp = oracle.fill_after_eos(tk, 2, dim, 3.0, v)
                act = F.fill_after_eos(torch.from_numpy(tk).to(device), 2, dim, 3.0, torch.from_numpy(v).to(device))
                assert act.dtype == torch.from_numpy(v).dtype and np.array_equal(act.cpu().numpy(), exp), (shape, dim, dt)
    # tokens and value of different shapes: `dim` counts the TOKENS' dimensions (the reference's
    # cumsum runs on the tokens; masked_fill then broadcasts the mask, _string.py:40-42)
    for tshape, vshape in [((7,), (5, 7)), ((7,), (3, 5, 7)), ((5, 1), (5, 6)), ((4, 1, 6), (3, 6)), ((3, 6), (4, 1, 6))]:
        tk = rng.integers(0, 3, tshape)
        v = rng.normal(size=vshape).astype(np.float32)
        for dim in range(-len(tshape), len(tshape)):
            exp = oracle.fill_after_eos(tk, 2, dim, -1.0, v)
            act = F.fill_after_eos(torch.from_numpy(tk).to(device), 2, dim, -1.0, torch.from_numpy(v).to(device))
            assert np.array_equal(act.cpu().numpy(), exp), (tshape, vshape, dim)
        with pytest.raises(IndexError):
            F.fill_after_eos(torch.from_numpy(tk).to(device), 2, len(tshape), -1.0, torch.from_numpy(v).to(device))
    # non-int64 tokens and a strided (transposed) view
    tk = rng.integers(0, 4, (33, 9))
    tt = torch.from_numpy(np.ascontiguousarray(tk.T)).to(device).t()
    assert not tt.is_contiguous()
    assert np.array_equal(F.fill_after_eos(tt, 1, 0).cpu().numpy(), oracle.fill_after_eos(tk, 1, 0))
    assert np.array_equal(F.fill_after_eos(tt.int(), 1, 1).cpu().numpy(), oracle.fill_after_eos(tk.astype(np.int32), 1, 1))
    assert np.array_equal(F.fill_after_eos(tt.float(), 1, 0, 7.5).cpu().numpy(),
                          oracle.fill_after_eos(tk.astype(np.float32), 1, 0, 7.5))
    # gradient: passes where the value was kept
    v = torch.randn((33, 9), device=device, requires_grad=True)
    out = F.fill_after_eos(tt, 1, 0, 0.25, v)
    (gv,) = torch.autograd.grad(out.sum(), v)
    kept = torch.from_numpy(oracle.fill_after_eos(tk, 1, 0, 0.0, np.ones((33, 9), np.float32))).to(device)
    assert torch.equal(gv, kept)


def test_optimal_completion_shape_sweep(device):
    """The tiled expansion over odd shapes: batches that do not fill a tile, one-word and
    many-word bitmasks, odd set widths C, both layouts, exclude_last, ragged lengths."""
    rng = np.random.default_rng(77)
    shapes = [(1, 1, 1, 2), (3, 5, 4, 3), (5, 33, 20, 4), (7, 64, 70, 9), (9, 65, 33, 40), (2, 130, 17, 3),
              (6, 600, 40, 25), (13, 31, 129, 2), (4, 200, 200, 200)]
    for (N, R, H, V) in shapes:
        ref = rng.integers(0, V, (R, N))
        hyp = rng.integers(0, V, (H, N))
        for kw in (dict(), dict(eos=V - 1, include_eos=False), dict(eos=0, exclude_last=True), dict(batch_first=True)):
            a, b = (ref.T.copy(), hyp.T.copy()) if kw.get("batch_first") else (ref, hyp)
            exp = oracle.optimal_completion(a, b, faithful=False, **kw)
            act = F.optimal_completion(torch.from_numpy(a).to(device), torch.from_numpy(b).to(device), warn=False, **kw)
            assert act.shape == exp.shape and np.array_equal(act.cpu().numpy(), exp), (N, R, H, V, kw)


@pytest.mark.parametrize("R", [1, 2, 31, 32, 33, 63, 64, 65, 200, 480, 511, 512])
def test_optimal_completion_bit_parallel_rows(device, R, switch):
    """Uniform costs and references of up to 512 tokens take the bit-parallel mask kernel
    (csrc/lev_bitpar.hip, oc_bitpar_kernel): one case per number of 32-column blocks in use, ragged
    lengths on both sides (eos anywhere, including position 0), vocabularies from two tokens (every
    row minimum tied many times) to more tokens than positions, batches that leave utterance slots
    of the last wave empty.  Checked against the oracle on a slice and against the row-synchronous
    kernel (PDT_OC_BITPAR=0) on everything."""
    rng = np.random.default_rng(9000 + R)
    for it, H in enumerate((1, max(1, R - 1), R + 37 if R < 300 else 90)):
        N = int(rng.integers(1, 11))
        V = int(rng.choice([2, 3, 7, 50, 2000]))
        ref = rng.integers(0, V, (R, N))
        hyp = rng.integers(0, V, (H, N))
        if it == 1 and R > 2:  # long common runs: carries that ripple through many blocks
            hyp[: min(R, H)] = ref[: min(R, H)]
            hyp[rng.integers(0, H, 3), rng.integers(0, N, 3)] = V
        for kw in (dict(), dict(eos=0), dict(eos=V - 1, include_eos=False, exclude_last=True),
                   dict(batch_first=True, ins_cost=2.5, del_cost=2.5, sub_cost=2.5)):
            a, b = (ref.T.copy(), hyp.T.copy()) if kw.get("batch_first") else (ref, hyp)
            ta, tb = torch.from_numpy(a).to(device), torch.from_numpy(b).to(device)
            switch("PDT_OC_BITPAR", "1")
            act = F.optimal_completion(ta, tb, warn=False, **kw)
            switch("PDT_OC_BITPAR", "0")
            row = F.optimal_completion(ta, tb, warn=False, **kw)
            assert act.shape == row.shape and torch.equal(act, row), (R, H, N, V, kw)
            if R * H * N <= 40000:
                exp = oracle.optimal_completion(a, b, faithful=False, **kw)
                assert np.array_equal(act.cpu().numpy(), exp), (R, H, N, V, kw)


# ---- bit-parallel unit-cost kernels (csrc/lev_bitpar.hip) ------------------------------------
_BITPAR_OPS = ["error_rate", "edit_distance", "prefix_error_rates", "prefix_edit_distances"]


@pytest.mark.parametrize("H", [1, 31, 32, 33, 64, 65, 129, 256, 257, 511, 513, 1000, 1024])
def test_bitpar_block_count_sweep(device, H):
    """Uniform costs run on the bit-parallel kernels whenever the hypothesis has at most 1024
    tokens: one case per number of lanes an utterance takes (1, 2, 4, ..., 32), ragged lengths on
    both sides, batch sizes that leave lanes of the last wave without an utterance."""
    rng = np.random.default_rng(H)
    for it, R in enumerate((max(1, H - 3), H + 70 if H < 600 else 300, 1300 if H <= 65 else 40)):
        N = int(rng.integers(1, 12))
        V = int(rng.choice([2, 3, 40]))
        ref = rng.integers(0, V, (R, N))
        hyp = rng.integers(0, V, (H, N))
        eos = None
        if it != 1:
            eos = V
            for n in range(N):
                if rng.random() < 0.8:
                    ref[int(rng.integers(0, R)), n] = eos
                if rng.random() < 0.8:
                    hyp[int(rng.integers(0, H)), n] = eos
            if N > 1:
                ref[0, 0] = eos  # an empty reference
                hyp[0, N - 1] = eos  # an empty hypothesis
        c = [1.0, 2.5][it % 2]
        for name in _BITPAR_OPS:
            kw = dict(eos=eos, include_eos=bool(it & 1), ins_cost=c, del_cost=c, sub_cost=c,
                      norm=bool(rng.integers(0, 2)))  # fmt: skip
            if name.startswith("prefix"):
                kw["exclude_last"] = bool(rng.integers(0, 2))
                kw["padding"] = -3
            exp = getattr(oracle, name)(ref, hyp, faithful=False, **kw)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                act = getattr(F, name)(torch.from_numpy(ref).to(device), torch.from_numpy(hyp).to(device),
                                       warn=False, **kw).cpu().numpy()  # fmt: skip
            _assert_same(exp, act, (name, H, R, N, V, kw))


def test_bitpar_extreme_token_values(device):
    """The class table is built from a sort of keys derived from the int64 tokens: the extremes of
    the range, negatives and neighbours that differ in the top or bottom bit only."""
    rng = np.random.default_rng(3)
    base = np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).min + 1, -(2**40), -1, 0, 1, 2**31,
                     2**32, 2**62, np.iinfo(np.int64).max - 1, np.iinfo(np.int64).max], dtype=np.int64)  # fmt: skip
    N, R, H = 9, 200, 190
    ref = base[rng.integers(0, len(base), (R, N))]
    hyp = base[rng.integers(0, len(base), (H, N))]
    for name in _BITPAR_OPS:
        exp, act = _call_both(name, ref, hyp, device, eos=0)
        _assert_same(exp, act, name)


def test_classification_by_presence_map_and_by_sort_agree(device):
    """lev_classify ranks vocabulary indices below 8192 through a presence map and anything else
    through a sorted table (lev_classes.hpp), decided per utterance: utterances on both sides of the
    boundary in one batch, look-up tokens outside the map's range (negative, >= 8192, beyond int32)
    against in-range class tokens, every word of the map in use -- against the oracle; and the same
    sequences shifted by 2^40 (all through the sort) give the same distances and the same
    completions."""
    rng = np.random.default_rng(77)
    N, R, H = 24, 300, 280
    ref = rng.integers(0, 8192, (R, N))
    hyp = rng.integers(0, 8192, (H, N))
    hyp[:, 0:4] = rng.integers(8100, 8292, (H, 4))      # some hypothesis tokens beyond the map
    ref[:, 4:8] = rng.integers(-50, 60, (R, 4))          # negative reference tokens, small hypothesis
    hyp[:, 4:8] = rng.integers(0, 60, (H, 4))
    ref[:, 8:12] = rng.integers(8000, 8400, (R, 4))      # reference across the boundary, hypothesis inside
    hyp[:, 8:12] = rng.integers(8000, 8192, (H, 4))
    ref[::7, 12] = 2**33 + 5                              # a look-up whose low 32 bits are a present token
    hyp[:, 12] = rng.integers(0, 9, H)
    ref[:, 12] = np.where(ref[:, 12] > 2**33, ref[:, 12], rng.integers(0, 9, R))
    hyp[:, 13] = np.arange(H) * 29 % 8192                 # spread over every word of the map
    ref[:, 13] = np.arange(R) * 31 % 8192
    for name in NAMES:
        for kw in ({}, {"eos": 8191, "include_eos": True}):
            exp, act = _call_both(name, ref, hyp, device, **kw)
            _assert_same(exp, act, (name, kw))
    small_r, small_h = rng.integers(0, 40, (R, N)), rng.integers(0, 40, (H, N))
    for name in NAMES:
        a = getattr(F, name)(torch.from_numpy(small_r).to(device), torch.from_numpy(small_h).to(device), warn=False)
        b = getattr(F, name)(torch.from_numpy(small_r + 2**40).to(device), torch.from_numpy(small_h + 2**40).to(device), warn=False)
        if name == "optimal_completion":
            b = torch.where(b >= 0, b - 2**40, b)  # (the padding value stays)
        assert torch.equal(a, b), name


def test_lev_workspace_is_optional(device):
    """pdt_lev gives the same bits with and without its workspace (bit-parallel kernels against the
    cell-by-cell ones), and reports no workspace for hypotheses beyond 1024 tokens."""
    from pydrobert_amd import _cabi

    L = _cabi.lib()
    assert L.pdt_lev_workspace_bytes(100, 1025, 8) == 0
    assert L.pdt_lev_workspace_bytes(5000, 1024, 8) > 0
    rng = np.random.default_rng(5)
    N, R, H, V = 37, 300, 280, 7
    ref = torch.from_numpy(rng.integers(0, V, (R, N))).to(device)
    hyp = torch.from_numpy(rng.integers(0, V, (H, N))).to(device)
    outs = []
    for use_ws in (True, False):
        nbytes = int(L.pdt_lev_workspace_bytes(R, H, N))
        assert nbytes > 0
        ws = torch.empty(nbytes, device=device, dtype=torch.uint8) if use_ws else None
        for mode, shape in ((_cabi.MODE_FINAL, (N,)), (_cabi.MODE_PREFIX, (H + 1, N))):
            out = torch.empty(shape, device=device, dtype=torch.float)
            rc = L.pdt_lev(
                _cabi.ptr(ref), R, ref.stride(0), ref.stride(1), _cabi.ptr(hyp), H, hyp.stride(0),
                hyp.stride(1), N, 1, 3, 1, 1.0, 1.0, 1.0, 1, mode, 0, -1.0, 1,
                _cabi.ptr(out), out.stride(0) if mode == _cabi.MODE_PREFIX else 0,
                out.stride(-1), 0, 0, 0, _cabi.ptr(ws), nbytes if use_ws else 0,
                _cabi.stream_ptr(device),
            )  # fmt: skip
            assert rc == 0
            outs.append(out.cpu())
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
    exp = oracle.error_rate(ref.cpu().numpy(), hyp.cpu().numpy(), eos=3, include_eos=True, norm=True)
    assert np.array_equal(exp, outs[0].numpy())


def test_classification_is_reused_only_for_the_same_inputs(device):
    """Inside `reuse_classification()` error_rate followed by prefix_error_rates on one (ref, hyp) pair
    classifies once (pdt_lev_classified through the host's one-entry, one-hit cache).  The entry must
    not outlive its inputs' CONTENTS as far as the version counter sees them: in-place edits, other
    eos handling, other tensors at recycled addresses, another stream -- every call against the
    oracle; warnings repeat with the cached bits.  Leaving the block drops the entry."""
    import warnings

    from pydrobert_amd import _string

    rng = np.random.default_rng(31)
    R, H, N, V = 70, 66, 9, 5

    def check(tr, th, **kw):
        r, h = tr.cpu().numpy(), th.cpu().numpy()
        assert np.array_equal(F.error_rate(tr, th, warn=False, **kw).cpu().numpy(), oracle.error_rate(r, h, **kw))
        assert np.array_equal(
            F.prefix_error_rates(tr, th, warn=False, **kw).cpu().numpy(), oracle.prefix_error_rates(r, h, **kw)
        )
        assert np.array_equal(F.edit_distance(tr, th, warn=False, **kw).cpu().numpy(), oracle.edit_distance(r, h, **kw))

    tr = torch.from_numpy(rng.integers(0, V, (R, N))).to(device)
    th = torch.from_numpy(rng.integers(0, V, (H, N))).to(device)
    with _string.reuse_classification():
        F.error_rate(tr, th, eos=4, include_eos=True, warn=False)
        assert _string._CLASSIFIED[device.index][0][0] == tr.data_ptr()  # an entry stands ...
        F.prefix_error_rates(tr, th, eos=4, include_eos=True, warn=False)
        assert device.index not in _string._CLASSIFIED  # ... for one hit
        check(tr, th, eos=4, include_eos=True)
        check(tr, th, eos=4, include_eos=False)  # other length rule: not the cached tables
        th[3] = (th[3] + 1) % V  # in-place edit: same address, new version
        check(tr, th, eos=4, include_eos=False)
        tr.add_(1).remainder_(V)
        check(tr, th, eos=4, include_eos=False)
        for _ in range(6):  # fresh tensors, freed every round: the allocator recycles their addresses
            a = torch.from_numpy(rng.integers(0, V, (R, N))).to(device)
            b = torch.from_numpy(rng.integers(0, V, (H, N))).to(device)
            check(a, b, eos=None)
            del a, b
        side = torch.cuda.Stream(device)
        with torch.cuda.stream(side):
            check(tr, th, eos=2, include_eos=True)
        side.synchronize()
        check(tr, th, eos=2, include_eos=True)
        # warnings: the second call of a pair reports what the classifying call found
        nr = torch.full((R, N), 1, device=device)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            F.error_rate(nr, th, eos=4, include_eos=True, norm=True)
            n1 = len(w)
            F.prefix_error_rates(nr, th, eos=4, include_eos=True, norm=True)
        assert n1 >= 1 and len(w) == 2 * n1, [str(x.message)[:40] for x in w]
        F.error_rate(tr, th, warn=False)
    assert not _string._CLASSIFIED  # nothing stays pinned outside the block


def test_no_reuse_by_default_and_writes_behind_the_version_counter(device):
    """By default every string operator classifies its own inputs: nothing is kept between calls, so a
    write the version counter does not see (`.data.copy_`) between error_rate and prefix_error_rates
    cannot serve stale tables."""
    from pydrobert_amd import _string, switches

    assert switches.get("PDT_LEV_CACHE") == 0
    rng = np.random.default_rng(5)
    R, H, N, V = 90, 80, 7, 6
    tr = torch.from_numpy(rng.integers(0, V, (R, N))).to(device)
    th = torch.from_numpy(rng.integers(0, V, (H, N))).to(device)
    F.error_rate(tr, th, warn=False)
    assert not _string._CLASSIFIED
    v0 = th._version
    th.data.copy_(torch.from_numpy(rng.integers(0, V, (H, N))).to(device))
    assert th._version == v0  # the hazard: contents changed, identity did not
    act = F.prefix_error_rates(tr, th, warn=False).cpu().numpy()
    assert np.array_equal(act, oracle.prefix_error_rates(tr.cpu().numpy(), th.cpu().numpy()))


def test_string_operators_on_inference_tensors(device):
    """ref / hyp created under torch.inference_mode() (the usual evaluation set-up) carry no version
    counter; every operator runs on them, with and without the opt-in reuse."""
    from pydrobert_amd import _string

    rng = np.random.default_rng(8)
    R, H, N, V = 60, 55, 6, 7
    r, h = rng.integers(0, V, (R, N)), rng.integers(0, V, (H, N))
    for reuse in (False, True):
        with torch.inference_mode():
            tr, th = torch.from_numpy(r).to(device), torch.from_numpy(h).to(device)
            assert tr.is_inference()
            ctx = _string.reuse_classification() if reuse else contextlib.nullcontext()
            with ctx:
                for name in ("error_rate", "prefix_error_rates", "edit_distance", "prefix_edit_distances",
                             "optimal_completion"):
                    act = getattr(F, name)(tr, th, eos=V - 1, warn=False).cpu().numpy()
                    exp = getattr(oracle, name)(r, h, eos=V - 1)
                    assert act.shape == exp.shape and np.array_equal(act, exp), (name, reuse)
                hs = torch.from_numpy(rng.integers(0, V, (H, N, 3))).to(device)
                loss = F.minimum_error_rate_loss(torch.randn(N, 3, device=device), tr, hs, eos=V - 1, warn=False)
                assert torch.isfinite(loss)


def test_bitpar_empty_sequences(device):
    """No hypothesis tokens / no reference tokens at all (zero-length dimensions), one utterance."""
    rng = np.random.default_rng(9)
    for R, H, N in ((0, 5, 3), (7, 0, 2), (0, 0, 1), (1, 1, 1), (40, 33, 1)):
        ref = rng.integers(0, 4, (R, N))
        hyp = rng.integers(0, 4, (H, N))
        for name in _BITPAR_OPS:
            kw = dict(norm=True) if "rate" in name else dict(norm=False)
            exp = getattr(oracle, name)(ref, hyp, faithful=False, **kw)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                act = getattr(F, name)(torch.from_numpy(ref).to(device), torch.from_numpy(hyp).to(device),
                                       warn=False, **kw).cpu().numpy()  # fmt: skip
            _assert_same(exp, act, (name, R, H, N))


@pytest.mark.parametrize("R", [10000, 20000])
def test_bitpar_long_reference_short_hypothesis(device, R):
    """A reference far longer than the hypothesis: its look-up pairs fill most of a CU's LDS
    (R = 10 000: one utterance per wave) or do not fit (R = 20 000: the cell-by-cell kernel)."""
    rng = np.random.default_rng(R)
    N, H, V = 3, 50, 6
    ref = rng.integers(0, V, (R, N))
    hyp = rng.integers(0, V, (H, N))
    ref[R // 2, 1] = V  # one reference ends half way
    for name in ("error_rate", "prefix_edit_distances"):
        exp = getattr(oracle, name)(ref, hyp, eos=V, faithful=False)
        act = getattr(F, name)(torch.from_numpy(ref).to(device), torch.from_numpy(hyp).to(device), eos=V, warn=False)
        _assert_same(exp, act.cpu().numpy(), (name, R))
