"""GPU: the HIP path against the golden vectors captured from the live reference
(tests/golden/*.npz) and the TensorFlow-addons arrays the reference's tests ship."""
import os
import warnings

import numpy as np
import pytest
import torch

from pydrobert_amd import functional as F
from pydrobert_amd import modules as M

from _toy_lm import BigramLM

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def T(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def test_string_goldens_bit_exact(device):
    warnings.simplefilter("ignore")
    g = load("string_s1")
    ref, hyp = T(g["ref"], device), T(g["hyp"], device)
    for i, c in enumerate(g["costs"]):
        for norm in (0, 1):
            kw = dict(norm=bool(norm), ins_cost=float(c[0]), del_cost=float(c[1]), sub_cost=float(c[2]), warn=False)
            assert np.array_equal(F.error_rate(ref, hyp, **kw).cpu().numpy(), g["er_c{}_n{}".format(i, norm)])
            assert np.array_equal(F.edit_distance(ref, hyp, **kw).cpu().numpy(), g["ed_c{}_n{}".format(i, norm)])
    g = load("string_s2")
    V = int(g["eos"])
    for inc in (0, 1):
        for bf in (0, 1):
            ref, hyp = (T(g["ref"].T, device), T(g["hyp"].T, device)) if bf else (T(g["ref"], device), T(g["hyp"], device))
            tag = "i{}_b{}".format(inc, bf)
            kw = dict(eos=V, include_eos=bool(inc), batch_first=bool(bf), warn=False)
            a = F.error_rate(ref, hyp, ins_cost=3.0, del_cost=3.0, sub_cost=4.0, **kw)
            assert np.array_equal(a.cpu().numpy(), g["er_" + tag])
            a = F.edit_distance(ref, hyp, ins_cost=2.0, del_cost=0.5, sub_cost=1.0, **kw)
            assert np.array_equal(a.cpu().numpy(), g["ed_" + tag])
            for ex in (0, 1):
                for norm in (0, 1):
                    t2 = tag + "_x{}_n{}".format(ex, norm)
                    a = F.prefix_error_rates(ref, hyp, norm=bool(norm), exclude_last=bool(ex), padding=-100,
                                             ins_cost=1.0, del_cost=2.0, sub_cost=3.0, **kw)  # fmt: skip
                    assert np.array_equal(a.cpu().numpy(), g["per_" + t2]), t2
                    a = F.prefix_edit_distances(ref, hyp, norm=bool(norm), exclude_last=bool(ex), padding=-7, **kw)
                    assert np.array_equal(a.cpu().numpy(), g["ped_" + t2]), t2
                a = F.optimal_completion(ref, hyp, exclude_last=bool(ex), padding=-100, **kw)
                assert np.array_equal(a.cpu().numpy(), g["oc_" + tag + "_x{}".format(ex)])


def test_string_edge_shapes_like_the_live_reference(device):
    """Empty hypothesis / reference, one utterance: what the live reference returns or raises
    (``string_edge.npz``; ``optimal_completion`` of an empty hypothesis is ONE row, with and without
    ``exclude_last``, _string.py:271-278, :286)."""
    import _string_edge

    class impl:
        pass

    for name in ("optimal_completion", "prefix_error_rates", "prefix_edit_distances", "error_rate", "edit_distance"):
        setattr(impl, name, staticmethod(lambda *a, _f=getattr(F, name), **k: _f(*a, warn=False, **k)))
    n = _string_edge.replay(impl, lambda a: T(a, device), lambda t: t.cpu().numpy())
    assert n > 150


def test_sclite_known_answer(device):
    g = load("sclite")
    errs = F.error_rate(T(g["ref"], device), T(g["hyp"], device), eos=-1, norm=False,
                        ins_cost=3.0, del_cost=3.0, sub_cost=4.0, warn=False).cpu().numpy()  # fmt: skip
    ref_lens = (g["ref"] != -1).sum(0)
    for e, l, want in zip(errs, ref_lens, g["per_utt"]):
        assert "{:.03f}".format(e / l) == "{:.03f}".format(want)
    assert "{:.03f}".format(errs.sum() / ref_lens.sum()) == "{:.03f}".format(float(g["total"]))


def _valid_eq(y, yl, gy):
    S = gy.shape[0]
    mask = np.arange(S)[:, None, None] < yl[None]
    return np.array_equal(np.where(mask, y[:S], 0), gy)


def test_ctc_search_goldens(device):
    g = load("ctc_search")
    lg, lens, K = T(g["logits"], device), T(g["lens"], device), int(g["width"])
    y, yl, yp = (x.cpu().numpy() for x in M.CTCPrefixSearch(K)(lg, lens))
    assert np.array_equal(yl, g["y_lens"]) and _valid_eq(y, yl, g["y"])
    assert np.allclose(yp, g["y_probs"], rtol=1e-5, atol=0)
    lm = BigramLM(T(g["lm_table"], device)).to(device)
    for name, vm in (("fusion", False), ("valid", True)):
        y, yl, yp = (x.cpu().numpy() for x in M.CTCPrefixSearch(K, 0.3, lm, valid_mixture=vm)(lg, lens))
        assert np.array_equal(yl, g["y_lens_" + name]), name
        assert _valid_eq(y, yl, g["y_" + name]), name
        assert np.allclose(yp, g["y_probs_" + name], rtol=1e-5, atol=0), name


def test_ctc_advance_golden(device):
    g = load("ctc_advance")
    out = F.ctc_prefix_search_advance(
        (T(g["ext"], device), T(g["nonext"], device), T(g["blank"], device)), int(g["width"]),
        (T(g["nb_prev"], device), T(g["b_prev"], device)), T(g["y_prev"], device),
        T(g["y_prev_last"], device), T(g["y_prev_lens"], device), T(g["prev_is_prefix"], device))  # fmt: skip
    y, last, lens, (nb, b), isp, src, non = [
        tuple(z.cpu().numpy() for z in x) if isinstance(x, tuple) else x.cpu().numpy() for x in out
    ]
    assert np.array_equal(lens, g["y_next_lens"]) and _valid_eq(y, lens, g["y_next"])
    assert np.array_equal(last, g["y_next_last"]) and np.array_equal(src, g["next_src"])
    assert np.array_equal(non, g["next_is_nonext"]) and np.array_equal(isp, g["next_is_prefix"])
    assert np.allclose(nb, g["nb_next"], rtol=1e-5) and np.allclose(b, g["b_next"], rtol=1e-5)


def test_beam_advance_and_search_goldens(device):
    g = load("beam_advance")
    for tag in "abc":
        ypl = T(g[tag + "_yprevlens"], device) if tag + "_yprevlens" in g.files else None
        out = F.beam_search_advance(T(g[tag + "_lpt"], device), int(g[tag + "_width"]), T(g[tag + "_lpp"], device),
                                    T(g[tag + "_yprev"], device), ypl)  # fmt: skip
        K = g[tag + "_ynext"].shape[2]
        assert np.array_equal(out[0][..., :K].cpu().numpy(), g[tag + "_ynext"]), tag
        assert np.array_equal(out[1].cpu().numpy(), g[tag + "_lens"])
        assert np.array_equal(out[2].cpu().numpy(), g[tag + "_lp"])
        assert np.array_equal(out[3].cpu().numpy(), g[tag + "_src"])
    g = load("beam_search")
    lm = BigramLM(T(g["lm_table"], device)).to(device)
    for tag, (kw, call) in {
        "eos": (dict(width=4, eos=0), dict(batch_size=5, max_iters=12)),
        "all": (dict(width=3, eos=2, finish_all_paths=True), dict(batch_size=2, max_iters=10)),
        "noeos": (dict(width=5), dict(batch_size=3, max_iters=6)),
    }.items():
        y, yl, lp = (x.cpu().numpy() for x in M.BeamSearch(lm, **kw).to(device)(dict(), **call))
        assert y.shape == g["y_" + tag].shape, (tag, y.shape, g["y_" + tag].shape)
        assert np.array_equal(yl, g["lens_" + tag]), tag
        assert np.allclose(lp, g["lp_" + tag], rtol=1e-5, atol=1e-6), tag
        mask = np.arange(y.shape[0])[:, None, None] < yl[None]
        assert np.array_equal(np.where(mask, y, 0), np.where(mask, g["y_" + tag], 0)), tag


def test_image_goldens(device):
    g = load("image")
    for o in (1, 2, 3):
        a = F.polyharmonic_spline(T(g["sp_c"], device), T(g["sp_f"], device), T(g["sp_q"], device), o).cpu().numpy()
        assert np.allclose(a, g["sp_o{}".format(o)], atol=1e-3), o
        a = F.warp_1d_grid(T(g["w1_src"], device), T(g["w1_flow"], device), T(g["w1_lens"], device), 7, o).cpu().numpy()
        valid = np.arange(7)[None] < g["w1_lens"][:, None]
        assert np.abs(a - g["w1_o{}".format(o)])[valid].max() < 1e-4, o
        params = tuple(T(g["sa_p{}".format(i)], device) for i in range(8))
        a = F.spec_augment_apply_parameters(T(g["sa_feats"], device), params, o, T(g["sa_lens"], device)).cpu().numpy()
        valid = np.arange(50)[None, :, None] < g["sa_lens"][:, None, None]
        assert (np.abs(a - g["sa_o{}".format(o)]) * valid).max() < 5e-3, o
    for pinned in (0, 1, 2):
        w, f = F.sparse_image_warp(T(g["siw_img"], device), T(g["siw_src"], device), T(g["siw_dst"], device),
                                   pinned_boundary_points=pinned)  # fmt: skip
        assert np.allclose(w.cpu().numpy(), g["siw_w{}".format(pinned)], atol=1e-3)
        assert np.allclose(f.cpu().numpy(), g["siw_f{}".format(pinned)], atol=1e-3)
        n = F.sparse_image_warp(T(g["siw_img"], device), T(g["siw_src"], device), T(g["siw_dst"], device),
                                pinned_boundary_points=pinned, include_flow=False)  # fmt: skip
        assert np.allclose(n.cpu().numpy(), g["siw_n{}".format(pinned)], atol=1e-3)


def test_tensorflow_addons_arrays(device):
    ld = lambda n: np.load(os.path.join(G, "tfa_" + n + ".npy"))  # noqa: E731
    x, y, q = (T(ld("spline_" + n), device) for n in "xyq")
    for o in (1, 2, 3):
        assert np.allclose(F.polyharmonic_spline(x, y, q, o).cpu().numpy(), ld("spline_o{}".format(o)), atol=1e-3)
    a = F.dense_image_warp(T(ld("dense_img"), device), T(ld("dense_flow"), device)).cpu().numpy()
    assert np.allclose(a, ld("dense_warped"), atol=1e-3)
    for pinned in (0, 2):
        w, f = M.SparseImageWarp(pinned_boundary_points=pinned)(
            T(ld("sparse_img"), device), T(ld("sparse_src"), device), T(ld("sparse_dst"), device))  # fmt: skip
        assert np.allclose(w.cpu().numpy(), ld("sparse_warped_{}".format(pinned)), atol=1e-3)
        assert np.allclose(f.cpu().numpy(), ld("sparse_flow_{}".format(pinned)), atol=1e-3)


def test_search_sweep_with_language_model(device):
    """48 small searches with a bigram LM in the loop, outputs captured from the live reference
    (tests/golden/search_sweep.npz): CTCPrefixSearch with shallow fusion / valid mixture over
    random widths, betas and ragged lens; BeamSearch over eos / finish_all_paths / batch_size /
    max_iters.  Pins the host-side frame loops around the step kernels."""
    from _toy_lm import BigramLM

    g = np.load(os.path.join(G, "search_sweep.npz"))
    for i in range(24):
        tag = "ctc%d_" % i
        K, beta, vm = g[tag + "cfg"]
        lens = g[tag + "lens"]
        lm = BigramLM(torch.from_numpy(g[tag + "table"]).to(device))
        mod = M.CTCPrefixSearch(int(K), float(beta), lm, valid_mixture=bool(vm))
        y, yl, yp = mod(
            torch.from_numpy(g[tag + "logits"]).to(device),
            None if lens[0] == -1 else torch.from_numpy(lens).to(device),
        )
        assert torch.equal(yl.cpu(), torch.from_numpy(g[tag + "y_lens"])), tag
        assert np.allclose(g[tag + "y_probs"], yp.cpu().numpy(), rtol=1e-5, atol=1e-30), (tag, np.abs(yp.cpu().numpy() / g[tag + "y_probs"] - 1).max())
        mask = torch.arange(y.shape[0], device=device).view(-1, 1, 1) < yl.unsqueeze(0)
        assert torch.equal(torch.where(mask, y, torch.zeros_like(y)).cpu(), torch.from_numpy(g[tag + "y"])), tag
    for i in range(24):
        tag = "beam%d_" % i
        K, eos, fin, N, iters = (int(x) for x in g[tag + "cfg"])
        lm = BigramLM(torch.from_numpy(g[tag + "table"]).to(device))
        mod = M.BeamSearch(lm, K, None if eos == -1000 else eos, bool(fin), -7).to(device)
        y, yl, lp = mod(dict(), None if N == -1 else N, iters)
        ey, eyl, elp = (torch.from_numpy(g[tag + k]) for k in ("y", "y_lens", "lp"))
        assert y.shape == ey.shape and torch.equal(yl.cpu(), eyl), (tag, y.shape, ey.shape)
        assert np.allclose(elp.numpy(), lp.cpu().numpy(), rtol=1e-5, atol=1e-6, equal_nan=True), tag
        S = y.shape[0]
        mask = torch.arange(S).view([S] + [1] * (y.dim() - 1)) < eyl.unsqueeze(0)
        assert torch.equal(torch.where(mask, y.cpu(), ey), ey), tag
