"""GPU parity of the spline / warp / SpecAugment kernels vs the numpy (float64) oracle.

Tolerances: the kernels evaluate gathers in float32 like torch's grid_sample; spline systems
are solved in float64.  1e-4 absolute on O(1) data (the reference's own test tolerance,
tests/test_img.py:197,223)."""
import numpy as np
import pytest
import torch

import oracle
from pydrobert_amd import functional as F
from pydrobert_amd import modules as M

pytestmark = pytest.mark.gpu
ATOL = 1e-4


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.mark.parametrize("order", [1, 2, 3])
def test_polyharmonic_spline(device, order):
    rng = np.random.default_rng(order)
    for (N, T, I, O, Q) in [(3, 6, 2, 2, 20), (20, 10, 1, 1, 50), (2, 30, 3, 4, 300)]:
        c = rng.uniform(-2, 2, (N, T, I)).astype(np.float32)
        f = rng.normal(size=(N, T, O)).astype(np.float32)
        q = rng.uniform(-2, 2, (N, Q, I)).astype(np.float32)
        exp = oracle.polyharmonic_spline(c, f, q, order)
        act = F.polyharmonic_spline(_t(c, device), _t(f, device), _t(q, device), order).cpu().numpy()
        assert np.allclose(exp, act, atol=ATOL * max(1.0, np.abs(exp).max())), np.abs(exp - act).max()
        # interpolation property: the spline passes through its control points
        back = F.polyharmonic_spline(_t(c, device), _t(f, device), _t(c, device), order).cpu().numpy()
        assert np.allclose(back, f, atol=1e-3)


@pytest.mark.parametrize("order", [1, 2, 3])
def test_warp_1d_grid(device, order):
    src = np.array([10.0, 12.0, 15.0, 3.0, 6.0, 0.0], np.float32)
    flow = np.array([2.0, -3.0, 1.5, 1.0, -2.0, 4.0], np.float32)
    lens = np.array([20.0, 25.0, 30.0, 7.0, 12.0, 30.0], np.float32)
    exp = oracle.warp_1d_grid(src, flow, lens, 30, order)
    act = M.Warp1DGrid(30, order)(_t(src, device), _t(flow, device), _t(lens, device)).cpu().numpy()
    valid = np.arange(30)[None] < lens[:, None]
    assert np.abs(exp - act)[valid].max() < 1e-5
    # max_length inferred from lengths (:279)
    assert F.warp_1d_grid(_t(src, device), _t(flow, device), _t(lens, device)).shape == (6, 30)


def _draw(rng, N, T, Fq, lens, MT=2, MF=2, freq=True):
    W = np.minimum(lens / 2 - 1, 5.0)
    w_0 = (rng.uniform(size=N) * (lens - 2 * W) + W).astype(np.float32)
    w = (rng.uniform(size=N) * W - W / 2).astype(np.float32)  # keeps the target away from the ends
    v_0 = (rng.uniform(size=N) * (Fq - 4) + 2).astype(np.float32) if freq else np.zeros(0, np.float32)
    v = (rng.uniform(size=N) * 2 - 1).astype(np.float32) if freq else np.zeros(0, np.float32)
    t = rng.integers(0, 6, (N, MT))
    t_0 = (rng.uniform(size=(N, MT)) * (lens[:, None] - t)).astype(np.int64)
    f = rng.integers(0, 4, (N, MF))
    f_0 = (rng.uniform(size=(N, MF)) * (Fq - f)).astype(np.int64)
    return w_0, w, v_0, v, t_0, t, f_0, f


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("freq", [False, True])
def test_spec_augment_apply_parameters(device, order, freq):
    rng = np.random.default_rng(10 * order + freq)
    N, T, Fq = 6, 50, 10
    feats = rng.normal(size=(N, T, Fq)).astype(np.float32)
    lens = np.array([50, 37, 20, 25, 44, 31])
    params = _draw(rng, N, T, Fq, lens.astype(np.float64), freq=freq)
    exp = oracle.spec_augment_apply_parameters(feats, params, order, lens)
    act = F.spec_augment_apply_parameters(
        _t(feats, device), tuple(_t(p, device) for p in params), order, _t(lens, device)
    ).cpu().numpy()
    valid = np.arange(T)[None, :, None] < lens[:, None, None]
    err = np.abs(exp - act) * valid
    assert err.max() < ATOL, err.max()
    # masked bands are exactly zero
    t_0, t = params[4], params[5]
    for n in range(N):
        for m in range(t.shape[1]):
            assert (act[n, t_0[n, m] : t_0[n, m] + t[n, m]] == 0).all()


@pytest.mark.parametrize("order", [1, 2, 3])
def test_spec_augment_time_warp_in_one_launch(device, order):
    """``spec_augment_apply_parameters`` with a time warp and no frequency warp is ONE kernel (round 5):
    warp_1d_grid's three-knot spline is solved in closed form inside the pass over the features
    (csrc/img_warp.hip warp_1d_spline).  Against the two-operator route through an explicit grid (the
    Gauss-Jordan solve of ``warp_1d_grid``), against the float64 oracle on the valid frames, for lengths
    down to 1 / 2 / 3 frames, with and without masks, strided features (the grid route again), and the
    gradient of the fused operator against the grid route's."""
    rng = np.random.default_rng(60 + order)
    N, T, Fq = 24, 90, 12
    feats = rng.normal(size=(N, T, Fq)).astype(np.float32)
    lens = rng.integers(8, T + 1, N)
    lens[:6] = [1, 2, 3, T, T - 1, 4]
    # (targets kept away from the ends, as _draw does: a target clamped onto an end knot -- an eps away
    # from it -- makes the 5 x 5 system singular to working precision, and every solver, the float64
    # oracle's included, returns its own numbers there; the tiny lengths are checked for sanity only)
    Wd = np.maximum(np.minimum(lens / 2 - 1, 8.0), 0.0)
    w_0 = (rng.random(N) * (lens - 2 * Wd) + Wd).astype(np.float32)
    w = (rng.random(N) * Wd - Wd / 2).astype(np.float32)
    t_0 = rng.integers(0, T - 5, (N, 2)); t = rng.integers(0, 5, (N, 2))
    f_0 = rng.integers(0, Fq - 2, (N, 1)); f = rng.integers(0, 3, (N, 1))
    e = torch.empty(0)
    x, ln = _t(feats, device), _t(lens, device)
    for masks in (False, True):
        mk = (_t(t_0, device), _t(t, device), _t(f_0, device), _t(f, device)) if masks else (e, e, e, e)
        params = (_t(w_0, device), _t(w, device), e, e) + mk
        act = F.spec_augment_apply_parameters(x, params, order, ln)
        grid = F.warp_1d_grid(params[0], params[1], ln, T, order)
        two = torch.ops.pydrobert_amd.spec_augment_apply(
            x, grid, None, *(m.long().contiguous() if m.numel() else None for m in mk))
        exp = oracle.spec_augment_apply_parameters(feats, (w_0, w, None, None) + ((t_0, t, f_0, f) if masks else (None,) * 4),
                                                    order, lens)
        valid = np.arange(T)[None, :, None] < lens[:, None, None]
        a, b = act.cpu().numpy(), two.cpu().numpy()
        ok = lens >= 8
        assert np.isfinite(a).all()
        assert np.abs(np.where(valid, a - exp, 0))[ok].max() < 1e-4, order
        # the two solves agree far inside the oracle tolerance
        assert np.abs(np.where(valid, a - b, 0))[ok].max() < 2e-5, order
    # strided features: the general kernel through the grid, same numbers as on a contiguous copy
    xs = torch.randn(N, T, 2 * Fq, device=device)[:, :, ::2]
    params = (_t(w_0, device), _t(w, device), e, e, e, e, e, e)
    vm = torch.from_numpy(valid & ok[:, None, None]).to(device)  # (beyond a length an order >= 2 spline is noise)
    d = F.spec_augment_apply_parameters(xs, params, order, ln) - F.spec_augment_apply_parameters(xs.contiguous(), params, order, ln)
    assert float((d * vm).abs().max()) < 2e-5
    # gradient of the fused operator = gradient through the explicit grid
    xg = x.clone().requires_grad_(True)
    g_out = torch.randn(N, T, Fq, device=device)
    (g1,) = torch.autograd.grad(F.spec_augment_apply_parameters(xg, params, order, ln), xg, g_out)
    grid = F.warp_1d_grid(params[0], params[1], ln, T, order)
    (g2,) = torch.autograd.grad(torch.ops.pydrobert_amd.spec_augment_apply(xg, grid, None, None, None, None, None), xg, g_out)
    assert torch.allclose(g1[_t(ok, device)], g2[_t(ok, device)], atol=1e-4)
    # lengths out of range: the verdict comes after the launch, the error is the reference's
    with pytest.raises(RuntimeError, match="values of lengths"):
        F.spec_augment_apply_parameters(x, params, order, torch.full((N,), T + 1, device=device))


def test_spec_augment_forward_behind_one_operator(device):
    """Training-mode ``SpecAugment.forward`` with a time warp and no frequency warp runs the draw and the
    one-pass application behind one operator (round 5): the same numbers as ``apply_parameters`` of
    ``draw_parameters`` from the same generator state, gradients included; a subclass with its own draw
    is served through it; bad lengths raise the reference's error."""
    N, T, Fq = 12, 120, 16
    feats = torch.randn(N, T, Fq, device=device)
    lens = torch.randint(30, T + 1, (N,), device=device)
    for kw in (dict(), dict(num_freq_mask=0), dict(max_time_mask=0), dict(interpolation_order=2)):
        sa = M.SpecAugment(max_time_warp=12.0, max_freq_warp=0.0, **kw).to(device)
        for ln in (lens, None):
            torch.manual_seed(7)
            a = sa(feats, ln)
            torch.manual_seed(7)
            b = sa.apply_parameters(feats, sa.draw_parameters(feats, ln), ln)
            assert torch.equal(a, b), kw
    sa = M.SpecAugment(max_time_warp=12.0, max_freq_warp=0.0).to(device)
    x = feats.clone().requires_grad_(True)
    torch.manual_seed(9)
    (g1,) = torch.autograd.grad(sa(x, lens), x, torch.ones_like(x))
    torch.manual_seed(9)
    (g2,) = torch.autograd.grad(sa.apply_parameters(x, sa.draw_parameters(x, lens), lens), x, torch.ones_like(x))
    assert torch.allclose(g1, g2, atol=1e-5)

    class NoWarp(M.SpecAugment):
        def draw_parameters(self, feats, lengths=None):
            p = super().draw_parameters(feats, lengths)
            return (p[0], torch.zeros_like(p[1])) + tuple(p[2:])

    nw = NoWarp(max_time_warp=12.0, max_freq_warp=0.0, max_time_mask=0, max_freq_mask=0).to(device)
    vm = (torch.arange(T, device=device)[None, :] < lens[:, None]).unsqueeze(-1)
    assert float(((nw(feats, lens) - feats) * vm).abs().max()) < 1e-4  # (a zero flow: the identity on the valid frames)
    with pytest.raises(RuntimeError, match="values of lengths"):
        sa(feats, torch.full((N,), T + 3, device=device))
    with pytest.raises(RuntimeError, match="values of lengths"):
        sa(feats, torch.zeros(N, dtype=torch.long, device=device))


def test_spec_augment_masks_only_and_identity(device):
    rng = np.random.default_rng(3)
    N, T, Fq = 3, 40, 8
    feats = rng.normal(size=(N, T, Fq)).astype(np.float32)
    e = np.zeros(0, np.float32)
    t_0, t = np.array([[3], [10], [0]]), np.array([[4], [0], [40]])
    params = (e, e, e, e, t_0, t, np.zeros(0, np.int64), np.zeros(0, np.int64))
    act = F.spec_augment_apply_parameters(_t(feats, device), tuple(_t(p, device) for p in params), 1).cpu().numpy()
    exp = oracle.spec_augment_apply_parameters(feats, params, 1)
    assert np.array_equal(exp, act)
    sa = M.SpecAugment().eval()
    x = _t(feats, device)
    assert sa(x) is x
    # strided (non-contiguous) input
    xt = _t(np.ascontiguousarray(feats.transpose(1, 0, 2)), device).transpose(0, 1)
    act2 = F.spec_augment_apply_parameters(xt, tuple(_t(p, device) for p in params), 1).cpu().numpy()
    assert np.array_equal(exp, act2)


def test_spec_augment_module_statistics_and_grad(device):
    """Ranges / counts of the drawn parameters (as the reference's tests/test_img.py:226-281
    does) and the gradient w.r.t. feats against torch's own grid_sample graph."""
    torch.manual_seed(0)
    N, T, Fq = 16, 200, 20
    feats = torch.randn(N, T, Fq, device=device)
    lens = torch.randint(100, T + 1, (N,), device=device)
    sa = M.SpecAugment(max_time_warp=10.0, max_freq_warp=2.0, max_time_mask=20, max_freq_mask=5,
                       max_time_mask_proportion=0.1, num_time_mask=3, num_time_mask_proportion=0.02,
                       num_freq_mask=2, interpolation_order=1)  # fmt: skip
    w_0, w, v_0, v, t_0, t, f_0, f = sa.draw_parameters(feats, lens)
    lf = lens.float()
    assert w_0.shape == (N,) and (w_0 >= 0).all() and (w_0 <= lf).all() and (w.abs() <= 10).all()
    assert (v.abs() <= 2).all() and (v_0 >= 0).all() and (v_0 <= Fq).all()
    assert t.shape == (N, 3) and (t <= 20).all() and (t_0 + t <= lens.unsqueeze(1)).all()
    assert ((t > 0).sum(1) <= (lf * 0.02).clamp(max=3).floor()).all()
    assert f.shape == (N, 2) and (f <= 5).all() and (f_0 + f <= Fq).all()
    out = sa(feats, lens)
    assert out.shape == feats.shape and torch.isfinite(out).all()
    # gradient
    x = feats.clone().requires_grad_(True)
    y = F.spec_augment_apply_parameters(x, (w_0, w, v_0, v, t_0, t, f_0, f), 1, lens)
    g_out = torch.randn_like(y)
    (g,) = torch.autograd.grad(y, x, g_out)
    exp = oracle.spec_augment_apply_parameters
    eps_dir = torch.randn_like(feats)
    # directional derivative check (the op is linear in feats)
    y2 = F.spec_augment_apply_parameters(feats + eps_dir, (w_0, w, v_0, v, t_0, t, f_0, f), 1, lens)
    y1 = F.spec_augment_apply_parameters(feats, (w_0, w, v_0, v, t_0, t, f_0, f), 1, lens)
    lhs = ((y2 - y1) * g_out).sum()
    rhs = (g * eps_dir).sum()
    assert torch.allclose(lhs, rhs, rtol=1e-3, atol=1e-2), (lhs.item(), rhs.item())


def test_spec_augment_draw_kernel_maps_draws_like_the_reference(device):
    """The one-kernel draw (pydrobert_amd::spec_augment_draw): given the SAME uniform draws it returns
    the live reference's parameters bit for bit (tests/golden/spec_draw.npz: torch.rand pinned while the
    reference ran), and on a large random batch the oracle's restatement of that mapping; the functional
    returns zero-size pairs for disabled groups like the reference."""
    from test_oracle_golden import spec_draw_cases

    def run(u, T, Fq, cfg, lens):
        out = torch.ops.pydrobert_amd.spec_augment_draw(
            _t(u, device), None if lens is None else _t(np.asarray(lens), device), T, Fq, float(cfg["max_time_warp"]),
            float(cfg["max_freq_warp"]), cfg["max_time_mask"], cfg["max_freq_mask"], float(cfg["max_time_mask_proportion"]),
            cfg["num_time_mask"], float(cfg["num_time_mask_proportion"]), cfg["num_freq_mask"], False)
        return [o.cpu().numpy() for o in out]

    for u, T, Fq, cfg, lens, exp in spec_draw_cases():
        for o, e in zip(run(u, T, Fq, cfg, lens), exp):
            assert o.size == e.size and (o.size == 0 or np.array_equal(o.reshape(e.shape), e)), cfg
    rng = np.random.default_rng(12)
    cfg = dict(max_time_warp=80.0, max_freq_warp=3.0, max_time_mask=100, max_freq_mask=27, max_time_mask_proportion=0.04,
               num_time_mask=2, num_time_mask_proportion=1.0, num_freq_mask=2)
    N, T, Fq = 2048, 1000, 80
    u = rng.random((N, 12)).astype(np.float32)
    lens = rng.integers(1, T + 1, N)
    exp = oracle.spec_augment_parameters_from_uniforms(u, T, Fq, lengths=lens, **cfg)
    for o, e in zip(run(u, T, Fq, cfg, lens), exp):
        assert np.array_equal(o, e)
    feats = torch.zeros((4, 30, 6), device=device)
    out = F.spec_augment_draw_parameters(feats, 0.0, 0.0, 5, 0, 0.5, 2, 1.0, 3)
    assert [o.numel() for o in out] == [0, 0, 0, 0, 8, 8, 0, 0]


@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
@pytest.mark.parametrize("padding", ["border", "zeros", "reflection"])
def test_dense_image_warp(device, mode, padding):
    rng = np.random.default_rng(5)
    img = rng.normal(size=(2, 3, 9, 7)).astype(np.float32)
    flow = rng.normal(size=(2, 9, 7, 2)).astype(np.float32) * 2.5
    if mode == "nearest":  # keep away from .5 rounding boundaries
        flow = np.round(flow * 4) / 4 + 0.1
    for indexing in ("hw", "wh"):
        exp = oracle.dense_image_warp(img, flow, indexing, mode, padding)
        act = M.DenseImageWarp(indexing, mode, padding)(_t(img, device), _t(flow, device)).cpu().numpy()
        assert np.allclose(exp, act, atol=1e-5), (indexing, np.abs(exp - act).max())


@pytest.mark.parametrize("pinned", [0, 1, 2])
@pytest.mark.parametrize("include_flow", [True, False])
def test_sparse_image_warp(device, pinned, include_flow):
    rng = np.random.default_rng(7 + pinned)
    N, C, H, W, Mp = 2, 2, 12, 9, 4
    img = rng.normal(size=(N, C, H, W)).astype(np.float32)  # (pixels of both signs: the fast kernels move their bits)
    src = (rng.uniform(size=(N, Mp, 2)) * [H - 1, W - 1]).astype(np.float32)
    dst = (src + rng.normal(size=(N, Mp, 2))).astype(np.float32)
    for indexing in ("hw", "wh"):
        s, d = (src, dst) if indexing == "hw" else (src[..., ::-1].copy(), dst[..., ::-1].copy())
        for order in (1, 2, 3):
            exp = oracle.sparse_image_warp(img, s, d, indexing, order, pinned_boundary_points=pinned,
                                           include_flow=include_flow)  # fmt: skip
            act = F.sparse_image_warp(_t(img, device), _t(s, device), _t(d, device), indexing, order,
                                      pinned_boundary_points=pinned, include_flow=include_flow)  # fmt: skip
            if include_flow:
                assert np.allclose(exp[1], act[1].cpu().numpy(), atol=2e-4), np.abs(exp[1] - act[1].cpu().numpy()).max()
                exp, act = exp[0], act[0]
            assert np.allclose(exp, act.cpu().numpy(), atol=5e-4), (indexing, order, np.abs(exp - act.cpu().numpy()).max())
    # no control points: identity (:543-548)
    e = torch.zeros((N, 0, 2), device=device)
    out = F.sparse_image_warp(_t(img, device), e, e, include_flow=False)
    assert torch.equal(out.cpu(), torch.from_numpy(img))


def test_float64_images_are_sampled_in_float64(device):
    """A float64 image goes through the float64 instantiation of the gather (pdt_*_f64): the live
    reference's outputs (tests/golden/image_f64.npz) to 1e-12 for dense_image_warp in every mode /
    padding / indexing, to the float32 spline's accuracy for sparse_image_warp with its flow; a larger
    random case against the oracle; the adjoint in float64; no narrowing warning."""
    import warnings
    from test_oracle_golden import load as _load

    g = _load("image_f64")
    img, flow = _t(g["img"], device), _t(g["flow"], device)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for mode in ("bilinear", "nearest"):
            for pad in ("border", "zeros", "reflection"):
                for ind in ("hw", "wh"):
                    a = F.dense_image_warp(img, flow, ind, mode, pad)
                    e = g["dense_{}_{}_{}".format(mode, pad, ind)]
                    assert a.dtype == torch.double
                    assert np.allclose(a.cpu().numpy(), e, atol=1e-12, rtol=0), (mode, pad, ind, np.abs(a.cpu().numpy() - e).max())
        for order in (1, 2, 3):
            w, f = F.sparse_image_warp(img, _t(g["src"], device), _t(g["dst"], device), field_interpolation_order=order)
            assert w.dtype == torch.double and f.dtype == torch.float
            assert np.allclose(w.cpu().numpy(), g["sparse_w{}".format(order)], atol=2e-3), order
            assert np.allclose(f.cpu().numpy(), g["sparse_f{}".format(order)], atol=1e-3), order
        rng = np.random.default_rng(31)
        N, C, H, W = 3, 2, 37, 29
        big = rng.normal(size=(N, C, H, W))
        fl = (rng.normal(size=(N, H, W, 2)) * 6).astype(np.float32)
        for pad in ("border", "zeros", "reflection"):
            a = F.dense_image_warp(_t(big, device), _t(fl, device), "hw", "bilinear", pad).cpu().numpy()
            e = oracle.dense_image_warp(big, fl, "hw", "bilinear", pad)
            assert np.allclose(a, e, atol=1e-12, rtol=0), (pad, np.abs(a - e).max())
        # the same flow applied to the float32 copy differs from the float64 result by float32 rounding
        a32 = F.dense_image_warp(_t(big.astype(np.float32), device), _t(fl, device)).cpu().numpy()
        a64 = F.dense_image_warp(_t(big, device), _t(fl, device)).cpu().numpy()
        assert 1e-9 < np.abs(a32 - a64).max() < 1e-4
        # adjoint (the op is linear in the image), float64 atomics
        x = _t(big, device)
        lhs, rhs, _ = _adjoint_gap(lambda t: F.dense_image_warp(t, _t(fl, device)), x)
        assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(lhs)), (lhs, rhs)
        src = _t((rng.uniform(size=(N, 4, 2)) * [H - 1, W - 1]).astype(np.float32), device)
        dst = src + _t(rng.normal(size=(N, 4, 2)).astype(np.float32), device)
        lhs, rhs, _ = _adjoint_gap(lambda t: F.sparse_image_warp(t, src, dst, pinned_boundary_points=1, include_flow=False), x)
        assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(lhs)), (lhs, rhs)


def _adjoint_gap(op, x):
    """<op(x + d) - op(x), g> vs <d, op^T g> for an operator that is linear in x."""
    x = x.clone().requires_grad_(True)
    y = op(x)
    g = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, x, g)
    d = torch.randn_like(x)
    lhs = ((op(x.detach() + d) - y.detach()) * g).sum()
    rhs = (gx * d).sum()
    return lhs.item(), rhs.item(), gx


@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
@pytest.mark.parametrize("padding", ["border", "zeros", "reflection"])
def test_dense_image_warp_backward(device, mode, padding):
    """HIP adjoint kernel against autograd through torch's own grid_sample (what the reference
    differentiates, _img.py:436)."""
    torch.manual_seed(11)
    N, C, H, W = 3, 2, 14, 11
    img = torch.randn(N, C, H, W, device=device)
    flow = torch.randn(N, H, W, 2, device=device) * 3
    if mode == "nearest":
        flow = (flow * 4).round() / 4 + 0.1
    lhs, rhs, gx = _adjoint_gap(lambda x: F.dense_image_warp(x, flow, "hw", mode, padding), img)
    assert abs(lhs - rhs) <= 1e-3 * max(1.0, abs(lhs)), (lhs, rhs)
    # direct comparison with grid_sample's backward
    x = img.clone().requires_grad_(True)
    hh, ww = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
    base = torch.stack((ww, hh), 2).unsqueeze(0).float()
    grid = (2 * base - 2 * flow.flip(-1) + 1.0) / torch.tensor([W, H], device=device).float() - 1.0
    y = torch.nn.functional.grid_sample(x, grid, mode=mode, padding_mode=padding, align_corners=False)
    torch.manual_seed(12)
    g = torch.randn_like(y)
    (exp,) = torch.autograd.grad(y, x, g)
    x2 = img.clone().requires_grad_(True)
    (act,) = torch.autograd.grad(F.dense_image_warp(x2, flow, "hw", mode, padding), x2, g)
    assert torch.allclose(exp, act, atol=1e-4), (exp - act).abs().max().item()


@pytest.mark.parametrize("include_flow", [True, False])
def test_sparse_image_warp_backward(device, include_flow):
    torch.manual_seed(13)
    N, C, H, W, Mp = 2, 3, 12, 10, 5
    img = torch.rand(N, C, H, W, device=device)
    src = torch.rand(N, Mp, 2, device=device) * torch.tensor([H - 1.0, W - 1.0], device=device)
    dst = src + torch.randn(N, Mp, 2, device=device)

    def op(x):
        out = F.sparse_image_warp(x, src, dst, pinned_boundary_points=1, include_flow=include_flow)
        return out[0] if include_flow else out

    lhs, rhs, _ = _adjoint_gap(op, img)
    assert abs(lhs - rhs) <= 1e-3 * max(1.0, abs(lhs)), (lhs, rhs)


def test_spec_augment_backward_matches_grid_sample(device):
    """Gradient of the fused apply kernel's adjoint against autograd through the reference's
    formulation: grid_sample over the outer product of the two 1-D grids, then masks."""
    torch.manual_seed(14)
    N, T, Fq = 5, 60, 12
    feats = torch.randn(N, T, Fq, device=device)
    lens = torch.tensor([60, 45, 33, 60, 20], device=device)
    for params in (
        (torch.tensor([20.0, 10, 15, 30, 8], device=device), torch.tensor([3.0, -2, 1.5, -4, 2], device=device),
         torch.tensor([5.0, 6, 4, 7, 5], device=device), torch.tensor([1.0, -1, 0.5, 1.5, -0.5], device=device),
         torch.tensor([[3], [0], [10], [50], [2]], device=device), torch.tensor([[4], [2], [0], [5], [3]], device=device),
         torch.tensor([[1], [8], [3], [0], [6]], device=device), torch.tensor([[2], [3], [1], [0], [4]], device=device)),
        (torch.tensor([20.0, 10, 15, 30, 8], device=device), torch.tensor([3.0, -2, 1.5, -4, 2], device=device),
         torch.empty(0), torch.empty(0), torch.empty(0), torch.empty(0), torch.empty(0), torch.empty(0)),
        (torch.empty(0), torch.empty(0), torch.empty(0), torch.empty(0),
         torch.tensor([[3], [0], [10], [50], [2]], device=device), torch.tensor([[4], [2], [0], [5], [3]], device=device),
         torch.empty(0), torch.empty(0)),
    ):  # fmt: skip
        w_0, w, v_0, v, t_0, t, f_0, f = params
        x = feats.clone().requires_grad_(True)
        y = F.spec_augment_apply_parameters(x, params, 1, lens)
        g = torch.randn_like(y)
        (act,) = torch.autograd.grad(y, x, g)
        # the reference's graph
        x2 = feats.clone().requires_grad_(True)
        tg = (2 * torch.arange(T, device=device).float() + 1) / T - 1
        fg = (2 * torch.arange(Fq, device=device).float() + 1) / Fq - 1
        tgrid = F.warp_1d_grid(w_0, w, lens, T, 1) if w_0.numel() else tg.expand(N, T)
        fgrid = (
            F.warp_1d_grid(v_0, v, torch.full((N,), Fq, device=device), Fq, 1) if v_0.numel() else fg.expand(N, Fq)
        )
        y2 = x2
        if w_0.numel() or v_0.numel():
            grid = torch.stack([fgrid.unsqueeze(1).expand(N, T, Fq), tgrid.unsqueeze(2).expand(N, T, Fq)], 3)
            y2 = torch.nn.functional.grid_sample(
                x2.unsqueeze(1), grid, mode="bilinear", padding_mode="border", align_corners=False
            ).squeeze(1)
        if t_0.numel():
            ar = torch.arange(T, device=device).view(1, T, 1)
            m = ((ar >= t_0.unsqueeze(1)) & (ar < (t_0 + t).unsqueeze(1))).any(2, keepdim=True)
            y2 = y2.masked_fill(m, 0.0)
        if f_0.numel():
            ar = torch.arange(Fq, device=device).view(1, Fq, 1)
            m = ((ar >= f_0.unsqueeze(1)) & (ar < (f_0 + f).unsqueeze(1))).any(2).unsqueeze(1)
            y2 = y2.masked_fill(m, 0.0)
        assert torch.allclose(y, y2, atol=1e-5)
        (exp,) = torch.autograd.grad(y2, x2, g)
        assert torch.allclose(exp, act, atol=1e-4), (exp - act).abs().max().item()


@pytest.mark.parametrize("monotone", [True, False])
def test_spec_augment_rows_backward_any_grid(device, monotone):
    """The gather-form adjoint (time grid + masks, F % 4 == 0) for a non-decreasing grid and for
    an arbitrary one (where its row ranges degrade to full scans), both against autograd
    through grid_sample; tiles: T > 256 rows."""
    torch.manual_seed(21 + monotone)
    N, T, Fq = 3, 700, 8
    feats = torch.randn(N, T, Fq, device=device)
    tgrid = torch.rand(N, T, device=device) * 2.4 - 1.2  # some samples clip at both borders
    if monotone:
        tgrid = tgrid.sort(1).values
    t_0 = torch.tensor([[5, 300], [0, 650], [100, 100]], device=device)
    t = torch.tensor([[10, 40], [3, 50], [0, 7]], device=device)
    f_0 = torch.tensor([[1], [6], [0]], device=device)
    f = torch.tensor([[2], [2], [0]], device=device)
    x = feats.clone().requires_grad_(True)
    y = torch.ops.pydrobert_amd.spec_augment_apply(x, tgrid, None, t_0, t, f_0, f)
    g = torch.randn_like(y)
    (act,) = torch.autograd.grad(y, x, g)
    x2 = feats.clone().requires_grad_(True)
    fg = (2 * torch.arange(Fq, device=device).float() + 1) / Fq - 1
    grid = torch.stack([fg.view(1, 1, Fq).expand(N, T, Fq), tgrid.unsqueeze(2).expand(N, T, Fq)], 3)
    y2 = torch.nn.functional.grid_sample(
        x2.unsqueeze(1), grid, mode="bilinear", padding_mode="border", align_corners=False
    ).squeeze(1)
    ar = torch.arange(T, device=device).view(1, T, 1)
    y2 = y2.masked_fill(((ar >= t_0.unsqueeze(1)) & (ar < (t_0 + t).unsqueeze(1))).any(2, keepdim=True), 0.0)
    ar = torch.arange(Fq, device=device).view(1, Fq, 1)
    y2 = y2.masked_fill(((ar >= f_0.unsqueeze(1)) & (ar < (f_0 + f).unsqueeze(1))).any(2).unsqueeze(1), 0.0)
    # float32 coordinate rounding at T = 700 moves taps by ~T * 2^-24 rows: compare at 2e-3
    assert torch.allclose(y, y2, atol=2e-3)
    (exp,) = torch.autograd.grad(y2, x2, g)
    assert torch.allclose(exp, act, atol=2e-3), (exp - act).abs().max().item()
    # deterministic: the same call twice gives the same bits
    (again,) = torch.autograd.grad(torch.ops.pydrobert_amd.spec_augment_apply(x, tgrid, None, t_0, t, f_0, f), x, g)
    assert torch.equal(act, again)


# ---------------------------------------------------------------------------------------
# gradients with respect to points / values / flow (the reference's ops are plain torch graphs)
# ---------------------------------------------------------------------------------------
def _torch_phi(r, k):
    eps = float(torch.finfo(torch.float).eps)
    return r**k if k % 2 else r**k * torch.log(r.clamp(min=eps))


def _torch_spline(c, f, x, k, reg=0.0):
    """The reference's spline as a float64 torch graph (_img.py:67-130), for autograd."""
    N, T, I = c.shape
    cdist = lambda a, b: torch.cdist(a, b, compute_mode="donot_use_mm_for_euclid_dist")  # noqa: E731 (exact differences)
    A = _torch_phi(cdist(c, c), k) + reg * torch.eye(T, dtype=c.dtype)
    B = torch.cat([c, torch.ones_like(c[..., :1])], 2)
    M_ = torch.cat([torch.cat([A, B], 2), torch.cat([B.transpose(1, 2), c.new_zeros(N, I + 1, I + 1)], 2)], 1)
    sol = torch.linalg.solve(M_, torch.cat([f, f.new_zeros(N, I + 1, f.shape[2])], 1))
    return _torch_phi(cdist(x, c), k) @ sol[:, :T] + torch.cat([x, torch.ones_like(x[..., :1])], 2) @ sol[:, T:]


@pytest.mark.parametrize("order", [1, 2, 3])
def test_polyharmonic_spline_gradients(device, order):
    rng = np.random.default_rng(40 + order)
    for reg in (0.0, 0.05):
        N, T, I, O, Q = 2, 7, 2, 3, 11
        c = (rng.uniform(size=(N, T, I)) * 4).astype(np.float32)
        f = rng.normal(size=(N, T, O)).astype(np.float32)
        x = (rng.uniform(size=(N, Q, I)) * 4).astype(np.float32)
        G = rng.normal(size=(N, Q, O)).astype(np.float32)
        ref = [torch.from_numpy(a).double().requires_grad_(True) for a in (c, f, x)]
        (_torch_spline(*ref, order, reg) * torch.from_numpy(G).double()).sum().backward()
        dev = [_t(a, device).requires_grad_(True) for a in (c, f, x)]
        out = F.polyharmonic_spline(*dev, order, reg)
        (out * _t(G, device)).sum().backward()
        for name, a, b in zip("cfx", dev, ref):
            err = (a.grad.cpu().double() - b.grad).abs().max() / b.grad.abs().max()
            assert err < 1e-3, (name, order, reg, float(err))


def test_polyharmonic_spline_many_points(device):
    """Systems beyond the LDS (more than ~140 unknowns) are eliminated in the workspace."""
    rng = np.random.default_rng(9)
    N, T, I, O, Q = 2, 180, 2, 2, 40
    c = rng.uniform(-3, 3, (N, T, I)).astype(np.float32)
    f = rng.normal(size=(N, T, O)).astype(np.float32)
    x = rng.uniform(-3, 3, (N, Q, I)).astype(np.float32)
    for order in (1, 3):
        exp = oracle.polyharmonic_spline(c, f, x, order)
        act = F.polyharmonic_spline(_t(c, device), _t(f, device), _t(x, device), order).cpu().numpy()
        assert np.allclose(exp, act, atol=2e-3 * max(1.0, np.abs(exp).max())), np.abs(exp - act).max()
        back = F.polyharmonic_spline(_t(c, device), _t(f, device), _t(c, device), order).cpu().numpy()
        assert np.allclose(back, f, atol=5e-3)


def test_warp_1d_grid_gradients(device):
    # (knots kept off the integer frames: with order 1 the spline has a kink wherever a query
    # point meets a knot, and every implementation picks its own subgradient there)
    src = torch.tensor([10.3, 12.7, 15.2, 3.4])
    flow = torch.tensor([2.1, -3.3, 1.5, 1.2])
    lens = torch.tensor([20.0, 25.0, 30.0, 7.0])
    T = 30
    for order in (1, 2):
        s, f = src.double().requires_grad_(True), flow.double().requires_grad_(True)
        eps = float(torch.finfo(torch.float).eps)
        ss = torch.min(s, lens.double() - 1).clamp_min(0)
        dd = torch.min(ss + f, lens.double() - 1).clamp_min(0)
        ss, dd = (2 * ss + 1) / T - 1, (2 * dd + 1) / T - 1
        lo = torch.full_like(ss, 1 / T - 1 - eps)
        up = (2 * lens.double() - 1) / T - 1 + eps
        t = ((2.0 * torch.arange(T) + 1) / T - 1).double().expand(4, T)
        grid = _torch_spline(torch.stack([lo, dd, up], 1).unsqueeze(-1), torch.stack([lo, ss, up], 1).unsqueeze(-1),
                             t.unsqueeze(-1), order).squeeze(-1)  # fmt: skip
        w = torch.randn(4, T, dtype=torch.double)
        (grid * w).sum().backward()
        s2, f2 = src.to(device).requires_grad_(True), flow.to(device).requires_grad_(True)
        g2 = F.warp_1d_grid(s2, f2, lens.to(device), T, order)
        (g2 * w.float().to(device)).sum().backward()
        assert torch.allclose(s2.grad.cpu().double(), s.grad, rtol=2e-3, atol=1e-5), order
        assert torch.allclose(f2.grad.cpu().double(), f.grad, rtol=2e-3, atol=1e-5), order


@pytest.mark.parametrize("include_flow", [True, False])
def test_warps_gradients_wrt_flow_and_points(device, include_flow):
    """dense_image_warp w.r.t. the flow and sparse_image_warp w.r.t. both point sets, against the
    reference's formulation as a float64 torch graph (spline -> grid -> grid_sample)."""
    rng = np.random.default_rng(21)
    N, C, H, W, Mp = 2, 2, 9, 7, 4
    img = torch.from_numpy(rng.uniform(size=(N, C, H, W)))
    flow = torch.from_numpy(rng.normal(size=(N, H, W, 2)) * 0.7)
    w = torch.from_numpy(rng.normal(size=(N, C, H, W)))
    hh, ww = torch.meshgrid(torch.arange(H, dtype=torch.double), torch.arange(W, dtype=torch.double), indexing="ij")
    xy = torch.stack((ww, hh), 2).unsqueeze(0)
    size = torch.tensor([W, H], dtype=torch.double)

    fl = flow.clone().requires_grad_(True)
    out = torch.nn.functional.grid_sample(img, (2 * xy - 2 * fl.flip(-1) + 1) / size - 1, mode="bilinear",
                                          padding_mode="border", align_corners=False)  # fmt: skip
    (out * w).sum().backward()
    fl2 = flow.float().to(device).requires_grad_(True)
    im2 = img.float().to(device).requires_grad_(True)
    out2 = F.dense_image_warp(im2, fl2)
    (out2 * w.float().to(device)).sum().backward()
    assert torch.allclose(fl2.grad.cpu().double(), fl.grad, rtol=1e-3, atol=1e-5)
    assert im2.grad is not None

    src = torch.from_numpy(rng.uniform(size=(N, Mp, 2)) * [H - 1, W - 1])
    dst = src + torch.from_numpy(rng.normal(size=(N, Mp, 2)) * 0.5)
    s, d = src.clone().requires_grad_(True), dst.clone().requires_grad_(True)
    sx, dx = s.flip(-1), d.flip(-1)
    query = torch.stack([ww.flatten(), hh.flatten()], 1).unsqueeze(0).expand(N, H * W, 2)
    if include_flow:
        fxy = _torch_spline(dx, dx - sx, query, 2).view(N, H, W, 2)
        grid = (2 * xy - 2 * fxy + 1) / size - 1
    else:
        grid = _torch_spline(dx, (2 * sx + 1) / size - 1, query, 2).view(N, H, W, 2)
    out = torch.nn.functional.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=False)
    (out * w).sum().backward()
    s2, d2 = src.float().to(device).requires_grad_(True), dst.float().to(device).requires_grad_(True)
    res = F.sparse_image_warp(img.float().to(device), s2, d2, include_flow=include_flow)
    warped = res[0] if include_flow else res
    (warped * w.float().to(device)).sum().backward()
    for name, a, b in (("source", s2, s), ("dest", d2, d)):
        err = (a.grad.cpu().double() - b.grad).abs().max() / b.grad.abs().max()
        assert err < 5e-3, (name, include_flow, float(err))


def test_float64_images_are_narrowed_out_loud(device):
    """The image kernels compute in float32; a float64 argument comes back as float64, close to the
    float32 result, and the narrowing is announced (once per process) instead of happening silently."""
    import warnings

    from pydrobert_amd import _img

    g = torch.Generator(device=device).manual_seed(5)
    img = torch.randn((2, 1, 12, 9), device=device, generator=g)
    flow = torch.randn((2, 12, 9, 2), device=device, generator=g)
    exp = F.dense_image_warp(img, flow)
    _img._NARROWING_WARNED = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        act = F.dense_image_warp(img.double(), flow.double())
        act2 = F.dense_image_warp(img.double(), flow.double())
    assert act.dtype == torch.double and torch.allclose(act.float(), exp, atol=1e-6) and torch.equal(act, act2)
    assert sum("float64" in str(x.message) for x in w) == 1
