"""ORACLE -- TEST INFRASTRUCTURE ONLY.

numpy restatement (float64 arithmetic, float32 in/out) of the reference's polyharmonic
spline, 1-D warp grid, dense / sparse image warps and SpecAugment application
(reference: src/pydrobert/torch/_img.py).  ``grid_sample`` is restated from its documented
semantics (bilinear / nearest; zeros / border / reflection; align_corners=False).

Parity status: PINNED -- against the live reference (tests/golden/make_golden.py) and the
TensorFlow-addons golden arrays the reference's tests ship (tests/golden/tfa_*.npy), to the
reference's own tolerances (1e-4 on valid frames; 1e-3 for the TF arrays).  The reference
solves the spline system in float32 with an mm-based cdist, so agreement beyond ~1e-4 is not
meaningful.
"""
import numpy as np

__all__ = [
    "spec_augment_parameters_from_uniforms",
    "dense_image_warp",
    "grid_sample",
    "pad_variable",
    "polyharmonic_spline",
    "sparse_image_warp",
    "spec_augment_apply_parameters",
    "warp_1d_grid",
]

_EPS32 = float(np.finfo(np.float32).eps)


def _np(x, dtype=None):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    return x if dtype is None else x.astype(dtype)


def _phi(r, k):
    """_img.py:59-64"""
    if k % 2:
        return r ** k
    return (r ** k) * np.log(np.maximum(r, _EPS32))


def _cdist(a, b):
    d = a[:, :, None, :] - b[:, None, :, :]
    return np.sqrt((d * d).sum(-1))


def polyharmonic_spline(train_points, train_values, query_points, order,
                        regularization_weight=0.0, full_matrix=True):  # fmt: skip
    """_img.py:67-150.  (N,T,I), (N,T,O), (N,Q,I) -> (N,Q,O).  The exact solution of the
    bordered system; ``full_matrix`` only selects between two float32 evaluation orders in the
    reference."""
    c = _np(train_points, np.float32).astype(np.float64)
    f = _np(train_values).astype(np.float64)
    x = _np(query_points, np.float32).astype(np.float64)
    N, T, I = c.shape
    O = f.shape[2]
    A = _phi(_cdist(c, c), order)
    if regularization_weight > 0.0:
        A = A + np.eye(T)[None] * regularization_weight
    B = np.concatenate([c, np.ones((N, T, 1))], 2)
    top = np.concatenate([A, B], 2)
    bot = np.concatenate([B.transpose(0, 2, 1), np.zeros((N, I + 1, I + 1))], 2)
    lhs = np.concatenate([top, bot], 1)
    rhs = np.concatenate([f, np.zeros((N, I + 1, O))], 1)
    wv = np.linalg.solve(lhs, rhs)
    w, v = wv[:, :T], wv[:, T:]
    phi = _phi(_cdist(x, c), order)
    x1 = np.concatenate([x, np.ones(x.shape[:2] + (1,))], 2)
    return (phi @ w + x1 @ v).astype(np.float32)


def warp_1d_grid(src, flow, lengths, max_length=None, interpolation_order=1):
    """_img.py:268-303 -> (N, T) float32 grid in [-1, 1]."""
    src = _np(src, np.float32).astype(np.float64)
    flow = _np(flow, np.float32).astype(np.float64)
    lengths = _np(lengths, np.float32).astype(np.float64)
    N = src.shape[0]
    T = int(np.ceil(lengths.max())) if max_length is None else int(max_length)
    src = np.maximum(np.minimum(src, lengths - 1), 0)
    dst = np.maximum(np.minimum(src + flow, lengths - 1), 0)
    src = (2.0 * src + 1.0) / T - 1.0
    dst = (2.0 * dst + 1.0) / T - 1.0
    lowers = np.full((N,), 1 / T - 1 - _EPS32)
    uppers = (2 * lengths - 1) / T - 1.0 + _EPS32
    s = np.stack([lowers, src, uppers], 1)[..., None]
    d = np.stack([lowers, dst, uppers], 1)[..., None]
    t = ((2.0 * np.arange(T) + 1.0) / T - 1.0)[None, :, None].repeat(N, 0)
    return polyharmonic_spline(d, s, t, interpolation_order)[..., 0]


def _reflect(x, lo2, hi2):
    """torch's reflect_coordinates with bounds given doubled (align_corners=False: -1, 2*size-1)."""
    if lo2 == hi2:
        return np.zeros_like(x)
    mn = lo2 / 2.0
    span = (hi2 - lo2) / 2.0
    x = np.abs(x - mn)
    extra = np.mod(x, span)
    flips = np.floor(x / span)
    return np.where(flips % 2 == 0, extra + mn, span - extra + mn)


def grid_sample(image, grid, mode="bilinear", padding_mode="zeros"):
    """torch.nn.functional.grid_sample(image (N,C,H,W), grid (N,Ho,Wo,2), align_corners=False)."""
    img = _np(image).astype(np.float64)
    # (a float32 image's grid is a float32 tensor; a float64 image's grid is float64: the reference
    # builds it from arange(..., dtype=image.dtype), _img.py:423-436)
    g = _np(grid).astype(np.float64) if _np(image).dtype == np.float64 else _np(grid).astype(np.float32).astype(np.float64)
    N, C, H, W = img.shape
    ix = ((g[..., 0] + 1) * W - 1) / 2
    iy = ((g[..., 1] + 1) * H - 1) / 2
    if padding_mode == "border":
        ix, iy = np.clip(ix, 0, W - 1), np.clip(iy, 0, H - 1)
    elif padding_mode == "reflection":
        ix = np.clip(_reflect(ix, -1, 2 * W - 1), 0, W - 1)
        iy = np.clip(_reflect(iy, -1, 2 * H - 1), 0, H - 1)
    out = np.zeros((N, C) + g.shape[1:3])
    nidx = np.arange(N)[:, None, None]

    def tap(yy, xx, wgt):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        xc, yc = np.clip(xx, 0, W - 1).astype(int), np.clip(yy, 0, H - 1).astype(int)
        vals = img[nidx, :, yc, xc]  # (N, Ho, Wo, C)
        return np.moveaxis(vals * (wgt * ok)[..., None], -1, 1)

    if mode == "nearest":
        return tap(np.rint(iy), np.rint(ix), np.ones_like(ix)).astype(_np(image).dtype)
    x0, y0 = np.floor(ix), np.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    out = (
        tap(y0, x0, (x1 - ix) * (y1 - iy))
        + tap(y0, x1, (ix - x0) * (y1 - iy))
        + tap(y1, x0, (x1 - ix) * (iy - y0))
        + tap(y1, x1, (ix - x0) * (iy - y0))
    )
    return out.astype(_np(image).dtype)


def dense_image_warp(image, flow, indexing="hw", mode="bilinear", padding_mode="border"):
    """_img.py:393-439: output[n,c,h,w] = image[n,c,h - flow_h, w - flow_w]."""
    image = _np(image)
    flow = _np(flow, np.float32).astype(np.float64)
    N, C, H, W = image.shape
    h, w = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    hw = np.stack([w, h], 2)[None]
    if indexing == "hw":
        flow = flow[..., ::-1]
    elif indexing != "wh":
        raise ValueError("Invalid indexing! must be one of 'wh' or 'hw'")
    grid = (2 * hw - 2 * flow + 1.0) / np.array([W, H], np.float64) - 1.0
    return grid_sample(image, grid, mode, padding_mode)


def _pinned_points(k, W, H, N):
    """_img.py:244-265"""
    r = np.linspace(0.0, 1.0, k + 1)
    wr, hr = (W - 1) * r, (H - 1) * r
    z = np.zeros(k + 1)
    bottom = np.stack([wr, z], 1)
    left = np.stack([z[1:-1], hr[1:-1]], 1)
    top = np.stack([wr, np.full(k + 1, H - 1.0)], 1)
    right = np.stack([np.full(k - 1, W - 1.0), hr[1:-1]], 1)
    return np.concatenate([bottom, left, top, right], 0)[None].repeat(N, 0)


def sparse_image_warp(image, source_points, dest_points, indexing="hw",
                      field_interpolation_order=2, field_regularization_weight=0.0,
                      field_full_matrix=True, pinned_boundary_points=0,
                      dense_interpolation_mode="bilinear", dense_padding_mode="border",
                      include_flow=True):  # fmt: skip
    """_img.py:520-714.  Returns warped, or (warped, flow (N,H,W,2)) when include_flow."""
    image = _np(image)
    src = _np(source_points, np.float32).astype(np.float64)
    dst = _np(dest_points, np.float32).astype(np.float64)
    if indexing == "hw":
        src, dst = src[..., ::-1], dst[..., ::-1]
    N, C, H, W = image.shape
    M = src.shape[1]
    if M == 0:
        return (image, np.zeros((N, H, W, 2), np.float32)) if include_flow else image
    if pinned_boundary_points > 0:
        pp = _pinned_points(pinned_boundary_points, W, H, N)
        src, dst = np.concatenate([src, pp], 1), np.concatenate([dst, pp], 1)
    h, w = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    q = np.stack([w.ravel(), h.ravel()], 1)[None].repeat(N, 0)
    if include_flow:
        flow = polyharmonic_spline(dst, dst - src, q, field_interpolation_order,
                                   field_regularization_weight, field_full_matrix)  # fmt: skip
        flow = flow.reshape(N, H, W, 2)
        warped = dense_image_warp(image, flow, "wh", dense_interpolation_mode, dense_padding_mode)
        if indexing == "hw":
            flow = flow[..., ::-1]
        return warped, np.ascontiguousarray(flow)
    vals = (2.0 * src + 1.0) / np.array([W, H], np.float64) - 1.0
    grid = polyharmonic_spline(dst, vals, q, field_interpolation_order,
                               field_regularization_weight, field_full_matrix)  # fmt: skip
    return grid_sample(image, grid.reshape(N, H, W, 2), dense_interpolation_mode, dense_padding_mode)


def spec_augment_apply_parameters(feats, params, interpolation_order, lengths=None):
    """_img.py:1142-1211.  params = (w_0, w, v_0, v, t_0, t, f_0, f); empty arrays skip a group."""
    feats = _np(feats)
    N, T, F = feats.shape
    lengths = np.full((N,), T, np.float64) if lengths is None else _np(lengths).astype(np.float64)
    w_0, w, v_0, v, t_0, t, f_0, f = [None if p is None else _np(p) for p in params]

    def has(a, b):
        return a is not None and a.size and b is not None and b.size

    out = feats
    tg = fg = None
    if has(w_0, w):
        tg = warp_1d_grid(w_0, w, lengths, T, interpolation_order)
    if has(v_0, v):
        fg = warp_1d_grid(v_0, v, np.full((N,), F), F, interpolation_order)
    if tg is not None or fg is not None:
        if tg is None:
            tg = (((2 * np.arange(T) + 1) / T - 1)[None].repeat(N, 0)).astype(np.float32)
        if fg is None:
            fg = (((2 * np.arange(F) + 1) / F - 1)[None].repeat(N, 0)).astype(np.float32)
        grid = np.stack(
            [np.broadcast_to(fg[:, None, :], (N, T, F)), np.broadcast_to(tg[:, :, None], (N, T, F))], 3
        )
        out = grid_sample(feats[:, None], grid, "bilinear", "border")[:, 0]
    mask = np.zeros((N, T, F), bool)
    if has(t_0, t):
        ar = np.arange(T)[None, :, None]
        mask |= ((ar >= t_0[:, None, :]) & (ar < (t_0 + t)[:, None, :])).any(2)[:, :, None]
    if has(f_0, f):
        ar = np.arange(F)[None, :, None]
        mask |= ((ar >= f_0[:, None, :]) & (ar < (f_0 + f)[:, None, :])).any(2)[:, None, :]
    return np.where(mask, np.zeros((), feats.dtype), out).astype(feats.dtype)


def pad_variable(x, lens, pad, mode="constant", value=0.0):
    """Loop restatement of the reference's ``pad_variable`` (_pad.py:108-149): per sequence,
    left padding, the sequence, right padding, then ``value`` up to the longest new length."""
    x = np.asarray(x)
    lens, pad = np.asarray(lens), np.asarray(pad)
    N = x.shape[0]
    new_lens = lens + pad.sum(0)
    Tp = int(new_lens.max()) if N else 0
    out = np.full((N, Tp) + x.shape[2:], value, dtype=x.dtype)
    for n in range(N):
        L, (a, b) = int(lens[n]), (int(pad[0, n]), int(pad[1, n]))
        seq = x[n, :L]
        if mode == "constant":
            left = np.full((a,) + x.shape[2:], value, dtype=x.dtype)
            right = np.full((b,) + x.shape[2:], value, dtype=x.dtype)
        elif mode == "reflect":
            if a >= L or b >= L:
                raise NotImplementedError("reflect padding must be shorter than the sequence")
            left = seq[1 : a + 1][::-1]
            right = seq[L - 1 - b : L - 1][::-1]
        elif mode == "replicate":
            if L < 1:
                raise RuntimeError("replicate padding needs non-empty sequences")
            left = np.repeat(seq[:1], a, 0)
            right = np.repeat(seq[-1:], b, 0)
        else:
            raise ValueError(mode)
        out[n, : a + L + b] = np.concatenate([left, seq, right], 0)
    return out


def spec_augment_parameters_from_uniforms(u, T, F, max_time_warp, max_freq_warp, max_time_mask, max_freq_mask,
                                          max_time_mask_proportion, num_time_mask, num_time_mask_proportion,
                                          num_freq_mask, lengths=None, double_feats=False):
    """spec_augment_draw_parameters (_img.py:1056-1139) as a function of its uniform draws: column c of
    ``u`` (N, R) is the c-th ``torch.rand`` value of an utterance in the reference's call order (w_0, w,
    v_0, v, t x MT, t_0 x MT, f x MF, f_0 x MF; disabled groups take no columns).  float32 expressions in
    the reference's order; returns the 8-tuple with zero-size arrays for disabled groups."""
    f32 = np.float32
    u = np.asarray(u, dtype=f32)
    N = u.shape[0]
    eps = 2.220446049250313e-16 if double_feats else 1.1920928955078125e-07
    omeps = f32(1 - eps)
    ln = np.full((N,), T, dtype=f32) if lengths is None else np.asarray(lengths).astype(f32)
    e = np.zeros((0,), f32)
    ei = np.zeros((0,), np.int64)
    w_0 = w = v_0 = v = e
    t_0 = t = f_0 = f = ei
    c = 0
    if max_time_warp != 0.0:
        Wt = np.clip(ln / f32(2) - f32(eps), f32(0), f32(max_time_warp)).astype(f32)
        w_0 = (u[:, c] * (ln - f32(2) * Wt) + Wt).astype(f32)
        w = (u[:, c + 1] * (f32(2) * Wt) - Wt).astype(f32)
        c += 2
    if max_freq_warp != 0.0:
        Vf = f32(min(max(F / 2 - eps, 0.0), max_freq_warp))
        v_0 = (u[:, c] * f32(f32(F) - f32(2) * Vf) + Vf).astype(f32)
        v = (u[:, c + 1] * (f32(2) * Vf) - Vf).astype(f32)
        c += 2
    if max_time_mask != 0 and max_time_mask_proportion != 0.0 and num_time_mask != 0 and num_time_mask_proportion != 0.0:
        MT = num_time_mask
        max_ = np.floor(np.minimum(ln * f32(max_time_mask_proportion), f32(max_time_mask))).astype(f32)
        nums_ = np.floor(np.minimum(ln * f32(num_time_mask_proportion), f32(num_time_mask))).astype(f32)
        t = (u[:, c : c + MT] * (max_ + omeps)[:, None]).astype(np.int64)
        t = np.where(nums_[:, None] <= np.arange(MT, dtype=f32)[None], 0, t)
        t_0 = (u[:, c + MT : c + 2 * MT] * ((ln[:, None] - t.astype(f32)) + omeps)).astype(np.int64)
        c += 2 * MT
    if max_freq_mask != 0 and num_freq_mask != 0:
        MF = num_freq_mask
        maxf = f32(min(max_freq_mask, F))
        f = (u[:, c : c + MF] * (maxf + omeps)).astype(np.int64)
        f_0 = (u[:, c + MF : c + 2 * MF] * ((f32(F) - f.astype(f32)) + omeps)).astype(np.int64)
    return w_0, w, v_0, v, t_0, t, f_0, f
