"""Randomised differential run, part 3: teacher-forced CTC step, image ops, MER loss, seq log probs."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "pydrobert-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, warnings
import oracle
from pydrobert_amd import _decoding as D, functional as F
warnings.simplefilter("ignore")
dev = "cuda"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60)
bad = n_cases = 0
t_say = time.time()
def T(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def cmp_step(act, exp):
    (y, last, lens, (nb, b), isp, src, non) = [tuple(z.cpu().numpy() for z in x) if isinstance(x, tuple) else x.cpu().numpy() for x in act]
    (ey, elast, elens, (enb, eb), eisp, esrc, enon) = exp
    if y.shape != ey.shape or not np.array_equal(lens, elens) or not np.array_equal(src, esrc): return "lens/src"
    if not np.array_equal(non, enon) or not np.array_equal(last, elast): return "non/last"
    if not np.array_equal(isp, eisp): return "isp"
    valid = np.isfinite(enb)
    if not np.array_equal(np.isfinite(nb), valid): return "finite"
    if not (np.allclose(nb[valid], enb[valid], rtol=2e-5, atol=0) and np.allclose(b[valid], eb[valid], rtol=2e-5, atol=0)): return "probs"
    m = np.arange(ey.shape[0])[:, None, None] < elens[None]
    if not np.array_equal(np.where(m, y, 0), np.where(m, ey, 0)): return "y"
    return None
while time.time() < t_end:
    n_cases += 1
    if time.time() - t_say > 45.0:  # (a line a minute: a silent run is taken for a hung one)
        t_say = time.time(); print("cases", n_cases, "mismatches", bad, flush=True)
    kind = rng.integers(0, 5)
    if kind == 0:  # teacher-forced CTC step
        V, W, N, Tn = int(rng.integers(1, 50)), int(rng.integers(1, 33)), int(rng.integers(1, 4)), int(rng.integers(1, 16))
        if rng.random() < 0.2:  # more than 32 prefixes: the radix-select form (csrc/advance_wide.hip)
            W = int(rng.integers(33, 120)); V = W + int(rng.integers(0, 60)); Tn = int(rng.integers(1, 8))
        # round 5: prefixes sharing one expand()ed row (one list for all), and the step that mixes a language
        # model's scores itself (pdt_ctc_prefix_search_advance_lm) -- both also with rows of hundreds of tokens
        form = "dense" if W > 32 else ["dense", "shared", "mixed"][int(rng.integers(0, 3))]
        if form != "dense" and rng.random() < 0.4:
            V = int(rng.integers(50, 700))
        beta, vmix = float(rng.random() * 0.9 + 0.05), bool(rng.random() < 0.5)
        if W > V + 1: continue
        nb, b = np.zeros((N, 1), np.float32), np.ones((N, 1), np.float32)
        y = np.zeros((0, N, 1), np.int64); last = lens = np.zeros((N, 1), np.int64); isp = np.ones((N, 1, 1), bool)
        for t in range(Tn):
            Kp = nb.shape[1]
            p = np.exp(rng.normal(size=(N, V + 1)) * 1.5).astype(np.float32); p /= p.sum(1, keepdims=True)
            nonext, blank = np.ascontiguousarray(p[:, :V]), np.ascontiguousarray(p[:, V])
            lm = np.exp(rng.normal(size=(N, Kp, V)) * 0.7).astype(np.float32); lm /= lm.sum(2, keepdims=True)
            ext = (lm ** 0.5 * nonext[:, None]).astype(np.float32)
            if form == "shared":
                ext = np.ascontiguousarray(np.broadcast_to(nonext[:, None], (N, Kp, V)))
            elif form == "mixed":
                scores = (rng.normal(size=(N, Kp, V)) * 2.0).astype(np.float32)  # any normalisation
                # (the mix by the package's own fusion_ext kernel -- checked against numpy to 1e-5 -- so that the
                # oracle ranks the SAME float32 extension probabilities the fused step forms inside itself)
                lsm = scores - scores.max(2, keepdims=True)
                lsm = lsm - np.log(np.exp(lsm).sum(2, keepdims=True))
                ext_np = ((1 - beta) * nonext[:, None] + beta * np.exp(lsm) * (1 - blank[:, None, None]) if vmix
                          else nonext[:, None] * np.exp(beta * lsm)).astype(np.float32)
                ext = torch.ops.pydrobert_amd.fusion_ext(T(scores).reshape(N * Kp, V), T(nonext), T(blank), beta, vmix).cpu().numpy()
                if not np.allclose(ext, ext_np, rtol=1e-5, atol=1e-30):
                    bad += 1; print("MISMATCH fusion_ext", V, W, N, t, np.abs(ext - ext_np).max())
            exp = oracle.ctc_prefix_search_advance((ext, nonext, blank), W, (nb, b), y, last, lens, isp)
            if form == "mixed":
                o = D._ctc_step_with_lm_scores(T(scores), beta, vmix, T(nonext), T(blank), W, T(nb), T(b), T(y), T(last), T(lens), T(isp))
                act = (o[0], o[1], o[2], (o[3], o[4]), o[5], o[6], o[7])
            elif form == "shared":
                act = F.ctc_prefix_search_advance((T(nonext).unsqueeze(1).expand(N, Kp, V), T(nonext), T(blank)), W, (T(nb), T(b)), T(y), T(last), T(lens), T(isp))
            else:
                act = F.ctc_prefix_search_advance((T(ext), T(nonext), T(blank)), W, (T(nb), T(b)), T(y), T(last), T(lens), T(isp))
            r = cmp_step(act, exp)
            if r:
                # near-ties between candidate masses are legitimate differences; flag only clear ones
                tot = (exp[3][0] + exp[3][1])
                srt = np.sort(tot[np.isfinite(tot)])[::-1]
                gaps = np.abs(np.diff(srt)) / np.maximum(srt[:-1], 1e-30) if srt.size > 1 else np.array([1.0])
                if gaps.size == 0 or gaps.min() > 1e-5:
                    bad += 1; print("MISMATCH ctc step", form, r, V, W, N, t)
                break
            y, last, lens, (nb, b), isp = exp[0], exp[1], exp[2], exp[3], exp[4]
    elif kind == 1:  # spec augment apply
        N, Tn, Fq = int(rng.integers(1, 5)), int(rng.integers(2, 400)), int(rng.integers(1, 5)) * (4 if rng.random() < 0.6 else 1)
        feats = rng.normal(size=(N, Tn, Fq)).astype(np.float32)
        lengths = rng.integers(max(1, Tn // 2), Tn + 1, N)
        has_tw, has_fw = rng.random() < 0.7, rng.random() < 0.3
        W_ = np.minimum(lengths / 2 - 1e-3, 5.0).clip(0)
        w_0 = (rng.random(N) * (lengths - 2 * W_) + W_).astype(np.float32) if has_tw else np.zeros(0, np.float32)
        w = ((rng.random(N) * 2 - 1) * W_).astype(np.float32) if has_tw else np.zeros(0, np.float32)
        Vf = min(max(Fq / 2 - 1e-3, 0), 1.5)
        v_0 = (rng.random(N) * (Fq - 2 * Vf) + Vf).astype(np.float32) if has_fw else np.zeros(0, np.float32)
        v = ((rng.random(N) * 2 - 1) * Vf).astype(np.float32) if has_fw else np.zeros(0, np.float32)
        mt, mf = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        t_ = rng.integers(0, 6, (N, mt)); t_0 = (rng.random((N, mt)) * (lengths[:, None] - t_ + 0.99)).astype(np.int64).clip(0)
        f_ = rng.integers(0, min(3, Fq) + 1, (N, mf)); f_0 = (rng.random((N, mf)) * (Fq - f_ + 0.99)).astype(np.int64)
        params = (w_0, w, v_0, v, t_0, t_, f_0, f_)
        exp = oracle.spec_augment_apply_parameters(feats, params, 1, lengths)
        act = F.spec_augment_apply_parameters(T(feats), tuple(T(p) for p in params), 1, T(lengths)).cpu().numpy()
        valid = np.arange(Tn)[None, :, None] < lengths[:, None, None]
        err = np.abs(exp - act) * valid
        if err.max() > 5e-3:
            bad += 1; print("MISMATCH spec", N, Tn, Fq, has_tw, has_fw, mt, mf, err.max())
    elif kind == 2:  # spline + 1d grid
        N, I, O, Q = int(rng.integers(1, 4)), int(rng.integers(1, 3)), int(rng.integers(1, 3)), int(rng.integers(1, 30))
        Tt = int(rng.integers(I + 2, 10))
        order = int(rng.integers(1, 4))
        c = (rng.random((N, Tt, I)) * 10).astype(np.float32); f = rng.normal(size=(N, Tt, O)).astype(np.float32)
        q = (rng.random((N, Q, I)) * 10).astype(np.float32)
        try:
            exp = oracle.polyharmonic_spline(c, f, q, order, 0.0)
        except np.linalg.LinAlgError:
            continue
        act = F.polyharmonic_spline(T(c), T(f), T(q), order).cpu().numpy()
        scale = max(1.0, np.abs(exp).max())
        if np.abs(exp - act).max() > 2e-3 * scale:
            bad += 1; print("MISMATCH spline", N, Tt, I, O, Q, order, np.abs(exp - act).max(), scale)
    elif kind == 3:  # MER loss
        N, S, R, H, V = int(rng.integers(1, 4)), int(rng.integers(2, 5)), int(rng.integers(1, 12)), int(rng.integers(1, 12)), int(rng.integers(2, 7))
        ref = rng.integers(0, V, (R, N)); hyp = rng.integers(0, V, (H, N, S)); lp = rng.normal(size=(N, S)).astype(np.float32)
        eos = None if rng.random() < 0.5 else int(rng.integers(0, V))
        red = ["mean", "sum", "none"][rng.integers(0, 3)]
        exp = oracle.minimum_error_rate_loss(lp, ref, hyp, eos=eos, reduction=red)
        act = F.minimum_error_rate_loss(T(lp), T(ref), T(hyp), eos=eos, reduction=red, warn=False).cpu().numpy()
        if not np.allclose(exp, act, rtol=1e-5, atol=1e-5):
            bad += 1; print("MISMATCH mer", N, S, R, H, V, eos, red)
    else:  # sequence log probs, any dim, eos
        shape = tuple(int(x) for x in rng.integers(1, 6, rng.integers(1, 4)))
        V = int(rng.integers(2, 9)); dim = int(rng.integers(-len(shape), len(shape)))
        lg = rng.normal(size=shape + (V,)).astype(np.float32); hyp = rng.integers(-1, V + 1, shape)
        eos = None if rng.random() < 0.5 else int(rng.integers(0, V))
        exp = oracle.sequence_log_probs(lg, hyp, dim, eos)
        act = F.sequence_log_probs(T(lg), T(hyp), dim, eos).cpu().numpy()
        if exp.shape != act.shape or not np.allclose(exp, act, rtol=1e-5, atol=1e-5):
            bad += 1; print("MISMATCH slp", shape, V, dim, eos)
print("cases", n_cases, "mismatches", bad)
