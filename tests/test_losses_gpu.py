"""GPU: fused optimal-completion distillation loss and minimum-error-rate loss vs the oracle
and the golden vectors (forward and gradient)."""
import os

import numpy as np
import pytest
import torch

import oracle
from pydrobert_amd import functional as F
from pydrobert_amd import modules as M

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def T(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def test_hocd_random_vs_oracle_and_autograd(device):
    rng = np.random.default_rng(17)
    for it in range(30):
        N, R, H, V = int(rng.integers(1, 6)), int(rng.integers(1, 40)), int(rng.integers(1, 30)), int(rng.integers(3, 50))
        bf = bool(rng.integers(0, 2))
        eos = None if rng.random() < 0.4 else int(rng.integers(0, V))
        ref = rng.integers(0, V, (N, R) if bf else (R, N))
        hyp = rng.integers(0, V, (N, H) if bf else (H, N))
        logits = rng.normal(size=hyp.shape + (V,)).astype(np.float32)
        w = None if rng.random() < 0.5 else rng.uniform(0.5, 2, V).astype(np.float32)
        for red in ("mean", "sum", "none"):
            exp = oracle.hard_optimal_completion_distillation_loss(
                logits, ref, hyp, eos=eos, batch_first=bf, weight=w, reduction=red)
            x = T(logits, device).requires_grad_(True)
            act = F.hard_optimal_completion_distillation_loss(
                x, T(ref, device), T(hyp, device), eos=eos, batch_first=bf,
                weight=None if w is None else T(w, device), reduction=red, warn=False)
            assert act.shape == exp.shape
            assert np.allclose(act.detach().cpu().numpy(), exp, rtol=1e-5, atol=1e-6), (it, red)
            # gradient against torch autograd through cross_entropy on the oracle's targets
            opt = torch.from_numpy(oracle.optimal_completion(ref, hyp, eos=eos, batch_first=bf, padding=-2, exclude_last=True))
            xc = torch.from_numpy(logits).double().requires_grad_(True)
            C = opt.shape[-1]
            ce = torch.nn.functional.cross_entropy(
                xc.unsqueeze(2).expand(-1, -1, C, -1).reshape(-1, V), opt.flatten(),
                weight=None if w is None else torch.from_numpy(w).double(), ignore_index=-2, reduction="none",
            ).view_as(opt) if C else torch.zeros(opt.shape, dtype=torch.double)
            pad = opt == -2
            l = ce.masked_fill(pad, 0.0).sum(2) / (~pad).sum(2).clamp_min(1)
            gw = torch.randn(l.shape, dtype=torch.double)
            (ge,) = torch.autograd.grad((l * gw).sum(), xc, allow_unused=True)
            lo = F.hard_optimal_completion_distillation_loss(
                x, T(ref, device), T(hyp, device), eos=eos, batch_first=bf,
                weight=None if w is None else T(w, device), reduction="none", warn=False)
            (ga,) = torch.autograd.grad((lo * gw.float().to(device)).sum(), x)
            ge = torch.zeros_like(xc) if ge is None else ge
            assert torch.allclose(ga.cpu().double(), ge, rtol=1e-4, atol=1e-5), (it, (ga.cpu().double() - ge).abs().max())


def test_hocd_long_reference(device):
    """A reference beyond 2048 tokens: class bitmasks wider than 64 words through the loss."""
    rng = np.random.default_rng(23)
    N, R, H, V = 2, 2600, 21, 700
    ref = rng.integers(0, V, (R, N))
    hyp = rng.integers(0, V, (H, N))
    logits = rng.normal(size=(H, N, V)).astype(np.float32)
    exp = oracle.hard_optimal_completion_distillation_loss(logits, ref, hyp, reduction="none")
    x = T(logits, device).requires_grad_(True)
    act = F.hard_optimal_completion_distillation_loss(x, T(ref, device), T(hyp, device), reduction="none", warn=False)
    assert np.allclose(act.detach().cpu().numpy(), exp, rtol=1e-5, atol=1e-6)
    (g,) = torch.autograd.grad(act.sum(), x)
    assert torch.isfinite(g).all() and float(g.abs().sum()) > 0


def test_loss_goldens(device):
    g = np.load(os.path.join(G, "losses.npz"))
    ref, hyp, w = T(g["ref"], device), T(g["hyp"], device), T(g["weight"], device)
    for tag, kw in {"a": dict(eos=0), "b": dict(eos=None, weight=w), "c": dict(eos=0, include_eos=False, weight=w)}.items():
        for red in ("mean", "sum", "none"):
            x = T(g["logits"], device).requires_grad_(True)
            mod = M.HardOptimalCompletionDistillationLoss(reduction=red, ignore_index=-2, **kw).to(device)
            loss = mod(x, ref, hyp, warn=False)
            (gr,) = torch.autograd.grad(loss.sum(), x)
            assert np.allclose(loss.detach().cpu().numpy(), g["hocd_{}_{}".format(tag, red)], rtol=1e-5, atol=1e-6)
            assert np.allclose(gr.cpu().numpy(), g["hocd_{}_{}_grad".format(tag, red)], rtol=1e-4, atol=1e-6)
    lp = T(g["mer_lp"], device).requires_grad_(True)
    for red in ("mean", "none"):
        for sub_avg in (True, False):
            loss = M.MinimumErrorRateLoss(eos=0, sub_avg=sub_avg, reduction=red)(lp, ref, T(g["mer_hyp"], device), warn=False)
            assert np.allclose(loss.detach().cpu().numpy(), g["mer_{}_{}".format(red, int(sub_avg))], rtol=1e-5, atol=1e-7)
    (gl,) = torch.autograd.grad(loss.sum(), lp)
    assert torch.isfinite(gl).all()


def test_loss_errors(device):
    logits = torch.zeros(3, 2, 4, device=device)
    tok = torch.zeros(3, 2, dtype=torch.long, device=device)
    with pytest.raises(RuntimeError, match="3 dimensional"):
        F.hard_optimal_completion_distillation_loss(logits[0], tok, tok)
    with pytest.raises(RuntimeError, match="must match hyp shape"):
        F.hard_optimal_completion_distillation_loss(logits, tok, tok[:2])
    with pytest.raises(RuntimeError, match="must be a class idx"):
        F.hard_optimal_completion_distillation_loss(logits, tok, tok, eos=9)
    with pytest.raises(RuntimeError, match="not class indices"):
        F.hard_optimal_completion_distillation_loss(logits, tok + 7, tok)
    with pytest.raises(RuntimeError, match="at least two samples"):
        F.minimum_error_rate_loss(torch.zeros(2, 1, device=device), tok, tok.unsqueeze(-1))
    with pytest.raises(ValueError):
        M.MinimumErrorRateLoss(reduction="avg")
