"""GPU: ctc_greedy_search, sequence_log_probs (forward + gradient), RandomWalk vs the oracle."""
import numpy as np
import pytest
import torch

import oracle
from pydrobert_amd import functional as F
from pydrobert_amd import modules as M

from _toy_lm import BigramLM

pytestmark = pytest.mark.gpu


def T(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def test_ctc_greedy_search_random(device):
    rng = np.random.default_rng(1)
    for it in range(60):
        Tn, N, V = int(rng.integers(1, 90)), int(rng.integers(1, 7)), int(rng.integers(2, 150))
        bf, pr = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        lg = rng.normal(size=(N, Tn, V) if bf else (Tn, N, V)).astype(np.float32)
        if pr:
            lg = (np.exp(lg) / np.exp(lg).sum(-1, keepdims=True)).astype(np.float32)
        lens = None if rng.random() < 0.5 else rng.integers(0, Tn + 1, N)
        bi = int(rng.integers(-V, V))
        exp = oracle.ctc_greedy_search(lg, lens, bi, bf, pr)
        act = M.CTCGreedySearch(bi, bf, pr)(T(lg, device), None if lens is None else T(lens, device))
        assert np.array_equal(act[2].cpu().numpy(), exp[2]), it
        assert np.array_equal(act[1].cpu().numpy(), exp[1]), it
        assert np.allclose(act[0].cpu().numpy(), exp[0], rtol=1e-5, atol=1e-5), it


def test_width_one_prefix_search_equals_greedy(device):
    """Reference tests/test_decoding.py:283-294: beam width 1 == greedy (when the best path's
    collapsed form dominates, which peaky logits guarantee)."""
    rng = np.random.default_rng(2)
    lg = rng.normal(size=(40, 6, 11)).astype(np.float32)
    np.put_along_axis(lg, rng.integers(0, 11, (40, 6, 1)), 15.0, 2)
    x = T(lg, device)
    _, paths, lens = F.ctc_greedy_search(x)
    y, yl, _ = F.ctc_prefix_search(x, 1)
    assert torch.equal(lens, yl[:, 0])
    for n in range(6):
        assert torch.equal(paths[: lens[n], n], y[: lens[n], n, 0])


def test_sequence_log_probs_forward_backward(device):
    rng = np.random.default_rng(3)
    for it in range(40):
        shape = tuple(int(x) for x in rng.integers(1, 6, int(rng.integers(1, 4))))
        V = int(rng.integers(2, 80))
        dim = int(rng.integers(-len(shape), len(shape)))
        hyp = rng.integers(-1, V + 1, shape)
        lg = rng.normal(size=shape + (V,)).astype(np.float32)
        eos = None if rng.random() < 0.5 else int(rng.integers(0, V))
        exp = oracle.sequence_log_probs(lg, hyp, dim, eos)
        x = T(lg, device).requires_grad_(True)
        act = M.SequenceLogProbabilities(dim, eos)(x, T(hyp, device))
        assert act.shape == exp.shape
        assert np.allclose(act.detach().cpu().numpy(), exp, rtol=1e-5, atol=1e-5), it
        # gradient vs torch autograd on the same masked definition
        xc = torch.from_numpy(lg).double().requires_grad_(True)
        h = torch.from_numpy(hyp)
        lsm = xc.log_softmax(-1)
        mask = (h < 0) | (h >= V)
        if eos is not None:
            d = dim % h.dim()
            is_eos = (h == eos)
            first = torch.where(is_eos.any(d), is_eos.long().argmax(d), torch.tensor(h.shape[d])) + 1
            ar = torch.arange(h.shape[d]).view([-1 if i == d else 1 for i in range(h.dim())])
            mask = mask | (ar >= first.unsqueeze(d))
        ref_out = lsm.gather(-1, h.masked_fill(mask, 0).unsqueeze(-1)).squeeze(-1).masked_fill(mask, 0.0).sum(dim)
        gw = torch.randn(ref_out.shape, dtype=torch.double)
        (ge,) = torch.autograd.grad((ref_out * gw).sum(), xc)
        (ga,) = torch.autograd.grad((act * gw.float().to(device)).sum(), x)
        assert torch.allclose(ga.cpu().double(), ge, rtol=1e-4, atol=1e-5), it


@pytest.mark.parametrize("V", [65, 512, 513, 1024, 1025, 2500])
def test_sequence_ops_row_forms(device, V):
    """Rows held in 8 registers per lane (V <= 512), in 16 (<= 1024) and streamed in two passes
    (beyond); sequences long enough for all 16 waves of a workgroup and for more than 64 steps per
    wave; eos in the first, a middle and no position."""
    rng = np.random.default_rng(V)
    S, N = (1100 if V == 65 else 70), 3
    lg = rng.normal(size=(S, N, V)).astype(np.float32) * 3
    hyp = rng.integers(0, V, (S, N))
    eos = V - 1
    hyp[hyp == eos] = 0
    hyp[0, 0] = eos
    hyp[S // 2, 1] = eos
    hyp[5, 2] = -1  # masked by value
    exp = oracle.sequence_log_probs(lg, hyp, 0, eos)
    x = T(lg, device).requires_grad_(True)
    act = F.sequence_log_probs(x, T(hyp, device), 0, eos)
    assert np.allclose(act.detach().cpu().numpy(), exp, rtol=2e-5, atol=1e-4)
    (ga,) = torch.autograd.grad(act.sum(), x)
    xc = torch.from_numpy(lg).double().requires_grad_(True)
    h = torch.from_numpy(hyp)
    first = torch.where((h == eos).any(0), (h == eos).long().argmax(0), torch.tensor(S)) + 1
    mask = (h < 0) | (torch.arange(S).unsqueeze(1) >= first.unsqueeze(0))
    ref = xc.log_softmax(-1).gather(-1, h.masked_fill(mask, 0).unsqueeze(-1)).squeeze(-1).masked_fill(mask, 0.0).sum()
    (ge,) = torch.autograd.grad(ref, xc)
    assert torch.allclose(ga.cpu().double(), ge, rtol=1e-4, atol=1e-5)
    lens = np.array([S, S // 3, 0])
    e_max, e_paths, e_lens = oracle.ctc_greedy_search(lg, lens, V - 1, False, False)
    a_max, a_paths, a_lens = F.ctc_greedy_search(T(lg, device), T(lens, device), V - 1)
    assert np.array_equal(a_lens.cpu().numpy(), e_lens) and np.array_equal(a_paths.cpu().numpy(), e_paths)
    assert np.allclose(a_max.cpu().numpy(), e_max, rtol=2e-5, atol=1e-3)


def test_sequence_log_probs_packed(device):
    rng = np.random.default_rng(4)
    S, N, V = 7, 4, 5
    lens = torch.tensor([7, 3, 5, 1])
    lg = torch.from_numpy(rng.normal(size=(S, N, V)).astype(np.float32)).to(device)
    hyp = torch.from_numpy(rng.integers(0, V, (S, N))).to(device)
    ps = torch.nn.utils.rnn.pack_padded_sequence(lg, lens, enforce_sorted=False)
    act = F.sequence_log_probs(ps, hyp, 0)
    hm = hyp.masked_fill(torch.arange(S, device=device).unsqueeze(1) >= lens.to(device).unsqueeze(0), -1)
    exp = oracle.sequence_log_probs(lg.cpu().numpy(), hm.cpu().numpy(), 0, None)
    assert np.allclose(act.cpu().numpy(), exp, rtol=1e-5, atol=1e-5)


def test_random_walk_statistics(device):
    """Sample frequencies follow the LM (reference tests/test_decoding.py:769-840, in spirit)."""
    torch.manual_seed(5)
    V = 4
    table = torch.tensor([[0.7, 0.1, 0.1, 0.1]] * (V + 1)).log().to(device)
    lm = BigramLM(table).to(device)
    walk = M.RandomWalk(lm, eos=3).to(device)
    y, lens, lp = walk(dict(), batch_size=4000, max_iters=20)
    assert y.shape[1] == 4000 and lens.shape == (4000,) and lp.shape == (4000,)
    first = torch.bincount(y[0], minlength=V).float() / 4000
    assert torch.allclose(first.cpu(), torch.tensor([0.7, 0.1, 0.1, 0.1]), atol=0.03)
    # log prob of a path = sum of its step log probs, up to and including its first eos
    n = 0
    toks = y[: lens[n], n].cpu()
    assert abs(lp[n].item() - table[0].cpu()[toks].sum().item()) < 1e-4
    y2, lp2 = F.random_walk_advance(table[:1].expand(8, V), torch.zeros(8, device=device),
                                    torch.zeros((0, 8), dtype=torch.long, device=device))  # fmt: skip
    assert y2.shape == (1, 8) and lp2.shape == (8,)


@pytest.mark.parametrize("is_probs", [False, True])
@pytest.mark.parametrize("batch_first", [False, True])
def test_ctc_greedy_search_max_gradient(device, is_probs, batch_first):
    """``max_`` stays in the graph as in the reference (_decoding.py:526-553)."""
    rng = np.random.default_rng(2 + is_probs)
    T, N, V = 9, 4, 6
    x = rng.normal(size=(T, N, V)).astype(np.float32)
    if is_probs:
        x = np.exp(x) / np.exp(x).sum(2, keepdims=True)
    if batch_first:
        x = np.ascontiguousarray(x.transpose(1, 0, 2))
    lens = torch.tensor([9, 4, 0, 7])
    w = torch.from_numpy(rng.normal(size=(N,)).astype(np.float32))
    a = torch.from_numpy(x).clone().requires_grad_(True)
    y = a if is_probs else a.log_softmax(2)
    y = y if batch_first else y.transpose(0, 1)
    best = y.max(2)[0]
    valid = torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)
    best = best.masked_fill(~valid, 1.0 if is_probs else 0.0)
    ((best.prod(1) if is_probs else best.sum(1)) * w).sum().backward()
    b = torch.from_numpy(x).to(device).requires_grad_(True)
    mx, paths, out_lens = F.ctc_greedy_search(b, lens.to(device), -1, batch_first, is_probs)
    assert mx.requires_grad
    (mx * w.to(device)).sum().backward()
    assert torch.allclose(b.grad.cpu(), a.grad, rtol=1e-4, atol=1e-6)
