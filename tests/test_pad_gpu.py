"""pad_variable / RandomShift on the GPU: against the live
reference's outputs (tests/golden/pad.npz), the oracle on random shapes / dtypes, and the
backward kernel against autograd through an equivalent gather graph."""
import os

import numpy as np
import pytest
import torch

import oracle
from pydrobert_amd import functional as F
from pydrobert_amd import modules as M

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pad.npz")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("mode", ["constant", "reflect", "replicate"])
def test_pad_variable_golden(mode):
    g = np.load(GOLDEN)
    act = M.PadVariable(mode, -1.5)(_t(g["x"]), _t(g["lens"]), _t(g["pad"]))
    assert np.array_equal(g["out_" + mode], act.cpu().numpy())
    act = F.pad_variable(_t(g["xi"]), _t(g["lens"]), _t(g["pad"]), mode, 7)
    assert act.dtype == torch.long and np.array_equal(g["outi_" + mode], act.cpu().numpy())


@pytest.mark.parametrize("mode", ["constant", "reflect", "replicate"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16, torch.int32, torch.uint8])
def test_pad_variable_random(mode, dtype):
    rng = np.random.default_rng(3)
    for shape in ((7, 33), (5, 20, 4), (3, 11, 2, 3), (4, 1, 5)):
        N, T = shape[:2]
        x = (rng.normal(size=shape) * 20).astype(np.float64)
        lens = rng.integers(1, T + 1, N)
        hi = np.maximum(lens - 1, 0) if mode == "reflect" else np.full(N, 12)
        pad = np.stack([rng.integers(0, hi + 1), rng.integers(0, hi + 1)])
        xt = torch.from_numpy(x).to(dtype)
        exp = oracle.pad_variable(xt.numpy(), lens, pad, mode, 3.0)
        act = F.pad_variable(xt.to(DEV), _t(lens), _t(pad), mode, 3.0)
        assert act.dtype == dtype and np.array_equal(exp, act.cpu().numpy()), (shape, mode)


def test_pad_variable_edge_cases():
    x = torch.randn((3, 5, 2), device=DEV)
    lens = torch.tensor([5, 3, 2], device=DEV)
    zero = torch.zeros((2, 3), dtype=torch.long, device=DEV)
    assert torch.equal(F.pad_variable(x, torch.tensor([5, 5, 5], device=DEV), zero), x)
    with pytest.raises(NotImplementedError, match="reflect"):
        F.pad_variable(x, lens, torch.tensor([[0, 3, 0], [0, 0, 0]], device=DEV), "reflect")
    with pytest.raises(RuntimeError, match="replicate"):
        F.pad_variable(x, torch.tensor([5, 0, 2], device=DEV), zero, "replicate")
    with pytest.raises(ValueError):
        F.pad_variable(x, lens[:2], zero)
    with pytest.raises(ValueError):
        F.pad_variable(x, lens, zero[:, :2])
    empty = F.pad_variable(x[:0], lens[:0], zero[:, :0])
    assert empty.shape == (0, 0, 2)


@pytest.mark.parametrize("mode", ["constant", "reflect", "replicate"])
def test_pad_variable_backward(mode):
    torch.manual_seed(4)
    N, T, Fq = 6, 12, 5
    x = torch.randn((N, T, Fq), device=DEV)
    lens = torch.tensor([12, 7, 3, 9, 12, 5], device=DEV)
    pad = torch.tensor([[2, 0, 1, 4, 0, 3], [1, 5, 2, 0, 0, 4]], device=DEV)
    x1 = x.clone().requires_grad_(True)
    y = F.pad_variable(x1, lens, pad, mode, 0.5)
    g = torch.randn_like(y)
    (act,) = torch.autograd.grad(y, x1, g)
    # equivalent differentiable graph: gather along time through the oracle's index map
    idx = np.tile(np.arange(T)[None], (N, 1))
    src = oracle.pad_variable(idx, lens.cpu().numpy(), pad.cpu().numpy(), mode, -1)
    src_t = torch.from_numpy(src).to(DEV)
    x2 = x.clone().requires_grad_(True)
    gathered = x2.gather(1, src_t.clamp(min=0).unsqueeze(2).expand(-1, -1, Fq))
    y2 = torch.where((src_t >= 0).unsqueeze(2), gathered, torch.full_like(gathered, 0.5))
    assert torch.equal(y, y2)
    (exp,) = torch.autograd.grad(y2, x2, g)
    assert torch.allclose(exp, act, atol=1e-5)


def test_random_shift():
    """Facts of the reference's tests/test_img.py:284-344: lengths grow within the bounds,
    the original sequence sits inside the output, eval mode is the identity."""
    torch.manual_seed(5)
    N, T, Fq = 50, 30, 4
    x = torch.rand((N, T, Fq), device=DEV) + 0.01
    lens = torch.randint(2, T + 1, (N,), device=DEV)
    for mode in ("reflect", "constant", "replicate"):
        shift = M.RandomShift((0.4, 0.6), mode, 0.0)
        out, out_lens = shift(x, lens)
        grow = out_lens - lens
        assert (grow >= 0).all() and (grow <= (0.4 * lens.float()).long() + (0.6 * lens.float()).long()).all()
        assert out.shape[:2] == (N, int(out_lens.max())) and out.shape[2:] == x.shape[2:]
        # beyond the new length: the fill value
        beyond = torch.arange(out.shape[1], device=DEV).unsqueeze(0) >= out_lens.unsqueeze(1)
        assert (out[beyond] == 0).all()
        if mode == "constant":  # the copy of x is the only non-zero stretch
            nz = (out.abs().sum(2) > 0).sum(1)
            assert torch.equal(nz, lens)
        shift.eval()
        same, same_lens = shift(x, lens)
        assert same is x and same_lens is lens
    with pytest.raises(NotImplementedError):
        M.RandomShift(1.5, "reflect")
    with pytest.raises(ValueError):
        M.RandomShift(-0.1)
    jit = torch.jit.script(M.RandomShift(0.3, "replicate"))
    out, out_lens = jit(x, lens)
    assert out.shape[1] == int(out_lens.max())
