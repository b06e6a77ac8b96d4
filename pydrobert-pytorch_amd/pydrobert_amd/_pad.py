"""Variable-length padding on MI355X (reference _pad.py:108-149, Module :152-238); the
movement behind :class:`RandomShift`.  One HIP pass (``csrc/pad_variable.hip``) instead of
the reference's masks + masked_scatter chain."""
import torch
from torch.library import custom_op, register_autograd

from . import _cabi, argcheck, config

__all__ = ["PadVariable", "pad_variable"]

_PAD_MODES = {"constant": 0, "reflect": 1, "replicate": 2}


@custom_op("pydrobert_amd::pad_variable", mutates_args=())
def _pad_variable_op(
    x: torch.Tensor, lens: torch.Tensor, pad: torch.Tensor, mode: str, value: float
) -> torch.Tensor:
    if x.dim() < 2:
        raise ValueError("Expected x to be at least two dimensional")
    shape = x.shape
    N, T = shape[0], shape[1]
    if lens.shape != (N,):
        raise ValueError(
            "For x of shape {}, lens should have shape ({},) but got {}".format(tuple(shape), N, tuple(lens.shape))
        )
    if pad.shape != (2, N):
        raise ValueError(
            "For x of shape {}, pad should have shape (2, {}), but got {}".format(tuple(shape), N, tuple(pad.shape))
        )
    if mode not in _PAD_MODES:
        raise ValueError("mode must be one of 'constant', 'reflect', 'replicate', got '{}'".format(mode))
    device = _cabi.require_hip(x, lens, pad)
    xc = x.detach().contiguous()
    ln = lens.detach().long().contiguous()
    pd = pad.detach().long().contiguous()
    F = 1
    for d in shape[2:]:
        F *= d
    with torch.cuda.device(device):
        # one read-back, like the reference's new_lens.max().item() (:128); the same trip
        # carries the two data checks of _get_padding_buffers (:52-56, :82-83)
        new_lens = ln + pd.sum(0)
        stats = torch.stack([
            new_lens.max() if N else ln.new_zeros(()),
            ((pd >= ln.unsqueeze(0)).any() if N else ln.new_zeros((), dtype=torch.bool)).long(),
            ((ln < 1).any() if N else ln.new_zeros((), dtype=torch.bool)).long(),
        ]).tolist()  # fmt: skip
        Tp, pad_ge_len, len_lt_1 = int(stats[0]), bool(stats[1]), bool(stats[2])
        if mode == "reflect" and pad_ge_len:
            raise NotImplementedError(
                "For reflect padding, all padding lengths must be less than the sequence length"
            )
        if mode == "replicate" and len_lt_1:
            raise RuntimeError("For replicate padding, all lens must be > 0")
        out = torch.empty((N, Tp) + tuple(shape[2:]), device=device, dtype=x.dtype)
        fill = torch.full((1,), value, device=device, dtype=x.dtype)
        rc = _cabi.lib().pdt_pad_variable(
            _cabi.ptr(xc) if xc.numel() else None, N, T, F, xc.element_size(), _cabi.ptr(ln),
            _cabi.ptr(pd), _PAD_MODES[mode], _cabi.ptr(fill), Tp,
            _cabi.ptr(out) if out.numel() else None, _cabi.stream_ptr(device),
        )  # fmt: skip
    _cabi.check(rc, "pdt_pad_variable")
    return out


@_pad_variable_op.register_fake
def _(x, lens, pad, mode, value):
    Tp = torch.library.get_ctx().new_dynamic_size()
    return x.new_empty((x.shape[0], Tp) + tuple(x.shape[2:]))


@custom_op("pydrobert_amd::pad_variable_backward", mutates_args=())
def _pad_variable_backward_op(
    grad_out: torch.Tensor, lens: torch.Tensor, pad: torch.Tensor, mode: str, T: int
) -> torch.Tensor:
    device = _cabi.require_hip(grad_out, lens, pad)
    g = grad_out.detach().float().contiguous()
    N, Tp = g.shape[0], g.shape[1]
    F = 1
    for d in g.shape[2:]:
        F *= d
    ln, pd = lens.detach().long().contiguous(), pad.detach().long().contiguous()
    with torch.cuda.device(device):
        grad = torch.empty((N, T) + tuple(g.shape[2:]), device=device, dtype=torch.float)
        rc = _cabi.lib().pdt_pad_variable_backward(
            _cabi.ptr(g) if g.numel() else None, N, T, F, _cabi.ptr(ln), _cabi.ptr(pd),
            _PAD_MODES[mode], Tp, _cabi.ptr(grad) if grad.numel() else None, _cabi.stream_ptr(device),
        )  # fmt: skip
    _cabi.check(rc, "pdt_pad_variable_backward")
    return grad.to(grad_out.dtype)


@_pad_variable_backward_op.register_fake
def _(grad_out, lens, pad, mode, T):
    return grad_out.new_empty((grad_out.shape[0], T) + tuple(grad_out.shape[2:]))


def _pad_setup_context(ctx, inputs, output):
    x, lens, pad, mode, _ = inputs
    ctx.save_for_backward(lens, pad)
    ctx.cfg = (mode, x.shape[1])


def _pad_backward(ctx, grad_out):
    lens, pad = ctx.saved_tensors
    mode, T = ctx.cfg
    g = torch.ops.pydrobert_amd.pad_variable_backward(grad_out, lens, pad, mode, T)
    return g, None, None, None, None


register_autograd("pydrobert_amd::pad_variable", _pad_backward, setup_context=_pad_setup_context)


def pad_variable(
    x: torch.Tensor,
    lens: torch.Tensor,
    pad: torch.Tensor,
    mode: str = "constant",
    value: float = config.DEFT_PAD_VALUE,
) -> torch.Tensor:
    """Functional version of :class:`PadVariable` (reference _pad.py:108-149)."""
    return torch.ops.pydrobert_amd.pad_variable(x, lens, pad, mode, value)


class PadVariable(torch.nn.Module):
    """Pad variable-length input by a variable amount on each side (reference _pad.py:152-238)."""

    __constants__ = ("mode", "value")

    def __init__(self, mode: str = "constant", value: float = config.DEFT_PAD_VALUE):
        mode = argcheck.is_in(mode, tuple(_PAD_MODES), "mode")
        value = argcheck.is_float(value, "value")
        super().__init__()
        self.mode, self.value = mode, value

    def extra_repr(self) -> str:
        s = "mode={}".format(self.mode)
        if self.mode == "constant":
            s += ", value={}".format(self.value)
        return s

    def forward(self, x: torch.Tensor, lens: torch.Tensor, pad: torch.Tensor) -> torch.Tensor:
        return pad_variable(x, lens, pad, self.mode, self.value)
