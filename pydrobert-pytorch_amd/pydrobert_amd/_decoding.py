"""Beam-search decoding on MI355X.

Host-side mirror of the reference's ``_decoding.py`` for the operators on the hot path:
``CTCPrefixSearch`` / ``ctc_prefix_search_advance`` and ``BeamSearch`` /
``beam_search_advance``.  The searches run in ``csrc/ctc_search.hip`` and
``csrc/beam_advance.hip`` through the C ABI (``include/pdt_amd.h``).
"""
from typing import Dict, Optional, Tuple

import torch

from . import _cabi, argcheck

__all__ = ["CTCPrefixSearch", "ctc_prefix_search"]

MAX_WIDTH = 32


def ctc_prefix_search(
    logits: torch.Tensor, width: int, lens: Optional[torch.Tensor] = None
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``CTCPrefixSearch(width)(logits, lens)`` without a language model, as ONE kernel.

    Reference: ``CTCPrefixSearch.forward`` (_decoding.py:1064-1202) with ``lm=None``.
    Returns ``(y (S, N, width) int64, y_lens (N, width) int64, y_probs (N, width))``; rows of
    ``y`` beyond ``y_lens`` are zero (the reference leaves them undefined).
    """
    if logits.dim() != 3:
        raise RuntimeError("logits must be 3 dimensional")  # :1073-1074
    device = _cabi.require_hip(logits, lens)
    T, N, Vp1 = logits.shape
    V = Vp1 - 1
    if V < 1:
        raise RuntimeError("logits must have at least one non-blank class")
    if width < 1:
        raise RuntimeError("width must be positive")
    if width > MAX_WIDTH:
        raise RuntimeError("width {} exceeds the MI355X kernel's limit of {}".format(width, MAX_WIDTH))
    logits = logits.detach()
    if logits.dtype != torch.float:
        logits = logits.float()
    if lens is None:
        S = T
    elif lens.dim() != 1:
        raise RuntimeError("lens must be 1 dimensional")  # :1084-1085
    elif lens.size(0) != N:
        raise RuntimeError("expected dim 0 of lens to be {}, got {}".format(N, lens.size(0)))
    else:
        lens = lens.long().contiguous()
        S = int(lens.max().item()) if N else 0  # the reference's len_max host read (:1089)
        S = max(0, min(S, T))
    L = _cabi.lib()
    with torch.cuda.device(device):
        y = torch.zeros((S, N, width), device=device, dtype=torch.long)
        y_lens = torch.empty((N, width), device=device, dtype=torch.long)
        y_probs = torch.empty((N, width), device=device, dtype=torch.float)
        ws = torch.empty(
            (int(L.pdt_ctc_prefix_search_workspace_bytes(T, N, width)),),
            device=device, dtype=torch.uint8,
        )  # fmt: skip
        rc = L.pdt_ctc_prefix_search(
            _cabi.ptr(logits), T, N, V, logits.stride(0), logits.stride(1), logits.stride(2),
            _cabi.ptr(lens), int(width), S, _cabi.ptr(y), _cabi.ptr(y_lens), _cabi.ptr(y_probs),
            _cabi.ptr(ws), _cabi.stream_ptr(device),
        )  # fmt: skip
    _cabi.check(rc, "pdt_ctc_prefix_search")
    return y, y_lens, y_probs


class CTCPrefixSearch(torch.nn.Module):
    """Beam search over CTC prefixes (reference _decoding.py:937-1204).

    ``lm=None`` (or ``beta == 0``) runs the fused MI355X kernel.
    """

    __constants__ = ["width", "beta", "valid_mixture"]

    def __init__(self, width: int, beta: float = 0.2, lm=None, valid_mixture: bool = False):
        width = argcheck.is_posi(width, name="width")
        beta = argcheck.is_closed01(beta, name="beta")
        valid_mixture = argcheck.is_bool(valid_mixture, "valid_mixture")
        super().__init__()
        self.width, self.beta, self.valid_mixture = width, beta, valid_mixture
        if lm is None:
            self.add_module("lm", None)
        else:
            self.lm = lm

    def reset_parameters(self) -> None:
        if self.lm is not None and hasattr(self.lm, "reset_parameters"):
            self.lm.reset_parameters()

    def extra_repr(self) -> str:
        return ", ".join("{}={}".format(x, getattr(self, x)) for x in self.__constants__)

    def forward(
        self,
        logits: torch.Tensor,
        lens: Optional[torch.Tensor] = None,
        initial_state: Optional[Dict[str, torch.Tensor]] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        if self.lm is None or not self.beta:
            return ctc_prefix_search(logits, self.width, lens)
        raise NotImplementedError("shallow fusion: use the step function")
