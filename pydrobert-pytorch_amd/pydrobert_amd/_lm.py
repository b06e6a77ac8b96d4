"""The language-model scoring interface the searches call.

Mirrors the three abstract classes of the reference (``_lm.py:45-400``): a user model
subclasses them; its forward pass (embedding / recurrent cell / ``Linear`` to vocabulary
logits) is ordinary PyTorch and runs on rocBLAS / hipBLASLt -- the only MFMA-shaped work on
this path.  Only the interface lives here.
"""
import abc
from typing import Any, Dict, Optional, Tuple

import torch

from . import argcheck

__all__ = [
    "ExtractableSequentialLanguageModel",
    "MixableSequentialLanguageModel",
    "SequentialLanguageModel",
]


class SequentialLanguageModel(torch.nn.Module, metaclass=abc.ABCMeta):
    """P(w) = prod_s P(w_s | w_<s): a model queried one position at a time (_lm.py:45-288).

    Subclasses implement :meth:`calc_idx_log_probs`; ``vocab_size`` fixes the last output
    dimension.  ``forward(hist, prev=None, idx=None)`` returns the log-probabilities of all
    positions ``(S + 1, N, V)`` when ``idx`` is None, else ``(log_probs (N, V), next_state)``
    for position(s) ``idx`` (an int, a 0-dim or an ``(N,)`` tensor; negative values count from
    the end).
    """

    __constants__ = ("vocab_size",)

    def __init__(self, vocab_size: int):
        vocab_size = argcheck.is_posi(vocab_size, "vocab_size")
        super().__init__()
        self.vocab_size = vocab_size

    def update_input(
        self, prev: Dict[str, torch.Tensor], hist: torch.Tensor
    ) -> Dict[str, torch.Tensor]:
        """Initialise / complete the state dictionary before any query (idempotent)."""
        return prev

    def extra_repr(self) -> str:
        return "vocab_size={}".format(self.vocab_size)

    @abc.abstractmethod
    def calc_idx_log_probs(
        self, hist: torch.Tensor, prev: Dict[str, torch.Tensor], idx: torch.Tensor
    ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """Log-probabilities ``(N, V)`` of the token at position ``idx`` given
        ``hist[:idx]``, and the state for the next position."""
        raise NotImplementedError()

    def calc_full_log_probs(
        self, hist: torch.Tensor, prev: Dict[str, torch.Tensor]
    ) -> torch.Tensor:
        out = []
        for i in range(hist.size(0) + 1):
            idx = torch.tensor(i, device=hist.device)
            lp, prev = self.calc_idx_log_probs(hist, prev, idx)
            out.append(lp)
        return torch.stack(out, 0)

    def forward(
        self,
        hist: torch.Tensor,
        prev: Optional[Dict[str, torch.Tensor]] = None,
        idx: Optional[Any] = None,
    ) -> Any:
        prev_: Dict[str, torch.Tensor] = dict()
        if prev is not None:
            prev_ = prev
        if hist.dim() != 2:
            raise RuntimeError("hist must be 2 dimensional")
        prev_ = self.update_input(prev_, hist)
        if idx is None:
            return self.calc_full_log_probs(hist, prev_)
        if isinstance(idx, int):
            return self._at(hist, prev_, torch.as_tensor(idx, dtype=torch.long, device=hist.device))
        elif isinstance(idx, torch.Tensor):
            return self._at(hist, prev_, idx.to(device=hist.device, dtype=torch.long))
        else:
            raise RuntimeError("idx must be an int or a tensor")

    def _at(
        self, hist: torch.Tensor, prev: Dict[str, torch.Tensor], idx: torch.Tensor
    ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        S, N = hist.size(0), hist.size(1)
        if not idx.numel():
            raise RuntimeError("idx_ must be at least one element")
        if idx.dim() == 1:
            if idx.size(0) == 1:
                idx = idx.squeeze(0)
            elif idx.size(0) != N:
                raise RuntimeError(
                    "Expected dim 0 of idx_ to be of size {}, got {}".format(N, idx.size(0))
                )
        if bool(((idx < -S - 1) | (idx > S)).any()):
            raise RuntimeError("All values in idx_ must be between ({}, {})".format(-S - 1, S))
        idx = (idx + S + 1) % (S + 1)
        return self.calc_idx_log_probs(hist, prev, idx)


class ExtractableSequentialLanguageModel(SequentialLanguageModel, metaclass=abc.ABCMeta):
    """A model whose state can follow a re-ordering of the batch (_lm.py:291-338):
    ``extract_by_src(prev, src)[...][n] = prev[...][src[n]]``."""

    @abc.abstractmethod
    def extract_by_src(
        self, prev: Dict[str, torch.Tensor], src: torch.Tensor
    ) -> Dict[str, torch.Tensor]:
        raise NotImplementedError()


class MixableSequentialLanguageModel(ExtractableSequentialLanguageModel, metaclass=abc.ABCMeta):
    """... and whose state can be chosen per batch element between two candidates
    (_lm.py:341-400): entry ``n`` comes from ``prev_true`` where ``mask[n]`` else
    ``prev_false``."""

    @abc.abstractmethod
    def mix_by_mask(
        self,
        prev_true: Dict[str, torch.Tensor],
        prev_false: Dict[str, torch.Tensor],
        mask: torch.Tensor,
    ) -> Dict[str, torch.Tensor]:
        raise NotImplementedError()
