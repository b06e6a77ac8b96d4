"""Module namespace -- mirrors ``pydrobert.torch.modules`` (modules.py:28-124) for the
operators on the MI355X hot path."""

from ._decoding import CTCPrefixSearch
from ._string import (
    EditDistance,
    ErrorRate,
    FillAfterEndOfSequence,
    OptimalCompletion,
    PrefixEditDistances,
    PrefixErrorRates,
)

__all__ = [
    "CTCPrefixSearch",
    "EditDistance",
    "ErrorRate",
    "FillAfterEndOfSequence",
    "OptimalCompletion",
    "PrefixEditDistances",
    "PrefixErrorRates",
]
