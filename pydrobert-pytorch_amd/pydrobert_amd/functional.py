"""Functional namespace -- mirrors ``pydrobert.torch.functional`` (functional.py:17-95)
for the operators on the MI355X hot path."""

from ._decoding import ctc_prefix_search
from ._string import (
    edit_distance,
    error_rate,
    fill_after_eos,
    optimal_completion,
    prefix_edit_distances,
    prefix_error_rates,
)

__all__ = [
    "ctc_prefix_search",
    "edit_distance",
    "error_rate",
    "fill_after_eos",
    "optimal_completion",
    "prefix_edit_distances",
    "prefix_error_rates",
]
