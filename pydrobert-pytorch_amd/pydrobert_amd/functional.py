"""Functional namespace -- mirrors ``pydrobert.torch.functional`` (functional.py:17-95)
for the operators on the MI355X hot path."""

from ._decoding import (
    beam_search_advance,
    ctc_prefix_search,
    ctc_prefix_search_advance,
)
from ._decoding import ctc_greedy_search, random_walk_advance, sequence_log_probs
from ._img import (
    dense_image_warp,
    polyharmonic_spline,
    random_shift,
    sparse_image_warp,
    spec_augment,
    spec_augment_apply_parameters,
    spec_augment_draw_parameters,
    warp_1d_grid,
)
from ._pad import pad_variable
from ._string import (
    hard_optimal_completion_distillation_loss,
    minimum_error_rate_loss,
    edit_distance,
    error_rate,
    fill_after_eos,
    optimal_completion,
    prefix_edit_distances,
    prefix_error_rates,
)

__all__ = [
    "pad_variable",
    "random_shift",
    "ctc_greedy_search",
    "random_walk_advance",
    "sequence_log_probs",
    "hard_optimal_completion_distillation_loss",
    "minimum_error_rate_loss",
    "beam_search_advance",
    "ctc_prefix_search",
    "ctc_prefix_search_advance",
    "dense_image_warp",
    "polyharmonic_spline",
    "sparse_image_warp",
    "spec_augment",
    "spec_augment_apply_parameters",
    "spec_augment_draw_parameters",
    "warp_1d_grid",
    "edit_distance",
    "error_rate",
    "fill_after_eos",
    "optimal_completion",
    "prefix_edit_distances",
    "prefix_error_rates",
]
