"""Module namespace -- mirrors ``pydrobert.torch.modules`` (modules.py:28-124) for the
operators on the MI355X hot path."""

from ._decoding import BeamSearch, CTCPrefixSearch
from ._img import (
    DenseImageWarp,
    PolyharmonicSpline,
    RandomShift,
    SparseImageWarp,
    SpecAugment,
    Warp1DGrid,
)
from ._pad import PadVariable
from ._decoding import CTCGreedySearch, RandomWalk, SequenceLogProbabilities
from ._lm import (
    ExtractableSequentialLanguageModel,
    ExtractableShallowFusionLanguageModel,
    LookupLanguageModel,
    MixableSequentialLanguageModel,
    MixableShallowFusionLanguageModel,
    SequentialLanguageModel,
    ShallowFusionLanguageModel,
)
from ._string import (
    HardOptimalCompletionDistillationLoss,
    MinimumErrorRateLoss,
    EditDistance,
    ErrorRate,
    FillAfterEndOfSequence,
    OptimalCompletion,
    PrefixEditDistances,
    PrefixErrorRates,
)

__all__ = [
    "PadVariable",
    "RandomShift",
    "DenseImageWarp",
    "PolyharmonicSpline",
    "SparseImageWarp",
    "SpecAugment",
    "Warp1DGrid",
    "CTCGreedySearch",
    "RandomWalk",
    "SequenceLogProbabilities",
    "HardOptimalCompletionDistillationLoss",
    "MinimumErrorRateLoss",
    "BeamSearch",
    "CTCPrefixSearch",
    "ExtractableSequentialLanguageModel",
    "ExtractableShallowFusionLanguageModel",
    "LookupLanguageModel",
    "MixableShallowFusionLanguageModel",
    "ShallowFusionLanguageModel",
    "MixableSequentialLanguageModel",
    "SequentialLanguageModel",
    "EditDistance",
    "ErrorRate",
    "FillAfterEndOfSequence",
    "OptimalCompletion",
    "PrefixEditDistances",
    "PrefixErrorRates",
]
