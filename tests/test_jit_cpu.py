"""Every Module of the path compiles under TorchScript (no GPU needed: scripting does not run
the kernels), and the kernels are visible as opaque ``pydrobert_amd::*`` operators.

Mirrors the reference's ``jit_type`` parametrisation (tests/test_string.py:39-42,
test_decoding.py:221-231, test_img.py:39-41) for the operators SURVEY.md section 8 names.
"""
import torch

from pydrobert_amd import modules as M

from _toy_lm import ScriptableBigramLM


def _modules():
    lm = ScriptableBigramLM(torch.randn(6, 5).log_softmax(-1))
    return [
        M.FillAfterEndOfSequence(0),
        M.EditDistance(eos=1),
        M.ErrorRate(),
        M.PrefixEditDistances(eos=3),
        M.PrefixErrorRates(),
        M.OptimalCompletion(eos=2),
        M.HardOptimalCompletionDistillationLoss(eos=1, weight=torch.ones(5)),
        M.MinimumErrorRateLoss(),
        M.CTCGreedySearch(),
        M.SequenceLogProbabilities(1, 3),
        M.CTCPrefixSearch(4),
        M.CTCPrefixSearch(4, 0.3, lm),
        M.BeamSearch(lm, 3, eos=0),
        M.RandomWalk(lm, eos=0),
        M.PolyharmonicSpline(2),
        M.Warp1DGrid(),
        M.Warp1DGrid(10, 2),
        M.DenseImageWarp(),
        M.SparseImageWarp(),
        M.SparseImageWarp(include_flow=False, pinned_boundary_points=2),
        M.SpecAugment(max_freq_warp=3.0),
        M.LookupLanguageModel(5, 0),
        M.LookupLanguageModel(5, -1, [{0: (0.0, 0.0), 1: (-1.0, -0.5)}, {(1, 0): -0.3}]),
        M.CTCPrefixSearch(3, 0.5, M.LookupLanguageModel(5, -1, [{0: (0.0, 0.0)}, {(0, 0): -0.3}])),
    ]


def test_every_module_scripts():
    for m in _modules():
        torch.jit.script(m)


def test_kernels_are_registered_operators():
    names = [
        "string_matching", "optimal_completion", "ocd_loss_rows", "ocd_loss_rows_backward",
        "beam_search_advance", "ctc_prefix_search_advance", "ctc_prefix_search",
        "ctc_greedy_search", "sequence_log_probs", "sequence_log_probs_backward",
        "polyharmonic_spline", "warp_1d_grid", "dense_image_warp", "dense_image_warp_backward",
        "sparse_image_warp", "spec_augment_apply", "spec_augment_apply_backward",
        "sparse_image_warp_backward", "lookup_lm_log_probs",
    ]  # fmt: skip
    for name in names:
        assert hasattr(torch.ops.pydrobert_amd, name), name


def test_scripted_graph_holds_the_opaque_operator():
    graph = str(torch.jit.script(M.ErrorRate(eos=0)).inlined_graph)
    assert "pydrobert_amd::string_matching" in graph
    graph = str(torch.jit.script(M.CTCPrefixSearch(8)).inlined_graph)
    assert "pydrobert_amd::ctc_prefix_search" in graph


def test_operators_refuse_cpu_tensors():
    # no CPU fallback behind the operators either
    import pytest

    ref = torch.zeros((3, 2), dtype=torch.long)
    with pytest.raises(RuntimeError):
        torch.jit.script(M.ErrorRate())(ref, ref)
