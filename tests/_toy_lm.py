"""A stateless bigram-table language model under this package's LM interface (the same model
tests/golden/make_golden.py defines under the reference's interface)."""
import torch

from pydrobert_amd.modules import MixableSequentialLanguageModel


class BigramLM(MixableSequentialLanguageModel):
    def __init__(self, table):
        super().__init__(table.shape[1])
        self.register_buffer("table", table)

    def calc_idx_log_probs(self, hist, prev, idx):
        V = self.vocab_size
        N = hist.shape[1]
        if idx.dim() == 0:
            idx = idx.expand(N)
        prev_tok = torch.full((N,), V, dtype=torch.long, device=hist.device)
        if hist.shape[0]:
            last = hist.gather(0, (idx - 1).clamp(min=0).unsqueeze(0)).squeeze(0).clamp(0, V - 1)
            prev_tok = torch.where(idx > 0, last, prev_tok)
        return self.table[prev_tok], prev

    def extract_by_src(self, prev, src):
        return prev

    def mix_by_mask(self, prev_true, prev_false, mask):
        return prev_true


from typing import Dict, Tuple  # noqa: E402


class ScriptableBigramLM(MixableSequentialLanguageModel):
    """The same model with the type annotations / exports TorchScript needs."""

    def __init__(self, table: torch.Tensor):
        super().__init__(table.shape[1])
        self.register_buffer("table", table)

    @torch.jit.export
    def calc_idx_log_probs(
        self, hist: torch.Tensor, prev: Dict[str, torch.Tensor], idx: torch.Tensor
    ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        V = self.vocab_size
        N = hist.shape[1]
        if idx.dim() == 0:
            idx = idx.expand(N)
        prev_tok = torch.full((N,), V, dtype=torch.long, device=hist.device)
        if hist.shape[0]:
            last = hist.gather(0, (idx - 1).clamp(min=0).unsqueeze(0)).squeeze(0).clamp(0, V - 1)
            prev_tok = torch.where(idx > 0, last, prev_tok)
        return self.table[prev_tok], prev

    @torch.jit.export
    def extract_by_src(
        self, prev: Dict[str, torch.Tensor], src: torch.Tensor
    ) -> Dict[str, torch.Tensor]:
        return prev

    @torch.jit.export
    def mix_by_mask(
        self,
        prev_true: Dict[str, torch.Tensor],
        prev_false: Dict[str, torch.Tensor],
        mask: torch.Tensor,
    ) -> Dict[str, torch.Tensor]:
        return prev_true


class CounterLM(MixableSequentialLanguageModel):
    """A model WITH state (the torch twin of oracle.CounterLM): every row of the flattened beam
    carries a counter of the scoring calls it has been through -- reordered by extract_by_src, kept or
    replaced by mix_by_mask exactly as the searches do -- and the scores depend on it, so a search that
    reorders the state wrongly decodes differently."""

    def __init__(self, table):
        super().__init__(table.shape[1])
        self.register_buffer("table", table)

    def update_input(self, prev, hist):
        if "count" not in prev:
            prev = {"count": torch.zeros((hist.size(1),), device=hist.device)}
        return prev

    def calc_idx_log_probs(self, hist, prev, idx):
        V = self.vocab_size
        N = hist.shape[1]
        if idx.dim() == 0:
            idx = idx.expand(N)
        prev_tok = torch.full((N,), V, dtype=torch.long, device=hist.device)
        if hist.shape[0]:
            last = hist.gather(0, (idx - 1).clamp(min=0).unsqueeze(0)).squeeze(0).clamp(0, V - 1)
            prev_tok = torch.where(idx > 0, last, prev_tok)
        x = self.table[prev_tok] * (1.0 + 0.1 * prev["count"]).unsqueeze(1)
        return x.log_softmax(-1), {"count": prev["count"] + 1.0}

    def extract_by_src(self, prev, src):
        return {"count": prev["count"].index_select(0, src)}

    def mix_by_mask(self, prev_true, prev_false, mask):
        return {"count": torch.where(mask, prev_true["count"], prev_false["count"])}
