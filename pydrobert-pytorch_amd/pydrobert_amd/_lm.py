"""The language-model scoring interface the searches call.

Mirrors the three abstract classes of the reference (``_lm.py:45-400``): a user model
subclasses them; its forward pass (embedding / recurrent cell / ``Linear`` to vocabulary
logits) is ordinary PyTorch and runs on rocBLAS / hipBLASLt -- the only MFMA-shaped work on
this path.  Only the interface lives here.
"""
import abc
import warnings
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from torch.library import custom_op

from . import _cabi, argcheck

__all__ = [
    "ExtractableSequentialLanguageModel",
    "ExtractableShallowFusionLanguageModel",
    "LookupLanguageModel",
    "MixableSequentialLanguageModel",
    "MixableShallowFusionLanguageModel",
    "SequentialLanguageModel",
    "ShallowFusionLanguageModel",
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


# ---------------------------------------------------------------------------------------
# the n-gram lookup model and shallow fusion
# ---------------------------------------------------------------------------------------
_INT_TYPES = ((torch.uint8, np.uint8), (torch.int16, np.int16), (torch.int32, np.int32),
              (torch.int64, np.int64))  # fmt: skip


def _smallest_int_type(max_value: int):
    for tt, nt in _INT_TYPES:
        if torch.iinfo(tt).max >= max_value:
            return tt, nt
    raise ValueError("value {} does not fit an int64".format(max_value))


def build_reverse_trie(vocab_size: int, sos: int, prob_dicts, destructive: bool = False, logger=None):
    """Flatten n-gram tables into the reference's reverse-trie buffers (_lm.py:609-677).

    Returns ``(logps, logbs, ids, offsets, max_ngram_nodes)`` with the reference's layout
    and dtypes, so state dicts are interchangeable with ``pydrobert.torch``'s
    ``LookupLanguageModel``.  Unlike the reference's insertion loop (:1012-1046) the levels are
    laid out with array operations: sort every order by reversed key, find each node's parent
    by position, and take child ranges from a ``searchsorted`` over the parent indices.
    """
    info = (lambda msg: None) if logger is None else logger.info
    if not len(prob_dicts):
        raise ValueError("prob_dicts must contain at least unigrams")
    if not destructive:
        prob_dicts = [d.copy() for d in prob_dicts]
    N, V = len(prob_dicts), vocab_size
    shift = 0 if (0 <= sos < V) else 1
    ninf = -float("inf")
    unigrams = set(range(V))
    if shift:
        unigrams.add(sos)
    # validation and completion, highest order first (reference :929-975): every suffix of an
    # entry must itself be an entry, with probability 0 and no back-off penalty if absent
    for n in range(N - 1, -1, -1):
        d = prob_dicts[n]
        if n == N - 1 and not d:
            raise ValueError("Final element in prob_dicts must not be empty")
        if n == 0:
            extra = set(d.keys()) - unigrams
            if extra:
                raise ValueError("Unexpected unigrams in prob_dicts: {} (are these ids?)".format(extra))
            missing = ninf if N == 1 else (ninf, 0.0)
            for key in unigrams - set(d.keys()):
                d[key] = missing
        else:
            lower = prob_dicts[n - 1]
            for seq in d:
                if len(seq) != n + 1:
                    raise ValueError(
                        "Key {0} in {1}-gram is not a sequence of length {1}".format(seq, n + 1)
                    )
                extra = set(seq) - unigrams
                if extra:
                    raise ValueError(
                        "Unexpected tokens in {}-gram in prob_dicts: {} (are these ids?)".format(n + 1, extra)
                    )
                suffix = seq[1] if n == 1 else seq[1:]
                if suffix not in lower:
                    info("{} is a suffix of {} but not an entry; adding (-inf, 0.0)".format(suffix, seq))
                    lower[suffix] = (ninf, 0.0)
    G = len(prob_dicts[-1])
    counts = [len(d) for d in prob_dicts]
    U = V + shift + (1 % N)
    O = sum(counts) - G + (N - 1)
    I, P = O + G - U, O + G
    id_type, _ = _smallest_int_type(U)
    logps = np.zeros(P, dtype=np.float32)
    logbs = np.zeros(O, dtype=np.float32)
    ids = np.zeros(I, dtype=np.int64)
    offsets = np.zeros(O, dtype=np.int64)

    def tok(t):  # sos outside the vocabulary is stored as id V (:977-989)
        return V if (shift and t == sos) else t

    uni = prob_dicts[0]
    if N == 1:
        logps[:] = [uni[sos if (shift and x == V) else x] for x in range(U)]
        return (torch.from_numpy(logps), torch.zeros(0), torch.zeros(0, dtype=id_type),
                torch.zeros(0, dtype=torch.uint8), G)  # fmt: skip
    vals = [uni[sos if (shift and x == V) else x] for x in range(U - 1)]
    logps[: U - 1] = [x[0] for x in vals]
    logbs[: U - 1] = [x[1] for x in vals]
    # level n (order n + 1): reversed keys, lexicographically sorted; the reversed key minus
    # its last element is the parent's reversed key
    prev_keys = np.arange(U - 1, dtype=np.int64).reshape(-1, 1)
    prev_start = 0
    start = U - 1  # index of level 1's dummy node
    for n in range(1, N):
        info("laying out {}-grams".format(n + 1))
        d = prob_dicts[n]
        last = n == N - 1
        cnt = len(d)
        keys = np.empty((cnt, n + 1), dtype=np.int64)
        lp = np.empty(cnt, dtype=np.float32)
        lb = np.zeros(cnt, dtype=np.float32)
        for i, (k, val) in enumerate(d.items()):
            keys[i] = [tok(t) for t in k[::-1]]
            if last:
                lp[i] = val
            else:
                lp[i], lb[i] = val
        order = np.lexsort(keys.T[::-1])
        keys, lp, lb = keys[order], lp[order], lb[order]
        # dummy node closing the previous level (:1005-1008)
        offsets[start] = cnt + 1
        logps[start] = np.nan
        logbs[start] = np.nan
        first = start + 1
        logps[first : first + cnt] = lp
        if not last:
            logbs[first : first + cnt] = lb
        ids[first - U : first - U + cnt] = keys[:, -1]
        # parents: position of keys[:, :-1] among the (sorted, unique) keys of the level above
        parent = _row_positions(prev_keys, keys[:, :-1])
        pcnt = prev_keys.shape[0]
        child_first = first + np.searchsorted(parent, np.arange(pcnt), side="left")
        offsets[prev_start : prev_start + pcnt] = child_first - (prev_start + np.arange(pcnt))
        prev_keys, prev_start, start = keys, first, first + cnt
    max_offset = int(offsets.max()) if O else 0
    off_type, _ = _smallest_int_type(max_offset)
    return (
        torch.from_numpy(logps), torch.from_numpy(logbs), torch.from_numpy(ids).to(id_type),
        torch.from_numpy(offsets).to(off_type), G,
    )  # fmt: skip


def _row_positions(sorted_rows: np.ndarray, rows: np.ndarray) -> np.ndarray:
    """Index in ``sorted_rows`` (lexicographically sorted, unique) of every row of ``rows``."""
    if rows.shape[1] == 1:
        return np.searchsorted(sorted_rows[:, 0], rows[:, 0])
    both = np.concatenate([sorted_rows, rows], 0)
    _, inv = np.unique(both, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    # every row of `rows` equals some row of `sorted_rows`, so the unique ranks of sorted_rows
    # are 0..len-1 in order
    return inv[sorted_rows.shape[0] :]


@custom_op("pydrobert_amd::lookup_lm_log_probs", mutates_args=())
def _lookup_lm_log_probs_op(
    hist: torch.Tensor,
    idx: Optional[torch.Tensor],
    logps: torch.Tensor,
    logbs: torch.Tensor,
    child_start: torch.Tensor,
    ids: torch.Tensor,
    succ_start: torch.Tensor,
    succ_tok: torch.Tensor,
    succ_node: torch.Tensor,
    vocab_size: int,
    max_ngram: int,
    sos: int,
) -> torch.Tensor:
    """(rows, V) log-probabilities: rows = B at positions ``idx``, or (S + 1) * B for every
    position when ``idx`` is None (csrc/lm_lookup.hip)."""
    if hist.dim() != 2:
        raise RuntimeError("hist must be 2 dimensional")
    S, B = hist.shape
    V, N = vocab_size, max_ngram
    device = _cabi.require_hip(hist, idx, logps, logbs, child_start, ids, succ_start, succ_tok, succ_node)
    rows = B if idx is not None else (S + 1) * B
    if idx is not None:
        if idx.numel() == 0:
            raise RuntimeError("idx cannot be empty")
        if idx.dim() > 1 or (idx.dim() == 1 and idx.size(0) not in (1, B)):
            raise RuntimeError("idx must be a scalar or have one entry per batch element")
        idx = idx.detach().to(dtype=torch.long).contiguous()
    h = hist.detach()
    if h.dtype != torch.long:
        h = h.long()
    shift = 0 if (0 <= sos < V) else 1
    have_index = succ_start.numel() == V + shift + 2  # (U + 1 entries; empty: search every entry)
    with torch.cuda.device(device):
        out = torch.empty((rows, V), device=device, dtype=torch.float)
        status = torch.zeros(1, device=device, dtype=torch.int32)
        rc = _cabi.lib().pdt_lookup_lm_log_probs(
            _cabi.ptr(h) if S and B else None, S, B, h.stride(0), h.stride(1),
            _cabi.ptr(idx), 0 if (idx is None or idx.numel() == 1) else 1, rows,
            _cabi.ptr(logps), _cabi.ptr(logbs), _cabi.ptr(child_start), _cabi.ptr(ids),
            _cabi.ptr(succ_start) if have_index else None, _cabi.ptr(succ_tok) if have_index else None,
            _cabi.ptr(succ_node) if have_index else None, V, N, V + shift + 1, sos, _cabi.ptr(out), _cabi.ptr(status), _cabi.stream_ptr(device),
        )  # fmt: skip
    _cabi.check(rc, "pdt_lookup_lm_log_probs")
    return out


@_lookup_lm_log_probs_op.register_fake
def _(hist, idx, logps, logbs, child_start, ids, succ_start, succ_tok, succ_node, vocab_size, max_ngram, sos):
    S, B = hist.shape
    return logps.new_empty((B if idx is not None else (S + 1) * B, vocab_size))


class LookupLanguageModel(MixableSequentialLanguageModel):
    """Back-off n-gram model over a fixed lookup table (reference _lm.py:518-1110).

    Buffers ``logps``, ``logbs``, ``ids``, ``offsets`` hold the reference's reverse trie in the
    reference's dtypes (state dicts are interchangeable); scoring runs in
    ``csrc/lm_lookup.hip`` on two derived, non-persistent int32 buffers (``child_start`` =
    node index + offset, ``ids_wide``) that follow the module across devices.
    """

    __constants__ = ("vocab_size", "sos", "shift", "max_ngram", "max_ngram_nodes", "max_direct_descendants")

    def __init__(self, vocab_size: int, sos: int, prob_dicts=None, destructive: bool = False,
                 logger=None, *, prob_list=None):  # fmt: skip
        sos = argcheck.is_int(sos, "sos")
        destructive = argcheck.is_bool(destructive, "destructive")
        if prob_list is not None:
            if prob_dicts is not None:
                raise ValueError("prob_list and prob_dicts cannot be specified simultaneously")
            warnings.warn("prob_list has been renamed to prob_dicts", DeprecationWarning)
            prob_dicts = prob_list
        super().__init__(vocab_size)
        self.sos = sos
        self.shift = 0 if (0 <= sos < vocab_size) else 1
        if prob_dicts is None:  # uniform unigram model (:714-724)
            logps = -torch.full((self.shift + vocab_size,), vocab_size, dtype=torch.float).log()
            logbs = torch.tensor([], dtype=torch.float)
            ids = torch.tensor([], dtype=torch.uint8)
            offsets = torch.tensor([], dtype=torch.uint8)
            self.max_ngram = 1
            self.max_ngram_nodes = self.shift + vocab_size
        else:
            self.max_ngram = len(prob_dicts)
            logps, logbs, ids, offsets, self.max_ngram_nodes = build_reverse_trie(
                vocab_size, sos, prob_dicts, destructive, logger
            )
        self.max_direct_descendants = self._infer_max_direct_descendants(offsets)
        self.register_buffer("logps", logps)
        self.register_buffer("logbs", logbs)
        self.register_buffer("ids", ids)
        self.register_buffer("offsets", offsets)
        self.register_buffer("child_start", torch.empty(0, dtype=torch.int32), persistent=False)
        self.register_buffer("ids_wide", torch.empty(0, dtype=torch.int32), persistent=False)
        self.register_buffer("succ_start", torch.empty(0, dtype=torch.int32), persistent=False)
        self.register_buffer("succ_tok", torch.empty(0, dtype=torch.int32), persistent=False)
        self.register_buffer("succ_node", torch.empty(0, dtype=torch.int32), persistent=False)
        self._widen()

    @torch.jit.unused
    def _widen(self) -> None:
        O = self.offsets.numel()
        dev = self.offsets.device
        self.child_start = self.offsets.to(torch.int32) + torch.arange(O, device=dev, dtype=torch.int32)
        self.ids_wide = self.ids.to(torch.int32)
        # forward index of the second level: the bigram nodes (children of the unigram nodes
        # 0 .. U - 2) grouped by their label -- the context token -- with the unigram they hang
        # under, i.e. the LAST token of the bigram, ascending inside a group
        U = self.vocab_size + self.shift + 1
        if self.max_ngram >= 2 and O >= U:
            cs = self.child_start[:U].long()
            counts = cs[1:] - cs[:-1]
            nodes = torch.arange(int(cs[0]), int(cs[-1]), device=dev)
            last_tok = torch.repeat_interleave(torch.arange(U - 1, device=dev), counts)
            label = self.ids_wide[nodes - U].long()
            order = torch.sort(label, stable=True)[1]
            start = torch.zeros(U + 1, dtype=torch.long, device=dev)
            start[1:] = torch.bincount(label, minlength=U)[:U].cumsum(0)
            self.succ_start = start.to(torch.int32)
            self.succ_tok = last_tok[order].to(torch.int32)
            self.succ_node = nodes[order].to(torch.int32)

    def extra_repr(self) -> str:
        return super().extra_repr() + ", max_ngram={}, sos={}".format(self.max_ngram, self.sos)

    @torch.jit.export
    def extract_by_src(self, prev: Dict[str, torch.Tensor], src: torch.Tensor) -> Dict[str, torch.Tensor]:
        return dict()

    @torch.jit.export
    def mix_by_mask(self, prev_true: Dict[str, torch.Tensor], prev_false: Dict[str, torch.Tensor],
                    mask: torch.Tensor) -> Dict[str, torch.Tensor]:  # fmt: skip
        return dict()

    @torch.jit.unused
    def _infer_max_direct_descendants(self, offsets: Optional[torch.Tensor] = None) -> int:
        offsets = self.offsets if offsets is None else offsets
        O = offsets.numel()
        if not O:
            return 0
        off = offsets.long().cpu()
        U = self.vocab_size + self.shift + 1
        S, i = 0, 0
        end = U - 1  # first dummy
        while True:
            # children of the real nodes i..end-1: (node+1 + off[node+1]) - (node + off[node])
            if end > i:
                S = max(S, int((off[i + 1 : end + 1] + 1 - off[i:end]).max()))
            i = end + 1
            if i >= O:
                break
            end = end + int(off[end])
        return S

    def _lookup(self, hist: torch.Tensor, idx: Optional[torch.Tensor]) -> torch.Tensor:
        V = self.vocab_size
        if self.max_ngram == 1:  # no history dependence (:446-448)
            rows = hist.size(1) if idx is not None else (hist.size(0) + 1) * hist.size(1)
            return self.logps[:V].expand(rows, V)
        return torch.ops.pydrobert_amd.lookup_lm_log_probs(
            hist, idx, self.logps, self.logbs, self.child_start, self.ids_wide, self.succ_start,
            self.succ_tok, self.succ_node, V, self.max_ngram, self.sos,
        )  # fmt: skip

    @torch.jit.export
    def calc_idx_log_probs(
        self, hist: torch.Tensor, prev: Dict[str, torch.Tensor], idx: torch.Tensor
    ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        return self._lookup(hist, idx), prev

    @torch.jit.export
    def calc_full_log_probs(self, hist: torch.Tensor, prev: Dict[str, torch.Tensor]) -> torch.Tensor:
        return self._lookup(hist, None).view(hist.size(0) + 1, hist.size(1), self.vocab_size)

    @torch.jit.export
    def calc_full_log_probs_chunked(
        self, hist: torch.Tensor, prev: Dict[str, torch.Tensor], chunk_size: int = 1
    ) -> torch.Tensor:
        """All positions at once (reference :806-848).  ``chunk_size`` only bounds the
        reference's temporaries; one launch has none to bound, so it is validated and ignored."""
        if chunk_size < 1:
            raise RuntimeError("expected chunk_size to be positive; got {}".format(chunk_size))
        return self.calc_full_log_probs(hist, prev)

    @torch.jit.unused
    def load_state_dict(self, state_dict: dict, **kwargs):
        """Accepts tables of any size (reference :850-905): the n-gram order and node counts are
        re-derived from the buffers' structure."""
        prefix = "Error(s) in loading state_dict for {}:\n".format(self.__class__.__name__)
        missing = {"offsets", "ids", "logps", "logbs"} - set(state_dict)
        if missing:
            raise RuntimeError('Missing key(s) in state_dict: "{}".'.format('", "'.join(sorted(missing))))
        offsets, ids = state_dict["offsets"], state_dict["ids"]
        logps, logbs = state_dict["logps"], state_dict["logbs"]
        V, shift = self.vocab_size, self.shift
        if ids.numel() and offsets.numel():
            U = V + shift + 1
            if len(offsets) < U:
                raise RuntimeError(
                    prefix + "Expected {} unigram probabilities, got {} (vocab_size and sos must be "
                    "correct!)".format(U - 1, len(offsets) - 1)
                )
            O = len(offsets)
            max_ngram, nodes, ptr = 1, U - 1, U - 1
            while ptr < O:
                step = int(offsets[ptr])
                if step <= 0:
                    raise RuntimeError(
                        prefix + "buffer contains unexpected value (are you sure you've set "
                        "vocab_size and sos correctly?)"
                    )
                ptr += step
                nodes = step - 1
                max_ngram += 1
            if ptr != O + nodes or len(logps) != O + nodes or len(ids) != O + nodes - U or len(logbs) != O:
                raise RuntimeError(prefix + "Unexpected buffer length")
            self.max_ngram, self.max_ngram_nodes = max_ngram, nodes
        else:
            if len(offsets) != len(ids):
                raise RuntimeError(prefix + "Incompatible trie buffers")
            if len(logps) != V + shift:
                raise RuntimeError(
                    prefix + "Expected {} unigram probabilities, got {} (vocab_size and sos must be "
                    "correct!)".format(V + shift, len(logps))
                )
            self.max_ngram, self.max_ngram_nodes = 1, V + shift
        self.max_direct_descendants = self._infer_max_direct_descendants(offsets)
        self.offsets = torch.empty_like(offsets, device=self.offsets.device)
        self.ids = torch.empty_like(ids, device=self.ids.device)
        self.logps = torch.empty_like(logps, device=self.logps.device)
        self.logbs = torch.empty_like(logbs, device=self.logbs.device)
        out = super().load_state_dict(state_dict, **kwargs)
        self._widen()
        return out


class ShallowFusionLanguageModel(SequentialLanguageModel):
    """``log S(v) = log P_first(v) + beta * log P_second(v)`` (reference _lm.py:1113-1262)."""

    __constants__ = ("vocab_size", "beta", "first_prefix", "second_prefix")

    def __init__(self, first: SequentialLanguageModel, second: SequentialLanguageModel,
                 beta: float = 0.0, first_prefix: str = "first.", second_prefix: str = "second."):  # fmt: skip
        beta = argcheck.is_float(beta, "beta")
        if not isinstance(first_prefix, str):
            raise ValueError("first_prefix ({}) is not a str".format(first_prefix))
        if not isinstance(second_prefix, str):
            raise ValueError("second_prefix ({}) is not a str".format(second_prefix))
        if first.vocab_size != second.vocab_size:
            raise ValueError(
                "first's vocab_size ({}) differs from second's vocab_size ({})".format(
                    first.vocab_size, second.vocab_size
                )
            )
        if not len(first_prefix) or not len(second_prefix):
            raise ValueError("prefixes cannot be empty")
        if first_prefix == second_prefix:
            raise ValueError("first_prefix matches second_prefix ('{}')".format(first_prefix))
        super().__init__(first.vocab_size)
        self.first, self.second, self.beta = first, second, beta
        self.first_prefix, self.second_prefix = first_prefix, second_prefix

    def extra_repr(self) -> str:
        return super().extra_repr() + (
            ", beta={}, first_prefix='{}', second_prefix='{}', first={}, second={}".format(
                self.beta, self.first_prefix, self.second_prefix, self.first, self.second
            )
        )

    def split_dicts(
        self, prev: Dict[str, torch.Tensor]
    ) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
        """State dictionary -> the two models' state dictionaries, by key prefix."""
        prev_first: Dict[str, torch.Tensor] = dict()
        prev_second: Dict[str, torch.Tensor] = dict()
        for k, v in prev.items():
            if k.startswith(self.first_prefix):
                prev_first[k[len(self.first_prefix) :]] = v
            elif k.startswith(self.second_prefix):
                prev_second[k[len(self.second_prefix) :]] = v
            else:
                raise RuntimeError(
                    "key '{}' from prev does not start with first_prefix '{}' nor second_prefix "
                    "'{}'".format(k, self.first_prefix, self.second_prefix)
                )
        return prev_first, prev_second

    def merge_dicts(
        self, prev_first: Dict[str, torch.Tensor], prev_second: Dict[str, torch.Tensor]
    ) -> Dict[str, torch.Tensor]:
        prev: Dict[str, torch.Tensor] = dict()
        for k, v in prev_first.items():
            prev[self.first_prefix + k] = v
        for k, v in prev_second.items():
            prev[self.second_prefix + k] = v
        return prev

    def update_input(self, prev: Dict[str, torch.Tensor], hist: torch.Tensor) -> Dict[str, torch.Tensor]:
        a, b = self.split_dicts(prev)
        return self.merge_dicts(self.first.update_input(a, hist), self.second.update_input(b, hist))

    def calc_idx_log_probs(
        self, hist: torch.Tensor, prev: Dict[str, torch.Tensor], idx: torch.Tensor
    ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        a, b = self.split_dicts(prev)
        lp_a, a = self.first.calc_idx_log_probs(hist, a, idx)
        lp_b, b = self.second.calc_idx_log_probs(hist, b, idx)
        return lp_a + self.beta * lp_b, self.merge_dicts(a, b)

    def calc_full_log_probs(self, hist: torch.Tensor, prev: Dict[str, torch.Tensor]) -> torch.Tensor:
        a, b = self.split_dicts(prev)
        return self.first.calc_full_log_probs(hist, a) + self.beta * self.second.calc_full_log_probs(hist, b)


class ExtractableShallowFusionLanguageModel(ShallowFusionLanguageModel, ExtractableSequentialLanguageModel):
    """Shallow fusion of two extractable models (reference _lm.py:1265-1303)."""

    def extract_by_src(self, prev: Dict[str, torch.Tensor], src: torch.Tensor) -> Dict[str, torch.Tensor]:
        a, b = self.split_dicts(prev)
        return self.merge_dicts(self.first.extract_by_src(a, src), self.second.extract_by_src(b, src))


class MixableShallowFusionLanguageModel(ExtractableShallowFusionLanguageModel, MixableSequentialLanguageModel):
    """Shallow fusion of two mixable models (reference _lm.py:1306-1345)."""

    def mix_by_mask(self, prev_true: Dict[str, torch.Tensor], prev_false: Dict[str, torch.Tensor],
                    mask: torch.Tensor) -> Dict[str, torch.Tensor]:  # fmt: skip
        at, bt = self.split_dicts(prev_true)
        af, bf = self.split_dicts(prev_false)
        return self.merge_dicts(self.first.mix_by_mask(at, af, mask), self.second.mix_by_mask(bt, bf, mask))
