"""Batch-axis sharding of the hot path over the GPUs of one node.

Every utterance is independent in every operator of the path, so the multi-GPU form is pure
data parallelism: one process per GPU (``torch.distributed``, backend "nccl" = RCCL over
xGMI), rank ``r`` owns the contiguous utterance block ``shard_bounds(N, world, r)`` and there
is NO collective on the data path.  The only exchange is the final gather of the small
per-utterance results (4 bytes per utterance for error counts), which mirrors how the
reference's own data parallelism averages metrics (training.py:887-910) and shards sample
indices (_dataloaders.py:124-127).
"""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

__all__ = [
    "corpus_error_rate",
    "gather_beams",
    "gather_utterance_values",
    "shard_bounds",
    "sharded_error_rate",
]


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of rank ``rank``; the remainder goes to the last rank (the reference's
    ``on_uneven_distributed="uneven"`` policy)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("invalid rank {} of {}".format(rank, world_size))
    per = n // world_size
    lo = rank * per
    hi = n if rank == world_size - 1 else lo + per
    return lo, hi


def _world(group):
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def gather_utterance_values(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather per-utterance values whose leading dim is this rank's shard.

    Shards follow :func:`shard_bounds`; unequal last shards are padded to the largest shard so
    that ONE ``all_gather_into_tensor`` moves everything (16 KB per rank at N=32768, W=8: latency
    bound, no ring tuning needed)."""
    world, rank = _world(group)
    if world == 1:
        return local
    sizes = [shard_bounds(n_total, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    if local.shape[0] != sizes[rank][1] - sizes[rank][0]:
        raise RuntimeError(
            "rank {} holds {} utterances, expected {}".format(rank, local.shape[0], sizes[rank][1] - sizes[rank][0])
        )
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], 0)
    out = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    if all(hi - lo == mx for lo, hi in sizes):
        return out
    return torch.cat([out[r * mx : r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


def sharded_error_rate(
    ref_shard: torch.Tensor,
    hyp_shard: torch.Tensor,
    n_total: int,
    compute: Optional[Callable[..., torch.Tensor]] = None,
    group=None,
    **kwargs,
) -> torch.Tensor:
    """Error rates of ALL ``n_total`` utterances on every rank: each rank scores its own
    ``(T, N_shard)`` block with ``compute`` (default: this package's ``error_rate`` kernel),
    then one all-gather of ``(N_shard,)`` float32."""
    if compute is None:
        from .functional import error_rate as compute
    local = compute(ref_shard, hyp_shard, **kwargs)
    return gather_utterance_values(local, n_total, group)


def corpus_error_rate(
    ref_shard: torch.Tensor,
    hyp_shard: torch.Tensor,
    ref_lens_shard: torch.Tensor,
    compute: Optional[Callable[..., torch.Tensor]] = None,
    group=None,
    **kwargs,
) -> torch.Tensor:
    """Corpus-level rate = total errors / total reference tokens, the quantity the reference's
    ``compute-torch-token-data-dir-error-rates`` accumulates (command_line.py:1135-1147): a
    2-element all-reduce(SUM)."""
    if compute is None:
        from .functional import error_rate as compute
    kwargs = dict(kwargs)
    kwargs["norm"] = False
    errs = compute(ref_shard, hyp_shard, **kwargs)
    acc = torch.stack([errs.sum().double(), ref_lens_shard.sum().double()])
    world, _ = _world(group)
    if world > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return (acc[0] / acc[1]).float()


def gather_beams(y: torch.Tensor, y_lens: torch.Tensor, y_probs: torch.Tensor, n_total: int, group=None):
    """Gather decoded beams ``y (S, N_shard, K)``: S is data dependent per shard, so one
    all-reduce(MAX) of a single int pads it first, then the payload goes in one all-gather per
    tensor (fully connected xGMI: every GPU sends its shard over its 7 links concurrently)."""
    world, _ = _world(group)
    if world == 1:
        return y, y_lens, y_probs
    s = torch.tensor([y.shape[0]], device=y.device, dtype=torch.long)
    dist.all_reduce(s, op=dist.ReduceOp.MAX, group=group)
    S = int(s.item())
    if y.shape[0] < S:
        y = torch.cat([y, y.new_zeros((S - y.shape[0],) + tuple(y.shape[1:]))], 0)
    yg = gather_utterance_values(y.transpose(0, 1).contiguous(), n_total, group).transpose(0, 1)
    return yg, gather_utterance_values(y_lens, n_total, group), gather_utterance_values(y_probs, n_total, group)
