"""CPU, world_size 2 over gloo: the batch-sharded path (shard -> local compute -> gather).
The local compute is the oracle here (no GPU in this container); on the GPU box the same host
code runs with the HIP kernels over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from pydrobert_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_error_rate(ref, hyp, **kw):
    return torch.from_numpy(oracle.error_rate(ref.numpy(), hyp.numpy(), **kw))


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)
        ref = torch.from_numpy(rng.integers(0, 6, (20, n_total)))
        hyp = torch.from_numpy(rng.integers(0, 6, (17, n_total)))
        lo, hi = D.shard_bounds(n_total, world, rank)
        full = D.sharded_error_rate(ref[:, lo:hi], hyp[:, lo:hi], n_total, compute=_oracle_error_rate, eos=5)
        ref_lens = torch.from_numpy(oracle.string_matching(ref.numpy(), hyp.numpy(), eos=5)[1])
        corpus = D.corpus_error_rate(ref[:, lo:hi], hyp[:, lo:hi], ref_lens[lo:hi], compute=_oracle_error_rate, eos=5)
        S = 4 + rank  # data-dependent row count per shard
        y = torch.full((S, hi - lo, 3), rank + 1, dtype=torch.long)
        yg, lg, pg = D.gather_beams(y, torch.full((hi - lo, 3), S), torch.full((hi - lo, 3), float(rank)), n_total)
        q.put((rank, full.numpy(), float(corpus), tuple(yg.shape), lg[:, 0].numpy(), pg[:, 0].numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [10, 11])
def test_sharded_error_rate_two_ranks(n_total):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.default_rng(123)
    ref = rng.integers(0, 6, (20, n_total))
    hyp = rng.integers(0, 6, (17, n_total))
    exp = oracle.error_rate(ref, hyp, eos=5)
    errs = oracle.error_rate(ref, hyp, eos=5, norm=False)
    ref_lens = oracle.string_matching(ref, hyp, eos=5)[1]
    for rank, full, corpus, yshape, lens_col, probs_col in res:
        assert np.array_equal(full, exp), rank
        assert abs(corpus - errs.sum() / ref_lens.sum()) < 1e-6
        assert yshape == (5, n_total, 3)
        half = n_total // 2
        assert (lens_col[:half] == 4).all() and (lens_col[half:] == 5).all()
        assert (probs_col[:half] == 0).all() and (probs_col[half:] == 1).all()


def test_shard_bounds():
    assert D.shard_bounds(10, 2, 0) == (0, 5) and D.shard_bounds(11, 2, 1) == (5, 11)
    assert [D.shard_bounds(32768, 8, r) for r in (0, 7)] == [(0, 4096), (28672, 32768)]
    covered = sorted(D.shard_bounds(13, 4, r) for r in range(4))
    assert covered[0][0] == 0 and covered[-1][1] == 13
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    with pytest.raises(ValueError):
        D.shard_bounds(4, 2, 2)


def test_single_process_passthrough():
    x = torch.arange(5.0)
    assert D.gather_utterance_values(x, 5) is x
