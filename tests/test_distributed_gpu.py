"""Two ranks over RCCL on a box with at least two GPUs (skipped otherwise: the round's GPU box has
one; the driver's 8-GPU node runs it): the batch-sharded path with the HIP kernels -- uneven
shards, the all-gather of per-utterance rates and of decoded beams -- against the same calls on
one rank."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(n_total):
    rng = np.random.default_rng(77)
    ref = rng.integers(0, 9, (24, n_total))
    hyp = rng.integers(0, 9, (21, n_total))
    lg = rng.normal(size=(30, n_total, 13)).astype(np.float32)
    np.put_along_axis(lg, rng.integers(0, 13, (30, n_total, 1)), 7.0, 2)
    lens = rng.integers(5, 31, n_total)
    return ref, hyp, lg, lens


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist

    from pydrobert_amd import distributed as D
    from pydrobert_amd import functional as F

    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        ref, hyp, lg, lens = _inputs(n_total)
        lo, hi = D.shard_bounds(n_total, world, rank)
        tr, th = torch.from_numpy(ref[:, lo:hi]).to(dev), torch.from_numpy(hyp[:, lo:hi]).to(dev)
        full = D.sharded_error_rate(tr, th, n_total, eos=8, warn=False)
        y, yl, yp = F.ctc_prefix_search(torch.from_numpy(lg[:, lo:hi]).to(dev), 4, torch.from_numpy(lens[lo:hi]).to(dev))
        yg, ylg, ypg = D.gather_beams(y, yl, yp, n_total)
        q.put((rank, full.cpu().numpy(), yg.cpu().numpy(), ylg.cpu().numpy(), ypg.cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("n_total", [12, 13])
def test_two_ranks_over_rccl_match_one_rank(n_total):
    import torch.multiprocessing as mp

    from pydrobert_amd import functional as F

    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ref, hyp, lg, lens = _inputs(n_total)
    dev = torch.device("cuda", 0)
    exp = F.error_rate(torch.from_numpy(ref).to(dev), torch.from_numpy(hyp).to(dev), eos=8, warn=False).cpu().numpy()
    ey, eyl, eyp = (x.cpu().numpy() for x in F.ctc_prefix_search(
        torch.from_numpy(lg).to(dev), 4, torch.from_numpy(lens).to(dev)))
    for rank, full, yg, ylg, ypg in res:
        assert np.array_equal(full, exp), rank
        assert np.array_equal(ylg, eyl) and np.array_equal(ypg, eyp), rank
        S = min(yg.shape[0], ey.shape[0])  # (a shard's S is its own longest utterance; rows beyond are zero)
        assert np.array_equal(yg[:S], ey[:S]) and not yg[S:].any() and not ey[S:].any(), rank
