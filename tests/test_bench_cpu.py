"""bench.py's launch path and CPU-baseline leg, without a GPU: `--gpus N` starts N ranks by
itself (gloo here; RCCL on the GPU node), refuses when ranks are missing or fail, and the
baseline record carries per-operator figures."""
import argparse
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + extra, env=e, capture_output=True, text=True, timeout=300)


def test_launcher_brings_up_two_ranks_over_gloo():
    r = _run(["--gpus", "2", "--rendezvous-only"], {"PDT_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1  # only rank 0 prints
    rec = json.loads(line[0])
    assert rec == {"rendezvous": "ok", "n_gpus": 2, "backend": "gloo"}


def test_launcher_refuses_more_ranks_than_devices():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 devices")
    r = _run(["--gpus", "2", "--rendezvous-only"])  # RCCL: one device per rank
    assert r.returncode != 0 and "device(s) are visible" in r.stderr


def test_launcher_propagates_a_failing_rank():
    import torch

    if torch.cuda.is_available():
        pytest.skip("the ranks would succeed on a GPU box")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"PDT_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0  # the workload needs a device: every rank fails, no JSON line
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "1", "--rendezvous-only"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_cpu_baseline_record():
    sys.path.insert(0, ROOT)
    import bench

    args = argparse.Namespace(T=24, V=12, beam=4, N=64, cpu_seconds=0.5)
    op_ms = {"error_rate": 1.0, "prefix_error_rates": 1.0, "optimal_completion": 1.0, "ctc_prefix_search": 1.0}
    rec = bench.cpu_baseline(args, True, op_ms)
    assert rec["kind"] == "port" and rec["unit"] == "utterances/s" and rec["value"] > 0
    assert set(rec["per_op"]) == set(op_ms) == set(rec["gpu_over_cpu"]) == set(rec["c_port"]["utt_per_s"])
    for op in op_ms:
        assert rec["per_op"][op]["reps"] >= 5
        assert rec["threads_8"]["per_op"][op]["utt_per_s_median"] > 0
