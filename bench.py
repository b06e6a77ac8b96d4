#!/usr/bin/env python
"""Benchmark of the sequence-level hot path on MI355X (contract: see the task prompt).

A step = one pass of the hot path over one synthetic batch that is already resident in
HBM: BASELINE.json's configs[1] shapes (N=4096 utterances per GPU, T=512, V=256)

    error_rate -> prefix_error_rates -> optimal_completion -> [CTC prefix beam search]
    -> all-gather of the per-shard error counts (multi-GPU only)

Prints ONE JSON line on rank 0.  `value` = utterances / s over all ranks (weak scaling:
every rank owns its own N utterances).  Per-op times come from HIP events recorded on the
launch stream inside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "pydrobert-pytorch_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--N", type=int, default=4096, help="utterances per GPU")
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--V", type=int, default=256)
    ap.add_argument("--beam", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_inputs(args, rank, device):
    """Synthetic tokens (and logits) of BASELINE config 2's shape; seed 0x5EED0002 + rank."""
    rng = np.random.default_rng(0x5EED0002 + rank)
    ref = torch.from_numpy(rng.integers(0, args.V, (args.T, args.N))).to(device)
    hyp = torch.from_numpy(rng.integers(0, args.V, (args.T, args.N))).to(device)
    return ref, hyp


def cpu_baseline(args, with_decode):
    """The oracle's reference-faithful C restatement (O(H*R^2) per utterance like
    _string.py:316-317; dense candidate tables per frame like _decoding.py:842-846) timed on
    ONE host core on a bounded sample of the same workload."""
    import oracle

    rng = np.random.default_rng(0x5EED0002)
    T, V = args.T, args.V
    n, done, t_used = 2, 0, 0.0
    while t_used < args.cpu_seconds and done < args.N:
        ref = rng.integers(0, V, (T, n))
        hyp = rng.integers(0, V, (T, n))
        lg = rng.normal(size=(T, n, V + 1)).astype(np.float32)
        np.put_along_axis(lg, rng.integers(0, V + 1, (T, n, 1)), 12.0, 2)
        t0 = time.perf_counter()
        oracle.error_rate(ref, hyp)
        oracle.prefix_error_rates(ref, hyp)
        oracle.optimal_completion(ref, hyp)
        if with_decode:
            oracle.ctc_prefix_search(lg, args.beam)
        t_used += time.perf_counter() - t0
        done += n
        n = min(n * 2, 16)
    return {
        "value": done / t_used,
        "unit": "utterances/s",
        "cores": 1,
        "kind": "port",
        "sample": "{} utterances of T={} V={}: the step's operators ({}) with the oracle's "
        "reference-faithful C restatement".format(
            done, T, V, "string ops + ctc_prefix_search" if with_decode else "string ops"
        ),
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        # RCCL is the data path.  PDT_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer
        # GPUs than ranks (RCCL refuses two ranks on one device); the gather is then host-staged.
        backend = os.environ.get("PDT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from pydrobert_amd import functional as F

    ref, hyp = make_inputs(args, rank, device)
    N, T = args.N, args.T
    gathered = torch.empty((world * N,), device=device, dtype=torch.float) if world > 1 else None

    ops = ["error_rate", "prefix_error_rates", "optimal_completion"]
    have_decode = (not args.no_decode) and hasattr(F, "ctc_prefix_search")
    logits = None
    if have_decode:
        ops.append("ctc_prefix_search")
        g = torch.Generator(device=device).manual_seed(0x5EED0003 + rank)
        logits = torch.randn((T, N, args.V + 1), device=device, generator=g)
        peak = torch.randint(0, args.V + 1, (T, N, 1), device=device, generator=g)
        logits.scatter_add_(2, peak, torch.full((T, N, 1), 12.0, device=device))
        del peak
    if world > 1:
        ops.append("all_gather")
    C_seen = [0]

    def step(events=None):
        k = 0

        def mark():
            nonlocal k
            if events is not None:
                events[k].record()
                k += 1

        mark()
        er = F.error_rate(ref, hyp, warn=False)
        mark()
        F.prefix_error_rates(ref, hyp, warn=False)
        mark()
        oc = F.optimal_completion(ref, hyp, warn=False)
        C_seen[0] = oc.shape[-1]
        mark()
        if have_decode:
            F.ctc_prefix_search(logits, args.beam)
            mark()
        if world > 1:
            if dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(gathered, er)
            else:
                parts = [torch.empty(N) for _ in range(world)]
                dist.all_gather(parts, er.cpu())
                gathered.copy_(torch.cat(parts))
            mark()
        return er

    for _ in range(args.warmup):
        step()
    evs = [
        [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
        for _ in range(args.steps)
    ]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(evs[s])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor(
            [dt], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64
        )
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    op_ms = {}
    for i, name in enumerate(ops):
        op_ms[name] = float(np.mean([evs[s][i].elapsed_time(evs[s][i + 1]) for s in range(args.steps)]))

    # algorithmic HBM bytes per utterance
    C = C_seen[0]
    alg_bytes = {
        "error_rate": 8 * 2 * T + 4,
        "prefix_error_rates": 8 * 2 * T + 4 * (T + 1),
        "optimal_completion": 8 * 2 * T + 8 * (T + 1) * C,
        "ctc_prefix_search": 4 * T * (args.V + 1) + 8 * T * args.beam + 12 * args.beam,
    }
    dom = max((o for o in ops if o in alg_bytes), key=lambda o: op_ms[o])
    achieved = alg_bytes[dom] * N / (op_ms[dom] * 1e-3) / 1e9
    kernel_names = {
        "error_rate": "pdt::lev_skewed_kernel<false>",
        "prefix_error_rates": "pdt::lev_skewed_kernel<false>",
        "optimal_completion": "pdt::lev_rowsync_kernel<false,false> + pdt::oc_expand_kernel",
        "ctc_prefix_search": "pdt::ctc_search_kernel<1, 4>",
    }
    # HBM bytes per launch from rocprofv3 PMC passes (profiles/), only for the profiled config
    traffic = None
    valu = None
    tpath = os.path.join(ROOT, "profiles", "r01_ctc_traffic.json")
    if dom == "ctc_prefix_search" and os.path.exists(tpath):
        rec = json.load(open(tpath))
        if rec["config"] == {"N": N, "T": T, "V": args.V, "beam": args.beam}:
            traffic = rec["hbm_bytes_per_launch"]
            if "sq" in rec:
                # what actually bounds this kernel: VALU issue.  A wave64 VALU instruction holds
                # its SIMD for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz (MI355X_MICROARCH.md).
                insts = rec["sq"]["SQ_INSTS_VALU_per_launch"]
                valu = {
                    "wave_insts_per_launch": insts,
                    "pipe_busy_frac": insts * 4 / (1024 * 2.4e9 * op_ms[dom] * 1e-3),
                    "source": "profiles/r01_ctc_sq_counters.csv (rocprofv3 --pmc SQ_INSTS_VALU)",
                }

    if rank == 0:
        out = {
            "metric": "utterances/sec (N=4096,T=512) error_rate+ctc_beam at 1/2/4/8 GPU",
            "value": world * N * args.steps / dt,
            "unit": "utterances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1] shapes per GPU: " + " + ".join(ops),
                "N_per_gpu": N, "T_ref": T, "T_hyp": T, "V": args.V, "beam": args.beam,
                "optimal_completion_C": C, "sharding": "batch axis, {} rank(s)".format(world),
            },
            "op_ms": op_ms,
            "roofline": {
                "kernel": kernel_names[dom],
                "op": dom,
                "algorithmic_bytes_per_launch": alg_bytes[dom] * N,
                "avg_launch_ms": op_ms[dom],
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "valu": valu,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, have_decode)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
