#!/usr/bin/env python
"""Benchmark of the sequence-level hot path on MI355X (contract: see the task prompt).

A step = one pass of the hot path over one synthetic batch that is already resident in
HBM: BASELINE.json's configs[1] shapes (N=4096 utterances per GPU, T=512, V=256)

    error_rate -> prefix_error_rates -> optimal_completion -> CTC prefix beam search (K=16)
    -> all-gather of the per-shard error rates (multi-GPU only)

`python bench.py --gpus N` starts its own N ranks (one process per GPU, RCCL) when it is not
already running under torchrun; rank 0 prints ONE JSON line.  `value` = utterances / s over all
ranks (weak scaling: every rank owns its own N utterances).  Per-op times come from HIP events
recorded on the launch stream inside the timed region.  After the timed region the line also
carries the other BASELINE configs (C3 search and step functions, C4 SpecAugment / sparse warp,
the C5 shard: error_rate + decode at V=5000 with its gather) as `other_configs`, and at N=1 the
CPU baseline (the reference's tensor algorithm on the host cores, per operator).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "pydrobert-pytorch_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
METRIC = "utterances/sec (N=4096,T=512) error_rate+ctc_beam at 1/2/4/8 GPU"


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
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configs")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="preflight: bring the ranks up, check one all-gather, run no workload (needs no GPU "
                         "with PDT_BENCH_BACKEND=gloo)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------
# launcher: one child process per rank, started before anything here touches a GPU
# ------------------------------------------------------------------------------------------
def launch_ranks(args):
    """Start args.gpus copies of this script as ranks 0..N-1 (the spawn-then-init pattern of the
    reference's own distributed tests, tests/test_dataloaders.py:818-904).  This process never
    initialises the GPU (device_count() does not); it only waits and forwards the exit code."""
    backend = os.environ.get("PDT_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        print("bench.py: --gpus {} but only {} device(s) are visible".format(args.gpus, have), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, deadline = 0, time.time() + 1500
    pending = set(range(args.gpus))
    while pending and time.time() < deadline:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print("bench.py: rank {} exited with {}; stopping the others".format(r, code), file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    for r in pending:  # timed out
        procs[r].kill()
        rc = rc or 3
    return rc


# ------------------------------------------------------------------------------------------
# inputs
# ------------------------------------------------------------------------------------------
def make_tokens(args, rank, device):
    """Synthetic tokens of BASELINE config 2's shape; seed 0x5EED0002 + rank (any rank can
    regenerate any other rank's shard: the gather check relies on it)."""
    rng = np.random.default_rng(0x5EED0002 + rank)
    ref = torch.from_numpy(rng.integers(0, args.V, (args.T, args.N))).to(device)
    hyp = torch.from_numpy(rng.integers(0, args.V, (args.T, args.N))).to(device)
    return ref, hyp


def peaky_logits(T, N, V, device, seed, chunk=64):
    """N(0,1) + 12 on one class per frame (blank included)."""
    g = torch.Generator(device=device).manual_seed(seed)
    lg = torch.empty((T, N, V + 1), device=device)
    for t0 in range(0, T, chunk):
        part = lg[t0:t0 + chunk]
        part.normal_(generator=g)
        peak = torch.randint(0, V + 1, (part.shape[0], N, 1), device=device, generator=g)
        part.scatter_add_(2, peak, torch.full((part.shape[0], N, 1), 12.0, device=device))
    return lg


def event_ms(fn, reps=5, warm=2):
    """Median HIP-event time of fn() on the current (= launch) stream."""
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


# ------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only)
# ------------------------------------------------------------------------------------------
def cpu_baseline(args, with_decode, op_ms):
    """the reference's tensor algorithm restated in torch CPU ops
    (oracle/torch_cpu.py -- what a user of the reference experiences), per operator, on a
    bounded sample, all host threads and 8 threads; the scalar C restatement on one core is
    reported separately ("c_port") and is not the denominator of any ratio."""
    import oracle
    from oracle import torch_cpu as tc

    T, V, K = args.T, args.V, args.beam
    rng = np.random.default_rng(0x5EED0002)
    # the box's CPU share, not the machine's core count: a GPU box gives 16 cores per GPU whatever
    # the affinity mask says, and torch threads beyond the share only fight each other
    host = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    host = max(1, min(host, int(os.environ.get("PDT_BENCH_CPU_THREADS", "16"))))
    # utterances per call, sized so one call takes ~0.1-1 s (optimal_completion's (H, n, R, R)
    # closure is 134 MB per utterance at T = 512)
    chunk = {"error_rate": 16, "prefix_error_rates": 16, "optimal_completion": 2, "ctc_prefix_search": 16}
    fns = {"error_rate": lambda r, h, lg: tc.error_rate(r, h),
           "prefix_error_rates": lambda r, h, lg: tc.prefix_error_rates(r, h),
           "optimal_completion": lambda r, h, lg: tc.optimal_completion(r, h),
           "ctc_prefix_search": lambda r, h, lg: tc.ctc_prefix_search(lg, K)}
    names = [o for o in fns if with_decode or o != "ctc_prefix_search"]
    budget = args.cpu_seconds / (2.0 * len(names))
    old_threads = torch.get_num_threads()
    res = {}
    for threads in sorted({host, min(8, host)}, reverse=True):
        torch.set_num_threads(threads)
        per = {}
        for name in names:
            print("bench.py: cpu baseline, {} threads, {}".format(threads, name), file=sys.stderr, flush=True)
            n = chunk[name]
            ref = torch.from_numpy(rng.integers(0, V, (T, n)))
            hyp = torch.from_numpy(rng.integers(0, V, (T, n)))
            lg = None
            if name == "ctc_prefix_search":
                lg = torch.from_numpy(rng.normal(size=(T, n, V + 1)).astype(np.float32))
                lg.scatter_add_(2, torch.from_numpy(rng.integers(0, V + 1, (T, n, 1))), torch.full((T, n, 1), 12.0))
            times, t_begin = [], time.perf_counter()
            for rep in range(2 + 5):  # 2 warm-ups, then >= 5 timed repetitions
                t0 = time.perf_counter()
                fns[name](ref, hyp, lg)
                if rep >= 2:
                    times.append(time.perf_counter() - t0)
            while time.perf_counter() - t_begin < budget and len(times) < 25:
                t0 = time.perf_counter()
                fns[name](ref, hyp, lg)
                times.append(time.perf_counter() - t0)
            per[name] = {"utt_per_s_median": n / float(np.median(times)), "utt_per_s_best": n / min(times),
                         "utterances_per_call": n, "reps": len(times)}
        step = 1.0 / sum(1.0 / p["utt_per_s_median"] for p in per.values())
        res[threads] = {"step_utt_per_s": step, "per_op": per}
    torch.set_num_threads(old_threads)
    # the scalar C restatement (oracle/pdt_oracle_*.c), one core, for comparison
    c_port = {}
    n = 8
    ref, hyp = rng.integers(0, V, (T, n)), rng.integers(0, V, (T, n))
    lg = rng.normal(size=(T, n, V + 1)).astype(np.float32)
    np.put_along_axis(lg, rng.integers(0, V + 1, (T, n, 1)), 12.0, 2)
    for name, fn in (("error_rate", lambda: oracle.error_rate(ref, hyp)),
                     ("prefix_error_rates", lambda: oracle.prefix_error_rates(ref, hyp)),
                     ("optimal_completion", lambda: oracle.optimal_completion(ref, hyp)),
                     ("ctc_prefix_search", lambda: oracle.ctc_prefix_search(lg, K))):
        if name not in names:
            continue
        t0 = time.perf_counter()
        fn()
        c_port[name] = n / (time.perf_counter() - t0)
    top = res[host]
    gpu = {o: args.N / (op_ms[o] * 1e-3) for o in names}
    return {
        "value": top["step_utt_per_s"],
        "unit": "utterances/s",
        "cores": host,
        "kind": "port",
        "sample": "torch-CPU restatement of the reference's tensor algorithm (oracle/torch_cpu.py), T={} V={} K={}: "
                  "per operator 2 warm-ups + >=5 timed calls of 16 utterances (optimal_completion: 2); "
                  "value = utterances/s of the whole step from the per-op medians".format(T, V, K),
        "per_op": top["per_op"],
        "threads_8": res[min(8, host)],
        "gpu_over_cpu": {o: gpu[o] / top["per_op"][o]["utt_per_s_median"] for o in names},
        "c_port": {"cores": 1, "utt_per_s": c_port,
                   "what": "scalar C restatement (oracle/pdt_oracle_*.c), 8 utterances, one call per operator"},
    }


# ------------------------------------------------------------------------------------------
# the other BASELINE configs, after the timed region
# ------------------------------------------------------------------------------------------
def roof(alg_bytes, ms, kernel, note=None):
    """Roofline entry of one configuration: algorithmic HBM bytes of one call (SURVEY section 8(d)'s
    per-unit figure x the units of the call), the measured time of that call, and the fraction of the
    8 TB/s HBM peak the two give."""
    r = {"algorithmic_bytes": int(alg_bytes), "ms": ms, "achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "bound": "hbm", "kernel": kernel}
    r["frac"] = r["achieved"] / HBM_PEAK_GBS
    if note:
        r["note"] = note
    return r


def summary_of(entry):
    """The few numbers of one `other_configs` entry a reader needs: its times and roofline fraction(s)."""
    s = {k: round(v, 4) for k, v in entry.items() if k.endswith("ms") and isinstance(v, float)}
    for k, v in entry.items():
        if k.endswith("roofline") and isinstance(v, dict) and "frac" in v:
            s["frac" if k == "roofline" else k[:-len("roofline")] + "frac"] = round(v["frac"], 4)
            if v.get("bound") == "mfma":
                s["bound"] = "mfma"
    return s


class _Reported(dict):
    """`other_configs`: a configuration is reported on stderr the moment it is recorded."""

    def __init__(self, say):
        super().__init__()
        self._say = say

    def __setitem__(self, name, entry):
        super().__setitem__(name, entry)
        self._say(name + " " + " ".join("{}={}".format(k, v) for k, v in summary_of(entry).items()))


def search_bytes(T, N, V, K):
    """4 T (V + 1) of logits + 8 T K + 12 K of outputs per utterance."""
    return (4 * T * (V + 1) + 8 * T * K + 12 * K) * N


def other_configs(F, M, device, world, rank, dist, gather_check, say=lambda msg: None):
    out = _Reported(say)
    K = 16
    # C5 shard: N=4096 x T=512, V=5000 per GPU -- error_rate + fused CTC decode + the gathers
    T, N, V = 512, 4096, 5000
    rng = np.random.default_rng(0x5EED0005 + rank)
    ref = torch.from_numpy(rng.integers(0, V, (T, N))).to(device)
    hyp = torch.from_numpy(rng.integers(0, V, (T, N))).to(device)
    lg = peaky_logits(T, N, V, device, 0x5EED0006 + rank)
    from pydrobert_amd import distributed as D

    def c5():
        er = F.error_rate(ref, hyp, warn=False)
        y, yl, yp = F.ctc_prefix_search(lg, K)
        if world > 1 and dist.get_backend() == "nccl":
            er = D.gather_utterance_values(er, world * N)
            yl = D.gather_utterance_values(yl, world * N)
            yp = D.gather_utterance_values(yp, world * N)
        return er, y, yl, yp

    er, y, yl, yp = c5()
    ok = bool(torch.isfinite(yp[:, 0]).all()) and bool((yl <= T).all()) and er.shape[0] == world * N
    if world > 1 and dist.get_backend() == "nccl":
        ok = ok and gather_check(er[rank * N:(rank + 1) * N], F.error_rate(ref, hyp, warn=False))
    ms_dec = event_ms(lambda: F.ctc_prefix_search(lg, K), reps=3, warm=0)
    ms_er = event_ms(lambda: F.error_rate(ref, hyp, warn=False), reps=3, warm=1)
    ms_all = event_ms(c5, reps=3, warm=0)
    out["C5_shard"] = {
        "workload": "per GPU: error_rate + ctc_prefix_search, N=4096 T=512 V=5000 K=16, "
                    "all-gather of rates / lens / probs over {} rank(s)".format(world),
        "ms": ms_all, "error_rate_ms": ms_er, "decode_ms": ms_dec, "utt_per_s_job": world * N / ms_all * 1e3,
        "decode_GBs": lg.numel() * 4 / ms_dec / 1e6, "checked": ok,
        "roofline": roof(search_bytes(T, N, V, K), ms_dec, ctc_kernel_name(V, K),
                         "the decode of the shard; error_rate (8 196 B per utterance) is issue-bound, not HBM-bound"),
        "error_rate_roofline": roof((8 * 2 * T + 4) * N, ms_er, "pdt::lev_classify_kernel<8> + pdt::lev_bitpar_kernel"),
    }
    del lg, ref, hyp
    if rank != 0:
        return out
    # the headline search on OTHER draws of the same distribution: a launch ends with its slowest utterance,
    # and an utterance in thousands that leaves the lean tier in every frame used to set the time
    # (EXPERIMENTS.md section 10.12: 1.64-2.04 ms over these draws before, 1.59-1.73 after)
    T, N, V = 512, 4096, 256
    draws = {}
    for seed, chunk in ((3, 64), (3, T), (4, 64), (4, T)):
        lg = peaky_logits(T, N, V, device, seed, chunk=chunk)
        draws["seed {} in {}-frame chunks".format(seed, chunk)] = event_ms(lambda: F.ctc_prefix_search(lg, K), reps=5, warm=2)
        del lg
    out["C2_ctc_prefix_search_other_draws"] = {
        "workload": "ctc_prefix_search N=4096 T=512 V=256 K=16 on four other draws of the bench's input distribution",
        "ms": max(draws.values()), "ms_min": min(draws.values()), "ms_by_draw": draws,
        "roofline": roof(search_bytes(T, N, V, K), max(draws.values()), ctc_kernel_name(V, K), "the slowest of the four draws"),
    }
    # C3: fused search N=1024, T=1000, V=1000
    T, N, V = 1000, 1024, 1000
    lg = peaky_logits(T, N, V, device, 0x5EED0003)
    ms = event_ms(lambda: F.ctc_prefix_search(lg, K), reps=3, warm=1)
    out["C3_search"] = {"workload": "ctc_prefix_search N=1024 T=1000 V=1000 K=16", "ms": ms,
                        "utt_per_s": N / ms * 1e3, "GBs": lg.numel() * 4 / ms / 1e6,
                        "roofline": roof(search_bytes(T, N, V, K), ms, ctc_kernel_name(V, K))}
    # C3: the bare step functions with S=100 rows of real history
    S = 100
    nb, b = torch.zeros((N, 1), device=device), torch.ones((N, 1), device=device)
    yh = torch.zeros((0, N, 1), dtype=torch.long, device=device)
    last = lens = torch.zeros((N, 1), dtype=torch.long, device=device)
    isp = torch.ones((N, 1, 1), dtype=torch.bool, device=device)
    step_args = None
    for t in range(S + 1):
        p = lg[t].softmax(1)
        nonext, blank = p[:, :V].contiguous(), p[:, V].contiguous()
        step_args = ((nonext.unsqueeze(1).expand(N, nb.shape[1], V), nonext, blank), K, (nb, b), yh, last, lens, isp)
        if t < S:
            yh, last, lens, (nb, b), isp, _, _ = F.ctc_prefix_search_advance(*step_args)
    ms = event_ms(lambda: F.ctc_prefix_search_advance(*step_args))
    step_bytes = (4 * (V + 1) + 8 * S * K + 8 * (S + 1) * K + K * K + 5 * 8 * K) * N
    out["C3_ctc_prefix_search_advance"] = {"workload": "one step, N=1024 K=16 V=1000 S=100", "ms": ms,
                                           "roofline": roof(step_bytes, ms, "pdt::ctc_advance_kernel")}
    g = torch.Generator(device=device).manual_seed(4)
    lpt = torch.randn((N, K, V), device=device, generator=g).log_softmax(-1)
    lpp = torch.randn((N, K), device=device, generator=g)
    yb = torch.randint(0, V, (S, N, K), device=device, generator=g)
    ybl = torch.full((N, K), S, device=device)
    ms = event_ms(lambda: F.beam_search_advance(lpt, K, lpp, yb, ybl))
    out["C3_beam_search_advance"] = {"workload": "one step, N=1024 K=16 V=1000 S=100", "ms": ms,
                                     "roofline": roof((4 * K * V + 8 * S * K + 8 * (S + 1) * K + 4 * 8 * K) * N, ms,
                                                      "pdt::beam_advance_flat_kernel")}
    del lg, lpt, yb, step_args
    # C4: SpecAugment N=2048 x 1000 x 80
    N, T, Fq = 2048, 1000, 80
    g = torch.Generator(device=device).manual_seed(7)
    feats = torch.randn((N, T, Fq), device=device, generator=g)
    lens = torch.randint(500, T + 1, (N,), device=device, generator=g)
    sa = M.SpecAugment(max_time_warp=80.0, max_freq_warp=0.0, max_time_mask=100, max_freq_mask=27,
                       max_time_mask_proportion=0.04, num_time_mask=2, num_time_mask_proportion=1.0,
                       num_freq_mask=2, interpolation_order=1)
    params = sa.draw_parameters(feats, lens)
    ms = event_ms(lambda: sa.apply_parameters(feats, params, lens))
    out["C4_spec_augment_apply"] = {"workload": "N=2048 T=1000 F=80, 2 time + 2 freq masks + time warp", "ms": ms,
                                    "utt_per_s": N / ms * 1e3, "GBs": 2 * feats.numel() * 4 / ms / 1e6,
                                    "roofline": roof(2 * 4 * feats.numel(), ms,
                                                     "pdt::spec_augment_rows_kernel (one launch: the time warp's spline solved in closed form inside it)")}
    ms = event_ms(lambda: sa(feats, lens))
    out["C4_SpecAugment_forward"] = {"workload": "draw + apply", "ms": ms, "utt_per_s": N / ms * 1e3,
                                     "roofline": roof(2 * 4 * feats.numel(), ms,
                                                      "pdt::spec_augment_draw_kernel + pdt::spec_augment_rows_kernel behind one operator")}
    img = feats.view(N, 1, T, Fq)
    src = torch.rand((N, 3, 2), device=device, generator=g) * torch.tensor([T - 1.0, Fq - 1.0], device=device)
    dst = src + torch.randn((N, 3, 2), device=device, generator=g)
    ms = event_ms(lambda: F.sparse_image_warp(img, src, dst, pinned_boundary_points=1, include_flow=False),
                  reps=3, warm=1)
    out["C4_sparse_image_warp"] = {"workload": "(2048,1,1000,80), 3 control + 4 pinned points, order 2", "ms": ms,
                                   "img_per_s": N / ms * 1e3, "GBs": 2 * feats.numel() * 4 / ms / 1e6,
                                   "roofline": roof(2 * 4 * feats.numel(), ms,
                                                    "pdt::spline_solve_kernel + pdt::warp_table_kernel + pdt::sparse_warp_bands_kernel<2, border, 7, 4>")}
    return out


def synthetic_bigram_dicts(V, successors=20, seed=0x5EED0007):
    """n-gram tables of a back-off bigram model over V tokens (+ an out-of-vocabulary sos): every token
    a unigram, `successors` random explicit bigrams per context token."""
    rng = np.random.default_rng(seed)
    uni = rng.normal(size=V + 1) - np.log(V)
    bo = rng.normal(size=V + 1) * 0.1 - 0.5
    d1 = {v: (float(uni[v]), float(bo[v])) for v in range(V + 1)}
    d2 = {}
    for a in range(V + 1):
        for b_ in rng.choice(V, successors, replace=False):
            d2[(a, int(b_))] = float(rng.normal() - 3.0)
    return [d1, d2]


def synthetic_trigram_dicts(V, successors=20, thirds=5, seed=0x5EED0017):
    """n-gram tables of a back-off TRIGRAM model over V tokens (+ an out-of-vocabulary sos): the unigrams and
    explicit bigrams of synthetic_bigram_dicts (now with back-off weights) and `thirds` random explicit
    trigrams under every explicit bigram -- (V + 1) * successors * thirds of them."""
    rng = np.random.default_rng(seed)
    d1, d2 = synthetic_bigram_dicts(V, successors)
    bo2 = rng.normal(size=len(d2)) * 0.1 - 0.4
    d2 = {k: (v, float(bo2[i])) for i, (k, v) in enumerate(d2.items())}
    d3 = {}
    for (a, b_) in d2:
        for c in rng.choice(V, thirds, replace=False):
            d3[(a, b_, int(c))] = float(rng.normal() - 2.5)
    return [d1, d2, d3]


def speechlike_logits(T, N, V, device, seed, dicts, rate=0.045):
    """Logits for the searches WITH a language model: N(0,1) + 12 on one class per frame like
    peaky_logits, but the peak is the blank in most frames and otherwise the next token of a sequence
    drawn along the model's explicit bigrams (`rate` tokens per frame).  Probability-space prefix masses
    (the reference's arithmetic, _decoding.py:1093-1202) then stay above the float32 underflow for the
    whole input: with a new uniformly random token in every frame -- SURVEY 8(d)'s input for the search
    WITHOUT a model -- every mass is exactly 0 after ~70 frames of shallow fusion (p_lm^0.2 ~ 0.25 per
    token) and the rest of the search is a tie among zeros."""
    rng = np.random.default_rng(seed)
    succ = {}
    for (a, b_) in dicts[1]:
        succ.setdefault(a, []).append(b_)
    width = max(len(v) for v in succ.values())
    table = np.zeros((V + 1, width), dtype=np.int64)
    for a in range(V + 1):
        opts = succ.get(a) or [0]
        table[a] = np.resize(np.asarray(opts), width)
    emit = rng.random((T, N)) < rate
    n_tok = int(emit.sum(0).max()) if N else 0
    seq = np.empty((max(n_tok, 1), N), dtype=np.int64)
    cur = np.full((N,), V, dtype=np.int64)  # the model's sos context
    for k in range(seq.shape[0]):
        cur = table[cur, rng.integers(0, width, N)]
        seq[k] = cur
    which = np.maximum(np.cumsum(emit, 0) - 1, 0)
    peak = np.where(emit, np.take_along_axis(seq, which, 0), V)
    g = torch.Generator(device=device).manual_seed(seed)
    lg = torch.randn((T, N, V + 1), device=device, generator=g)
    lg.scatter_add_(2, torch.from_numpy(peak).to(device).unsqueeze(2), torch.full((T, N, 1), 12.0, device=device))
    return lg


def synthetic_bigram_lm(M, V, device, successors=20, seed=0x5EED0007):
    """The LookupLanguageModel over synthetic_bigram_dicts (sos = V)."""
    return M.LookupLanguageModel(V, V, synthetic_bigram_dicts(V, successors, seed)).to(device)


def make_gru_lm(M, V, hidden=256):
    class GruLM(M.MixableSequentialLanguageModel):
        """1-layer GRU-cell model, hidden 256 -> Linear(256, V): the dense logit GEMM of SURVEY 8(d)."""

        def __init__(self, V, hidden=256):
            super().__init__(V)
            self.hidden = hidden
            self.embed = torch.nn.Embedding(V + 1, hidden)
            self.cell = torch.nn.GRUCell(hidden, hidden)
            self.out = torch.nn.Linear(hidden, V)

        def update_input(self, prev, hist):
            if "hidden" not in prev:
                prev = {"hidden": torch.zeros((hist.size(1), self.hidden), device=hist.device)}
            return prev

        def calc_idx_log_probs(self, hist, prev, idx):
            Vv, Np = self.vocab_size, hist.size(1)
            if idx.dim() == 0:
                idx = idx.expand(Np)
            tok = torch.full((Np,), Vv, dtype=torch.long, device=hist.device)
            if hist.size(0):
                lastt = hist.gather(0, (idx - 1).clamp(min=0).unsqueeze(0)).squeeze(0).clamp(0, Vv - 1)
                tok = torch.where(idx > 0, lastt, tok)
            h = self.cell(self.embed(tok), prev["hidden"])
            return self.out(h).log_softmax(-1), {"hidden": h}

        def extract_by_src(self, prev, src):
            return {"hidden": prev["hidden"].index_select(0, src)}

        def mix_by_mask(self, prev_true, prev_false, mask):
            return {"hidden": torch.where(mask.unsqueeze(1), prev_true["hidden"], prev_false["hidden"])}

    return GruLM(V, hidden)


def lm_configs(F, M, device, args, ref, hyp, say=lambda msg: None):
    """The runs SURVEY section 8(d) asks for besides the headline step: C2 with ragged lengths, the
    step with the reference's default warn=True, C3 with a language model in the loop (the shipped
    n-gram model; a GRU-cell model whose logit layer is the path's only GEMM), BeamSearch end to end."""
    out = _Reported(say)
    T, N, V = args.T, args.N, args.V
    # C2 ragged: eos = V written at len ~ U{T/2 .. T}
    g = torch.Generator(device=device).manual_seed(11)
    rr, hh = ref.clone(), hyp.clone()
    for x in (rr, hh):
        lens = torch.randint(T // 2, T + 1, (N,), device=device, generator=g)
        x[lens.clamp(max=T - 1), torch.arange(N, device=device)] = V
    rag = {}
    for name in ("error_rate", "prefix_error_rates", "optimal_completion"):
        fn = getattr(F, name)
        rag[name + "_ms"] = event_ms(lambda: fn(rr, hh, eos=V, warn=False), reps=3, warm=1)
    C = F.optimal_completion(rr, hh, eos=V, warn=False).shape[-1]
    rag["optimal_completion_C"] = C
    rag["optimal_completion_GBs"] = 8 * (T + 1) * C * N / rag["optimal_completion_ms"] / 1e6
    rag["roofline"] = roof((8 * 2 * T + 8 * (T + 1) * C) * N, rag["optimal_completion_ms"],
                           "pdt::lev_classify_kernel<8> + pdt::oc_bitpar_kernel<3> + pdt::oc_expand_tiles_kernel<8>",
                           "optimal_completion of the ragged pair (the HBM-write-bound operator of the three)")
    rag["ms"] = rag["error_rate_ms"] + rag["prefix_error_rates_ms"] + rag["optimal_completion_ms"]
    rag["workload"] = "C2 shapes, eos={} at len ~ U{{T/2..T}} in ref and hyp".format(V)
    out["C2_ragged"] = rag
    # the string operators of the step with the reference's default arguments (warn=True: one host
    # read of the status word per call, like the reference's own .any() checks)
    def strings_default():
        F.error_rate(ref, hyp)
        F.prefix_error_rates(ref, hyp)
        F.optimal_completion(ref, hyp)

    def strings_nowarn():
        F.error_rate(ref, hyp, warn=False)
        F.prefix_error_rates(ref, hyp, warn=False)
        F.optimal_completion(ref, hyp, warn=False)

    out["step_default_warn"] = {
        "workload": "error_rate + prefix_error_rates + optimal_completion of the step, default arguments (warn=True)",
        "ms": event_ms(strings_default, reps=5, warm=1), "ms_warn_false": event_ms(strings_nowarn, reps=5, warm=1),
    }
    # C3 with the shipped n-gram model in the loop (shallow fusion).  Two inputs: "speechlike" -- blank-
    # dominated frames, tokens drawn along the model's bigrams, every prefix mass stays above the
    # float32 underflow for all 1000 frames (speechlike_logits) -- and SURVEY 8(d)'s input for the search
    # without a model (a new uniformly random token peaks in every frame), on which every mass is
    # exactly 0 from frame ~70 on: the second is kept for continuity with rounds 2-3 and is NOT a
    # meaningful decode.
    T3, N3, V3, K = 1000, 1024, 1000, 16
    dicts = synthetic_bigram_dicts(V3)
    lm = M.LookupLanguageModel(V3, V3, [d.copy() for d in dicts]).to(device)
    lg_speech = speechlike_logits(T3, N3, V3, device, 0x5EED0009, dicts)
    lg = peaky_logits(T3, N3, V3, device, 0x5EED0003)
    search = M.CTCPrefixSearch(K, 0.2, lm)
    lm_kernel = "pdt::ctc_lm_table_kernel<16, 16> (factor table of the bigram model; csrc/ctc_lm_table.hip)"
    with torch.no_grad():
        search(lg[:8])
        ms_speech = event_ms(lambda: search(lg_speech), reps=3, warm=1)
        alive = float((search(lg_speech)[2][:, 0] > 0).float().mean())
        ms = event_ms(lambda: search(lg), reps=3, warm=0)
        dead = float((search(lg)[2][:, 0] == 0).float().mean())
    out["C3_search_lookup_lm"] = {
        "workload": "CTCPrefixSearch(16, beta=0.2, LookupLanguageModel bigram, 20 explicit successors per token), "
                    "N=1024 T=1000 V=1000, speechlike logits (~45 tokens per utterance among blank frames)",
        "ms": ms_speech, "ms_per_frame": ms_speech / T3, "utt_per_s": N3 / ms_speech * 1e3,
        "GBs": lg.numel() * 4 / ms_speech / 1e6, "utterances_with_positive_best_mass": alive, "reps": 3,
        "roofline": roof(search_bytes(T3, N3, V3, K), ms_speech, lm_kernel),
        "survey_input": {
            "workload": "the same search on SURVEY 8(d)'s no-model input (a new random token peaks in every frame): "
                        "masses underflow to 0 after ~70 frames of shallow fusion; rounds 2-3 quoted this number",
            "ms": ms, "utterances_with_zero_best_mass": dead, "reps": 3,
        },
    }

    # the same search with a TRIGRAM model: its (V + 1)^2 contexts as a 4 GB factor table (round 5; the
    # frame-by-frame kernel of csrc/ctc_lm_step.hip, which served orders above two before, beside it)
    dicts3 = synthetic_trigram_dicts(V3)
    lm3 = M.LookupLanguageModel(V3, V3, [d.copy() for d in dicts3]).to(device)
    search3 = M.CTCPrefixSearch(K, 0.2, lm3)
    from pydrobert_amd import switches as _sw

    with torch.no_grad():
        search3(lg_speech[:8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        search3(lg_speech[:2])  # (the table: built on the first call with this model)
        torch.cuda.synchronize()
        ms3 = event_ms(lambda: search3(lg_speech), reps=3, warm=1)
        alive3 = float((search3(lg_speech)[2][:, 0] > 0).float().mean())
        _sw.set("PDT_CTC_LM_TABLE", 0)
        ms3_step = event_ms(lambda: search3(lg_speech), reps=1, warm=0)
        _sw.set("PDT_CTC_LM_TABLE", 1)
    from pydrobert_amd import _decoding as _dec3

    tab3 = _dec3._FACTOR_TABLES.get(lm3)
    out["C3_search_lookup_lm_order3"] = {
        "workload": "CTCPrefixSearch(16, beta=0.2, LookupLanguageModel TRIGRAM: 20 explicit bigrams per token, 5 explicit "
                    "trigrams per bigram), N=1024 T=1000 V=1000, speechlike logits",
        "ms": ms3, "ms_per_frame": ms3 / T3, "utt_per_s": N3 / ms3 * 1e3, "reps": 3,
        "utterances_with_positive_best_mass": alive3,
        "factor_table_bytes": None if tab3 is None else int(tab3[1].numel() * 4),
        "ms_frame_kernel_route": ms3_step,
        "roofline": roof(search_bytes(T3, N3, V3, K), ms3, lm_kernel.replace("bigram", "trigram")),
    }
    del lm3, search3, tab3
    _dec3._FACTOR_TABLES.clear()
    torch.cuda.empty_cache()

    torch.manual_seed(5)
    gru = make_gru_lm(M, V3).to(device)
    search = M.CTCPrefixSearch(K, 0.2, gru)
    with torch.no_grad():
        search(lg_speech[:8])
        ms = event_ms(lambda: search(lg_speech), reps=3, warm=0)
        h = torch.randn((N3 * K, 256), device=device)
        gemm_ms = event_ms(lambda: gru.out(h), reps=5, warm=2)
    out["C3_search_gru_lm"] = {
        "workload": "CTCPrefixSearch(16, beta=0.2, GRU-cell LM hidden 256 -> Linear(256, 1000)), N=1024 T=1000 V=1000, "
                    "speechlike logits (an untrained model: p_lm ~ 1/V per token, masses still underflow by frame ~400)",
        "ms": ms, "ms_per_frame": ms / T3, "utt_per_s": N3 / ms * 1e3, "GBs": lg.numel() * 4 / ms / 1e6, "reps": 3,
        "logit_gemm_ms_per_frame": gemm_ms, "logit_gemm_share": gemm_ms * T3 / ms,
        "logit_gemm_TFLOPs": 2.0 * N3 * K * 256 * V3 / gemm_ms / 1e9,
        "gemm_kernel": "hipBLASLt / rocBLAS through torch.nn.Linear (kernel name: profiles/r03_gru_lm_kernel_stats.csv)",
        "roofline": {"bound": "mfma", "kernel": "Tensile GEMM (N K x 256 x V, fp32) inside torch.nn.Linear",
                     "achieved": 2.0 * N3 * K * 256 * V3 / gemm_ms / 1e9, "peak": 157.3, "unit": "TFLOP/s",
                     "frac": 2.0 * N3 * K * 256 * V3 / gemm_ms / 1e9 / 157.3,
                     "note": "the path's only GEMM, priced against the fp32 matrix peak (MI355X_MICROARCH.md); the "
                             "frame loop around it is launch- and glue-bound, see ms_per_frame"},
    }
    del lg, lg_speech
    # BeamSearch end to end: N=1024 paths x 16, V=1000, 100 iterations of the n-gram model
    bs = M.BeamSearch(lm, K, eos=0).to(device)
    from pydrobert_amd import _decoding as _dec

    with torch.no_grad():
        bs(None, 8, 4)
        _dec._BIGRAM_TABLES.clear()
        t0 = time.perf_counter()
        bs._bigram_table(device)
        torch.cuda.synchronize()
        table_ms = (time.perf_counter() - t0) * 1e3
        ms = event_ms(lambda: bs(None, N3, 100), reps=3, warm=1)
    # what the search really moves through HBM: the model's table once, a (source, token) word per beam entry and
    # iteration written and read back, the int64 paths written once (the score rows come out of the 4 MB table
    # in L2 / Infinity Cache; the step-by-step routes copied the (t, N, K) history every iteration)
    hist_bytes = 2 * 4 * N3 * K * 100 + 8 * N3 * K * 100
    out["BeamSearch_end_to_end"] = {
        "workload": "BeamSearch(LookupLanguageModel bigram, width 16, eos=0), batch 1024, 100 iterations, V=1000",
        "ms": ms, "ms_per_iteration": ms / 100, "paths_per_s": N3 * K / ms * 1e3, "reps": 3,
        "table_build_ms": table_ms,
        "table_build_note": "the model's dense (context, token) table + row statistics, built once per model (not in `ms`)",
        "row_bytes_from_cache": 4.0 * N3 * K * V3 * 100,
        "roofline": roof(hist_bytes + 4 * (V3 + 1) * V3, ms, "pdt::beam_search_table_kernel<true> + pdt::beam_search_walk_kernel (every iteration in one launch)",
                         "HBM bytes = one read of the table + the trie + the paths; 6.5 GB of score rows come from the "
                         "table in cache and are not HBM traffic: the search is bound by the per-iteration chain of a workgroup"),
    }
    return out


def ctc_kernel_name(V, W):
    """The instantiation the library launches for this row length (include/pdt_amd.h)."""
    from pydrobert_amd import _cabi

    plan = (ctypes.c_int32 * 5)()
    if _cabi.lib().pdt_ctc_prefix_search_plan(V, W, plan) != 0:
        return "pdt::ctc_search_kernel"
    if plan[3] == 3:  # long rows held in the producers' registers (ctc_rowreg.hip)
        nr = plan[4]
        return "pdt::ctc_rowreg_kernel<{}, {}, {}, {}>".format(nr, nr - (8 if nr <= 80 else 16 if nr <= 128 else 32), plan[0],
                                                                16 if W == 16 else -1)
    nt = V // 64 if (plan[3] == 1 and V // 64 == 4) else -1
    # (the default shape -- width 16, 256 tokens, contiguous rows -- has its constants compiled in)
    wc = 16 if (nt == 4 and W == 16) else -1
    vc = 256 if (wc == 16 and V == 256) else -1
    # (... and its producer takes two frames per pass unless PDT_CTC_PAIR=0)
    from pydrobert_amd import switches

    pair = "true" if (vc == 256 and plan[1] == 4 and switches.get("PDT_CTC_PAIR") != 0) else "false"
    return "pdt::ctc_search_kernel<{}, {}, {}, {}, {}, {}, {}>".format(
        plan[0], nt, "true" if plan[3] == 1 else "false", "true" if plan[3] == 2 else "false", wc, vc, pair)


def rendezvous_only(args, world, rank):
    """Preflight of the launch path alone: N ranks, one all-gather, no kernels."""
    import torch.distributed as dist

    backend = os.environ.get("PDT_BENCH_BACKEND", "nccl")
    dev = torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        dist.init_process_group(backend)
    mine = torch.full((3,), float(rank), device=dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(parts, mine)
    else:
        parts = [mine]
    ok = all(bool((p == r).all()) for r, p in enumerate(parts))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"rendezvous": "ok" if ok else "failed", "n_gpus": world, "backend": backend}), flush=True)
    if not ok:
        raise SystemExit(4)


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    if args.rendezvous_only:
        return rendezvous_only(args, world, rank)
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
        if dist.get_world_size() != args.gpus:
            raise SystemExit("asked for {} ranks, {} came up".format(args.gpus, dist.get_world_size()))

    from pydrobert_amd import functional as F
    from pydrobert_amd import modules as M

    ref, hyp = make_tokens(args, rank, device)
    N, T = args.N, args.T
    gathered = torch.empty((world * N,), device=device, dtype=torch.float) if world > 1 else None

    ops = ["error_rate", "prefix_error_rates", "optimal_completion"]
    have_decode = not args.no_decode
    logits = None
    if have_decode:
        ops.append("ctc_prefix_search")
        logits = peaky_logits(T, N, args.V, device, 0x5EED0003 + rank, chunk=T)
    if world > 1:
        ops.append("all_gather")
    C_seen = [0]

    def gather(er):
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(gathered, er)
        else:
            parts = [torch.empty(N) for _ in range(world)]
            dist.all_gather(parts, er.cpu())
            gathered.copy_(torch.cat(parts))

    def agree(flag):
        """True on every rank only if `flag` holds on every rank."""
        if world == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def gather_check(got, want):
        return agree(torch.equal(got, want))

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
            gather(er)
            mark()
        return er

    for _ in range(args.warmup):
        step()
    if world > 1:
        # once, before timing: the gathered vector is the concatenation of every rank's shard
        # (each rank regenerates every shard's tokens from the rank's seed and scores them itself)
        step()
        good = True
        for r in range(world):
            rr, hh = make_tokens(args, r, device)
            good = good and torch.equal(gathered[r * N:(r + 1) * N], F.error_rate(rr, hh, warn=False))
        if not agree(good):
            raise SystemExit("rank {}: gathered error rates differ from the shards'".format(rank))
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

    # per-rank spread of the op times (the line's op_ms are rank 0's; a SCALE record then shows whether
    # the slowest rank or the exchange sets the step): min / max over ranks of every op, the gather's
    # own event time among them
    op_ms_ranks = None
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, op_ms)
        op_ms_ranks = {o: {"min": min(r[o] for r in per_rank), "max": max(r[o] for r in per_rank)} for o in ops}

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
        "error_rate": "pdt::lev_classify_kernel<8> + pdt::lev_bitpar_kernel",
        "prefix_error_rates": "pdt::lev_classify_kernel<8> + pdt::lev_bitpar_kernel (every operator classifies its own "
                              "inputs; reuse across the pair is opt-in: pydrobert_amd._string.reuse_classification)",
        "optimal_completion": "pdt::lev_classify_kernel<8> + pdt::oc_bitpar_kernel<3> + pdt::oc_expand_tiles_kernel<8>",
        "ctc_prefix_search": ctc_kernel_name(args.V, args.beam),
    }
    # HBM bytes per launch and instruction counts from the rocprofv3 PMC passes (profiles/), only
    # when they were collected for this very configuration
    traffic, valu = None, None
    tpath = os.path.join(ROOT, "profiles", "r05_ctc_traffic.json")
    for older in ("r04_ctc_traffic.json", "r03_ctc_traffic.json"):
        if not os.path.exists(tpath):
            tpath = os.path.join(ROOT, "profiles", older)
    tname = os.path.basename(tpath)
    if dom == "ctc_prefix_search" and os.path.exists(tpath):
        rec = json.load(open(tpath))
        if rec.get("config") == {"N": N, "T": T, "V": args.V, "beam": args.beam}:
            traffic = rec.get("hbm_bytes_per_launch")
            if "sq" in rec:
                valu = {
                    "wave_insts_per_launch": rec["sq"]["SQ_INSTS_VALU_per_launch"],
                    "pipe_busy_frac": rec["sq"].get("valu_pipe_busy_frac_measured"),
                    "cycles_per_wave_inst": rec["sq"].get("valu_cycles_per_inst_measured"),
                    "profiled_ms": rec.get("avg_ms_profiled"),
                    "source": "profiles/" + tname + " (SQ_INSTS_VALU and SQ_ACTIVE_INST_VALU PMC passes over this "
                              "configuration: busy = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs against GRBM_GUI_ACTIVE / 8 "
                              "cycles of the launch; a committed record, not collected in this run)",
                }

    def say(msg):
        if rank == 0:
            print("bench.py: " + msg, file=sys.stderr, flush=True)

    say("timed region done: {:.3f} ms per step".format(dt / args.steps * 1e3))
    extra = None
    if not args.no_extra and have_decode:
        # (one stderr line per configuration as it finishes, and `config_summary` as the LAST key of the
        #  JSON line: whatever tail of either stream a driver keeps carries every BASELINE config)
        extra = other_configs(F, M, device, world, rank, dist, gather_check, say)
        say("other configs done")
        if rank == 0:
            extra.update(lm_configs(F, M, device, args, ref, hyp, say))
            say("ragged / default-argument / language-model configs done")

    if rank == 0:
        out = {
            "metric": METRIC,
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
                "workload": "BASELINE configs[1] shapes on every GPU (its own N utterances): " + " + ".join(ops),
                "N_per_gpu": N, "T_ref": T, "T_hyp": T, "V": args.V, "beam": args.beam,
                "optimal_completion_C": C, "sharding": "batch axis, {} rank(s)".format(world),
                "gather_verified": world > 1,
            },
            "op_ms": op_ms,
            "op_ms_over_ranks": op_ms_ranks,
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
                "traffic_source": None if traffic is None else
                    "profiles/" + tname + ": FETCH_SIZE / WRITE_SIZE PMC passes over this configuration "
                    "(2 x FETCH + WRITE), a committed record -- not collected in this run",
                "valu": valu,
                "note": "priced against HBM as the contract asks; the kernel's own limiter is vector-"
                        "instruction issue (valu.pipe_busy_frac; DESIGN.md section 4.3)",
            },
        }
        if extra is not None:
            out["other_configs"] = dict(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, have_decode, op_ms)
            say("cpu baseline done")
        if extra is not None:  # LAST key on purpose (see above)
            out["config_summary"] = {name: summary_of(entry) for name, entry in extra.items()}
            out["config_summary"]["headline"] = {"ms_per_step": round(dt / args.steps * 1e3, 4), "frac": round(achieved / HBM_PEAK_GBS, 4),
                                                 **{k + "_ms": round(v, 4) for k, v in op_ms.items()}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be positive")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()
