#!/usr/bin/env python
"""Randomised sweep of EVERY C entry point of the oracle through its Python wrappers -- the wrappers
size the buffers, so a disagreement between a wrapper's allocation and the C code's writes (the
round-4 heap overflow: mask mode, H == 0, exclude_last) is a sanitizer report.  numpy only.

    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 PDT_ORACLE_SANITIZE=1 \\
        python tests/fuzz/oracle_sweep.py SEED CALLS

Degenerate shapes are drawn on purpose: H / R / N / T / S of 0 and 1, beams wider than the candidate
set (K > V + 1, K > K'.V), utterances of length 0, empty references, tokens outside the vocabulary.
Exceptions the wrappers raise for shapes the reference raises on (IndexError / RuntimeError) are
counted, not errors.  ``tests/test_oracle_sanitizers.py`` runs this under ASan + UBSan."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import oracle  # noqa: E402


def small(rng, hi, p0=0.12, p1=0.12):
    u = rng.random()
    if u < p0:
        return 0
    if u < p0 + p1:
        return 1
    return int(rng.integers(0, hi + 1))


def string_call(rng):
    R, H, N = small(rng, 12), small(rng, 12), small(rng, 5, 0.05, 0.3)
    V = int(rng.integers(1, 5))
    ref = rng.integers(0, V + 1, (R, N))
    hyp = rng.integers(0, V + 1, (H, N))
    bf = bool(rng.integers(0, 2))
    if bf:
        ref, hyp = np.ascontiguousarray(ref.T), np.ascontiguousarray(hyp.T)
    eos = None if rng.random() < 0.4 else int(rng.integers(-1, V + 1))
    costs = [(1, 1, 1), (2, 2, 2), (3, 3, 4), (2, 0.5, 1), (0.1, 0.7, 1.3)][int(rng.integers(0, 5))]
    kw = dict(eos=eos, include_eos=bool(rng.integers(0, 2)), batch_first=bf, ins_cost=costs[0],
              del_cost=costs[1], sub_cost=costs[2], faithful=bool(rng.integers(0, 2)))  # fmt: skip
    op = int(rng.integers(0, 5))
    ex = bool(rng.integers(0, 2))
    if op == 0:
        oracle.error_rate(ref, hyp, norm=bool(rng.integers(0, 2)), **kw)
    elif op == 1:
        oracle.edit_distance(ref, hyp, norm=bool(rng.integers(0, 2)), **kw)
    elif op == 2:
        oracle.prefix_error_rates(ref, hyp, norm=bool(rng.integers(0, 2)), exclude_last=ex, **kw)
    elif op == 3:
        oracle.prefix_edit_distances(ref, hyp, norm=bool(rng.integers(0, 2)), exclude_last=ex, **kw)
    else:
        oracle.optimal_completion(ref, hyp, exclude_last=ex, **kw)


def ctc_search_call(rng):
    T, N, V = small(rng, 9), small(rng, 4, 0.05, 0.3), int(rng.integers(1, 6))
    K = int(rng.integers(1, 2 * V + 4))  # (K > V + 1 among them)
    logits = (3 * rng.standard_normal((T, N, V + 1))).astype(np.float32)
    lens = None if rng.random() < 0.3 else rng.integers(0, T + 1, N)
    oracle.ctc_prefix_search(logits, K, lens)


def ctc_step_call(rng):
    N, Kp, V, S = small(rng, 3, 0.05, 0.3), int(rng.integers(1, 5)), int(rng.integers(1, 6)), small(rng, 6)
    W = int(rng.integers(1, Kp * (V + 1) + 3))
    ext = rng.random((N, Kp, V)).astype(np.float32)
    nonext = rng.random((N, V)).astype(np.float32)
    blank = rng.random((N,)).astype(np.float32)
    nb, b = rng.random((N, Kp)).astype(np.float32), rng.random((N, Kp)).astype(np.float32)
    y_prev = rng.integers(-1, V + 1, (S, N, Kp))  # (tokens outside the vocabulary: clamped, :779, :808)
    last = rng.integers(-1, V + 1, (N, Kp))
    lens = rng.integers(0, S + 1, (N, Kp))
    isp = rng.integers(0, 2, (N, Kp, Kp)).astype(bool)
    oracle.ctc_prefix_search_advance((ext, nonext, blank), W, (nb, b), y_prev, last, lens, isp)


def beam_step_call(rng):
    N, Kp, V, S = small(rng, 3, 0.05, 0.3), int(rng.integers(1, 5)), int(rng.integers(1, 6)), small(rng, 6)
    W = int(rng.integers(1, Kp * V + 3))
    lpt = rng.standard_normal((N, Kp, V)).astype(np.float32)
    lpp = rng.standard_normal((N, Kp)).astype(np.float32)
    y_prev = rng.integers(0, V, (S, N, Kp))
    lens = None if rng.random() < 0.4 else rng.integers(0, S + 1, (N, Kp))
    oracle.beam_search_advance(lpt, W, lpp, y_prev, lens)


def main(seed, calls):
    rng = np.random.default_rng(seed)
    kinds = [string_call] * 5 + [ctc_search_call] * 2 + [ctc_step_call] * 2 + [beam_step_call]
    done = refused = 0
    for _ in range(calls):
        try:
            kinds[int(rng.integers(0, len(kinds)))](rng)
        except (IndexError, RuntimeError):  # (shapes the reference raises on)
            refused += 1
        done += 1
    print("oracle_sweep: {} calls ({} on shapes the reference refuses), no sanitizer report".format(done, refused))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 12000)
