"""Helpers shared by the language-model tests: rebuild the n-gram tables stored in
tests/golden/lm.npz and draw random tables in the reference's test style
(tests/test_lm.py:218-243)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lm.npz")


def golden():
    return np.load(GOLDEN)


def dicts_from_golden(g, tag):
    V, sos, N = (int(x) for x in g[tag + "_cfg"][:3])
    dicts = []
    for n in range(N):
        keys, vals = g["{}_keys{}".format(tag, n)], g["{}_vals{}".format(tag, n)]
        d = {}
        for k, v in zip(keys, vals):
            key = int(k[0]) if n == 0 else tuple(int(x) for x in k)
            d[key] = float(v[0]) if n == N - 1 else (float(v[0]), float(v[1]))
        dicts.append(d)
    return V, sos, N, dicts


def random_dicts(rng, V, N, density=0.5, sos=None):
    """Every possible n-gram kept with probability ``density`` (values ~ N(0, 1))."""
    dicts = []
    for n in range(1, N + 1):
        d = {}
        for flat in range(V**n):
            if rng.random() >= density:
                continue
            key, x = [], flat
            for _ in range(n):
                key.append(x % V)
                x //= V
            key = key[0] if n == 1 else tuple(key)
            d[key] = float(rng.normal()) if n == N else (float(rng.normal()), float(rng.normal()))
        dicts.append(d)
    if sos is not None:
        dicts[0][sos] = -99.0 if N == 1 else (-99.0, 0.0)
    return dicts
