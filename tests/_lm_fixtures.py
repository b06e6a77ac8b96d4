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


LM_SEARCH_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lm_search.npz")


def lm_search_golden():
    """tests/golden/lm_search.npz (searches of the live reference with LookupLanguageModels and a
    stateful toy model in the loop) and its n-gram tables as {tag: (V, sos, order, dicts)}."""
    g = np.load(LM_SEARCH_GOLDEN)
    tags = [str(t) for t in g["model_tags"]]
    models = {}
    for tag in tags:
        V, sos, N = (int(x) for x in g[tag + "_cfg"])
        dicts = []
        for n in range(N):
            keys, vals = g["{}_keys{}".format(tag, n)], g["{}_vals{}".format(tag, n)]
            d = {}
            for k, v in zip(keys, vals):
                key = int(k[0]) if n == 0 else tuple(int(x) for x in k)
                d[key] = float(v[0]) if n == N - 1 else (float(v[0]), float(v[1]))
            dicts.append(d)
        models[tag] = (V, sos, N, dicts)
    return g, tags, models


def same_paths(y, lens, ey, elens):
    """Token tensors equal inside the reported lengths (the reference leaves the rest undefined)."""
    y, lens, ey, elens = (np.asarray(x) for x in (y, lens, ey, elens))
    if y.shape != ey.shape or not np.array_equal(lens, elens):
        return False
    inside = np.arange(y.shape[0]).reshape((-1,) + (1,) * (y.ndim - 1)) < lens[None]
    return np.array_equal(np.where(inside, y, 0), np.where(inside, ey, 0))
