"""Fuzz of round 4's two new search kernels (python tests/fuzz/fuzz_r04.py [seed] [cases]).

1. ctc_prefix_search without a model: the register-row form (PDT_CTC_ROWREG=1, and =2: from 128 tokens)
   against the LDS / workspace rows (=0) -- torch.equal on all three outputs -- over random vocabularies
   (instantiation boundaries included), widths 1..32, ragged lens, masked (-inf) tokens, flat and peaky
   rows, exact ties; a sample also against the oracle.
2. CTCPrefixSearch with a bigram LookupLanguageModel: the factor-table search against the oracle's
   restatement of the reference loop (tokens and lengths exact on tie-free cases, probabilities 1e-5) and
   against the three-kernel route.
"""
import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd"); sys.path.insert(0, "tests")
import oracle
from pydrobert_amd import functional as F, modules as M, switches
from _lm_fixtures import random_dicts
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
edges = [64 * k + d for k in (2, 5, 8, 16, 24, 40, 72, 80, 96, 128, 160, 224, 256) for d in (-1, 0, 1, 63)]
bad = 0
for it in range(cases):
    V = int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(120, 6000))
    W = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 32])); T = int(rng.integers(1, 30)); N = int(rng.integers(1, 7))
    kind = it % 4
    lg = rng.normal(size=(T, N, V + 1)).astype(np.float32) * (0.01 if kind == 1 else 1.0)
    if kind != 1:
        np.put_along_axis(lg, rng.integers(0, V + 1, (T, N, 1)), float(rng.choice([3.0, 8.0, 13.0])), 2)
    if kind == 2:  # masked tokens
        lg[:, :, rng.integers(0, V, max(1, V // 3))] = -np.inf
    if kind == 3:  # exact ties: a few distinct values only
        lg = np.round(lg * 2) / 2
    lens = torch.from_numpy(rng.integers(0, T + 1, N)).to(dev) if rng.random() < 0.5 else None
    x = torch.from_numpy(lg).to(dev)
    outs = []
    for mode in (0, 1, 2):
        switches.set("PDT_CTC_ROWREG", mode)
        outs.append(F.ctc_prefix_search(x, W, lens))
    for m in (1, 2):
        if not all(torch.equal(a, b) for a, b in zip(outs[0], outs[m])):
            bad += 1
            print("ROWREG MISMATCH mode", m, "case", it, "V", V, "W", W, "T", T, "N", N, "kind", kind, flush=True)
    if it % 5 == 0 and kind == 0 and W <= V + 1:
        ey, eyl, eyp = oracle.ctc_prefix_search(lg, W, None if lens is None else lens.cpu().numpy())
        y, yl, yp = (o.cpu().numpy() for o in outs[1])
        fin = np.isfinite(eyp)
        alive = fin.any() and eyp[fin].min() > 1e-30  # (below: denormals and ties among zeros)
        if alive and not (np.array_equal(yl[fin], eyl[fin]) and np.array_equal(y, ey) and np.allclose(yp[fin], eyp[fin], rtol=1e-5)):
            bad += 1
            print("ORACLE MISMATCH case", it, V, W, T, N, flush=True)
switches.set("PDT_CTC_ROWREG", 1)
print("rowreg fuzz: %d cases, %d mismatches" % (cases, bad), flush=True)
bad2 = 0
for it in range(cases // 2):
    V = int(rng.choice([4, 9, 40, 150, 600])); W = int(rng.choice([1, 2, 4, 8, 16, 32]))
    if W > V + 1:
        continue
    T = int(rng.integers(1, 40)); N = int(rng.integers(1, 6))
    sos = int(rng.choice([-1, 0, V - 1])); vm = bool(rng.integers(0, 2)); beta = float(rng.choice([0.1, 0.3, 0.7]))
    dicts = random_dicts(rng, V, 2, 0.5 if V ** 2 < 3000 else (0.1 if V < 200 else 0.01), sos if sos < 0 else None)
    for v in range(V):
        dicts[0].setdefault(v, (float(rng.normal()), float(rng.normal())))
    lm = M.LookupLanguageModel(V, sos, [d.copy() for d in dicts]).to(dev)
    olm = oracle.NGramLM(V, sos, dicts)
    lg = rng.normal(size=(T, N, V + 1)).astype(np.float32)
    np.put_along_axis(lg, rng.integers(0, V + 1, (T, N, 1)), float(rng.choice([2.0, 5.0, 9.0])), 2)
    lens_np = rng.integers(0, T + 1, N) if rng.random() < 0.6 else None
    lens = None if lens_np is None else torch.from_numpy(lens_np).to(dev)
    search = M.CTCPrefixSearch(W, beta, lm, valid_mixture=vm)
    x = torch.from_numpy(lg).to(dev)
    switches.set("PDT_CTC_LM_TABLE", 1)
    y, yl, yp = (o.cpu().numpy() for o in search(x, lens))
    switches.set("PDT_CTC_LM_TABLE", 0); switches.set("PDT_CTC_LM_FUSED", 0)
    ty, tyl, typ = (o.cpu().numpy() for o in search(x, lens))
    switches.set("PDT_CTC_LM_FUSED", 1)
    ey, eyl, eyp = oracle.ctc_prefix_search_lm(lg, W, lens_np, olm, beta, vm)
    srt = -np.sort(-eyp, 1)
    # (masses near the float32 underflow lose their digits: a product of T factors below ~1e-30 is
    # compared on nothing; near ties are the reference's to break)
    tie_free = np.isfinite(eyp).all() and eyp.min() > 1e-30 and (
        W == 1 or ((srt[:, :-1] - srt[:, 1:]) > 1e-4 * np.abs(srt[:, :-1])).all())
    inside = np.arange(y.shape[0])[:, None, None] < yl[None]
    ok_o = np.array_equal(yl, eyl) and np.array_equal(np.where(inside, y, 0), ey) and np.allclose(yp, eyp, rtol=1e-5, atol=0)
    tin = np.arange(ty.shape[0])[:, None, None] < tyl[None]
    ok_t = np.array_equal(yl, tyl) and np.array_equal(np.where(inside, y, 0), np.where(tin, ty, 0)) and np.allclose(yp, typ, rtol=1e-5, atol=0)
    if tie_free and not (ok_o and ok_t):
        bad2 += 1
        print("LM TABLE MISMATCH case", it, "V", V, "W", W, "T", T, "N", N, "sos", sos, "vm", vm, "beta", beta, "oracle", ok_o, "three", ok_t,
              "lens equal", np.array_equal(yl, eyl), "max rel", float(np.abs(yp / eyp - 1).max()), flush=True)
print("lm table fuzz: %d mismatches" % bad2, flush=True)
sys.exit(1 if bad or bad2 else 0)
