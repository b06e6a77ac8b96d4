"""Fuzz of the lean tier's side paths (mid tier, exact re-ranking, all-pairs prefix update) against the full
tiers / per-descendant loops: the same search with PDT_CTC_LEAN_EXTRA = 1 and 0, torch.equal on all outputs.
python profiles/tools/fuzz_tiers.py [seed] [cases]"""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F, switches
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
bad = 0
for it in range(cases):
    V = int(rng.choice([256, 256, 257, 300, 128, 64, 1000, 5000, 40])); W = int(rng.choice([16, 16, 16, 8, 12, 5, 2, 32]))
    T = int(rng.choice([20, 64, 130, 300, 600])); N = int(rng.integers(1, 40))
    if V >= 1000: N = min(N, 6); T = min(T, 130)
    kind = it % 5
    lg = rng.normal(size=(T, N, V + 1)).astype(np.float32) * float(rng.choice([0.3, 1.0, 2.0]))
    peak = rng.integers(0, V + 1, (T, N, 1))
    if kind == 3:
        peak = np.where(rng.random((T, N, 1)) < float(rng.choice([0.8, 0.95])), V, peak)
    np.put_along_axis(lg, peak, float(rng.choice([4.0, 8.0, 12.0])), 2)
    if kind == 0:
        lg = np.round(lg * 2) / 2
    elif kind == 1:
        for _ in range(T * N // 3):
            t, n = rng.integers(0, T), rng.integers(0, N); a, b = rng.integers(0, V, 2)
            lg[t, n, b] = lg[t, n, a]
    elif kind == 2:
        lg = np.round(lg * 1024) / 1024
    elif kind == 4:
        lg[:, :, rng.integers(0, V, max(1, V // 4))] = -np.inf
    lens = torch.from_numpy(rng.integers(0, T + 1, N)).to(dev) if rng.random() < 0.5 else None
    x = torch.from_numpy(lg).to(dev)
    outs = []
    for extra in (1, 0):
        switches.set("PDT_CTC_LEAN_EXTRA", extra)
        outs.append(F.ctc_prefix_search(x, W, lens))
    if not all(torch.equal(p, q) for p, q in zip(*outs)):
        bad += 1
        (y1, l1, p1), (y0, l0, p0) = outs
        dn = ((l1 != l0) | (p1 != p0)).any(1).nonzero().flatten().tolist()
        dy = (y1 != y0).any(0).any(1).nonzero().flatten().tolist()
        n = (dn + dy)[0]
        print("MISMATCH case", it, "V", V, "W", W, "T", T, "N", N, "kind", kind, "utterances", sorted(set(dn + dy))[:6],
              "probs", p1[n].tolist(), p0[n].tolist(), "lens", l1[n].tolist(), l0[n].tolist(), flush=True)
switches.set("PDT_CTC_LEAN_EXTRA", 1)
print("tier fuzz: %d cases, %d mismatches" % (cases, bad), flush=True)

# ---- the same two settings through the searches with a bigram model (per-prefix lists: DENSE ctc_frame) ----
sys.path.insert(0, "tests")
from pydrobert_amd import modules as M
from _lm_fixtures import random_dicts
bad2 = n2 = 0
for it in range(cases // 5):
    V = int(rng.choice([9, 40, 150, 600])); W = int(rng.choice([2, 4, 8, 16, 16])); T = int(rng.choice([20, 80, 200])); N = int(rng.integers(1, 12))
    if W > V + 1:
        continue
    sos = int(rng.choice([-1, 0, V - 1])); vm = bool(rng.integers(0, 2)); beta = float(rng.choice([0.1, 0.3, 0.7]))
    dicts = random_dicts(rng, V, 2, 0.5 if V ** 2 < 3000 else (0.1 if V < 200 else 0.01), sos if sos < 0 else None)
    for v in range(V):
        dicts[0].setdefault(v, (float(rng.normal()), float(rng.normal())))
    lm = M.LookupLanguageModel(V, sos, dicts, destructive=True).to(dev)
    lg = rng.normal(size=(T, N, V + 1)).astype(np.float32)
    peak = rng.integers(0, V + 1, (T, N, 1))
    if it % 2:
        peak = np.where(rng.random((T, N, 1)) < 0.9, V, peak)
    np.put_along_axis(lg, peak, float(rng.choice([4.0, 9.0])), 2)
    if it % 3 == 0:
        lg = np.round(lg * 2) / 2
    x = torch.from_numpy(lg).to(dev)
    search = M.CTCPrefixSearch(W, beta, lm, valid_mixture=vm)
    outs = []
    for extra in (1, 0):
        switches.set("PDT_CTC_LEAN_EXTRA", extra)
        with torch.no_grad():
            outs.append(search(x))
    n2 += 1
    if not all(torch.equal(p, q) for p, q in zip(*outs)):
        bad2 += 1
        print("LM MISMATCH case", it, V, W, T, N, sos, vm, beta, flush=True)
switches.set("PDT_CTC_LEAN_EXTRA", 1)
print("tier fuzz with a bigram model: %d cases, %d mismatches" % (n2, bad2), flush=True)
