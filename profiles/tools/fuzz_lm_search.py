"""One-off fuzz of pdt_ctc_lookup_lm_search (the whole search from one call: history slots, cached
factor rows) against the host's frame loop around the one-kernel frames and, every fourth case, the
three-kernel route: random models of order 2-4, widths, vocabularies, ragged lengths, both mixes."""
import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd"); sys.path.insert(0, "tests")
from pydrobert_amd import modules as M
from pydrobert_amd import switches
from _lm_fixtures import random_dicts
dev = torch.device("cuda:0")
# (bigram models would otherwise take the factor-table search, whose fused softmax differs from torch's in
# the last bits: that route is fuzzed against the oracle with a tolerance by fuzz_r04.py)
switches.set("PDT_CTC_LM_TABLE", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for it in range(cases):
    order = int(rng.choice([2, 2, 3, 4]))
    V = int(rng.choice([4, 7, 12, 30] if order > 2 else [4, 9, 40, 150]))
    W = int(rng.choice([1, 2, 4, 8, 16, 32])); T = int(rng.integers(1, 40)); N = int(rng.integers(1, 9))
    sos = int(rng.choice([-1, 0, V - 1])); vm = bool(rng.integers(0, 2)); beta = float(rng.choice([0.1, 0.3, 0.7]))
    dicts = random_dicts(rng, V, order, 0.5 if V ** order < 3000 else 0.1, sos if sos < 0 else None)
    for v in range(V):
        dicts[0].setdefault(v, (float(rng.normal()), float(rng.normal())))
    lm = M.LookupLanguageModel(V, sos, dicts, destructive=True).to(dev)
    lg = rng.normal(size=(T, N, V + 1)).astype(np.float32)
    np.put_along_axis(lg, rng.integers(0, V + 1, (T, N, 1)), float(rng.choice([2.0, 5.0, 9.0])), 2)
    lens = torch.from_numpy(rng.integers(0, T + 1, N)).to(dev) if rng.random() < 0.6 else None
    search = M.CTCPrefixSearch(W, beta, lm, valid_mixture=vm)
    x = torch.from_numpy(lg).to(dev)
    switches.set("PDT_CTC_LM_FUSED", 1); switches.set("PDT_CTC_LM_SEARCH", 1)
    y, yl, yp = search(x, lens)
    switches.set("PDT_CTC_LM_SEARCH", 0)
    if it % 4 == 3: switches.set("PDT_CTC_LM_FUSED", 0)
    ey, eyl, eyp = search(x, lens)
    mask = torch.arange(y.shape[0], device=dev).view(-1, 1, 1) < yl.unsqueeze(0)
    ok = y.shape == ey.shape and torch.equal(yl, eyl) and torch.equal(yp, eyp) and torch.equal(torch.where(mask, y, ey), ey)
    if not ok:
        print("MISMATCH", it, order, V, W, T, N, sos, vm, beta, lens); sys.exit(1)
print("fuzz_lm_search: %d cases, no disagreement" % cases)
