"""One-off fuzz of the bit-parallel optimal-completion mask kernel against the row-synchronous one
(PDT_OC_BITPAR=0): random shapes, lengths, vocabularies, eos handling, layouts.  Prints the first
disagreement or the number of cases."""
import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F
from pydrobert_amd import switches
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for it in range(cases):
    R = int(rng.choice([1, 2, 5, 17, 31, 32, 33, 64, 100, 129, 255, 256, 300, 511, 512]))
    H = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 33, 100, 257, 600]))
    N = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 63, 130, 257]))
    V = int(rng.choice([1, 2, 3, 5, 30, 300, 40000]))
    ref = rng.integers(0, V, (R, N)); hyp = rng.integers(0, V, (H, N))
    if rng.random() < 0.3:
        k = min(R, H); hyp[:k] = ref[:k]
        for _ in range(3): hyp[rng.integers(0, H), rng.integers(0, N)] = rng.integers(0, V)
    kw = {}
    if rng.random() < 0.6:
        kw["eos"] = int(rng.integers(0, V)); kw["include_eos"] = bool(rng.integers(0, 2))
    kw["exclude_last"] = bool(rng.integers(0, 2)); kw["batch_first"] = bool(rng.integers(0, 2))
    a, b = (ref.T.copy(), hyp.T.copy()) if kw["batch_first"] else (ref, hyp)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    if rng.random() < 0.3:  # non-contiguous views
        ta = torch.stack([ta, ta], -1)[..., 0]
    switches.set("PDT_OC_BITPAR", 1); x = F.optimal_completion(ta, tb, warn=False, **kw)
    switches.set("PDT_OC_BITPAR", 0); y = F.optimal_completion(ta, tb, warn=False, **kw)
    if x.shape != y.shape or not torch.equal(x, y):
        print("MISMATCH", it, R, H, N, V, kw); sys.exit(1)
print("fuzz_oc: %d cases, no disagreement" % cases)
