"""The lean tier's mid / exact paths against the full tiers on tie-heavy inputs: run once with the shipped
library and once with a -DPDT_NO_MID_TIER -DPDT_NO_EXACT_LEAN build (PDT_AMD_LIB), each run saves its outputs;
`compare` loads both and demands torch.equal.  python tiers_equal.py run <tag> | compare <tagA> <tagB>"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import numpy as np, torch
if sys.argv[1] == "compare":
    a, b = torch.load("gpurun_out/tiers_%s.pt" % sys.argv[2]), torch.load("gpurun_out/tiers_%s.pt" % sys.argv[3])
    bad = sum(0 if all(torch.equal(x, y) for x, y in zip(p, q)) else 1 for p, q in zip(a, b))
    print("tiers: %d cases, %d differ" % (len(a), bad)); sys.exit(1 if bad else 0)
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
rng = np.random.default_rng(99)
outs = []
for it in range(60):
    V = int(rng.choice([256, 256, 300, 200, 1000, 40])); W = int(rng.choice([16, 16, 8, 32, 5])); T = int(rng.choice([40, 120, 300])); N = int(rng.integers(2, 40))
    kind = it % 3
    lg = rng.normal(size=(T, N, V + 1)).astype(np.float32)
    np.put_along_axis(lg, rng.integers(0, V + 1, (T, N, 1)), float(rng.choice([6.0, 9.0, 12.0])), 2)
    if kind == 0:    # a few distinct values only: exact ties everywhere
        lg = np.round(lg * 2) / 2
    elif kind == 1:  # duplicate logits in a fraction of the frames (two tokens exactly equal)
        for _ in range(T * N // 4):
            t, n = rng.integers(0, T), rng.integers(0, N); a, b = rng.integers(0, V, 2)
            lg[t, n, b] = lg[t, n, a]
    else:            # near ties: values on a 2^-12 grid
        lg = np.round(lg * 4096) / 4096
    lens = torch.from_numpy(rng.integers(T // 2, T + 1, N)).to(dev)
    outs.append(tuple(o.cpu() for o in F.ctc_prefix_search(torch.from_numpy(lg).to(dev), W, lens)))
os.makedirs("gpurun_out", exist_ok=True)
torch.save(outs, "gpurun_out/tiers_%s.pt" % sys.argv[2])
print("saved", len(outs))
