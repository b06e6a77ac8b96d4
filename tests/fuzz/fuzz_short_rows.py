"""Fuzz of the short-row CTC search instances (python tests/fuzz/fuzz_short_rows.py [seed] [cases]).

The default shape (V = 256, width 16, contiguous rows) and width 16 with V = 256..319 run instantiations
with those shapes compiled in (ctc_search.hip); the same values through a vocabulary axis with a stride
run the general kernels.  torch.equal on all three outputs over peaky / flat / masked (-inf) / tied rows,
ragged lengths, up to 400 frames; a sample against the oracle."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
import oracle
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(seed)
bad = 0
for it in range(cases):
    V = int(rng.choice([256, 256, 256, 257, 263, 300, 319, 255, 200, 128]))
    W = int(rng.choice([16, 16, 16, 15, 17, 8, 32, 1]))
    T = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 257, 400])); N = int(rng.integers(1, 9))
    kind = it % 4
    lg = rng.normal(size=(T, N, V + 1)).astype(np.float32) * (0.01 if kind == 1 else 1.0)
    if kind != 1:
        np.put_along_axis(lg, rng.integers(0, V + 1, (T, N, 1)), float(rng.choice([3.0, 8.0, 13.0])), 2)
    if kind == 2:
        lg[:, :, rng.integers(0, V, max(1, V // 3))] = -np.inf
    if kind == 3:
        lg = np.round(lg * 2) / 2
    lens = torch.from_numpy(rng.integers(0, T + 1, N)).to(dev) if rng.random() < 0.5 else None
    x = torch.from_numpy(lg).to(dev)
    wide = torch.zeros((T, N, 2 * (V + 1)), device=dev)
    wide[:, :, ::2] = x
    a = F.ctc_prefix_search(x, W, lens)
    b = F.ctc_prefix_search(wide[:, :, ::2], W, lens)
    if not all(torch.equal(p, q) for p, q in zip(a, b)):
        bad += 1
        print("INSTANCE MISMATCH case", it, "V", V, "W", W, "T", T, "N", N, "kind", kind, flush=True)
    if it % 4 == 0 and T <= 100 and W <= V + 1:
        ey, eyl, eyp = oracle.ctc_prefix_search(lg, W, None if lens is None else lens.cpu().numpy())
        y, yl, yp = (o.cpu().numpy() for o in a)
        fin = np.isfinite(eyp)
        alive = fin.any() and eyp[fin].min() > 1e-30
        if alive and not (np.array_equal(yl[fin], eyl[fin]) and np.array_equal(y, ey) and np.allclose(yp[fin], eyp[fin], rtol=1e-5)):
            bad += 1
            print("ORACLE MISMATCH case", it, V, W, T, N, flush=True)
print("short-row fuzz: %d cases, %d mismatches" % (cases, bad), flush=True)
