"""Randomised run: BeamSearch over bigram-table models, every route of the package against the others
(one launch for the whole search / a launch per iteration on the table / around the model's forward / the
step-by-step loop): y, lengths and log-probabilities.  python tests/fuzz/fuzz_beam_search.py SEED SECONDS"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "pydrobert-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, warnings
from pydrobert_amd import modules as M, switches
warnings.simplefilter("ignore")
dev = "cuda"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60)
bad = n_cases = 0
t_say = time.time()
ROUTES = {"search": dict(PDT_BEAM_FUSED=1, PDT_BEAM_TABLE=1, PDT_BEAM_SEARCH=1), "table": dict(PDT_BEAM_FUSED=1, PDT_BEAM_TABLE=1, PDT_BEAM_SEARCH=0),
          "fused": dict(PDT_BEAM_FUSED=1, PDT_BEAM_TABLE=0), "loop": dict(PDT_BEAM_FUSED=0)}
while time.time() < t_end:
    n_cases += 1
    if time.time() - t_say > 45.0:
        t_say = time.time(); print("cases", n_cases, "mismatches", bad, flush=True)
    V = int(rng.integers(65, 1100)) if rng.random() < 0.8 else int(rng.integers(2, 65))
    w_max = min(64, 256 // ((V + 63) // 64))
    W = int(rng.integers(1, min(16, w_max) + 1)) if rng.random() < 0.7 or w_max < 17 else int(rng.integers(17, w_max + 1))
    N, iters = int(rng.integers(1, 9)), int(rng.integers(1, 60))
    eos = None if rng.random() < 0.2 else int(rng.integers(0, V))
    fin, sos = bool(rng.random() < 0.5), (-1 if rng.random() < 0.3 else int(rng.integers(0, V)))
    # a sparse bigram table over full unigrams (vectorised draw: V^2 keys are too many for a Python loop)
    uni = {v: (float(a), float(b)) for v, (a, b) in enumerate(zip(rng.normal(size=V), rng.normal(size=V)))}
    if sos < 0:
        uni[sos] = (-99.0, float(rng.normal()))
    if eos is not None:
        uni[eos] = (uni[eos][0] + float(rng.uniform(1.0, 7.0)), uni[eos][1])
    ctx = rng.integers(-1 if sos < 0 else 0, V, 6 * V)
    big = {(int(c) if c >= 0 else sos, int(v)): float(x) for c, v, x in zip(ctx, rng.integers(0, V, 6 * V), rng.normal(size=6 * V) * 2)}
    if rng.random() < 0.3:  # exact ties in the table
        big = {k: round(x * 2) / 2 for k, x in big.items()}
    lm = M.LookupLanguageModel(V, sos, [uni, big]).to(dev)
    bs = M.BeamSearch(lm, W, eos=eos, finish_all_paths=fin, pad_value=int(rng.integers(-9, 3))).to(dev)
    outs = {}
    for name, sw in ROUTES.items():
        for k, v in sw.items():
            switches.set(k, v)
        with torch.no_grad():
            outs[name] = bs(dict(), N, iters)
    y, yl, lp = outs["search"]
    for name in ("table", "fused"):
        y1, yl1, lp1 = outs[name]
        if y.shape != y1.shape or not (torch.equal(y, y1) and torch.equal(yl, yl1) and torch.equal(lp, lp1)):
            bad += 1; print("MISMATCH search vs", name, V, W, N, iters, eos, fin, sos, flush=True)
            break
    y2, yl2, lp2 = outs["loop"]  # (the loop forms log_softmax with torch: probabilities to 1e-5, near ties may swap)
    if not torch.equal(yl, yl2) and torch.allclose(lp.sort(1)[0], lp2.sort(1)[0], rtol=1e-5, atol=1e-6):
        pass
    elif not (torch.equal(yl, yl2) and torch.allclose(lp, lp2, rtol=1e-5, atol=1e-5)):
        bad += 1; print("MISMATCH search vs loop", V, W, N, iters, eos, fin, sos, float((lp - lp2).abs().max()), flush=True)
print("cases", n_cases, "mismatches", bad)
