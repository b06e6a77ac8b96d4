"""Randomised differential run, part 2: step functions, losses, LM lookup, image ops."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "pydrobert-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, warnings
import oracle
from pydrobert_amd import functional as F, modules as M
from _lm_fixtures import random_dicts
warnings.simplefilter("ignore")
dev = "cuda"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60)
bad = n_cases = 0
t_say = time.time()
def T(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
lms = {}
while time.time() < t_end:
    n_cases += 1
    if time.time() - t_say > 45.0:  # (a line a minute: a silent run is taken for a hung one)
        t_say = time.time(); print("cases", n_cases, "mismatches", bad, flush=True)
    kind = rng.integers(0, 4)
    if kind == 0:  # beam_search_advance
        N, Kp, V, W = int(rng.integers(1, 6)), int(rng.integers(1, 20)), int(rng.integers(1, 60)), int(rng.integers(1, 40))
        if rng.random() < 0.2:  # wider than a wave: the radix-select form (csrc/advance_wide.hip)
            Kp, V, W = int(rng.integers(1, 140)), int(rng.integers(1, 300)), int(rng.integers(1, 200))
            if rng.random() < 0.5: Kp = max(Kp, 65)
            else: W = max(W, 65)
        flat = rng.random() < 0.35  # rows of more than 64 tokens in a workgroup's registers: the flat selection (round 5)
        if flat:
            Kp = int(rng.integers(1, 17))
            V, W = int(rng.integers(65, min(1100, 16384 // Kp) + 1)), int(rng.integers(1, 70))
        S = int(rng.integers(0, 9))
        lpt = rng.normal(size=(N, Kp, V)).astype(np.float32)
        if rng.random() < 0.3: lpt = np.round(lpt * 4) / 4  # exact ties: lowest flat index first
        lpp = rng.normal(size=(N, Kp)).astype(np.float32)
        if flat:
            what = rng.integers(0, 5)
            if what == 1:  # finished beams: -inf but for one token
                done = rng.random((N, Kp)) < rng.random()
                row = np.full(V, -np.inf, np.float32); row[int(rng.integers(0, V))] = 0.0
                lpt[done] = row
            elif what == 2:  # the best candidates crowd one lane
                lpt[:, :, int(rng.integers(0, 64))::64] += 25.0
            elif what == 3:  # -inf sprinkled, or nearly everywhere
                lpt = np.where(rng.random(lpt.shape) < rng.choice([0.3, 0.999]), -np.inf, lpt).astype(np.float32)
            elif what == 4:
                lpp[rng.random((N, Kp)) < 0.3] = -np.inf
        yp = rng.integers(0, V, (S, N, Kp))
        ypl = rng.integers(1 if S else 0, S + 1, (N, Kp)) if (rng.random() < 0.5) else None
        exp = oracle.beam_search_advance(lpt, W, lpp, yp, ypl)
        act = F.beam_search_advance(T(lpt), W, T(lpp), T(yp), None if ypl is None else T(ypl))
        K = min(W, Kp * V)
        ok = np.array_equal(exp[1][:, :K], act[1].cpu().numpy()[:, :K]) and np.array_equal(exp[3][:, :K], act[3].cpu().numpy()[:, :K]) \
            and np.array_equal(exp[2][:, :K], act[2].cpu().numpy()[:, :K])
        if ok and exp[0].shape == tuple(act[0].shape):
            m = np.arange(exp[0].shape[0])[:, None, None] < exp[1][None]
            m[:, :, K:] = False
            ok = np.array_equal(np.where(m, exp[0], 0), np.where(m, act[0].cpu().numpy(), 0))
        else:
            ok = False
        if not ok:
            bad += 1; print("MISMATCH beam_advance", N, Kp, V, W, S, ypl is not None)
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez("gpurun_out/fuzz_beam_%d.npz" % bad, lpt=lpt, lpp=lpp, yp=yp, W=W,
                     ypl=np.zeros(0) if ypl is None else ypl, act1=act[1].cpu().numpy(),
                     act2=act[2].cpu().numpy(), act3=act[3].cpu().numpy())
    elif kind == 1:  # LM lookup vs oracle brute force
        key = (int(rng.integers(2, 7)), int(rng.integers(1, 5)), int(rng.integers(0, 2)))
        if key not in lms:
            V, N, s = key
            sos = -1 if s else 0
            dicts = random_dicts(np.random.default_rng(sum(key)), V, N, 0.5, sos if sos < 0 else None)
            if not dicts[-1]:
                dicts[-1][tuple([0] * N) if N > 1 else 0] = -1.0
            lms[key] = (dicts, M.LookupLanguageModel(V, sos, [d.copy() for d in dicts]).to(dev), sos)
        dicts, lm, sos = lms[key]
        V, N, _ = key
        S, B = int(rng.integers(0, 7)), int(rng.integers(1, 6))
        toks = list(range(V)) + ([sos] if sos < 0 else [])
        hist = np.asarray(toks)[rng.integers(0, len(toks), (S, B))].reshape(S, B)
        idx = rng.integers(0, S + 1, B)
        act = lm(T(hist.astype(np.int64)), None, T(idx))[0].cpu().numpy()
        exp = oracle.backoff_log_probs(dicts, V, sos, hist, idx)
        if not np.allclose(exp, act, atol=1e-5, equal_nan=True):
            bad += 1; print("MISMATCH lm", key, S, B)
    elif kind == 2:  # hard OCD loss vs oracle
        N, R, H, V = int(rng.integers(1, 5)), int(rng.integers(1, 20)), int(rng.integers(1, 20)), int(rng.integers(2, 9))
        ref, hyp = rng.integers(0, V, (R, N)), rng.integers(0, V, (H, N))
        logits = rng.normal(size=(H, N, V)).astype(np.float32)
        eos = None if rng.random() < 0.4 else int(rng.integers(0, V))
        red = ["mean", "sum", "none"][rng.integers(0, 3)]
        exp = oracle.hard_optimal_completion_distillation_loss(logits, ref, hyp, eos=eos, reduction=red)
        act = F.hard_optimal_completion_distillation_loss(T(logits), T(ref), T(hyp), eos=eos, reduction=red, warn=False).cpu().numpy()
        if not np.allclose(exp, act, rtol=2e-5, atol=2e-5):
            bad += 1; print("MISMATCH hocd", N, R, H, V, eos, red, np.abs(exp - act).max())
    else:  # dense image warp
        N, C, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(1, 20)), int(rng.integers(1, 20))
        img = rng.normal(size=(N, C, H, W)).astype(np.float32)
        flow = (rng.normal(size=(N, H, W, 2)) * 3).astype(np.float32)
        mode = ["bilinear", "nearest"][rng.integers(0, 2)]
        if mode == "nearest":
            flow = (np.round(flow * 4) / 4 + 0.1).astype(np.float32)
        padm = ["border", "zeros", "reflection"][rng.integers(0, 3)]
        ind = ["hw", "wh"][rng.integers(0, 2)]
        exp = oracle.dense_image_warp(img, flow, ind, mode, padm)
        act = F.dense_image_warp(T(img), T(flow), ind, mode, padm).cpu().numpy()
        if not np.allclose(exp, act, atol=2e-4):
            bad += 1; print("MISMATCH dense warp", N, C, H, W, mode, padm, ind, np.abs(exp - act).max())
print("cases", n_cases, "mismatches", bad)
