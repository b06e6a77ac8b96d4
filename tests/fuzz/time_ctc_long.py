"""CTC search over long rows: the register-row form (PDT_CTC_ROWREG=1, =2: one producer) against the
LDS-row form (=0) -- same results at a sample against the oracle, then the C5 shard and C3 timings."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import numpy as np, torch
import oracle
from pydrobert_amd import functional as F, switches
from bench import peaky_logits, event_ms

dev = torch.device("cuda:0")
modes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2").split(",")]
rng = np.random.default_rng(3)
for V, K in ((600, 16), (1000, 16), (2047, 8), (3000, 32), (5000, 16), (5119, 16)):
    T, N = 40, 6
    lg = rng.normal(size=(T, N, V + 1)).astype(np.float32)
    np.put_along_axis(lg, rng.integers(0, V + 1, (T, N, 1)), 11.0, 2)
    lens = rng.integers(0, T + 1, N)
    exp = oracle.ctc_prefix_search(lg, K, lens)
    for m in modes:
        switches.set("PDT_CTC_ROWREG", m)
        y, yl, yp = (x.cpu().numpy() for x in F.ctc_prefix_search(torch.from_numpy(lg).to(dev), K, torch.from_numpy(lens).to(dev)))
        ok = np.array_equal(yl, exp[1]) and np.array_equal(y, exp[0]) and np.allclose(yp, exp[2], rtol=1e-5)
        print("V", V, "K", K, "mode", m, "ok" if ok else "MISMATCH", flush=True)
for name, (T, N, V) in (("C5", (512, 4096, 5000)), ("C3", (1000, 1024, 1000)), ("V2000", (512, 4096, 2000))):
    lg = peaky_logits(T, N, V, dev, 0x5EED0006)
    outs = []
    for m in modes:
        switches.set("PDT_CTC_ROWREG", m)
        outs.append(F.ctc_prefix_search(lg, 16))
        ms = event_ms(lambda: F.ctc_prefix_search(lg, 16), reps=3, warm=1)
        print(name, "mode", m, "%.3f ms" % ms, "%.2f TB/s" % (lg.numel() * 4 / ms / 1e9), flush=True)
    for o in outs[1:]:
        print("  same as mode", modes[0], all(torch.equal(a, b) for a, b in zip(outs[0], o)))
    del lg, outs
