"""error_rate + prefix_error_rates + optimal_completion at the C2 shape (every operator classifies its own inputs: the default)."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F, _string
dev = torch.device("cuda:0")
rng = np.random.default_rng(0x5EED0002)
T, N, V = 512, 4096, 256
ref = torch.from_numpy(rng.integers(0, V, (T, N))).to(dev)
hyp = torch.from_numpy(rng.integers(0, V, (T, N))).to(dev)
def fn():
    F.error_rate(ref, hyp, warn=False); F.prefix_error_rates(ref, hyp, warn=False); F.optimal_completion(ref, hyp, warn=False)
for _ in range(3): fn()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort(); print("error_rate + prefix_error_rates + optimal_completion ms min %.3f median %.3f" % (ts[0], ts[len(ts)//2]))
