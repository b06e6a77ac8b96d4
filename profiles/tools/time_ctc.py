import sys, torch, time, os
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
T, N, V, K = int(os.environ.get("T", 512)), int(os.environ.get("N", 4096)), int(os.environ.get("V", 256)), 16
g = torch.Generator(device=dev).manual_seed(3)
logits = torch.randn((T, N, V + 1), device=dev, generator=g)
peak = torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g)
logits.scatter_add_(2, peak, torch.full((T, N, 1), float(os.environ.get("SCALE", 12.0)), device=dev))
for _ in range(5):
    F.ctc_prefix_search(logits, K)
torch.cuda.synchronize()
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    F.ctc_prefix_search(logits, K)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print("ctc ms min %.3f median %.3f max %.3f" % (ts[0], ts[len(ts) // 2], ts[-1]))
