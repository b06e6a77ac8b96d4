import sys, torch, os
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
T, N, V, K = 512, 4096, 256, 16
g = torch.Generator(device=dev).manual_seed(3)
logits = torch.randn((T, N, V + 1), device=dev, generator=g)
peak = torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g)
logits.scatter_add_(2, peak, torch.full((T, N, 1), 12.0, device=dev))
def run(sync, reps=30):
    for _ in range(5): F.ctc_prefix_search(logits, K)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); F.ctc_prefix_search(logits, K); b.record()
        if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[0], ts[len(ts)//2], ts[-1]
print("sync between reps   min %.3f med %.3f max %.3f" % run(True))
print("back to back        min %.3f med %.3f max %.3f" % run(False))
print("sync between reps   min %.3f med %.3f max %.3f" % run(True))
print("back to back (5)    min %.3f med %.3f max %.3f" % run(False, 5))
# keep outputs alive: no allocator reuse
keep = []
ev = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); keep.append(F.ctc_prefix_search(logits, K)); b.record(); ev.append((a, b))
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
print("back to back, outputs kept: min %.3f med %.3f max %.3f" % (ts[0], ts[5], ts[-1]))
