import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import numpy as np, torch
import bench
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
T, N, V, K = 512, 4096, 256, 16
def times(lg, reps=20, warm=5):
    for _ in range(warm): F.ctc_prefix_search(lg, K)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); F.ctc_prefix_search(lg, K); b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in ev]
    return "first %.3f min %.3f med %.3f" % (ts[0], min(ts), float(np.median(ts)))
lg = bench.peaky_logits(T, N, V, dev, 3); print("bench.peaky_logits:", times(lg))
g = torch.Generator(device=dev).manual_seed(3)
lg2 = torch.randn((T, N, V + 1), device=dev, generator=g)
peak = torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g)
lg2.scatter_add_(2, peak, torch.full((T, N, 1), 12.0, device=dev))
print("time_ctc input:", times(lg2))
print("bench.peaky_logits again:", times(lg))
print("stats", float(lg.mean()), float(lg2.mean()), float(lg.max()), float(lg2.max()), lg.stride(), lg2.stride(), lg.data_ptr() % 4096, lg2.data_ptr() % 4096)
c = lg.clone(); print("clone of peaky (new allocation):", times(c)); del c
tmp = lg.clone(); lg.copy_(lg2); print("time_ctc DATA in peaky's STORAGE:", times(lg)); lg2.copy_(tmp); print("peaky DATA in time_ctc's STORAGE:", times(lg2))
