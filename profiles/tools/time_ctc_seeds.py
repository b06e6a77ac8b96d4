"""The headline CTC search on statistically identical inputs drawn with different seeds / generation orders."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import numpy as np, torch
import bench
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
T, N, V, K = 512, 4096, 256, 16
def med(lg, reps=10, warm=3):
    for _ in range(warm): F.ctc_prefix_search(lg, K)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); F.ctc_prefix_search(lg, K); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for seed in (3, 4, 5):
    g = torch.Generator(device=dev).manual_seed(seed)
    lg = torch.randn((T, N, V + 1), device=dev, generator=g)
    peak = torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g)
    lg.scatter_add_(2, peak, torch.full((T, N, 1), 12.0, device=dev))
    print("randn, seed %d: %.3f" % (seed, med(lg)), flush=True); del lg
for seed, chunk in ((3, 64), (3, 512), (4, 64), (4, 512), (0x5EED0003, 512)):
    lg = bench.peaky_logits(T, N, V, dev, seed, chunk=chunk)
    print("peaky_logits seed %d chunk %d: %.3f" % (seed, chunk, med(lg)), flush=True); del lg
