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
lg = bench.peaky_logits(T, N, V, dev, 0x5EED0003, chunk=T); a = med(lg); del lg
lg = bench.peaky_logits(T, N, V, dev, 4, chunk=T); a2 = med(lg); del lg
lg = bench.speechlike_logits(T, N, V, dev, 5, bench.synthetic_bigram_dicts(V)); b = med(lg)
print("bench draw %.3f  seed 4 %.3f  speech-like %.3f" % (a, a2, b))
