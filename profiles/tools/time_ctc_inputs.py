"""The headline CTC search (N=4096, T=512, V=256, K=16) on other kinds of input than the bench's: blank-dominated
speech-like rows, flatter rows, and real-ish mixtures -- the short-list / lean-tier hit rates are input dependent."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import numpy as np, torch
import bench
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
T, N, V, K = 512, 4096, 256, 16
def t(lg): return bench.event_ms(lambda: F.ctc_prefix_search(lg, K), reps=5, warm=2)
lg = bench.peaky_logits(T, N, V, dev, 3); print("bench input (a new token peaks in every frame, +12): %.3f ms" % t(lg)); del lg
lg = bench.speechlike_logits(T, N, V, dev, 5, bench.synthetic_bigram_dicts(V)); print("speech-like (blank peaks in 95 %% of the frames): %.3f ms" % t(lg)); del lg
g = torch.Generator(device=dev).manual_seed(9)
for scale in (8.0, 5.0, 3.0):
    lg = torch.randn((T, N, V + 1), device=dev, generator=g)
    peak = torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g)
    lg.scatter_add_(2, peak, torch.full((T, N, 1), scale, device=dev))
    print("peak +%g: %.3f ms" % (scale, t(lg))); del lg
lg = torch.randn((T, N, V + 1), device=dev, generator=g) * 3.0; print("no peak, N(0, 9) logits: %.3f ms" % t(lg))
