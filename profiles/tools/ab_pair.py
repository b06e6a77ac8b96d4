"""A/B of the two-frames-per-pass producer (PDT_CTC_PAIR, csrc/ctc_search.hip) against the one-frame
form on the bench shape: same bits (torch.equal on y / lens / probs, several draws, odd frame counts,
ragged lengths, flat rows that miss the short-list window), then the times.
    gpurun -- 'python profiles/tools/ab_pair.py'"""
import os
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F, switches  # noqa: E402

dev = torch.device("cuda:0")
K = 16


def peaky(T, N, V, seed, scale=12.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    lg = torch.randn((T, N, V + 1), device=dev, generator=g)
    peak = torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g)
    lg.scatter_add_(2, peak, torch.full((T, N, 1), scale, device=dev))
    return lg


def both(lg, lens=None):
    out = []
    for pair in (1, 0):
        switches.set("PDT_CTC_PAIR", pair)
        out.append(F.ctc_prefix_search(lg, K, lens))
    switches.set("PDT_CTC_PAIR", 1)
    return out


bad = 0
cases = [(512, 4096, 3, 12.0), (511, 1024, 4, 12.0), (1, 64, 5, 12.0), (2, 64, 6, 12.0), (3, 64, 7, 12.0), (64, 512, 8, 6.0),
         (64, 512, 9, 3.0), (64, 512, 10, 0.0), (129, 512, 11, 20.0)]
for T, N, seed, scale in cases:
    lg = peaky(T, N, 256, seed, scale)
    g = torch.Generator(device=dev).manual_seed(seed + 100)
    for lens in (None, torch.randint(0, T + 1, (N,), device=dev, generator=g)):
        a, b = both(lg, lens)
        same = all(torch.equal(x, y) for x, y in zip(a, b))
        bad += 0 if same else 1
        print("T={} N={} scale={} lens={}: {}".format(T, N, scale, lens is not None, "same bits" if same else "DIFFERENT"), flush=True)
# masked rows: -inf logits on a third of the vocabulary
lg = peaky(100, 256, 256, 21)
lg[:, :, 5:90] = float("-inf")
a, b = both(lg)
same = all(torch.equal(x, y) for x, y in zip(a, b))
bad += 0 if same else 1
print("masked vocabulary:", "same bits" if same else "DIFFERENT")
print("MISMATCHES", bad)

lg = peaky(512, 4096, 256, 3)
for pair in (1, 0, 1, 0):
    switches.set("PDT_CTC_PAIR", pair)
    for _ in range(3):
        F.ctc_prefix_search(lg, K)
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        F.ctc_prefix_search(lg, K)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("PDT_CTC_PAIR={} ms min {:.3f} median {:.3f}".format(pair, ts[0], ts[len(ts) // 2]), flush=True)
