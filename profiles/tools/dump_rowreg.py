"""Outputs of the register-row CTC search on a set of inputs, saved for a comparison between two builds of the
library (PDT_AMD_LIB): python profiles/tools/dump_rowreg.py OUT.pt"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from pydrobert_amd import functional as F
from bench import peaky_logits, event_ms
dev = torch.device("cuda:0")
outs = []
for (T, N, V, seed, scale) in ((96, 256, 5000, 1, 12.0), (64, 128, 1000, 2, 12.0), (64, 128, 1000, 3, 4.0), (50, 64, 2500, 4, 0.0),
                               (40, 64, 8000, 5, 9.0), (33, 32, 12000, 6, 12.0), (80, 100, 330, 7, 6.0), (64, 64, 700, 8, 20.0)):
    g = torch.Generator(device=dev).manual_seed(seed)
    lg = torch.randn((T, N, V + 1), device=dev, generator=g)
    lg.scatter_add_(2, torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g), torch.full((T, N, 1), scale, device=dev))
    if seed == 3:
        lg = (lg * 2).round() / 2          # exact ties everywhere
    if seed == 4:
        lg[:, :, 100:900] = float("-inf")  # masked vocabulary
    lens = torch.randint(0, T + 1, (N,), device=dev, generator=g)
    for ln in (None, lens):
        outs.append([x.cpu() for x in F.ctc_prefix_search(lg, 16, ln)])
torch.save(outs, sys.argv[1])
lg = peaky_logits(512, 4096, 5000, dev, 0x5EED0006)
print(os.environ.get("PDT_AMD_LIB", "default"), "C5 decode ms", ["%.3f" % event_ms(lambda: F.ctc_prefix_search(lg, 16), reps=5, warm=2) for _ in range(3)])
del lg
lg = peaky_logits(1000, 1024, 1000, dev, 0x5EED0003)
print(os.environ.get("PDT_AMD_LIB", "default"), "C3 search ms", ["%.3f" % event_ms(lambda: F.ctc_prefix_search(lg, 16), reps=5, warm=2) for _ in range(3)])
