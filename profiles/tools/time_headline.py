"""The headline CTC search (N=4096, T=512, V=256, K=16) on the bench's input and four other draws, event-timed
(A/B of ctc_search.hip builds through PDT_AMD_LIB)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from pydrobert_amd import functional as F
from bench import peaky_logits, event_ms
dev = torch.device("cuda:0")
tag = os.environ.get("PDT_AMD_LIB", "default")[-28:]
for seed in (0x5EED0002, 3, 4):
    lg = peaky_logits(512, 4096, 256, dev, seed)
    print(tag, "seed", hex(seed), "ms", ["%.4f" % event_ms(lambda: F.ctc_prefix_search(lg, 16), reps=9, warm=3) for _ in range(3)])
    out = F.ctc_prefix_search(lg, 16)
    print(tag, "  checksum", int(out[0].sum()), int(out[1].sum()), float(out[2].double().sum()))
