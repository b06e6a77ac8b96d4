"""C5 shard and C3 search timings under the current library (A/B through PDT_AMD_LIB)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from pydrobert_amd import functional as F
from bench import peaky_logits, event_ms
dev = torch.device("cuda:0")
out = []
for T, N, V in ((512, 4096, 5000), (1000, 1024, 1000)):
    lg = peaky_logits(T, N, V, dev, 0x5EED0006)
    out.append("V=%d %s" % (V, ["%.3f" % event_ms(lambda: F.ctc_prefix_search(lg, 16), reps=5, warm=2) for _ in range(3)]))
    del lg
print(" | ".join(out))
