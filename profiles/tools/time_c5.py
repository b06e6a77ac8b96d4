"""C5 decode (N=4096, T=512, V=5000, K=16) and the C3 search, event-timed: for A/B runs of ctc_rowreg.hip builds (PDT_AMD_LIB)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from pydrobert_amd import functional as F
from bench import peaky_logits, event_ms
dev = torch.device("cuda:0")
lg = peaky_logits(512, 4096, 5000, dev, 0x5EED0006)
print(os.environ.get("PDT_AMD_LIB", "default")[-40:], "C5 decode ms", ["%.3f" % event_ms(lambda: F.ctc_prefix_search(lg, 16), reps=5, warm=2) for _ in range(4)])
del lg
lg = peaky_logits(1000, 1024, 1000, dev, 0x5EED0003)
print(os.environ.get("PDT_AMD_LIB", "default")[-40:], "C3 search ms", ["%.3f" % event_ms(lambda: F.ctc_prefix_search(lg, 16), reps=5, warm=2) for _ in range(3)])
